"""Stand-alone SH stage and the low-rank gradient exchange of the view-parallel step."""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import torch_oracle as O
from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads

pytestmark = pytest.mark.gpu
N, W, H, V, DEG = 6000, 160, 128, 4, 3
NAMES = ["means3D", "scales", "rotations", "opacities", "shs"]


def test_sh_to_rgb_matches_oracle(hip_device):
    from splatfields_amd.sh import sh_to_rgb
    sp = make_splats(5000, seed=3)
    sp["shs"] = sp["shs"] * 3.0  # make the clamp at 0 bite
    cam = make_camera(2, 64, 64)
    for deg in range(4):
        m = sp["means3D"].to(hip_device).requires_grad_(True)
        s = sp["shs"].to(hip_device).requires_grad_(True)
        col = sh_to_rgb(m, s, cam.camera_center.to(hip_device), deg)
        g = torch.randn(5000, 3, generator=torch.Generator().manual_seed(deg))
        (col * g.to(hip_device)).sum().backward()
        m64 = sp["means3D"].double().requires_grad_(True)
        s64 = sp["shs"].double().requires_grad_(True)
        ref = O.sh_to_rgb(deg, s64, m64, cam.camera_center.double())
        (ref * g.double()).sum().backward()
        assert (ref == 0).any()
        assert torch.allclose(col.detach().cpu().double(), ref.detach(), atol=2e-6)
        assert torch.allclose(s.grad.cpu().double(), s64.grad, atol=1e-5 * s64.grad.abs().max().item())
        mref = m64.grad if m64.grad is not None else torch.zeros_like(m64)  # degree 0 does not depend on the direction
        assert torch.allclose(m.grad.cpu().double(), mref, atol=1e-4 * max(mref.abs().max().item(), 1e-12))


def _rs(cam, dev, deg=DEG):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                         torch.ones(3, device=dev), 1.0, cam.world_view_transform, cam.full_proj_transform, deg,
                                         cam.camera_center, False, False)


def test_shs_path_equals_precomputed_colours_of_sh_stage(hip_device):
    from diff_gaussian_rasterization import GaussianRasterizer
    from splatfields_amd.sh import sh_to_rgb
    dev = hip_device
    sp = make_splats(N, seed=5, device=dev)
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    cam = make_camera(1, W, H, device=dev)
    res = []
    for split in (False, True):
        p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
        kw = dict(means3D=p["means3D"], means2D=torch.zeros_like(p["means3D"]), opacities=p["opacities"], scales=p["scales"],
                  rotations=p["rotations"])
        if split:
            kw["colors_precomp"] = sh_to_rgb(p["means3D"], p["shs"], cam.camera_center, DEG)
        else:
            kw["shs"] = p["shs"]
        c, r, d, a = GaussianRasterizer(_rs(cam, dev)).forward_ex(**kw)
        torch.autograd.backward((c, d, a), (gi, gd, ga))
        res.append((c.detach(), {k: v.grad.clone() for k, v in p.items()}))
    assert torch.allclose(res[0][0], res[1][0], atol=1e-6)
    for k in NAMES:
        scale = res[0][1][k].abs().max().item()
        assert torch.allclose(res[0][1][k], res[1][1][k], atol=2e-5 * scale), k


def _reference_grads(dev, n=N, V=V):
    """plain path: every view through the SH rasterizer, mean of the views, one backward (reference train.py:169-252)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    sp = make_splats(n, seed=7, device=dev)
    p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    for v in range(V):
        cam = make_camera(v, W, H, device=dev)
        c, r, d, a = GaussianRasterizer(_rs(cam, dev)).forward_ex(means3D=p["means3D"], means2D=torch.zeros_like(p["means3D"]),
                                                                  opacities=p["opacities"], shs=p["shs"], scales=p["scales"],
                                                                  rotations=p["rotations"])
        torch.autograd.backward((c, d, a), (gi / V, gd / V, ga / V))
    return {k: v.grad.detach().cpu() for k, v in p.items()}


def _regulariser(p):
    """a loss term that reaches the leaves OUTSIDE the rasterizer (scale / opacity / SH regularisers of a training loss)"""
    return 1e-3 * (p["scales"] ** 2).sum() + 2e-3 * p["opacities"].sum() + 1e-4 * (p["shs"][:, 1:] ** 2).sum() \
        + 1e-3 * (p["means3D"] * p["rotations"][:, :3]).sum()


def _gather_grads(dev, rank, world, V=V, slices=None, regularise=False):
    from splatfields_amd.view_parallel import sh_gather_step
    sp = make_splats(N, seed=7, device=dev)
    p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    cams = [make_camera(v, W, H, device=dev) for v in range(V)]

    def bwd(vi, c, d, a):
        if regularise:   # every view's loss carries the term / V, as `(loss / V).backward()` of a regularised loss does
            loss = (c * gi).sum() + (d * gd).sum() + (a * ga).sum() + _regulariser(p)
            (loss / V).backward()
            return
        torch.autograd.backward((c, d, a), (gi / V, gd / V, ga / V))

    sh_gather_step(p, cams, torch.ones(3, device=dev), DEG, bwd, rank=rank, world=world, slices=slices)
    torch.cuda.synchronize()
    return {k: v.grad.detach().cpu() for k, v in p.items()}


def _close(a, b):
    for k in NAMES:
        scale = b[k].abs().max().item()
        assert torch.allclose(a[k], b[k], atol=5e-5 * scale), (k, (a[k] - b[k]).abs().max().item() / scale)


def test_gather_step_single_rank_equals_plain_multiview(hip_device):
    _close(_gather_grads(hip_device, 0, 1), _reference_grads(hip_device))


def _worker(rank, world, port, q, views=V, slices=None, regularise=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")  # both ranks share the single GPU of the test box; gloo moves the tensors
    torch.cuda.set_device(dev)
    g = _gather_grads(dev, rank, world, views, slices, regularise)
    q.put((rank, {k: v.numpy().copy() for k, v in g.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_step_two_ranks_equals_plain_multiview(hip_device):
    ref = _reference_grads(hip_device)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, g in outs:
        _close({k: torch.from_numpy(v) for k, v in g.items()}, ref)


@pytest.mark.parametrize("slices", [1, 3, 4])
def test_sliced_gather_step_two_ranks_one_view_each(hip_device, slices):
    """One view per rank (the 8-GPU bench's shape): the per-splat backward runs in `slices` ranges and every range's all-gather /
    grouped all-reduce is issued behind it from inside the backward (view_parallel._SlicedGatherHook).  Same gradients as the
    single-process two-view step, on both ranks, for the unsliced exchange (1) and for ranges that do (4: 1536 rows) and do
    not (3: 2048 rows) divide the cloud evenly."""
    ref = _reference_grads(hip_device, V=2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 2, slices)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, g in outs:
        _close({k: torch.from_numpy(v) for k, v in g.items()}, ref)
    # both ranks hold bit-identical gradients (what replicated Adam needs)
    for k in NAMES:
        assert (outs[0][1][k] == outs[1][1][k]).all(), k


def _reference_grads_regularised(dev, V):
    """single process: mean over the views of (view loss + regulariser), one backward per view"""
    from diff_gaussian_rasterization import GaussianRasterizer
    sp = make_splats(N, seed=7, device=dev)
    p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    for v in range(V):
        cam = make_camera(v, W, H, device=dev)
        c, r, d, a = GaussianRasterizer(_rs(cam, dev)).forward_ex(means3D=p["means3D"], means2D=torch.zeros_like(p["means3D"]),
                                                                  opacities=p["opacities"], shs=p["shs"], scales=p["scales"],
                                                                  rotations=p["rotations"])
        (((c * gi).sum() + (d * gd).sum() + (a * ga).sum() + _regulariser(p)) / V).backward()
    return {k: v.grad.detach().cpu() for k, v in p.items()}


@pytest.mark.parametrize("slices", [1, 4])
def test_gather_step_with_a_regulariser_outside_the_rasterizer(hip_device, slices):
    """a loss term that reaches scales / opacities / SH / means outside the rasterizer.  The sliced exchange
    (gradients written into the hook's buffers, autograd gets None from the rasterizer) must not lose it: both forms equal the
    single-process regularised step, on both ranks, bit-identical across the ranks."""
    ref = _reference_grads_regularised(hip_device, 2)
    plain = _reference_grads(hip_device, V=2)
    assert not torch.allclose(ref["scales"], plain["scales"], atol=1e-6 * plain["scales"].abs().max().item())   # the term matters
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 2, slices, True)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, g in outs:
        _close({k: torch.from_numpy(v) for k, v in g.items()}, ref)
    for k in NAMES:
        assert (outs[0][1][k] == outs[1][1][k]).all(), k


@pytest.mark.parametrize("slices", [3, 4])
def test_sliced_gather_step_four_ranks_with_a_regulariser(hip_device, slices):
    """more than two ranks.  Four gloo ranks (sharing the box's one GPU), one view each, the sliced
    exchange (all-gather of colour gradients + grouped all-reduce per splat range, issued from inside the backward), a cloud
    whose size (6000) is no multiple of ranks x 256, slice counts that do (4 x 1536) and do not (3 x 2048) divide it evenly, and
    a loss term outside the rasterizer: the single-process four-view step's gradients on every rank, bit-identical across the
    ranks."""
    assert N % (4 * 256) != 0
    ref = _reference_grads_regularised(hip_device, 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, q, 4, slices, True)) for r in range(4)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(r for r, _ in outs) == [0, 1, 2, 3]
    for rank, g in outs:
        _close({k: torch.from_numpy(v) for k, v in g.items()}, ref)
    for k in NAMES:
        for other in outs[1:]:
            assert (outs[0][1][k] == other[1][k]).all(), k


def test_gather_step_single_rank_with_a_regulariser(hip_device):
    _close(_gather_grads(hip_device, 0, 1, regularise=True), _reference_grads_regularised(hip_device, V))


def test_sliced_backward_equals_the_unsliced_one_bit_for_bit(hip_device):
    """sr_backward_blend + sr_backward_splats over any partition == sr_backward (same kernels, same arithmetic per splat)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = hip_device
    sp = make_splats(N, seed=9, device=dev)
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    cam = make_camera(3, W, H, device=dev)

    class Hook:
        def __init__(self, slices):
            self.slices, self.buf, self.seen = slices, None, []

        def buffers(self, n):
            self.buf = {k: torch.full((n, c), float("nan"), device=dev) for k, c in
                        (("means3D", 3), ("scales", 3), ("rotations", 4), ("opacities", 1), ("colors", 3))}
            return self.buf

        def on_slice(self, j, lo, hi):
            self.seen.append((j, lo, hi))

    def run(hook):
        p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
        m2 = torch.zeros_like(p["means3D"], requires_grad=True)
        sink = []
        c, r, d, a = GaussianRasterizer(_rs(cam, dev)).forward_ex(means3D=p["means3D"], means2D=m2, opacities=p["opacities"], shs=p["shs"],
                                                                  scales=p["scales"], rotations=p["rotations"], color_grad_sink=sink,
                                                                  slice_hook=hook)
        torch.autograd.backward((c, d, a), (gi, gd, ga))
        if hook is None:
            return {"means3D": p["means3D"].grad, "scales": p["scales"].grad, "rotations": p["rotations"].grad,
                    "opacities": p["opacities"].grad, "colors": sink[0], "means2D": m2.grad}
        assert all(p[k].grad is None for k in NAMES)          # the hook's owner sets .grad
        return dict(hook.buf, means2D=m2.grad)

    whole = run(None)
    for k in (1, 2, 5):
        h = Hook(k)
        part = run(h)
        assert [x[0] for x in h.seen] == list(range(len(h.seen))) and h.seen[0][1] == 0 and h.seen[-1][2] == N
        assert all(lo % 256 == 0 for _, lo, _ in h.seen) and all(a[2] == b[1] for a, b in zip(h.seen, h.seen[1:]))
        for name in whole:
            assert torch.equal(part[name], whole[name]), (k, name)


# ---- SH-sharded step: colours / colour gradients travel by all-to-all, each rank owns a slice of the SH tensor ----
def _sharded_grads(dev, rank, world, n=N):
    from splatfields_amd.view_parallel import sh_sharded_step
    sp = make_splats(n, seed=7, device=dev)
    p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    cams = [make_camera(v, W, H, device=dev) for v in range(V)]

    def bwd(vi, c, d, a):
        torch.autograd.backward((c, d, a), (gi / V, gd / V, ga / V))

    lo, hi, d_shs = sh_sharded_step(p, cams, torch.ones(3, device=dev), DEG, bwd, rank=rank, world=world)
    torch.cuda.synchronize()
    assert p["shs"].grad is None
    out = {k: p[k].grad.detach().cpu() for k in NAMES if k != "shs"}
    return out, (lo, hi, None if d_shs is None else d_shs.detach().cpu())


def _close_sharded(out, shard, ref):
    for k in NAMES:
        if k == "shs":
            continue
        scale = ref[k].abs().max().item()
        assert torch.allclose(out[k], ref[k], atol=5e-5 * scale), (k, (out[k] - ref[k]).abs().max().item() / scale)
    lo, hi, d_shs = shard
    if hi > lo:
        scale = ref["shs"].abs().max().item()
        assert torch.allclose(d_shs, ref["shs"][lo:hi], atol=5e-5 * scale), (d_shs - ref["shs"][lo:hi]).abs().max().item() / scale


def test_sharded_step_single_rank_equals_plain_multiview(hip_device):
    out, shard = _sharded_grads(hip_device, 0, 1)
    assert shard[0] == 0 and shard[1] == N
    _close_sharded(out, shard, _reference_grads(hip_device))


def _worker_sharded(rank, world, port, q, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")  # both ranks share the single GPU of the test box; gloo moves the tensors
    torch.cuda.set_device(dev)
    out, (lo, hi, d_shs) = _sharded_grads(dev, rank, world, n)
    q.put((rank, {k: v.numpy().copy() for k, v in out.items()}, lo, hi, None if d_shs is None else d_shs.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [N, N - 1])   # N - 1 is odd: the last shard is one row short (padding path)
def test_sharded_step_two_ranks_equals_plain_multiview(hip_device, n):
    ref = _reference_grads(hip_device, n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q, n)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    covered = 0
    for rank, g, lo, hi, d_shs in outs:
        _close_sharded({k: torch.from_numpy(v) for k, v in g.items()}, (lo, hi, None if d_shs is None else torch.from_numpy(d_shs)), ref)
        covered += hi - lo
    assert covered == n   # the shards tile the splats
