"""The RCCL calls of the view-parallel steps, executed on ONE MI355X: a 1-rank "nccl" process group with
view_parallel.FORCE_COLLECTIVES makes every collective the multi-GPU steps issue (ReduceOp.AVG all-reduce in place and
packed, all_gather_into_tensor, device all_to_all_single, packed SUM all-reduce) run through RCCL, and the results must
equal the plain single-process multi-view step.  (The 2-rank equality tests run over gloo: tests/test_view_parallel_gloo.py,
tests/test_gpu_sh_gather.py; a multi-GPU node is only available to the round-end driver.)"""
import math
import os
import socket

import pytest
import torch
import torch.distributed as dist

from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads

pytestmark = pytest.mark.gpu
N, W, H, V, DEG = 5000, 144, 112, 2, 3
NAMES = ["means3D", "scales", "rotations", "opacities", "shs"]


@pytest.fixture(scope="module")
def rccl_group(hip_device):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(hip_device)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=hip_device)
    from splatfields_amd import view_parallel as vp
    vp.FORCE_COLLECTIVES = True
    yield
    vp.FORCE_COLLECTIVES = False
    dist.destroy_process_group()


def _reference(sp, cams, gi, gd, ga, dev):
    """single-process loop over the views, mean loss, one backward (reference train.py:169-252)"""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
    for cam in cams:
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3, device=dev), 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, DEG, cam.camera_center, False, False)
        color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
            means3D=p["means3D"], means2D=torch.zeros_like(p["means3D"]), opacities=p["opacities"], shs=p["shs"],
            scales=p["scales"], rotations=p["rotations"])
        torch.autograd.backward((color, depth, alpha), (gi / len(cams), gd / len(cams), ga / len(cams)))
    return {k: p[k].grad.clone() for k in NAMES}


def test_every_exchange_scheme_runs_on_rccl(hip_device, rccl_group):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from splatfields_amd import view_parallel as vp
    dev = hip_device
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    sp = make_splats(N, seed=11, device=dev)
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    cams = [make_camera(k, W, H, device=dev) for k in range(V)]
    ref = _reference(sp, cams, gi, gd, ga, dev)
    bg = torch.ones(3, device=dev)

    def bwd(vi, color, depth, alpha):
        torch.autograd.backward((color, depth, alpha), (gi / V, gd / V, ga / V))

    def close(a, b, name):
        scale = b.abs().max().clamp_min(1e-30)
        assert ((a - b).abs().max() / scale).item() <= 1e-5, name

    # 1. all-gather of colour gradients + packed SUM all-reduce
    p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
    vp.sh_gather_step(p, cams, bg, DEG, bwd, rank=0, world=1)
    for k in NAMES:
        close(p[k].grad, ref[k], ("gather", k))

    # 1b. one view per rank: the SLICED exchange -- per range an all-gather and ONE grouped (ncclGroupStart / End) in-place
    #     all-reduce of the four geometric gradients, issued from inside the backward
    ref1 = _reference(sp, cams[:1], gi, gd, ga, dev)
    for slices in (1, 4):
        p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
        vp.sh_gather_step(p, cams[:1], bg, DEG, lambda vi, c, d, a: torch.autograd.backward((c, d, a), (gi, gd, ga)), rank=0, world=1,
                          slices=slices)
        torch.cuda.synchronize()
        for k in NAMES:
            close(p[k].grad, ref1[k], ("gather sliced", slices, k))

    # 2. SH sharded by splat range: two device all-to-all + packed all-reduce
    p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
    lo, hi, d_shs = vp.sh_sharded_step(p, cams, bg, DEG, bwd, rank=0, world=1)
    assert (lo, hi) == (0, N)
    close(d_shs, ref["shs"], ("shard", "shs"))
    for k in NAMES[:4]:
        close(p[k].grad, ref[k], ("shard", k))

    # 3. plain scheme: ReduceOp.AVG all-reduce of every gradient (in place for the large one, packed for the small ones),
    #    including the "active SH bands only" variant
    for active in (None, 4):
        p = {k: sp[k].clone().requires_grad_(True) for k in NAMES}
        deg = DEG if active is None else 1
        for cam in cams:
            rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0, cam.world_view_transform,
                                               cam.full_proj_transform, deg, cam.camera_center, False, False)
            color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
                means3D=p["means3D"], means2D=torch.zeros_like(p["means3D"]), opacities=p["opacities"], shs=p["shs"],
                scales=p["scales"], rotations=p["rotations"])
            torch.autograd.backward((color, depth, alpha), (gi / V, gd / V, ga / V))
        before = {k: p[k].grad.clone() for k in NAMES}
        old = vp.PACK_BELOW_BYTES
        vp.PACK_BELOW_BYTES = 200_000   # shs.grad (5000 x 16 x 3 x 4 B) goes alone, the rest packed
        try:
            vp.allreduce_gradients(list(p.values()), 1, sh_param=p["shs"], sh_active_coeffs=active)
        finally:
            vp.PACK_BELOW_BYTES = old
        torch.cuda.synchronize()
        for k in NAMES:
            close(p[k].grad, before[k], ("allreduce", k, active))   # mean over one rank
        if active is None:
            for k in NAMES:
                close(p[k].grad, ref[k], ("allreduce-vs-ref", k))
        else:
            assert (p["shs"].grad[:, active:] == 0).all()   # inactive bands carry no gradient

    # 4. view_parallel_step (loss all-reduced with the packed small gradients)
    params = [sp[k].clone().requires_grad_(True) for k in NAMES]

    def render_loss(cam):
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, DEG, cam.camera_center, False, False)
        color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
            means3D=params[0], means2D=torch.zeros_like(params[0]), opacities=params[3], shs=params[4], scales=params[1],
            rotations=params[2])
        return (color * gi).sum() + (depth * gd).sum() + (alpha * ga).sum()

    loss = vp.view_parallel_step(params, cams, render_loss, rank=0, world=1)
    assert torch.isfinite(loss)
    for t, k in zip(params, NAMES):
        close(t.grad, ref[k], ("view_parallel_step", k))
