"""Multi-GPU path on CPU: 2 gloo ranks shard the views of one step and all-reduce per-splat gradients;
the result must equal the single-process multi-view step of the reference (train.py:158-252)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import torch_oracle as O
from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
from splatfields_amd.view_parallel import allreduce_gradients, field_view_parallel_step, shard_views, view_parallel_step

N, W, H, VIEWS = 120, 40, 32, 3


def build():
    sp = make_splats(N, seed=8, mean_scale=0.1, dtype=torch.float64)
    names = ["means3D", "scales", "rotations", "opacities", "shs"]
    params = [sp[k].clone().requires_grad_(True) for k in names]
    gi, _, _ = make_upstream_grads(H, W, dtype=torch.float64)

    def render_loss(view):
        cam = make_camera(view, W, H)
        st = O.settings_from_camera(cam, torch.ones(3, dtype=torch.float64), 2)
        m3, sc, ro, op, sh = params
        out = O.rasterize(m3, None, op, shs=sh, scales=sc, rotations=ro, settings=st)
        return (out.color * gi).sum() * 1e3 + ((out.alpha - 0.5) ** 2).mean()

    return params, render_loss


def reference_step():
    params, render_loss = build()
    loss = sum(render_loss(v) for v in range(VIEWS)) / VIEWS  # train.py:242: mean over the views
    loss.backward()  # train.py:252
    return loss.detach(), [p.grad.clone() for p in params]


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    params, render_loss = build()
    loss = view_parallel_step(params, list(range(VIEWS)), render_loss)
    q.put((rank, loss.numpy().copy(), [p.grad.numpy().copy() for p in params]))  # plain arrays: no fd passing
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_shard_views_round_robin():
    assert shard_views(list(range(7)), 1, 3) == [1, 4]
    assert sum((shard_views(list(range(7)), r, 3) for r in range(3)), []) != []
    assert sorted(sum((shard_views(list(range(7)), r, 3) for r in range(3)), [])) == list(range(7))


def test_two_rank_step_equals_single_process_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref_loss, ref_grads = reference_step()
    results = [(r, torch.from_numpy(l), [torch.from_numpy(g) for g in gs]) for r, l, gs in results]
    for rank, loss, grads in results:
        assert torch.allclose(loss, ref_loss, rtol=1e-12, atol=1e-14), rank
        for g, r in zip(grads, ref_grads):
            assert torch.allclose(g, r, rtol=1e-10, atol=1e-16), rank
    # both ranks hold identical (all-reduced) gradients
    for a, b in zip(results[0][2], results[1][2]):
        assert torch.equal(a, b)


def test_four_rank_step_with_fewer_views_than_ranks_equals_single_process_step():
    """world_size 4 over 3 views: rank 3 renders nothing and must still join every collective with zeros; all four ranks end
    with the single-process gradients (more than two ranks, ragged work)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ref_loss, ref_grads = reference_step()
    results = sorted(((r, torch.from_numpy(l), [torch.from_numpy(g) for g in gs]) for r, l, gs in results), key=lambda x: x[0])
    assert [r for r, _, _ in results] == [0, 1, 2, 3]
    for rank, loss, grads in results:
        assert torch.allclose(loss, ref_loss, rtol=1e-12, atol=1e-14), rank
        for g, r in zip(grads, ref_grads):
            assert torch.allclose(g, r, rtol=1e-10, atol=1e-16), rank
    for other in results[1:]:
        for a, b in zip(results[0][2], other[2]):
            assert torch.equal(a, b)


def _avg_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = torch.nn.Parameter(torch.zeros(5))
    p.grad = torch.full((5,), float(rank + 1))
    allreduce_gradients([p], world)
    q.put(p.grad.numpy().copy())
    dist.destroy_process_group()


def test_allreduce_gradients_averages():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_avg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    for o in outs:
        assert torch.equal(torch.from_numpy(o), torch.full((5,), 1.5))


def _ragged_worker(rank, world, port, q):
    """rank 1 has no gradient for one parameter (it rendered no view that reached it) and only the leading SH bands carry
    gradient: every rank must still issue the same collectives (zero-filled), and only the active bands travel."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    means = torch.nn.Parameter(torch.zeros(6, 3))
    opac = torch.nn.Parameter(torch.zeros(6, 1))
    shs = torch.nn.Parameter(torch.zeros(6, 16, 3))
    means.grad = torch.full((6, 3), float(rank + 1))
    if rank == 0:
        opac.grad = torch.full((6, 1), 4.0)          # rank 1: opac.grad stays None
    shs.grad = torch.zeros(6, 16, 3)
    shs.grad[:, :4] = float(10 * (rank + 1))          # active_sh_degree = 1: bands 4..15 are exactly zero everywhere
    allreduce_gradients([means, opac, shs], world, sh_param=shs, sh_active_coeffs=4)
    q.put((means.grad.numpy().copy(), opac.grad.numpy().copy(), shs.grad.numpy().copy()))
    dist.destroy_process_group()


def test_allreduce_gradients_with_missing_grads_and_active_sh_bands():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for m, o, s in outs:
        assert torch.equal(torch.from_numpy(m), torch.full((6, 3), 1.5))
        assert torch.equal(torch.from_numpy(o), torch.full((6, 1), 2.0))          # (4 + 0) / 2
        s = torch.from_numpy(s)
        assert torch.equal(s[:, :4], torch.full((6, 4, 3), 15.0)) and (s[:, 4:] == 0).all()


# ---- neural path: the splat attributes are the output of a (replicated) network; only attribute gradients are exchanged ----

def build_field():
    """a small stand-in network (positions, time) -> attributes, and the oracle rasterizer on precomputed colours"""
    g = torch.Generator().manual_seed(3)
    sp = make_splats(N, seed=8, mean_scale=0.1, dtype=torch.float64)
    xyz = sp["means3D"].clone().requires_grad_(True)
    w1 = (0.3 * torch.randn(3, 16, generator=g, dtype=torch.float64)).requires_grad_(True)
    w2 = (0.3 * torch.randn(16, 14, generator=g, dtype=torch.float64)).requires_grad_(True)
    leaves = [xyz, w1, w2]
    gi, _, _ = make_upstream_grads(H, W, dtype=torch.float64)

    def compute_splats():
        h = torch.tanh(xyz @ w1) @ w2
        return {"means3D": xyz + 0.05 * h[:, 0:3], "scales": sp["scales"] * torch.exp(0.1 * h[:, 3:6]),
                "rotations": torch.nn.functional.normalize(sp["rotations"] + 0.1 * h[:, 6:10], dim=-1),
                "opacity": torch.sigmoid(h[:, 10:11]), "rgb": torch.sigmoid(h[:, 11:14]), "frame": 3}

    def render_loss(s, view):
        cam = make_camera(view, W, H)
        st = O.settings_from_camera(cam, torch.ones(3, dtype=torch.float64), 0)
        out = O.rasterize(s["means3D"], None, s["opacity"], colors_precomp=s["rgb"], scales=s["scales"], rotations=s["rotations"], settings=st)
        return (out.color * gi).sum() * 1e3 + ((out.alpha - 0.5) ** 2).mean()

    return leaves, compute_splats, render_loss


def field_reference_step():
    leaves, compute_splats, render_loss = build_field()
    s = compute_splats()
    loss = sum(render_loss(s, v) for v in range(VIEWS)) / VIEWS
    loss.backward()
    return loss.detach(), [p.grad.clone() for p in leaves]


def field_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    leaves, compute_splats, render_loss = build_field()
    loss = field_view_parallel_step(compute_splats, list(range(VIEWS)), render_loss)
    q.put((rank, loss.numpy().copy(), [p.grad.numpy().copy() for p in leaves]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_neural_step_equals_single_process_step():
    """configs[4]'s N > 1 leg: views sharded, network replicated, one packed all-reduce of the attribute gradients, then the
    network backward on every rank: network and position gradients equal the single-process loop and each other."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=field_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref_loss, ref_grads = field_reference_step()
    results = sorted((r, torch.from_numpy(l), [torch.from_numpy(g) for g in gs]) for r, l, gs in results)
    for rank, loss, grads in results:
        assert torch.allclose(loss, ref_loss, rtol=1e-12, atol=1e-14), rank
        for g, r in zip(grads, ref_grads):
            assert torch.allclose(g, r, rtol=1e-9, atol=1e-15), rank
    for a, b in zip(results[0][2], results[1][2]):
        assert torch.equal(a, b)


def test_neural_step_without_a_process_group_is_the_plain_step():
    leaves, compute_splats, render_loss = build_field()
    loss = field_view_parallel_step(compute_splats, list(range(VIEWS)), render_loss, rank=0, world=1)
    ref_loss, ref_grads = field_reference_step()
    assert torch.allclose(loss, ref_loss, rtol=1e-12)
    for p, r in zip(leaves, ref_grads):
        assert torch.allclose(p.grad, r, rtol=1e-10, atol=1e-16)
