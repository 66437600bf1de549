"""View-parallel training step across the GPUs of one node (new design; the reference is single-GPU).

The reference renders the views of one iteration in a Python loop over the *same* splat tensors and
averages their losses before one backward (train.py:158-169, :242, :252).  Views are independent, so
they shard: one process per GPU, splat tensors replicated, rank r renders views r, r+G, r+2G, ...;
each rank back-propagates its local mean loss, then ONE sum all-reduce per gradient tensor over
RCCL/xGMI (`torch.distributed`, backend "nccl") followed by a 1/G scale reproduces the gradient of
the mean over all views.  No packing copy: each dense gradient tensor is reduced in place (the SH
gradient, 192 B/splat, dominates the payload: 236 B/splat with SH, 56 B/splat with precomputed
colours -- SURVEY.md §8e).

Densification statistics (`viewspace_points.grad`, `radii`) are per view; this harness keeps the
reference's "last view wins" rule (train.py:178, :282, :307) on every rank for its own last view.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Sequence

import torch
import torch.distributed as dist


def shard_views(views: Sequence, rank: int, world: int) -> list:
    """Round-robin deal of this step's view list (reference train.py:158-163 builds the list)."""
    return [v for i, v in enumerate(views) if i % world == rank]


def allreduce_gradients(params: Iterable[torch.Tensor], world: int, group=None) -> None:
    """In-place sum all-reduce of every ``.grad`` followed by the 1/world scale.
    Collectives are issued asynchronously back to back (largest tensor first, so its ring starts
    while the small ones are queued) and waited together."""
    if world <= 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    grads.sort(key=lambda g: -g.numel())
    # RCCL averages in the collective itself (no extra pass over 236 B/splat); gloo (CPU tests) has no AVG
    avg = dist.get_backend(group) == "nccl"
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    works = [dist.all_reduce(g, op=op, group=group, async_op=True) for g in grads]
    for w in works:
        w.wait()
    if not avg:
        scale = 1.0 / world
        for g in grads:
            g.mul_(scale)


def sh_gather_step(params: dict, cams: Sequence, bg, sh_degree: int, backward_fn: Callable, *, scaling_modifier: float = 1.0,
                   rank: int = None, world: int = None, group=None) -> None:
    """View-parallel step for the SH colour path with the low-rank gradient exchange.

    ``params``: dict of leaf tensors ``means3D, scales, rotations, opacities, shs`` (replicated on every rank).
    ``cams``: the V cameras of this step (same list on every rank; rank r renders cams[r::world]).
    ``backward_fn(view_index, color, depth, alpha)`` must back-propagate the loss of that view *already divided by V*
    (e.g. ``(loss / V).backward()`` or ``torch.autograd.backward(outputs, upstream_grads / V)``).

    Afterwards every rank holds in ``p.grad`` the gradient of the mean loss over all V views -- the same result as
    all-reducing all five gradient tensors, but the 192 B/splat SH gradient never crosses xGMI: for one view it is
    basis(view direction) (x) dL/dcolour, so ranks all-gather the 12 B/splat colour gradients of all views and rebuild
    the sum locally (sr_sh_backward).  Wire traffic per rank drops from ~413 to ~161 bytes per splat."""
    import math
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from . import sh as shmod
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    V = len(cams)
    if V % world != 0:
        raise ValueError("the number of views must be a multiple of the number of ranks")
    names = ["means3D", "scales", "rotations", "opacities"]
    for p in params.values():
        p.grad = None
    means3D, shs = params["means3D"], params["shs"]
    dev = means3D.device
    n = means3D.shape[0]
    mine = list(range(rank, V, world))
    dcol_views = []
    for slot, vi in enumerate(mine):
        cam = cams[vi]
        rs = GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
        sink = []
        color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
            means3D=means3D, means2D=torch.zeros_like(means3D, requires_grad=True), opacities=params["opacities"],
            shs=shs, scales=params["scales"], rotations=params["rotations"], color_grad_sink=sink)
        # the rasterizer's backward accumulates the 4 small gradients (incl. the view-direction term in means3D) and hands
        # over the clamp-masked colour gradient instead of writing 192 B/splat of SH gradient
        backward_fn(vi, color, depth, alpha)
        dcol_views.append(sink.pop())
    for k in names:
        if params[k].grad is None:
            params[k].grad = torch.zeros_like(params[k])
    dcol_local = dcol_views[0][None] if len(dcol_views) == 1 else torch.stack(dcol_views)
    campos_all = torch.stack([c.camera_center.to(device=dev, dtype=torch.float32).reshape(3) for c in cams])
    if world > 1:
        gathered = torch.empty(world, len(mine), n, 3, dtype=torch.float32, device=dev)
        works = [dist.all_gather_into_tensor(gathered.view(-1), dcol_local.view(-1), group=group, async_op=True)]
        works += [dist.all_reduce(params[k].grad, op=dist.ReduceOp.SUM, group=group, async_op=True)
                  for k in sorted(names, key=lambda q: -params[q].numel())]
        for w in works:
            w.wait()
        # gathered[r, slot] is view r + slot*world
        order = [r + sl * world for r in range(world) for sl in range(len(mine))]
        dcol_all = gathered.reshape(world * len(mine), n, 3)
        campos_used = campos_all if order == list(range(V)) else campos_all[torch.tensor(order, device=dev)]
    else:
        dcol_all, campos_used = dcol_local, campos_all
    params["shs"].grad = shmod.sh_backward(means3D, shs, campos_used, dcol_all, sh_degree, want_shs=True)


def view_parallel_step(params: List[torch.Tensor], views: Sequence, render_loss: Callable, *, rank: int = None,
                       world: int = None, group=None) -> torch.Tensor:
    """One data-parallel step.  ``render_loss(view) -> scalar loss`` renders one view with the shared
    parameters.  After the call every rank holds, in ``p.grad``, the gradient of
    ``mean_{v in views} render_loss(v)`` -- exactly what the single-process loop of the reference
    computes -- and the returned tensor is that mean loss (all-reduced)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    for p in params:
        p.grad = None
    mine = shard_views(views, rank, world)
    total = None
    for v in mine:
        l = render_loss(v)
        total = l if total is None else total + l
    n_views = len(views)
    if total is not None:
        # local contribution to the global mean; summed (not averaged) across ranks below
        (total / n_views).backward()
        local = (total / n_views).detach()
    else:
        local = torch.zeros((), device=params[0].device, dtype=params[0].dtype)
    if world > 1:
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        works = [dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True)
                 for p in sorted(params, key=lambda q: -q.numel())]
        local = local.clone()
        works.append(dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()
    return local
