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
