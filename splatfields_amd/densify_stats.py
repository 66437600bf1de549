"""Per-view densification bookkeeping: the consumers of the rasterizer's ``radii`` and ``viewspace_points.grad``
(reference train.py:280-286 and scene/gaussian_model.py:427-438) as ONE kernel.

The reference writes

    gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])
    gaussians.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
    gaussians.denom[update_filter] += 1

with ``visibility_filter = update_filter = radii > 0``: boolean-mask indexing, i.e. ``nonzero`` (a host synchronisation)
plus about ten small kernels per iteration.  ``densification_stats`` does the same updates in place with one launch of
``sr_densification_stats`` and no synchronisation."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor, xyz_gradient_accum: Optional[torch.Tensor],
                        denom: Optional[torch.Tensor], max_radii2D: Optional[torch.Tensor]) -> None:
    """In-place update of ``xyz_gradient_accum`` [N,1] / ``denom`` [N,1] / ``max_radii2D`` [N] (float32, contiguous; any of
    them may be None) for the splats with ``radii > 0``, from ``viewspace_grad`` [N,3] (``viewspace_points.grad``) and
    ``radii`` [N] int32 of one rendered view."""
    lib = _lib.load()
    if not viewspace_grad.is_cuda:
        raise RuntimeError("densification_stats has no CPU path: tensors must be on a HIP ('cuda') device")
    dev = viewspace_grad.device
    n = viewspace_grad.shape[0]
    g = viewspace_grad.detach()
    if g.dtype is not torch.float32 or not g.is_contiguous():
        g = g.to(torch.float32).contiguous()
    r = radii if (radii.dtype is torch.int32 and radii.is_contiguous()) else radii.to(torch.int32).contiguous()
    if g.dim() != 2 or g.shape[1] != 3 or r.numel() != n:
        raise RuntimeError("viewspace_grad must be [N,3] and radii [N]")
    for name, t in (("xyz_gradient_accum", xyz_gradient_accum), ("denom", denom), ("max_radii2D", max_radii2D)):
        if t is not None and (t.dtype is not torch.float32 or not t.is_contiguous() or t.numel() != n or t.device != dev):
            raise RuntimeError(f"{name} must be a contiguous float32 tensor with N elements on the same device (updated in place)")
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    with torch.cuda.device(dev):
        _lib.check(lib.sr_densification_stats(n, p(g), p(r), p(xyz_gradient_accum), p(denom), p(max_radii2D),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
