"""SH -> RGB as a stand-alone differentiable stage (include/splatraster.h: sr_sh_forward / sr_sh_backward).

Semantics are those of the rasterizer's own SH evaluation (reference
utils/sh_utils.py:57-112 and extract_geo.py:40-44): ``rasterizer(shs=shs)`` and
``rasterizer(colors_precomp=sh_to_rgb(means3D, shs, campos, degree))`` produce the same image and the same gradients.
Splitting the stage out is what lets the view-parallel step exchange 12-byte colour gradients instead of 192-byte SH
gradients (splatfields_amd/view_parallel.py)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def sh_forward(means3D: torch.Tensor, shs: torch.Tensor, campos: torch.Tensor, sh_degree: int):
    """-> (colors[N,3] f32, clamped[N] u8).  No autograd."""
    lib = _lib.load()
    if not means3D.is_cuda:
        raise RuntimeError("sh_forward has no CPU path: tensors must be on a HIP ('cuda') device")
    dev = means3D.device
    m = means3D.detach().to(torch.float32).contiguous()
    s = shs.detach().to(torch.float32).contiguous()
    c = campos.detach().to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
    n, k = m.shape[0], s.shape[1]
    colors = torch.empty(n, 3, dtype=torch.float32, device=dev)
    clamped = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.sr_sh_forward(n, k, int(sh_degree), _p(m), _p(s), _p(c), _p(colors), _p(clamped), _stream(dev)))
    return colors, clamped


def sh_forward_views(means3D: torch.Tensor, shs: torch.Tensor, campos_views: torch.Tensor, sh_degree: int, want_keep: bool = True,
                     out: torch.Tensor = None):
    """Colours of the same splats for V cameras in one pass: -> (colors [V,N,3], keep [V,N,3] or None); ``keep`` is 1 where
    a channel was not clamped to 0 and 0 where it was (multiply the colour gradient by it before ``sh_backward``).
    ``out``: optional contiguous float32 [V,N,3] tensor to write the colours into (e.g. a communication buffer)."""
    lib = _lib.load()
    if not means3D.is_cuda:
        raise RuntimeError("sh_forward_views has no CPU path: tensors must be on a HIP ('cuda') device")
    dev = means3D.device
    m = means3D.detach().to(torch.float32).contiguous()
    s = shs.detach().to(torch.float32).contiguous()
    cp = campos_views.detach().to(device=dev, dtype=torch.float32).contiguous().reshape(-1, 3)
    n, k, v = m.shape[0], s.shape[1], cp.shape[0]
    if out is not None:
        if out.dtype is not torch.float32 or not out.is_contiguous() or tuple(out.shape) != (v, n, 3) or out.device != dev:
            raise RuntimeError("out must be a contiguous float32 [V,N,3] tensor on the splats' device")
        colors = out
    else:
        colors = torch.empty(v, n, 3, dtype=torch.float32, device=dev)
    keep = torch.empty(v, n, 3, dtype=torch.float32, device=dev) if want_keep else None
    with torch.cuda.device(dev):
        _lib.check(lib.sr_sh_forward_views(n, k, int(sh_degree), v, _p(m), _p(s), _p(cp), _p(colors), _p(keep), _stream(dev)))
    return colors, keep


def mask_clamped(dcolors: torch.Tensor, clamped: torch.Tensor) -> torch.Tensor:
    """Zeroes the colour gradient of channels that were clamped to 0 in the forward (as the rasterizer's backward does)."""
    bits = torch.tensor([1, 2, 4], dtype=torch.uint8, device=clamped.device)
    return dcolors * ((clamped[:, None] & bits) == 0).to(dcolors.dtype)


def sh_backward(means3D, shs, campos_views, dcolors_views, sh_degree: int, *, scale: float = 1.0, want_shs=True,
                means_grad: torch.Tensor = None, accumulate_means: bool = True, out: torch.Tensor = None):
    """campos_views [V,3]; dcolors_views [V,N,3] (already masked).  Returns dL/dshs [N,K,3] (or None); adds the gradient
    through the view directions into ``means_grad`` when given.  ``out``: write dL/dshs into this contiguous float32 [N,K,3]
    tensor (e.g. a row range of a larger gradient) instead of allocating it."""
    lib = _lib.load()
    dev = means3D.device
    m = means3D.detach().to(torch.float32).contiguous()
    s = shs.detach().to(torch.float32).contiguous()
    cp = campos_views.detach().to(device=dev, dtype=torch.float32).contiguous().reshape(-1, 3)
    dc = dcolors_views.detach().to(torch.float32).contiguous().reshape(cp.shape[0], m.shape[0], 3)
    n, k, v = m.shape[0], s.shape[1], cp.shape[0]
    d_shs = (out if out is not None else torch.empty_like(s)) if want_shs else None
    if out is not None and not (out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == tuple(s.shape)):
        raise RuntimeError("out must be a contiguous float32 tensor of the shape of shs")
    if means_grad is not None and not (means_grad.is_contiguous() and means_grad.dtype == torch.float32):
        raise RuntimeError("means_grad must be a contiguous float32 tensor")
    with torch.cuda.device(dev):
        _lib.check(lib.sr_sh_backward(n, k, int(sh_degree), v, _p(m), _p(s), _p(cp), _p(dc), float(scale), _p(d_shs),
                                      _p(means_grad), int(bool(accumulate_means)), _stream(dev)))
    return d_shs


class _ShToRgb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, shs, campos, sh_degree):
        colors, clamped = sh_forward(means3D, shs, campos, sh_degree)
        ctx.save_for_backward(means3D, shs, campos, clamped)
        ctx.sh_degree = int(sh_degree)
        return colors

    @staticmethod
    def backward(ctx, grad_colors):
        means3D, shs, campos, clamped = ctx.saved_tensors
        dcol = mask_clamped(grad_colors.to(torch.float32).contiguous(), clamped)
        d_means = torch.zeros(means3D.shape[0], 3, dtype=torch.float32, device=means3D.device)
        d_shs = sh_backward(means3D, shs, campos.reshape(1, 3), dcol[None], ctx.sh_degree, means_grad=d_means,
                            accumulate_means=False)
        return d_means, d_shs, None, None


def sh_to_rgb(means3D, shs, campos, sh_degree: int) -> torch.Tensor:
    """Differentiable SH -> RGB: max(SH(dir) + 0.5, 0) with dir = normalize(means3D - campos)."""
    return _ShToRgb.apply(means3D, shs, campos, sh_degree)
