"""Tri-plane feature encoder of the deform network: the per-point half of the reference's `VarTriPlaneEncoder`
(scene/tripFields.py:383-436) on HIP kernels (csrc/triplane.hip).

The reference encoder has two halves: a plane GENERATOR per axis pair (`Tensorial2D`, :176-204: a diffusers-style
`TimeVAEDecoder` that turns a fixed 8 x 20 x 20 noise map into a 16 x 320 x 320 plane) and the LOOKUP every point does
(`forward`, :430-436: `F.grid_sample` of the three planes at the xy / yz / zx projections of the point + `cat`).  The lookup
is what the hot path of a 4-D training step runs per splat; it is built here, forward and backward, with the interface
`SplatFields` expects of its encoder (`out_dim`; `encoder(x[None]) -> [1, N, out_dim]`).  The generator stays a
caller-supplied module (`plane_source`): the reference builds it from diffusers / mmgen blocks that exist neither in its
checkout nor in this image.  Without one, the planes are a learnable parameter of the sampler itself (a decoder-free
tri-plane of the same shape), which is what the default `SplatFields()` constructs.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch
from torch import nn

from . import _lib


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _TriPlaneLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, planes: torch.Tensor, pts: torch.Tensor):
        lib = _lib.load()
        if not planes.is_cuda or not pts.is_cuda:
            raise RuntimeError("splatfields_amd.triplane has no CPU path: tensors must be on a HIP ('cuda') device")
        if planes.dim() != 4 or planes.shape[0] != 3 or planes.shape[1] % 4 != 0:
            raise ValueError("planes must be [3, C, H, W] with C a multiple of 4")
        if pts.dim() != 2 or pts.shape[1] != 3:
            raise ValueError("points must be [N, 3]")
        dev = planes.device
        _, ch, h, w = planes.shape
        n = pts.shape[0]
        p32 = planes.detach().to(torch.float32).contiguous()
        x32 = pts.detach().to(device=dev, dtype=torch.float32).contiguous()
        hwc = torch.empty(3, h, w, ch, dtype=torch.float32, device=dev)
        out = torch.empty(n, 3 * ch, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_triplane_forward(n, ch, h, w, _ptr(p32), _ptr(hwc), _ptr(x32), _ptr(out),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        ctx.save_for_backward(hwc, x32)
        ctx.shape = (ch, h, w)
        ctx.dtypes = (planes.dtype, pts.dtype)
        return out if planes.dtype == torch.float32 else out.to(planes.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        hwc, x32 = ctx.saved_tensors
        ch, h, w = ctx.shape
        dev, n = hwc.device, x32.shape[0]
        g32 = g.detach().to(torch.float32).contiguous()
        need_planes, need_pts = ctx.needs_input_grad
        d_planes = torch.empty(3, ch, h, w, dtype=torch.float32, device=dev) if need_planes else None
        d_pts = torch.empty(n, 3, dtype=torch.float32, device=dev) if need_pts else None
        fixed = torch.empty(lib.sr_triplane_backward_workspace(n, ch, h, w), dtype=torch.uint8, device=dev) if need_planes else None
        with torch.cuda.device(dev):
            _lib.check(lib.sr_triplane_backward(n, ch, h, w, _ptr(hwc), _ptr(x32), _ptr(g32), _ptr(d_planes), _ptr(d_pts), _ptr(fixed),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if d_planes is not None and ctx.dtypes[0] != torch.float32:
            d_planes = d_planes.to(ctx.dtypes[0])
        if d_pts is not None and ctx.dtypes[1] != torch.float32:
            d_pts = d_pts.to(ctx.dtypes[1])
        return d_planes, d_pts


def triplane_lookup(planes: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    """planes [3, C, H, W], pts [N, 3] -> [N, 3 C]: what reference scene/tripFields.py:432-435 computes with
    `F.grid_sample(planes, stack([pts[..., (0,1)], pts[..., (1,2)], pts[..., (2,0)]]))` and fuse mode 'cat'.
    Differentiable with respect to both arguments; gradients are bit-reproducible."""
    return _TriPlaneLookup.apply(planes, pts)


class TriPlaneSampler(nn.Module):
    """Encoder module with the interface of the reference's `VarTriPlaneEncoder` as `SplatFields` uses it
    (utils/time_utils.py:313-334, :450): `out_dim`, `n_planes`, `axis`, `forward(input_pts[B, N, 3], alpha_ratio=1.0,
    input_time=None, frame_id=None) -> [B, N, out_dim]`.

    plane_source: module or callable `frame_id -> [3, C, H, W]` (e.g. the reference's three `Tensorial2D` generators behind
        `get_planes`), registered as a sub-module when it is one so that its parameters train;
    otherwise `planes` [3, out_ch, resolution, resolution] is a learnable parameter (initialised like the reference's
        noise map, N(0, 1), scaled by `init_scale`)."""

    def __init__(self, out_ch: int = 16, resolution: int = 320, fuse_mode: str = "cat", plane_source: Optional[Callable] = None,
                 init_scale: float = 0.1):
        super().__init__()
        if fuse_mode not in ("cat", "add", "mean"):
            raise NotImplementedError(fuse_mode)
        self.img_channels, self.fuse_mode = int(out_ch), fuse_mode
        self.space_axis = [[0, 1], [1, 2], [2, 0]]   # xy, yz, zx (reference scene/tripFields.py:399)
        if plane_source is not None:
            self.plane_source = plane_source           # nn.Module instances register themselves
        else:
            self.plane_source = None
            self.planes = nn.Parameter(init_scale * torch.randn(3, out_ch, resolution, resolution))

    @property
    def axis(self):
        return self.space_axis

    @property
    def n_planes(self) -> int:
        return 3

    @property
    def out_dim(self) -> int:
        return self.n_planes * self.img_channels if self.fuse_mode == "cat" else self.img_channels

    def get_planes(self, frame_id=None) -> torch.Tensor:
        if self.plane_source is None:
            return self.planes
        src = self.plane_source
        planes = src.get_planes(frame_id=frame_id) if hasattr(src, "get_planes") else src(frame_id)
        if planes.dim() != 4 or planes.shape[0] != 3 or planes.shape[1] != self.img_channels:
            raise ValueError(f"plane_source must return [3, {self.img_channels}, H, W], got {tuple(planes.shape)}")
        return planes

    def forward(self, input_pts: torch.Tensor, alpha_ratio: float = 1.0, input_time=None, frame_id=None) -> torch.Tensor:
        planes = self.get_planes(frame_id)
        lead = input_pts.shape[:-1]
        feat = triplane_lookup(planes, input_pts.reshape(-1, 3))                       # [B N, 3 C], plane-major
        if self.fuse_mode != "cat":
            feat = feat.view(-1, 3, self.img_channels).sum(dim=1)                      # the reference sums for 'add' AND 'mean' (:424-425)
        return feat.view(*lead, feat.shape[-1])
