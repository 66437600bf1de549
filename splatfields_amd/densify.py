"""Densification / pruning of the splat set on the device.

`densify_and_prune_tensors` is the functional core (tensors in, tensors out); `densify_and_prune` applies it to an object
with the attributes of the reference's ``GaussianModel`` (reference scene/gaussian_model.py:27-61: ``_xyz``,
``_features_dc``, ``_features_rest``, ``_opacity``, ``_scaling``, ``_rotation``, ``xyz_gradient_accum``, ``denom``,
``max_radii2D``, ``percent_dense``, ``optimizer`` with one named parameter group per tensor) and performs the same
optimizer surgery as the reference (:272-353), in ONE pass instead of three rebuilds of every tensor and moment.

Semantics restated (file:line of the reference):
  * grads = xyz_gradient_accum / denom, NaN -> 0                                             (:412-413)
  * clone  where |grads| >= max_grad and max(scale) <= percent_dense * extent                (:394-409)
  * split  where  grads  >= max_grad and max(scale) >  percent_dense * extent, N = 2 children with
           xyz = build_rotation(q) @ normal(0, scale) + xyz, scale / 1.6, everything else copied; the parent is removed (:355-380)
  * prune  opacity < min_opacity, and -- if max_screen_size -- max_radii2D > max_screen_size or max(scale) > 0.1 extent (:418-423)
    (`densification_postfix` :349-353 has zeroed max_radii2D by then, so the screen-size test of the reference never fires;
    `screen_test_on_accumulated_radii=True` tests the radii accumulated before the densification instead)
  * statistics are reset for every row, Adam moments of new rows are zero                    (:317-320, :349-353)
Row order of the result = the reference's: surviving originals, clones, first children, second children.
The normal samples come from ``unit_normals`` [2, N, 3] (row k of splat i feeds its k-th child), default ``torch.randn``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib

PARAM_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")   # the reference's optimizer group names


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def densify_and_prune_tensors(params: Dict[str, torch.Tensor], moments: Optional[Dict[str, tuple]], grad_accum: torch.Tensor,
                              denom: torch.Tensor, max_radii2D: Optional[torch.Tensor], max_grad: float, min_opacity: float,
                              extent: float, max_screen_size, percent_dense: float = 0.01,
                              unit_normals: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                              screen_test_on_accumulated_radii: bool = False):
    """params: name -> tensor for PARAM_NAMES (raw parameters, any trailing shape, N rows each); moments: name ->
    (exp_avg, exp_avg_sq) or None.  Returns (new_params, new_moments, counts) with counts = dict(kept, clones, children, total)."""
    lib = _lib.load()
    xyz = params["xyz"]
    if not xyz.is_cuda:
        raise RuntimeError("splatfields_amd.densify has no CPU path: tensors must be on a HIP ('cuda') device")
    dev, n = xyz.device, xyz.shape[0]
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    src = {k: f32(params[k]) for k in PARAM_NAMES}
    scale_cols = src["scaling"].reshape(n, -1).shape[1]
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if unit_normals is None:
            unit_normals = torch.randn(2, n, 3, device=dev, dtype=torch.float32, generator=generator)
        unit_normals = f32(unit_normals)
        if tuple(unit_normals.shape) != (2, n, 3):
            raise ValueError("unit_normals must be [2, N, 3]")
        ws = torch.empty(lib.sr_densify_workspace_bytes(n), dtype=torch.uint8, device=dev)
        dest = torch.empty(4, max(n, 1), dtype=torch.int32, device=dev)
        counts = (C.c_longlong * 5)()
        radii = f32(max_radii2D).reshape(-1) if (screen_test_on_accumulated_radii and max_radii2D is not None) else None
        _lib.check(lib.sr_densify_plan(n, _ptr(src["scaling"]), scale_cols, _ptr(src["opacity"]), _ptr(f32(grad_accum).reshape(-1)),
                                       _ptr(f32(denom).reshape(-1)), _ptr(radii), float(max_grad), float(min_opacity), float(extent),
                                       float(percent_dense), float(max_screen_size or 0.0), _ptr(ws), _ptr(dest), counts, stream))
        kept, clones, c1, c2, total = [int(c) for c in counts]

        def gather(t, mode):
            row = t[0].numel() if n > 0 else int(torch.tensor(t.shape[1:]).prod().item()) if t.dim() > 1 else 1
            out = torch.empty((total,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
            _lib.check(lib.sr_densify_gather(n, row, _ptr(t), _ptr(out), _ptr(dest), mode, _ptr(src["scaling"]), scale_cols,
                                             _ptr(src["rotation"]), _ptr(unit_normals), stream))
            return out

        mode_of = {"xyz": 2, "scaling": 3}
        new_params = {k: gather(src[k], mode_of.get(k, 0)).to(params[k].dtype) for k in PARAM_NAMES}
        new_moments = None
        if moments is not None:
            new_moments = {}
            for k in PARAM_NAMES:
                if moments.get(k) is None:
                    new_moments[k] = None
                else:
                    new_moments[k] = tuple(gather(f32(m), 1).to(m.dtype) for m in moments[k])
    return new_params, new_moments, dict(kept=kept, clones=clones, children=c1 + c2, total=total)


def densify_and_prune(gaussians, max_grad: float, min_opacity: float, extent: float, max_screen_size, *,
                      unit_normals: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                      screen_test_on_accumulated_radii: bool = False) -> dict:
    """Drop-in for ``GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size)`` (reference
    scene/gaussian_model.py:411-425, called from train.py:287-289): rebuilds the six parameters, both Adam moments of each
    and the three statistics tensors of `gaussians` in place."""
    from torch import nn
    attr = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
            "rotation": "_rotation"}
    opt = gaussians.optimizer
    groups = {g["name"]: g for g in opt.param_groups if g.get("name") in attr}
    params = {k: groups[k]["params"][0] for k in attr}
    moments = {}
    for k, p in params.items():
        st = opt.state.get(p, None)
        moments[k] = (st["exp_avg"], st["exp_avg_sq"]) if st else None
    new_params, new_moments, counts = densify_and_prune_tensors(
        {k: p.detach() for k, p in params.items()}, moments, gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D,
        max_grad, min_opacity, extent, max_screen_size, getattr(gaussians, "percent_dense", 0.01), unit_normals, generator,
        screen_test_on_accumulated_radii)
    for k, g in groups.items():
        old = g["params"][0]
        st = opt.state.pop(old, None)
        newp = nn.Parameter(new_params[k].requires_grad_(True))
        g["params"][0] = newp
        if st is not None:
            st["exp_avg"], st["exp_avg_sq"] = new_moments[k]
            opt.state[newp] = st
        setattr(gaussians, attr[k], newp)
    m, dev = counts["total"], new_params["xyz"].device
    gaussians.xyz_gradient_accum = torch.zeros((m, 1), device=dev)
    gaussians.denom = torch.zeros((m, 1), device=dev)
    gaussians.max_radii2D = torch.zeros((m,), device=dev)
    return counts
