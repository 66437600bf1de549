"""ctypes wrapper + build recipe of oracle/raster_ref.c (CPU oracle #2).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
SRC = HERE / "raster_ref.c"
LIB = HERE / "_build" / "libraster_ref.so"        # working precision float: the published pipeline's arithmetic
LIB64 = HERE / "_build" / "libraster_ref64.so"    # the same statements in double (-DREF_REAL=double): rounding-free reference


def build(force: bool = False) -> Path:
    for lib, flags in ((LIB, []), (LIB64, ["-DREF_REAL=double"])):
        if force or not lib.exists() or lib.stat().st_mtime < SRC.stat().st_mtime:
            lib.parent.mkdir(exist_ok=True)
            subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", *flags, str(SRC), "-o", str(lib), "-lm"], check=True)
    return LIB


class RefView(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int), ("sh_coeffs", C.c_int),
                ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16), ("campos", C.c_float * 3),
                ("bg", C.c_float * 3)]


_libs = {}


def _load(precision: str = "fp32"):
    if precision not in _libs:
        build()
        lib = C.CDLL(str({"fp32": LIB, "fp64": LIB64}[precision]))
        lib.ref_rasterize.restype = C.c_int
        lib.ref_rasterize_ex.restype = C.c_int
        _libs[precision] = lib
    return _libs[precision]


def _np(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().to(torch.float32).numpy())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rasterize(inputs: dict, settings, *, use_sh=True, g_img=None, g_depth=None, g_alpha=None, tile_window=None,
              threads: int = 1, precision: str = "fp32", fragile: bool = False, margin: float = 2e-4, xy_ulps: float = 2.0):
    """Forward (and backward when g_img is given).  `settings` is an OracleSettings-like tuple.
    Returns (out dict of torch tensors, grads dict or None, num_rendered).
    precision: "fp32" (the published arithmetic) or "fp64" (the same statements in double).
    fragile=True adds out["fragile"] ([H,W] bool: a threshold decision of the pixel sits within the fp32 margin, the
    rule of torch_oracle.py:288-293 plus near-equal depths) and out["splat_flag"] ([N] bool: the splat is blended,
    or nearly blended, into a fragile pixel -- its gradient sums contain that pixel), out["cond_bound"] ([5,H,W]: r, g, b,
    depth, alpha -- first-order bound on what `xy_ulps` float ulps of rounding in the splats' stored screen-space centres do
    to the pixel, raster_ref.c) and out["radius_raw"] ([N]: 3 sqrt(lambda_max) before the ceil)."""
    lib = _load(precision)
    H, W = int(settings.image_height), int(settings.image_width)
    means = _np(inputs["means3D"]); n = means.shape[0]
    opac = _np(inputs["opacities"]).reshape(-1)
    scales, rots = _np(inputs["scales"]), _np(inputs["rotations"])
    shs = _np(inputs["shs"]) if use_sh else None
    cols = None if use_sh else _np(inputs["colors_precomp"])
    K = shs.shape[1] if shs is not None else 0
    v = RefView()
    v.H, v.W, v.tanfovx, v.tanfovy = H, W, settings.tanfovx, settings.tanfovy
    v.scale_modifier, v.sh_degree, v.sh_coeffs = settings.scale_modifier, int(settings.sh_degree), K
    v.viewmatrix[:] = _np(settings.viewmatrix).reshape(-1).tolist()
    v.projmatrix[:] = _np(settings.projmatrix).reshape(-1).tolist()
    v.campos[:] = _np(settings.campos).reshape(-1).tolist()
    v.bg[:] = _np(settings.bg).reshape(-1).tolist()
    color = np.zeros((3, H, W), np.float32); depth = np.zeros((1, H, W), np.float32); alpha = np.zeros((1, H, W), np.float32)
    radii = np.zeros(n, np.int32)
    nr = C.c_longlong(0)
    gi, gd, ga = _np(g_img), _np(g_depth), _np(g_alpha)
    bwd = gi is not None
    d = {}
    if bwd:
        d = dict(means3D=np.zeros((n, 3), np.float32), means2D=np.zeros((n, 3), np.float32),
                 opacities=np.zeros((n, 1), np.float32), scales=np.zeros((n, 3), np.float32),
                 rotations=np.zeros((n, 4), np.float32))
        if use_sh:
            d["shs"] = np.zeros((n, K, 3), np.float32)
        else:
            d["colors_precomp"] = np.zeros((n, 3), np.float32)
    tw = tile_window or (0, 0, 0, 0)
    common = (C.byref(v), C.c_int(n), _p(means), _p(opac), _p(scales), _p(rots), _p(shs), _p(cols),
              _p(color), _p(depth), _p(alpha), _p(radii), C.byref(nr),
              _p(gi), _p(gd), _p(ga),
              _p(d.get("means3D")), _p(d.get("means2D")), _p(d.get("opacities")), _p(d.get("scales")),
              _p(d.get("rotations")), _p(d.get("shs")), _p(d.get("colors_precomp")),
              C.c_int(tw[0]), C.c_int(tw[1]), C.c_int(tw[2]), C.c_int(tw[3]), C.c_int(threads))
    frag = flag = None
    if fragile:
        frag = np.zeros((H, W), np.uint8); flag = np.zeros(max(n, 1), np.uint8)
        bound = np.zeros((5, H, W), np.float32); rraw = np.zeros(max(n, 1), np.float32)
        rc = lib.ref_rasterize_ex(*common, _p(frag), _p(flag), C.c_float(margin), _p(bound), C.c_float(xy_ulps), _p(rraw))
    else:
        rc = lib.ref_rasterize(*common)
    assert rc == 0
    out = dict(color=torch.from_numpy(color), depth=torch.from_numpy(depth), alpha=torch.from_numpy(alpha),
               radii=torch.from_numpy(radii))
    if fragile:
        out["fragile"] = torch.from_numpy(frag).bool()
        out["splat_flag"] = torch.from_numpy(flag[:n]).bool()
        out["cond_bound"] = torch.from_numpy(bound)
        out["radius_raw"] = torch.from_numpy(rraw[:n])
    grads = {k: torch.from_numpy(a) for k, a in d.items()} if bwd else None
    return out, grads, int(nr.value)
