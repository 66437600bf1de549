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
LIB = HERE / "_build" / "libraster_ref.so"


def build(force: bool = False) -> Path:
    if force or not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
        LIB.parent.mkdir(exist_ok=True)
        subprocess.run(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", str(SRC), "-o", str(LIB), "-lm"], check=True)
    return LIB


class RefView(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int), ("sh_coeffs", C.c_int),
                ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16), ("campos", C.c_float * 3),
                ("bg", C.c_float * 3)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(build()))
        _lib.ref_rasterize.restype = C.c_int
    return _lib


def _np(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().to(torch.float32).numpy())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def rasterize(inputs: dict, settings, *, use_sh=True, g_img=None, g_depth=None, g_alpha=None, tile_window=None,
              threads: int = 1):
    """Forward (and backward when g_img is given).  `settings` is an OracleSettings-like tuple.
    Returns (out dict of torch tensors, grads dict or None, num_rendered)."""
    lib = _load()
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
    rc = lib.ref_rasterize(C.byref(v), C.c_int(n), _p(means), _p(opac), _p(scales), _p(rots), _p(shs), _p(cols),
                           _p(color), _p(depth), _p(alpha), _p(radii), C.byref(nr),
                           _p(gi), _p(gd), _p(ga),
                           _p(d.get("means3D")), _p(d.get("means2D")), _p(d.get("opacities")), _p(d.get("scales")),
                           _p(d.get("rotations")), _p(d.get("shs")), _p(d.get("colors_precomp")),
                           C.c_int(tw[0]), C.c_int(tw[1]), C.c_int(tw[2]), C.c_int(tw[3]), C.c_int(threads))
    assert rc == 0
    out = dict(color=torch.from_numpy(color), depth=torch.from_numpy(depth), alpha=torch.from_numpy(alpha),
               radii=torch.from_numpy(radii))
    grads = {k: torch.from_numpy(a) for k, a in d.items()} if bwd else None
    return out, grads, int(nr.value)
