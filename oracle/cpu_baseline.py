"""bench.py's `cpu_baseline` leg: the C oracle (kind "port") timed on the host cores of the GPU box.
TEST/MEASUREMENT INFRASTRUCTURE ONLY -- it is a reported baseline, never the product path."""
from __future__ import annotations

import math
import os
import time

import torch


def run_cpu_baseline(n_splats: int, height: int, width: int, use_sh: bool, sh_degree: int, target_seconds: float, keep: dict = None):
    """`keep` (a dict): when the timed sample covered the WHOLE image, the oracle's outputs and gradients of that run are left
    in it (`out`, `grads`, `view` = 0): bench.py's `parity` block compares the HIP path with them -- no second oracle run."""
    from oracle import c_oracle
    from oracle import torch_oracle as O
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    cores = os.cpu_count() or 1
    sp = make_splats(n_splats, seed=1234)
    cam = make_camera(0, width, height)
    st = O.settings_from_camera(cam, torch.ones(3), sh_degree)
    gi, gd, ga = make_upstream_grads(height, width)
    gx, gy = (width + 15) // 16, (height + 15) // 16

    last = {}

    def timed(window):
        t0 = time.perf_counter()
        out, grads, _ = c_oracle.rasterize(sp, st, use_sh=use_sh, g_img=gi, g_depth=gd, g_alpha=ga, tile_window=window, threads=cores)
        dt = time.perf_counter() - t0
        last.update(out=out, grads=grads, window=window)
        return dt

    # calibrate on a centred crop of tile rows, then take the largest crop that fits the time budget:
    # every splat is always preprocessed and binned; only the blended tile window is bounded.
    rows = max(1, gy // 16)
    y0 = (gy - rows) // 2
    t_small = timed((0, y0, gx, y0 + rows))
    scale = max(1.0, target_seconds / max(t_small, 1e-3))
    rows_full = int(min(gy, max(rows, math.floor(rows * scale * 0.8))))
    if rows_full >= gy - 2:
        rows_full = gy  # the whole image fits the budget
    y0 = (gy - rows_full) // 2
    window = (0, y0, gx, y0 + rows_full)
    t = timed(window)
    if keep is not None and rows_full == gy:
        keep.update(out=last["out"], grads=last["grads"], view=0)
    px = min(height, (y0 + rows_full) * 16) - y0 * 16
    frac = rows_full / gy
    return {"value": n_splats * px * width / t, "unit": "splat*px/s", "cores": cores, "kind": "port",
            "seconds": t,
            "sample": f"C oracle (oracle/raster_ref.c, OpenMP {cores} threads), fwd+bwd of view 0 of the same workload: "
                      f"all {n_splats} splats preprocessed and binned, tile rows {y0}..{y0 + rows_full} of {gy} "
                      f"({frac:.0%} of the image, {px}x{width} px) blended and back-propagated"}


def run_torch_oracle_config0():
    """the PyTorch-CPU render path (the differentiable torch oracle in fp32 -- the stand-in for "the
    reference's PyTorch-CPU render path", which the reference does not have) at BASELINE.json configs[0]: 10 k splats, one
    256x256 camera, forward + autograd backward.  PyTorch's CPU kernels are run on at most 16 threads: the oracle issues
    ~10^5 small tensor ops, and a fork-join over 256 host threads per op takes minutes where 16 threads take seconds
    (`cores` reports the threads actually used)."""
    from oracle import torch_oracle as O
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    cores = min(16, os.cpu_count() or 1)
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    n, w, h = 10_000, 256, 256
    sp = make_splats(n, seed=1234)
    st = O.settings_from_camera(make_camera(1, w, h), torch.ones(3), 3)
    gi, gd, ga = make_upstream_grads(h, w)
    t0 = time.perf_counter()
    O.fwd_bwd(sp, st, gi, gd, ga, use_sh=True, dtype=torch.float32)
    t = time.perf_counter() - t0
    torch.set_num_threads(prev)
    return {"value": n * w * h / t, "unit": "splat*px/s", "cores": cores, "kind": "port", "seconds": t,
            "sample": f"oracle/torch_oracle.py (PyTorch CPU, fp32, autograd backward), {n} splats, {w}x{h}, one fwd+bwd, "
                      f"torch threads = {cores}"}
