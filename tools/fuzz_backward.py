#!/usr/bin/env python3
"""Randomised cross-check of the two backward blend kernels (and of bit-reproducibility) over image sizes, splat counts,
footprints, colour paths and upstream-gradient combinations.  GPU diagnostic: python tools/fuzz_backward.py [cases] [seed]"""
import math
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import grad_error, make_scene, run_hip  # noqa: E402
from splatfields_amd.rasterizer import set_backward_kernel  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    big = len(sys.argv) > 3 and sys.argv[3] == "big"   # larger clouds and images
    dev = torch.device("cuda:0")
    worst, bad = 0.0, 0
    for it in range(cases):
        n = int(math.exp(rnd.uniform(0, math.log(400000 if big else 30000))))
        w, h = (rnd.randint(100, 1000), rnd.randint(100, 900)) if big else (rnd.randint(5, 330), rnd.randint(5, 270))
        scale = math.exp(rnd.uniform(math.log(0.003), math.log(0.5)))
        use_sh = rnd.random() < 0.6
        wd, wa = rnd.random() < 0.7, rnd.random() < 0.7
        bg = tuple(rnd.random() for _ in range(3))
        sp, cam, st, grads = make_scene(n, w, h, mean_scale=scale, view=rnd.randint(0, 7), bg=bg, sh_degree=rnd.randint(0, 3),
                                        seed=1000 + it)
        res = {}
        for kernel in ("quads", "wave", "quads"):
            set_backward_kernel(kernel)
            _, g = run_hip(sp, st, grads, dev, use_sh=use_sh, with_depth=wd, with_alpha=wa)
            res.setdefault(kernel, []).append(g)
        set_backward_kernel(None)
        e = max(grad_error(res["quads"][0][k], res["wave"][0][k]) for k in res["wave"][0])
        repro = all(torch.equal(res["quads"][0][k], res["quads"][1][k]) for k in res["wave"][0])
        finite = all(torch.isfinite(v).all() for v in res["quads"][0].values())
        worst = max(worst, e)
        flag = "" if (e <= 2e-4 and repro and finite) else "  <-- CHECK"
        bad += bool(flag)
        print(f"{it:3d} n={n:6d} {w:3d}x{h:3d} scale={scale:.4f} sh={int(use_sh)} d={int(wd)} a={int(wa)} "
              f"diff={e:.2e} repro={repro} finite={finite}{flag}", flush=True)
    print(f"worst mfma-vs-wave difference {worst:.3e}; flagged cases: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
