#!/usr/bin/env python3
"""GPU diagnostic: gradients of the MFMA backward blend (blend_bwd.hip) against round 1's pixel-per-lane kernel
(set_backward_kernel("wave")) and against the C oracle, on a few scene regimes.  Usage: python tools/bwd_compare.py [fast]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import grad_error, make_scene, run_hip  # noqa: E402


def grads_with(kernel, sp, st, grads, dev, use_sh=True):
    from splatfields_amd.rasterizer import set_backward_kernel
    set_backward_kernel(kernel)
    try:
        out, g = run_hip(sp, st, grads, dev, use_sh=use_sh)
    finally:
        set_backward_kernel(None)
    return out, g


def main():
    dev = torch.device("cuda:0")
    scenes = [
        ("4k 160x120", dict(n=4000, width=160, height=120), {}),
        ("10k 256x256", dict(n=10000, width=256, height=256), {}),
        ("6k 250x187 rgb", dict(n=6000, width=250, height=187, bg=(0.0, 0.0, 0.0), view=5), dict(use_sh=False)),
        ("dense 9k 96x64 s=.25", dict(n=9000, width=96, height=64, mean_scale=0.25, view=3), {}),
        ("2k 128x96 s=.03", dict(n=2000, width=128, height=96, mean_scale=0.03), {}),
        ("300k 800x800", dict(n=300000, width=800, height=800), {}),
        ("1M 800x800", dict(n=1000000, width=800, height=800), {}),
    ]
    if len(sys.argv) > 1 and sys.argv[1] == "fast":
        scenes = scenes[:5]
    worst = 0.0
    for name, kw, opt in scenes:
        sp, cam, st, grads = make_scene(**kw)
        use_sh = opt.get("use_sh", True)
        _, g_old = grads_with("wave", sp, st, grads, dev, use_sh)
        t0 = time.time()
        _, g_new = grads_with(None, sp, st, grads, dev, use_sh)
        _, g_new2 = grads_with(None, sp, st, grads, dev, use_sh)
        line = []
        for k in g_new:
            e = grad_error(g_new[k], g_old[k])
            worst = max(worst, e)
            rep = bool(torch.equal(g_new[k], g_new2[k]))
            nan = int(torch.isnan(g_new[k]).sum())
            line.append(f"{k}={e:.2e}{'' if rep else ' NONREPRO'}{'' if nan == 0 else ' NAN%d' % nan}")
        print(f"[{name}] new vs wave: " + " ".join(line), flush=True)
        if kw["n"] <= 10000:
            from oracle import c_oracle
            cout, cg, _ = c_oracle.rasterize(sp, st, use_sh=use_sh, g_img=grads[0], g_depth=grads[1], g_alpha=grads[2])
            print(f"    vs C oracle  new: " + " ".join(f"{k}={grad_error(g_new[k], cg[k]):.2e}" for k in cg)
                  + "\n    vs C oracle wave: " + " ".join(f"{k}={grad_error(g_old[k], cg[k]):.2e}" for k in cg), flush=True)
    print("worst new-vs-wave relative difference: %.3e" % worst)


if __name__ == "__main__":
    main()
