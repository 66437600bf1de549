#!/usr/bin/env python3
"""Randomised check of the forward that does not wait for its instance count (sr_forward_async) over image sizes, splat counts
and footprints: a camera that was rendered before is launched without a host wait and must give the bits of the waiting path;
when the cloud then grows or shrinks under the same camera, the launch either still holds (same bits as a waiting render of the
grown cloud) or is detected at the backward (RasterizerOverflow, nothing applied) and the re-run gives those bits.
GPU diagnostic: python tools/fuzz_async.py [cases] [seed]"""
import math
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from splatfields_amd import rasterizer as rz  # noqa: E402
from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads  # noqa: E402

NAMES = ["means3D", "scales", "rotations", "opacities"]


def step(sp, rs, grads, use_sh):
    from diff_gaussian_rasterization import GaussianRasterizer
    names = NAMES + (["shs"] if use_sh else ["colors_precomp"])
    leaf = {k: sp[k].detach().clone().requires_grad_(True) for k in names}
    m2 = torch.zeros_like(leaf["means3D"], requires_grad=True)
    c, r, d, a = GaussianRasterizer(rs).forward_ex(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"],
                                                   shs=leaf["shs"] if use_sh else None,
                                                   colors_precomp=None if use_sh else leaf["colors_precomp"],
                                                   scales=leaf["scales"], rotations=leaf["rotations"])
    loss = (c * grads[0]).sum() + (d * grads[1]).sum() + (a * grads[2]).sum()
    loss.backward()
    torch.cuda.synchronize()
    out = [c.detach(), r, d.detach(), a.detach(), m2.grad] + [leaf[k].grad for k in names]
    return out


def same(x, y):
    return all(torch.equal(a, b) for a, b in zip(x, y))


def main():
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device("cuda:0")
    bad = overflows = asyncs = 0
    for it in range(cases):
        n = int(math.exp(rnd.uniform(math.log(50), math.log(120000))))
        w, h = rnd.randint(16, 640), rnd.randint(16, 480)
        scale = math.exp(rnd.uniform(math.log(0.004), math.log(0.4)))
        use_sh = rnd.random() < 0.6
        sp = make_splats(n, seed=3000 + it, device=dev, mean_scale=scale)
        cam = make_camera(rnd.randint(0, 7), w, h, device=dev)   # persistent camera tensors, as a training loop has them
        grads = make_upstream_grads(h, w, device=dev)
        rs = GaussianRasterizationSettings(h, w, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3, device=dev), 1.0,
                                           cam.world_view_transform, cam.full_proj_transform, rnd.randint(0, 3), cam.camera_center,
                                           False, False)
        rz.set_async_forward(False)
        ref = step(sp, rs, grads, use_sh)
        rz.set_async_forward(True)
        rz.host_sync_counters(reset=True)
        got = step(sp, rs, grads, use_sh)
        c = rz.host_sync_counters()
        ok = same(got, ref) and c["async_forwards"] == (1 if n > 0 else 0) and c["forward_host_waits"] == 0
        # the cloud changes under the same camera and the same count
        f = rnd.choice([0.7, 1.1, 1.3, 1.6, 2.5])
        sp2 = dict(sp, scales=sp["scales"] * f)
        rz.set_async_forward(False)
        pack = rz._ViewPack.get(rs, dev, 16 if use_sh else 0)
        before = dict(pack.seen)
        ref2 = step(sp2, rs, grads, use_sh)
        rz.set_async_forward(True)
        pack.seen.clear(); pack.seen.update(before)           # the waiting render refreshed the record: make it stale again
        key = (dev.index, n, h, w)
        rz._CAPACITY[key] = rz._round_capacity(before[n][0])  # ... and the capacity: as if only the first cloud had been seen
        outcome = "held"
        try:
            got2 = step(sp2, rs, grads, use_sh)
        except rz.RasterizerOverflow:
            outcome = "overflow"
            overflows += 1
            got2 = step(sp2, rs, grads, use_sh)               # the estimates are corrected: the re-run is asynchronous and fits
        ok2 = same(got2, ref2)
        asyncs += 1
        flag = "" if (ok and ok2) else "  <-- CHECK"
        bad += bool(flag)
        print(f"{it:3d} n={n:6d} {w:3d}x{h:3d} scale={scale:.4f} sh={int(use_sh)} inst={rz.LAST_INSTANCES:8d} x{f:.1f} -> {outcome:8s} "
              f"same={ok} same_after_change={ok2}{flag}", flush=True)
    print(f"cases {cases}, overflows detected and recovered {overflows}, flagged {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
