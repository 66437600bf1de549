#!/usr/bin/env python3
"""Diagnostic (GPU box): the non-fragile pixels beyond 1e-4 of one view, with what they look like."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle
from tests.helpers import make_scene, run_hip

view = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prec = sys.argv[2] if len(sys.argv) > 2 else "fp64"
margins = [2e-4, 1e-3, 5e-3]
sp, cam, st, grads = make_scene(1_000_000, 800, 800, view=view)
out, g = run_hip(sp, st, grads, torch.device("cuda:0"))
refs = {}
for m in margins:
    refs[m], _, _ = c_oracle.rasterize(sp, st, use_sh=True, threads=min(128, os.cpu_count()), precision=prec, fragile=True, margin=m)
ref = refs[margins[0]]
print("radii mismatches", int((out["radii"] != ref["radii"]).sum()), "of", out["radii"].numel())
idx = (out["radii"] != ref["radii"]).nonzero().flatten()[:5]
print("  ", [(int(i), int(out["radii"][i]), int(ref["radii"][i])) for i in idx])
for k in ("color", "depth", "alpha"):
    a, b = out[k].double(), ref[k].double()
    err = (a - b).abs(); rel = err / b.abs().clamp_min(1e-3)
    bad = (rel > 1e-4) & ~ref["fragile"][None].expand_as(rel)
    print(k, "offenders", int(bad.sum()), "by margin:", {m: int(((rel > 1e-4) & ~refs[m]["fragile"][None].expand_as(rel)).sum()) for m in margins},
          "fragile share", {m: round(float(refs[m]["fragile"].float().mean()), 4) for m in margins})
    nz = bad.nonzero()
    order = rel[bad].argsort(descending=True)[:12]
    for j in order:
        c, y, x = [int(t) for t in nz[j]]
        print("   ch%d (%d,%d) ref %.6g hip %.6g abs %.3g rel %.3g  alpha_ref %.6g" % (c, y, x, b[c, y, x], a[c, y, x], err[c, y, x], rel[c, y, x], ref["alpha"][0, y, x]))
    # distribution of |ref| among offenders
    if bad.any():
        v = b.abs()[bad]
        print("   |ref| of offenders: min %.3g median %.3g max %.3g; abs err median %.3g max %.3g" % (v.min(), v.median(), v.max(), err[bad].median(), err[bad].max()))
