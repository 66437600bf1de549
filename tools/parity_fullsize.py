#!/usr/bin/env python3
"""Full-size parity survey (GPU box): every bench view x both colour paths of a workload against the C oracle in float and
in double, with the oracle's fragile-pixel mask and splat flags.  Prints one JSON line per comparison and appends them to
gpurun_out/parity_fullsize.jsonl.  TEST INFRASTRUCTURE (uses oracle/)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--views", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--paths", default="sh,precomp")
    ap.add_argument("--precisions", default="fp32,fp64")
    ap.add_argument("--mean-scale", type=float, default=None)
    ap.add_argument("--margin", type=float, default=2e-4)
    ap.add_argument("--xy-ulps", type=float, default=2.0)
    ap.add_argument("--out", default="gpurun_out/parity_fullsize.jsonl")
    a = ap.parse_args()
    from oracle import c_oracle, parity as P
    from tests.helpers import make_scene, run_hip
    dev = torch.device("cuda:0")
    threads = min(128, os.cpu_count() or 8)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    for view in [int(x) for x in a.views.split(",")]:
        sp, cam, st, grads = make_scene(a.splats, a.width, a.height, view=view, mean_scale=a.mean_scale)
        for path in a.paths.split(","):
            use_sh = path == "sh"
            out, g = run_hip(sp, st, grads, dev, use_sh=use_sh)
            for prec in a.precisions.split(","):
                ref, rg, _ = c_oracle.rasterize(sp, st, use_sh=use_sh, g_img=grads[0], g_depth=grads[1], g_alpha=grads[2],
                                                threads=threads, precision=prec, fragile=True, margin=a.margin, xy_ulps=a.xy_ulps)
                fig = P.compare_flagged(out, g, ref, rg, oracle_precision=prec)
                line = {"splats": a.splats, "size": [a.width, a.height], "view": view, "path": path, "oracle": prec,
                        "radii": fig["radii"], "unexplained": fig["unexplained"],
                        "conditioning_limited": fig["image_conditioning_limited"],
                        "max_err_over_allowance": max(v.get("max_err_over_allowance", 0.0) for v in fig["images"].values()),
                        "image_robust_max_rel": fig["image_robust_max_rel"], "image_robust_above_1e-4": fig["image_robust_above_1e-4"],
                        "fragile_share": fig["images"]["color"]["fragile_share"],
                        "fragile_max_abs": max(v["fragile_max_abs"] for v in fig["images"].values()),
                        "flagged_splat_share": fig["flagged_splat_share"],
                        "grad": {k: [v["max_rel_to_tensor_max"], v["max_unflagged"], v["unexplained"], v["splats_above_1e-3"]]
                                 for k, v in fig["gradients"].items()}}
                s = json.dumps(line)
                print(s, flush=True)
                with open(a.out, "a") as f:
                    f.write(s + "\n")


if __name__ == "__main__":
    main()
