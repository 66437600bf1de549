#!/usr/bin/env python3
"""Work counters of the backward blend for one fwd+bwd of a synthetic workload (needs a library built with
-DSR_BWD_STATS, selected through SPLATRASTER_LIB).  Prints one JSON object."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def backward_work(n, width, height, mean_scale=None, use_sh=True, view=1, device="cuda:0"):
    from splatfields_amd import _lib, rasterizer as rz
    from tests.helpers import make_scene, run_hip
    lib = _lib.load()
    out8 = (C.c_ulonglong * 16)()
    sp, cam, st, grads = make_scene(n, width, height, mean_scale=mean_scale, view=view)
    lib.sr_debug_backward_stats(out8, 1)
    run_hip(sp, st, grads, torch.device(device), use_sh=use_sh)
    lib.sr_debug_backward_stats(out8, 1)
    e, qe, bk, pe, pb, ch = [int(out8[i]) for i in range(6)]
    return {"splats": n, "width": width, "height": height, "mean_scale": mean_scale, "tile_instances": int(rz.LAST_INSTANCES),
            "list_entries_replayed": e, "quad_entry_pairs": qe, "buckets": bk, "pairs_evaluated": pe, "pairs_blended": pb,
            "chunks": ch, "lane_efficiency": (pb / pe) if pe else None,
            "bucket_fill": (qe / (16.0 * bk)) if bk else None,
            "phase_cycles_share": (lambda ph: {k: round(x / max(sum(ph), 1), 4) for k, x in zip(
                ["preamble", "test", "barrier1", "assign+scatter", "barrier2", "replay", "barrier3", "combine"], ph)})(
                [int(out8[8 + i]) for i in range(8)]),
            "wave_cycles_total": sum(int(out8[8 + i]) for i in range(8))}


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--splats", type=int, default=1_000_000)
    p.add_argument("--width", type=int, default=800)
    p.add_argument("--height", type=int, default=800)
    p.add_argument("--mean-scale", type=float, default=None)
    a = p.parse_args()
    print(json.dumps(backward_work(a.splats, a.width, a.height, a.mean_scale)))
