#!/usr/bin/env python3
"""Condense the output of tools/profile.sh into the files kept under profiles/.

    python tools/summarize_profile.py gpurun_out/prof8 profiles/r01_v5 [--traffic profiles/traffic.json]

writes <prefix>_bench.json, <prefix>_kernel_stats.csv, <prefix>_pmc.json and (with --traffic) the per-stage HBM
bytes `bench.py` reads for `roofline.traffic`.  HBM bytes follow MI355X_MICROARCH.md's rocprofv3 section:
FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate passes, and gfx950 counts wide reads at half size, so
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import argparse
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splatfields_amd.build import source_hash  # noqa: E402  (the kernel sources these counters were collected from)

# kernel name prefix -> pipeline stage of bench.py / sr_profile_stage_name
STAGE_OF = [
    ("sr::k_preprocess_backward", "preprocess_backward"),
    ("sr::k_preprocess", "preprocess"),
    ("sr::k_count_tiles", "scan"),
    ("sr::k_colscan", "scan"),
    ("sr::k_scan_small", "scan"),
    ("sr::k_emit", "emit"),
    ("sr::k_sort_tiles", "sort_tiles"),
    ("sr::k_render_forward", "render_forward"),
    ("sr::k_render_backward", "render_backward"),
]


def short_name(kernel):
    k = re.sub(r"^void ", "", kernel)
    return re.sub(r"\(.*$", "", k)


def per_launch(path):
    """{kernel: {counter: mean value per launch}} (a counter value is summed over its dimensions per dispatch)."""
    per_dispatch = defaultdict(float)
    for row in csv.DictReader(open(path)):
        if "sr::" not in row["Kernel_Name"]:
            continue
        per_dispatch[(short_name(row["Kernel_Name"]), row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
    acc = defaultdict(lambda: defaultdict(list))
    for (k, c, _), v in per_dispatch.items():
        acc[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}, \
           {k: len(next(iter(cs.values()))) for k, cs in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("prefix")
    ap.add_argument("--traffic")
    ap.add_argument("--sq", help="write the per-launch SQ counters of the two blend kernels (bench.py's roofline_valu) here")
    a = ap.parse_args()

    shutil.copy(os.path.join(a.src, "bench.json"), a.prefix + "_bench.json")
    shutil.copy(os.path.join(a.src, "stats", "r_kernel_stats.csv"), a.prefix + "_kernel_stats.csv")

    out = defaultdict(dict)
    launches = {}
    for sub, unit in (("pmc_fetch", "_KB_per_launch"), ("pmc_write", "_KB_per_launch"), ("pmc_sq", "_per_launch"),
                      ("pmc_sq2", "_per_launch")):
        if not os.path.exists(os.path.join(a.src, sub, "r_counter_collection.csv")):
            continue
        vals, n = per_launch(os.path.join(a.src, sub, "r_counter_collection.csv"))
        for k, cs in vals.items():
            for c, v in cs.items():
                out[k][c + unit] = v
        if sub == "pmc_fetch":
            launches = n
    for k, d in out.items():
        if "FETCH_SIZE_KB_per_launch" in d and "WRITE_SIZE_KB_per_launch" in d:
            d["hbm_bytes_per_launch_corrected"] = (2 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024
    json.dump(out, open(a.prefix + "_pmc.json", "w"), indent=1, sort_keys=True)

    if a.sq:
        bench = json.load(open(os.path.join(a.src, "bench.json")))
        cfg = bench["config"]
        sq = {"_workload": {"splats": cfg["splats"], "width": cfg["width"], "height": cfg["height"], "color": "sh", "sh_degree": 3,
                            "mean_scale": None},
              "_source": "%s_pmc.json (rocprofv3 --pmc SQ_* passes of tools/profile.sh, means per launch)" % a.prefix,
              "_source_hash": source_hash()}
        # average launch duration of the same kernels in the (counter-free) kernel-trace pass: kernel cycles / this = the clock
        avg_ns = {}
        for row in csv.DictReader(open(a.prefix + "_kernel_stats.csv")):
            avg_ns[short_name(row["Name"])] = float(row["AverageNs"])
        for k, d in out.items():
            for prefix, s in (("sr::k_render_forward", "render_forward"), ("sr::k_render_backward", "render_backward")):
                if k.startswith(prefix) and (s not in sq or d.get("SQ_WAVE_CYCLES_per_launch", 0) > sq[s].get("SQ_WAVE_CYCLES", 0)):
                    sq[s] = {c[:-len("_per_launch")]: v for c, v in d.items() if c.startswith("SQ_")}
                    sq[s]["kernel"] = k
                    if k in avg_ns:
                        sq[s]["avg_launch_ns_kernel_trace"] = avg_ns[k]
        json.dump(sq, open(a.sq, "w"), indent=1, sort_keys=True)

    if a.traffic:
        bench = json.load(open(os.path.join(a.src, "bench.json")))
        # launches of the bench pass per step: a kernel that runs once per step has `per_step` launches recorded;
        # rarer kernels (the long-list sort classes) are weighted by their launch count.
        per_step = max(launches.values())
        stage = defaultdict(float)
        for k, d in out.items():
            for prefix, s in STAGE_OF:
                if k.startswith(prefix):
                    stage[s] += d.get("hbm_bytes_per_launch_corrected", 0.0) * launches.get(k, 0) / per_step
                    break
        t = dict(stage)
        t["_note"] = ("HBM bytes per step per pipeline stage from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                      "KB units), (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 reports wide reads at half size "
                      "(MI355X_MICROARCH.md, HBM section); %s_pmc.json holds the raw counters" % a.prefix)
        cfg = bench["config"]
        t["_workload"] = {"splats": cfg["splats"], "width": cfg["width"], "height": cfg["height"], "color": "sh",
                          "sh_degree": 3}
        t["_source_hash"] = source_hash()
        json.dump(t, open(a.traffic, "w"), indent=1)
        for s, v in stage.items():
            print("%-20s %8.1f MB" % (s, v / 1e6))


if __name__ == "__main__":
    main()
