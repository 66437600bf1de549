#!/usr/bin/env python3
"""Benchmark of the hot path: forward+backward splat rasterization of one view per GPU.

Contract (task statement §④): `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver
launches it with torch.distributed.run, one rank per GPU.  A *step* = each rank renders ONE view of
the same splat cloud (forward + backward through the drop-in facade -> C ABI -> HIP kernels) and,
for N>1, the per-splat gradients are sum-all-reduced over RCCL (view-parallel training step,
reference train.py:169-252).  Per-GPU work is fixed as N grows ("weak" scaling).

metric  = splats*px rasterized per second (fwd+bwd) = n_gpus * N_splats * H * W / t_step
workload = BASELINE.json's metric configuration: 1 M synthetic splats, 800x800, SH degree 3.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PRE_CYCLES = 6   # untimed passes over the 8-view cycle before the W warm-up steps (see main)
HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md:35 (6.29e12 measured-achievable)
FP32_PEAK = 157.3e12  # FLOP/s, vector fp32 (same guide, :40)
SIMDS = 1024          # 256 CUs x 4 SIMD-32
VALU_ISSUE_CYCLES = 2.0   # a wave64 VALU instruction occupies its SIMD-32 for two cycles (same guide, "Wave scheduling")
FLOP_PER_PAIR = {"render_forward": 30.0, "render_backward": 90.0}   # SURVEY.md Appendix C, per blended (pixel, splat) pair


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--splats", type=int, default=1_000_000)
    p.add_argument("--width", type=int, default=800)
    p.add_argument("--height", type=int, default=800)
    p.add_argument("--color", choices=["sh", "precomp"], default="sh")
    p.add_argument("--sh-degree", type=int, default=3)
    p.add_argument("--cpu-baseline", choices=["auto", "none"], default="auto")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time of the bounded baseline sample")
    p.add_argument("--dp-mode", choices=["shard", "gather", "allreduce"], default="gather",
                   help="N>1, SH path.  'gather' (default) = all-gather of 12 B/splat colour gradients, the SH gradient is rebuilt "
                        "on every rank (sh_gather_step, ~161 B/splat on the wire): afterwards EVERY rank holds the full gradient "
                        "of all five tensors -- what the reference's replicated Adam consumes; 'allreduce' = the same result by "
                        "sum-all-reduce of all five gradient tensors (~413 B/splat); 'shard' = SH coefficients sharded by splat "
                        "range, colours and colour gradients travel by all-to-all (sh_sharded_step, ~98 B/splat): the SH gradient "
                        "then exists only on its owner (a sharded optimizer), so it is a different deliverable and never the "
                        "default.  A mode that fails on any rank falls back to the next one on all ranks.")
    p.add_argument("--mean-scale", type=float, default=None, help="mean splat scale in world units (default: SURVEY.md 8d rule, "
                                                                   "0.35 N^(-1/3)); larger = denser tile lists")
    p.add_argument("--extra-workloads", choices=["auto", "none"], default="auto",
                   help="rank 0, 1 GPU: also time the precomputed-colour headline and two dense regimes (a few steps each) and "
                        "report them under other_workloads")
    p.add_argument("--inputs", choices=["boundary", "raw-split"], default="boundary",
                   help="'boundary' (default, the reference's call): activated tensors and the concatenated SH tensor; 'raw-split': "
                        "the optimiser's raw parameters (logits, log-scales, unnormalised quaternions) and the two SH tensors "
                        "(dc, rest) -- activations and concatenation fused into the kernels (GaussianRasterizer.forward_raw)")
    p.add_argument("--force-dp-path", action="store_true", help="run the chosen --dp-mode step function even with 1 GPU (overhead check)")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for smoke-testing the control flow)")
    p.add_argument("--single-device", action="store_true", help="smoke test: every rank uses cuda:0 (needs --backend gloo)")
    return p.parse_args()


def stage_bytes(stage: str, n: int, vis: float, R: float, hw: int, c_in: int) -> float:
    """Algorithmic HBM bytes of one launch of a pipeline stage (SURVEY.md Appendix C terms)."""
    return {
        # read mean 12 + scale 12 + rot 16 + opacity 4 + colour input; write radii 4; visible: 40 B state
        # (+ 36 B colour/direction Jacobian on the SH path, which spares the backward the SH read)
        "preprocess": n * (44 + c_in + 4) + vis * (40 + (36 if c_in > 12 else 0)),
        "scan": n / 256 * 8,
        # read depth/rect/count per splat, write one 8-byte key (depth bits, splat index) per instance
        "emit": n * 16 + R * 8,
        # read the 8-byte key, write the sorted splat index
        "sort_tiles": R * 12,
        # read id 4 + gather 40 B state per instance; write rgb 12, depth 4, alpha 4, final_T 4, n_contrib 4 per pixel
        "render_forward": R * 44 + hw * 28,
        # read id 4 + first-instance offset 4 + 40 B state, write 40 B gradient moments per instance;
        # read dL/drgb 12, dL/ddepth 4, dL/dalpha 4, final_T 4, n_contrib 4 per pixel
        "render_backward": R * 88 + hw * 28,
        # re-read inputs (SH path: the 36 B Jacobian instead of the coefficients), read 40 B per instance, write all gradients
        "preprocess_backward": n * (44 + (36 if c_in > 12 else c_in)) + R * 40 + vis * 40 + n * (12 + 12 + 12 + 16 + 4 + c_in),
    }[stage]


def pipeline_bytes(n: int, vis: float, R: float, hw: int, c_in: int) -> float:
    """B_alg of SURVEY.md §8d: N*(3*(44+C_in)+16) + V_vis*160 + R*164 + H*W*48."""
    return n * (3 * (44 + c_in) + 16) + vis * 160 + R * 164 + hw * 48



# What bounds each stage on MI355X (rocprofv3 SQ counters and the timing experiments of round 4):
# "hbm" = streams per-splat data at 75-80 % of the achievable bandwidth; "valu" = the two blend kernels (issue + dependent-issue
# latency of the per-pair arithmetic; their HBM figure is reported because it is the north star's yardstick, not because it
# bounds them); "latency" = the small dependent kernels of the bucketing chain (launch gaps, LDS atomics, barriers).
STAGE_BOUND = {"preprocess": "hbm", "preprocess_backward": "hbm", "render_forward": "valu", "render_backward": "valu",
               "scan": "latency", "emit": "latency", "sort_tiles": "valu"}


def roofline_block(stage_ms: dict, n: int, vis: float, R: float, hw: int, c_in: int, ms_per_step: float, ms_median: float,
                   traffic=None, traffic_why=None, valu=None) -> dict:
    """The `roofline` object of the JSON line: the dominant stage in the contract's terms (algorithmic bytes per launch / its
    average duration, against the 8 TB/s HBM peak), what really bounds that stage, and -- the figure the north star is stated in --
    the whole forward+backward pipeline's algorithmic bytes per step time."""
    dom = max(stage_ms, key=stage_ms.get)
    dom_bytes = stage_bytes(dom, n, vis, R, hw, c_in)
    dom_bw = dom_bytes / (stage_ms[dom] * 1e-3)
    b_alg = pipeline_bytes(n, vis, R, hw, c_in)
    out = {"bound": STAGE_BOUND[dom], "kernel": dom, "achieved": dom_bw / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
           "frac": dom_bw / HBM_PEAK, "frac_basis": "algorithmic bytes per launch / average launch duration / 8 TB/s HBM peak",
           "traffic": traffic, "traffic_note": traffic_why, "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": stage_ms[dom],
           "stage_bound": {k: STAGE_BOUND[k] for k in stage_ms},
           "pipeline_b_alg_bytes": b_alg, "pipeline_achieved": b_alg / (ms_per_step * 1e-3) / 1e9,
           "pipeline_frac": b_alg / (ms_per_step * 1e-3) / HBM_PEAK,
           "pipeline_frac_vs_6.29": b_alg / (ms_per_step * 1e-3) / 6.29e12,
           "pipeline_frac_from_median": b_alg / (ms_median * 1e-3) / HBM_PEAK if ms_median > 0 else None,
           "note": ("bound = what limits the dominant kernel; the blend kernels are bound by VALU issue and dependent-issue latency, "
                    "not by HBM: see roofline_valu (SQ counters) and blend_work (pairs evaluated / blended); achieved / frac are "
                    "the HBM-roofline figures of the task contract; `traffic` is L2<->fabric bytes incl. requests served by the "
                    "256 MB infinity cache; pipeline_* = SURVEY section 8d's B_alg over the whole fwd+bwd step (north star)")}
    if valu and valu.get(dom):
        out["valu_issue_frac"] = valu[dom].get("valu_issue_frac")
    return out


def checked_forward(rz, fwd):
    """`fwd()` launched without the mid-forward host wait where the camera is known (rasterizer.py: sr_forward_async), its ticket
    redeemed BEFORE the outputs are used: the step of a training loop that opts into the asynchronous launch.  A capacity promise
    that did not hold (never here: static splats, every camera rendered before) is rendered again."""
    with rz.async_forward():
        out = fwd()
        try:
            rz.resolve_pending()
        except rz.RasterizerOverflow:
            out = fwd()
            rz.resolve_pending()
    return out


def time_plain_workload(n, width, height, use_sh, mean_scale, sh_degree, steps, warmup, dev):
    """ms per fwd+bwd step and per-stage HIP-event times of one more single-GPU workload (same step as the headline)."""
    import math
    from splatfields_amd import _lib, rasterizer as rz
    from splatfields_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    lib = _lib.load()
    sp = make_splats(n, seed=1234, device=dev, mean_scale=mean_scale)
    names = ["means3D", "scales", "rotations", "opacities", "shs" if use_sh else "colors_precomp"]
    params = {k: sp[k].clone().requires_grad_(True) for k in names}
    gi, gd, ga = make_upstream_grads(height, width, device=dev)
    bg = torch.ones(3, device=dev)
    cams = [make_camera(k, width, height, device=dev) for k in range(8)]
    vis = [0.0]
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)   # only its gradient is used: allocated once

    def step(i, record=False):
        cam = cams[i % len(cams)]
        rs = GaussianRasterizationSettings(
            image_height=height, image_width=width, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
            scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree,
            campos=cam.camera_center, prefiltered=False, debug=False)
        for p in params.values():
            p.grad = None
        means2D.grad = None
        color, radii, depth, alpha = checked_forward(rz, lambda: GaussianRasterizer(rs).forward_ex(
            means3D=params["means3D"], means2D=means2D,
            opacities=params["opacities"], shs=params["shs"] if use_sh else None,
            colors_precomp=None if use_sh else params["colors_precomp"], scales=params["scales"], rotations=params["rotations"]))
        torch.autograd.backward((color, depth, alpha), (gi, gd, ga))
        if record:
            vis[0] = float((radii > 0).sum().item())

    import gc
    gc.collect()
    gc.disable()            # as for the headline (main): parked from the warm-up to the end of the timed steps
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    # same estimators as the headline: wall-clock mean of the K steps (ms_per_step) and the median of per-step HIP-event
    # times (ms_per_step_median: one host stall -- the allocator, a subprocess that just ran -- does not become the figure)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    ms_per_step = (time.perf_counter() - t0) / steps * 1e3
    gc.enable()
    gc.collect()
    for i in range(24):                        # (the collector's pause left the device idle)
        step(i)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    stream = torch.cuda.current_stream(dev)
    for i in range(steps):                     # second pass: per-step events (they cost the stream a few microseconds each)
        marks[i].record(stream)
        step(warmup + i)
    marks[steps].record(stream)
    torch.cuda.synchronize()
    ms_median = median([marks[i].elapsed_time(marks[i + 1]) for i in range(steps)])
    stage_ms = staged_pass(lib, steps, lambda i: step(warmup + i, record=(i == steps - 1)))
    R = float(rz.LAST_INSTANCES)
    c_in = 12 * 16 if use_sh else 12
    b_alg = pipeline_bytes(n, vis[0], R, height * width, c_in)
    binning = stage_ms["scan"] + stage_ms["emit"] + stage_ms["sort_tiles"]
    blend = stage_ms["render_forward"] + stage_ms["render_backward"]
    return {"splats": n, "width": width, "height": height, "mean_scale": mean_scale,
            "color": "sh%d" % sh_degree if use_sh else "precomp", "steps": steps, "ms_per_step": ms_per_step,
            "ms_per_step_median": ms_median, "stage_ms_sum": sum(stage_ms.values()),
            "value": n * height * width / (ms_per_step * 1e-3), "tile_instances": R, "instances_per_splat": R / n,
            "visible_splats": vis[0], "stage_ms": stage_ms, "binning_over_blend": binning / blend if blend > 0 else None,
            "sort_keys_per_s": R / (stage_ms["sort_tiles"] * 1e-3) if stage_ms["sort_tiles"] > 0 else None,
            "roofline_pipeline_frac": b_alg / (ms_per_step * 1e-3) / HBM_PEAK,
            "roofline_pipeline_frac_from_median": b_alg / (ms_median * 1e-3) / HBM_PEAK if ms_median > 0 else None}


def time_reference_pattern(n, width, height, sh_degree, steps, warmup, dev):
    """The reference's LITERAL per-view call pattern behind the unmodified drop-in (missing 3/4): the
    `GaussianModel` accessors (`exp`, `sigmoid`, `normalize`, `cat(dc, rest)`: scene/gaussian_model.py:64-86), `render()` with its
    second rasterizer call for the mask (gaussian_renderer/__init__.py:94-115: white colours on a fresh `bg_color*0.0`), one
    backward -- through `diff_gaussian_rasterization.GaussianRasterizer.forward` with the facade's defaults (every forward
    waits for its instance count, as [EXT]'s does).  Timed with the mask pass served from the first pass's alpha output
    (rasterizer.py: _serve_mask_call, the default) and with two full rasterizations (SPLATRASTER_MASK_SHORTCUT=0)."""
    import math
    from types import SimpleNamespace
    from splatfields_amd import rasterizer as rz
    from splatfields_amd.render import render
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    sp = make_splats(n, seed=1234, device=dev)
    raw = {"xyz": sp["means3D"], "opacity": torch.logit(sp["opacities"].clamp(1e-4, 1 - 1e-4)), "scaling": torch.log(sp["scales"]),
           "rotation": sp["rotations"] * 1.7, "f_dc": sp["shs"][:, :1].contiguous(), "f_rest": sp["shs"][:, 1:].contiguous()}
    P = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    gi, gd, ga = make_upstream_grads(height, width, device=dev)
    bg = torch.ones(3, device=dev)
    cams = [make_camera(k, width, height, device=dev) for k in range(8)]
    pipe = SimpleNamespace(debug=False)

    def step(i):
        for p_ in P.values():
            p_.grad = None
        gdict = {"means3D": P["xyz"], "active_sh_degree": sh_degree, "gaussian_opacity": torch.sigmoid(P["opacity"]),
                 "gaussian_scales": torch.exp(P["scaling"]), "gaussian_rotations": torch.nn.functional.normalize(P["rotation"]),
                 "gaussian_features": torch.cat((P["f_dc"], P["f_rest"]), dim=1)}
        res = render(cams[i % len(cams)], gdict, pipe, bg, two_pass=True)
        torch.autograd.backward((res["render"], res["depth"], res["opacity"]), (gi, gd, ga))

    def timed():
        import gc
        gc.collect(); gc.disable()
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        gc.enable()
        return ms

    prev_async = rz.set_async_forward(None)
    prev = rz.set_mask_shortcut(True)
    try:
        served0 = rz.MASK_CALLS_SERVED
        rz.host_sync_counters(reset=True)
        ms_fused = timed()
        counters = rz.host_sync_counters()
        served = rz.MASK_CALLS_SERVED - served0
        rz.set_mask_shortcut(False)
        ms_two = timed()
    finally:
        rz.set_mask_shortcut(prev)
        rz.set_async_forward(prev_async)
    return {"name": "reference call pattern, unmodified render(): accessors + cat + two rasterizer calls per view + one backward",
            "splats": n, "width": width, "height": height, "color": "sh%d" % sh_degree, "steps": steps,
            "ms_per_step": ms_fused, "value": n * height * width / (ms_fused * 1e-3),
            "ms_per_step_two_full_passes": ms_two, "speedup_from_serving_the_mask_pass": ms_two / ms_fused,
            "mask_calls_served": served, "rasterizations_per_step": (counters["forward_host_waits"] + counters["async_forwards"]) / (steps + warmup),
            "forward_host_waits_per_step": counters["forward_host_waits"] / (steps + warmup),
            "note": "the plain facade: every forward waits for its instance count on the host ([EXT] does the same); the mask pass "
                    "costs no rasterization and no host wait"}


def staged_pass(lib, n_steps: int, step_fn) -> dict:
    """Per-stage durations (HIP events on the launch stream inside the library, sr_profile_enable): `n_steps` steps, collected
    step by step; per stage the AVERAGE over the steps, leaving out steps that took more than three times the stage's median --
    one slow launch (a clock or allocator hiccup on the box: round 5's first run showed one 3 ms forward among 20 in a side
    workload, which made a 0.08 ms stage read 0.24) does not become the figure."""
    from splatfields_amd import _lib
    lib.sr_profile_enable(1)
    per = [[] for _ in range(_lib.PROFILE_STAGES)]
    for i in range(n_steps):
        step_fn(i)
        ms = (C.c_double * _lib.PROFILE_STAGES)()
        cnt = (C.c_longlong * _lib.PROFILE_STAGES)()
        lib.sr_profile_collect(ms, cnt)
        for k in range(_lib.PROFILE_STAGES):
            if cnt[k] > 0:
                per[k].append(ms[k])   # the stage's launches of this step together (e.g. the sort's two classes)
    torch.cuda.synchronize()
    lib.sr_profile_enable(0)
    def robust_mean(xs):
        if not xs:
            return 0.0
        m = median(xs)
        keep = [x for x in xs if x <= 3.0 * m] or xs
        return sum(keep) / len(keep)
    return {lib.sr_profile_stage_name(k).decode(): robust_mean(per[k]) for k in range(_lib.PROFILE_STAGES)}


def blend_work_counters(n, width, height, mean_scale):
    """pairs_evaluated / pairs_blended of the backward blend, counted by the SR_BWD_STATS build of the
    same sources on ONE fwd+bwd of view 1, in its own process (tools/bwd_stats.py with SPLATRASTER_LIB pointing at the
    counting build): a separate library, never the timed path."""
    import subprocess
    from splatfields_amd import build as b
    if not b.STATS_LIB_PATH.exists():
        return None
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bwd_stats.py"), "--splats", str(n), "--width", str(width), "--height", str(height)]
    if mean_scale is not None:
        cmd += ["--mean-scale", str(mean_scale)]
    env = dict(os.environ, SPLATRASTER_LIB=str(b.STATS_LIB_PATH))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=180)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    ph = d.pop("phase_cycles_share", {})
    share = {}
    for k, x in ph.items():
        k = "barrier" if k.startswith("barrier") else k
        share[k] = round(share.get(k, 0.0) + x, 4)
    return {"kernel": "render_backward", "list_entries_replayed": d["list_entries_replayed"], "quad_entry_pairs": d["quad_entry_pairs"],
            "buckets_of_16": d["buckets"], "pairs_evaluated": d["pairs_evaluated"], "pairs_blended": d["pairs_blended"],
            "lane_efficiency": d["lane_efficiency"], "bucket_fill": d["bucket_fill"], "list_chunks": d["chunks"],
            "wave_cycle_share_by_phase": share,
            "note": "pair = (pixel, list entry) on which alpha is evaluated; blended = alpha >= 1/255 and not behind the pixel's "
                    "last contributor; counted on one fwd+bwd of view 1 by the -DSR_BWD_STATS build (tools/bwd_stats.py), which runs three workgroups per CU "
                    "(its counters cost registers) where the product runs four: the phase shares are its own"}


def median(xs):
    xs = sorted(xs)
    n = len(xs)
    return 0.0 if n == 0 else (xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2]))


def recorded_counters(name: str, workload: dict, check_hash: bool = True):
    """A counter file under profiles/ (rocprofv3 --pmc passes condensed by tools/summarize_profile.py) if it was recorded for
    this workload AND for the kernel sources the library is built from (`_source_hash`); otherwise (None, reason)."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no recorded counters"
    try:
        pj = json.load(open(path))
    except Exception:
        return None, "unreadable counter file"
    want = {k: workload[k] for k in pj.get("_workload", {}) if k in workload}
    if pj.get("_workload") != want or not want:
        return None, "counters were recorded for another workload"
    if check_hash:
        from splatfields_amd.build import source_hash
        if pj.get("_source_hash") != source_hash():
            return None, "kernel sources changed since the counters were recorded (tools/profile.sh + tools/summarize_profile.py)"
    return pj, None


def valu_roofline(workload: dict, stage_ms: dict, pairs_blended=None, check_hash: bool = True):
    """Issue-side view of the two blend kernels (they are VALU-bound, not HBM-bound), against HARDWARE ceilings:

    * `valu_issue_frac` = wave-level VALU instructions x 2 cycles / 1024 SIMDs / kernel cycles: the share of the launch during
      which a SIMD's VALU port is at least occupied (every wave64 VALU instruction takes two cycles of its SIMD-32; DPP,
      v_cmp / v_cndmask and transcendental instructions take more, so 1.0 is not reachable with this instruction mix).
      Kernel cycles = SQ_BUSY_CYCLES / 32 shader engines, i.e. counted at the clock the launch really ran at
      (`clock_ghz` = cycles / duration of the traced launch), not at the 2.4 GHz maximum.
    * `useful_flop_frac` = blended (pixel, splat) pairs x FLOP per pair (30 forward / 90 backward)
      / live launch duration / 157.3 TFLOP/s: what the fp32 vector peak would need for the work that reaches the image.
    * `wave_cycle_shares`: where the resident wavefronts spend their cycles -- parked (s_waitcnt / barrier: SQ_WAIT_ANY),
      issue-stalled (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY) over SQ_WAVE_CYCLES.

    The counters are a recorded rocprofv3 measurement of this workload and these kernel sources (profiles/pmc_sq.json); the
    launch duration next to them is the live one."""
    pj, why = recorded_counters("pmc_sq.json", workload, check_hash)
    if pj is None:
        return None if why in ("no recorded counters", "counters were recorded for another workload") else {"stale": why}
    out = {"source": pj.get("_source"), "source_hash": pj.get("_source_hash"), "simds": SIMDS,
           "valu_issue_cycles_per_wave_inst": VALU_ISSUE_CYCLES, "fp32_peak_tflops": FP32_PEAK / 1e12}
    for stage in ("render_forward", "render_backward"):
        c = pj.get(stage)
        if not c:
            continue
        cycles = c["SQ_BUSY_CYCLES"] / 32.0            # summed over the 32 shader engines
        st = {"wave_valu_insts_per_launch": c["SQ_INSTS_VALU"], "wave_salu_insts_per_launch": c["SQ_INSTS_SALU"],
              "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "mfma_insts_per_launch": c.get("SQ_INSTS_MFMA"),
              "kernel_cycles": cycles, "cycles_per_valu_inst_per_simd": cycles * SIMDS / c["SQ_INSTS_VALU"],
              "valu_issue_frac": c["SQ_INSTS_VALU"] * VALU_ISSUE_CYCLES / SIMDS / cycles,
              "live_avg_launch_ms": stage_ms.get(stage)}
        st["frac"] = st["valu_issue_frac"]
        if c.get("avg_launch_ns_kernel_trace"):
            st["clock_ghz"] = cycles / c["avg_launch_ns_kernel_trace"]
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            st["wave_cycle_shares"] = {"parked": c.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stalled": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                                       "issuing": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
        if pairs_blended and stage_ms.get(stage):
            st["useful_flop_frac"] = pairs_blended * FLOP_PER_PAIR[stage] / (stage_ms[stage] * 1e-3) / FP32_PEAK
        out[stage] = st
    return out


def deform_network_probe():
    """configs[4] (4-D scene): the SplatFields-shaped deform network at 100 k splats, PyTorch-ROCm vs the fused MLP kernels, next
    to the rasterizer for the same splats -- tools/deform_net_probe.py in its own process (not part of the timed steps)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "deform_net_probe.py"), "--steps", "10"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    keep = ["splats", "image", "dtype", "net_fwd_bwd_ms", "net_fwd_bwd_fused_mlps_ms", "rasterizer_fwd_bwd_ms", "full_step_ms",
            "full_step_fused_mlps_ms", "net_forward_ms", "net_forward_fused_mlps_ms", "net_share_of_step", "fused_forward_max_rel_diff",
            "fused_training_max_rel_grad_diff_parameters", "fused_training_xyz_grad_median_rel_diff",
            "product_net_fwd_bwd_ms", "product_full_step_ms", "product_net_forward_ms", "product_decoderfree_net_fwd_bwd_ms",
            "product_decoderfree_full_step_ms", "product_decoderfree_net_forward_ms", "product_graphed_net_fwd_bwd_ms",
            "product_graphed_full_step_ms", "product_decoderfree_graphed_net_fwd_bwd_ms", "product_decoderfree_graphed_full_step_ms",
            "product_graphed_error", "product_decoderfree_graphed_error", "product_launches_per_step",
            "product_decoderfree_launches_per_step", "product_library_launches_per_step",
            "product_decoderfree_library_launches_per_step", "product_mfma_frac", "product_decoderfree_mfma_frac", "mlp_macs_per_splat"]
    res = {k: d[k] for k in keep if k in d}
    res["name"] = ("4-D config: deform network (stand-in of the reference's shapes), PyTorch-ROCm vs fused MLP kernels; product_* = "
                   "splatfields_amd.deform_field.SplatFields (tri-plane lookup, ResField composition and MLPs on HIP kernels) with the same "
                   "stand-in plane decoder, product_decoderfree_* = the sampler owns its planes")
    return res


def self_launch(args) -> int:
    """`python bench.py --gpus N` started WITHOUT a launcher: start the N ranks here (torch.distributed.run, one process per GPU,
    rendezvous on 127.0.0.1) with the same arguments and hand their output through -- a run that silently used one GPU and
    printed n_gpus: 1 would lose the first multi-GPU measurement to a launcher assumption."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE in the environment: launching {args.gpus} ranks "
          f"(torch.distributed.run, 127.0.0.1:{port})", file=sys.stderr, flush=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def collective_probe(n_splats: int, world: int, dev, iters: int = 5) -> dict:
    """Bus bandwidth of the exchange's collectives on this node, each alone (median of `iters`, after two warm-up calls):
    all-gather of one view's colour gradients per rank (12 B/splat in, world x 12 out) and the in-place sum all-reduce of the
    44 B/splat of geometric gradients.  bus bandwidth = ring-model wire bytes per GPU / time (the figure RCCL's own tests quote)."""
    import torch.distributed as dist
    res = {}
    col = torch.zeros(n_splats, 3, device=dev)
    gathered = torch.empty(world, n_splats, 3, device=dev)
    geo = torch.zeros(n_splats, 11, device=dev)
    f = (world - 1) / world

    def timed(fn, wire_bytes):
        ts = []
        for i in range(iters + 2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            dist.barrier()
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(a.elapsed_time(b))
        ms = median(ts)
        return {"ms": ms, "wire_bytes_per_gpu": wire_bytes, "bus_GBps": wire_bytes / (ms * 1e-3) / 1e9 if ms > 0 else None}

    res["all_gather_colour_gradients"] = timed(lambda: dist.all_gather_into_tensor(gathered.view(-1), col.view(-1)), f * gathered.numel() * 4)
    res["all_reduce_geometric_gradients"] = timed(lambda: dist.all_reduce(geo), 2 * f * geo.numel() * 4)
    return res


def headline_parity(view, n, height, width, use_sh, sh_degree, mean_scale, dev) -> dict:
    """One fwd+bwd of the HIP path on view `view` of the seeded workload compared with the C oracle's outputs and gradients, in
    float and in double (oracle/parity.py), with the oracle's own account of what two fp32 evaluations may differ in
    (tests/helpers.py: assert_parity_explained states the rule; the full-size GPU tests assert it for all 8 views and both
    colour paths): pixels whose threshold decisions sit within the fp32 margin of their threshold (`fragile`), splats blended
    into such a pixel (`splat_flag`: their gradient sums contain it), the first-order effect of float rounding of the splats'
    stored centres (`cond_bound`), radii where the ceil's argument is within rounding of an integer.  Anything else is
    `unexplained`."""
    import math
    from oracle import c_oracle
    from oracle import parity as P
    from oracle import torch_oracle as O
    from splatfields_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    sp = make_splats(n, seed=1234, device=dev, mean_scale=mean_scale)
    names = ["means3D", "scales", "rotations", "opacities", "shs" if use_sh else "colors_precomp"]
    leaf = {k: sp[k].clone().requires_grad_(True) for k in names}
    m2 = torch.zeros_like(leaf["means3D"], requires_grad=True)
    cam = make_camera(view, width, height, device=dev)
    gi, gd, ga = make_upstream_grads(height, width, device=dev)
    rs = GaussianRasterizationSettings(
        image_height=height, image_width=width, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
    color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
        means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"] if use_sh else None,
        colors_precomp=None if use_sh else leaf["colors_precomp"], scales=leaf["scales"], rotations=leaf["rotations"])
    torch.autograd.backward((color, depth, alpha), (gi, gd, ga))
    torch.cuda.synchronize()
    out = {"color": color.detach().cpu(), "depth": depth.detach().cpu(), "alpha": alpha.detach().cpu(), "radii": radii.cpu()}
    g = {k: v.grad.detach().cpu() for k, v in leaf.items()}
    g["means2D"] = m2.grad.detach().cpu()
    sp_cpu = {k: v.detach().cpu() for k, v in sp.items()}
    st = O.settings_from_camera(make_camera(view, width, height), torch.ones(3), sh_degree)
    gcpu = [t.cpu() for t in (gi, gd, ga)]
    res = {"explained": {}}
    total_unexplained = 0
    for prec in ("fp32", "fp64"):
        ref, rg, _ = c_oracle.rasterize(sp_cpu, st, use_sh=use_sh, g_img=gcpu[0], g_depth=gcpu[1], g_alpha=gcpu[2],
                                        threads=min(128, os.cpu_count() or 8), precision=prec, fragile=True,
                                        xy_ulps={"fp64": 4.0, "fp32": 6.0}[prec])   # tests/helpers.py: XY_ULPS
        if prec == "fp32":   # the round-5 figures (no account of fragile pixels), for continuity
            res.update(P.compare(out, g, ref, rg))
        fig = P.compare_flagged(out, g, ref, rg, oracle_precision=prec)
        total_unexplained += fig["unexplained"]
        res["explained"][prec] = {
            "unexplained": fig["unexplained"], "radii": fig["radii"],
            "image_robust_max_rel": fig["image_robust_max_rel"], "image_robust_above_1e-4": fig["image_robust_above_1e-4"],
            "image_conditioning_limited": fig["image_conditioning_limited"],
            "max_err_over_allowance": max(v.get("max_err_over_allowance", 0.0) for v in fig["images"].values()),
            "fragile_pixel_share": fig["images"]["color"]["fragile_share"],
            "fragile_max_abs": max(v["fragile_max_abs"] for v in fig["images"].values()),
            "flagged_splat_share": fig["flagged_splat_share"],
            "gradient_max_on_unflagged_splats": fig["gradient_max_unflagged"],
            "gradient_elements_beyond_1e-3_on_unflagged_splats": sum(v["unexplained"] for v in fig["gradients"].values())}
    res["unexplained"] = total_unexplained
    res["against"] = ("oracle/raster_ref.c, explicit backward, in float (libraster_ref.so: the arithmetic timed as cpu_baseline) and in "
                      "double (libraster_ref64.so): same inputs, view %d, whole image" % view)
    res["tolerance"] = ("north star: <= 1e-4 max relative image error (relative to max(|ref|, 1e-3); 2e-4 against the oracle in float, itself an fp32 evaluation) on pixels whose threshold decisions "
                        "do not sit within the fp32 margin of their threshold, plus the oracle's first-order bound for 4 float ulps (6 against "
                        "the oracle in float, where both sides round) of rounding in the splats' stored screen-space centres; gradient elements beyond 1e-3 of the tensor's maximum only "
                        "on splats blended into a fragile pixel; `unexplained` counts everything outside that, against the oracle in "
                        "float and in double")
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:   # a hard error either way: the line's n_gpus must be what was asked for
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started a different number of ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback for the product path)")
    dev = torch.device("cuda", 0 if args.single_device else local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from splatfields_amd import _lib, rasterizer as rz
    from splatfields_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    from splatfields_amd.view_parallel import allreduce_gradients, sh_gather_step, sh_sharded_step
    import math

    if world > 1 and dist.get_world_size() != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but {dist.get_world_size()} ranks joined the process group")
    lib = _lib.load()
    N, H, W = args.splats, args.height, args.width
    use_sh = args.color == "sh"
    c_in = 12 * 16 if use_sh else 12
    sp = make_splats(N, seed=1234, device=dev, mean_scale=args.mean_scale)
    names = ["means3D", "scales", "rotations", "opacities", "shs" if use_sh else "colors_precomp"]
    params = {k: sp[k].clone().requires_grad_(True) for k in names}
    raw_split = args.inputs == "raw-split" and use_sh
    if raw_split:
        raw = {"means3D": sp["means3D"], "opacity": torch.logit(sp["opacities"].clamp(1e-4, 1 - 1e-4)), "scaling": torch.log(sp["scales"]),
               "rotation": sp["rotations"], "f_dc": sp["shs"][:, :1].contiguous(), "f_rest": sp["shs"][:, 1:].contiguous()}
        params = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    bg = torch.ones(3, device=dev)
    cams = [make_camera(k, W, H, device=dev) for k in range(8)]
    stats = {"R": 0.0, "vis": 0.0, "n": 0, "vis_probe": float(N)}
    means2D = torch.zeros_like(params["means3D"], requires_grad=True)   # only its gradient is used (densification statistics)

    dp_active = (world > 1 or args.force_dp_path) and use_sh
    modes = {"shard": ["shard", "gather", "allreduce"], "gather": ["gather", "allreduce"], "allreduce": ["allreduce"]}[args.dp_mode]
    state = {"mode": modes[0] if dp_active else "allreduce"}

    def one_step_gather(step_idx: int, record: bool):
        step_cams = [cams[(step_idx * world + r) % len(cams)] for r in range(world)]
        seen = {}

        def bwd(vi, color, depth, alpha):
            seen["radii_vis"] = None
            torch.autograd.backward((color, depth, alpha), (gi / world, gd / world, ga / world))

        if state["mode"] == "shard":
            sh_sharded_step(params, step_cams, bg, args.sh_degree, bwd, rank=rank, world=world)
        else:
            sh_gather_step(params, step_cams, bg, args.sh_degree, bwd, rank=rank, world=world)
        if record:
            stats["R"] += rz.LAST_INSTANCES
            stats["vis"] += stats["vis_probe"]
            stats["n"] += 1

    def one_step(step_idx: int, record: bool = False):
        if state["mode"] in ("shard", "gather"):
            return one_step_gather(step_idx, record)
        cam = cams[(step_idx * world + rank) % len(cams)]
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
            bg=bg, scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
            sh_degree=args.sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
        for p in params.values():
            p.grad = None
        means2D.grad = None   # (the tensor itself is allocated once: a zeros_like per step is a 5.6 us fill kernel in the stream)
        if raw_split:
            fwd = lambda: GaussianRasterizer(rs).forward_raw(
                means3D=params["means3D"], means2D=means2D, opacity_logits=params["opacity"], shs=params["f_dc"],
                shs_rest=params["f_rest"], log_scales=params["scaling"], quaternions=params["rotation"])
        else:
            fwd = lambda: GaussianRasterizer(rs).forward_ex(
                means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                shs=params["shs"] if use_sh else None, colors_precomp=None if use_sh else params["colors_precomp"],
                scales=params["scales"], rotations=params["rotations"])
        # the training-loop step that opts into the asynchronous launch: ticket redeemed before the outputs are used
        color, radii, depth, alpha = checked_forward(rz, fwd)
        # loss = sum(color*G_img) + sum(depth*G_depth) + sum(alpha*G_alpha) is linear, so its upstream
        # gradients are the fixed G tensors: feed them directly instead of spending ~15 small PyTorch kernels
        # (mul/sum/add and their backward) on a stand-in loss that is not part of the rasterizer.
        torch.autograd.backward((color, depth, alpha), (gi, gd, ga))
        if world > 1:
            allreduce_gradients(list(params.values()), world)
        if record:
            stats["R"] += rz.LAST_INSTANCES
            stats["vis"] += float((radii > 0).sum().item())
            stats["n"] += 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if dp_active:   # visible splats of one view, for the byte model (the exchange steps do not hand the radii out)
        cam0 = cams[0]
        rs0 = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam0.FoVx * 0.5), tanfovy=math.tan(cam0.FoVy * 0.5), bg=bg,
            scale_modifier=1.0, viewmatrix=cam0.world_view_transform, projmatrix=cam0.full_proj_transform,
            sh_degree=args.sh_degree, campos=cam0.camera_center, prefiltered=False, debug=False)
        with torch.no_grad():
            _, radii0, _ = GaussianRasterizer(rs0)(means3D=sp["means3D"], means2D=torch.zeros_like(sp["means3D"]),
                                                   opacities=sp["opacities"], shs=sp["shs"], scales=sp["scales"],
                                                   rotations=sp["rotations"])
        stats["vis_probe"] = float((radii0 > 0).sum().item())
    while dp_active and state["mode"] != "allreduce":
        # make sure every rank can run this exchange; otherwise all ranks fall back to the next mode together
        ok = torch.ones(1, device=dev)
        try:
            one_step(0)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print(f"[bench] rank {rank}: --dp-mode {state['mode']} failed ({e!r}); falling back", file=sys.stderr)
            ok.zero_()
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() != 0:
            break
        state["mode"] = modes[modes.index(state["mode"]) + 1]
    t_start = time.perf_counter()
    from splatfields_amd.view_parallel import ExchangeStats
    # Before the W warmup steps every view of the cycle is rendered once (untimed, like the library build): the buffer sizes of
    # the facade follow the largest instance count seen, and with W smaller than the number of views the first timed steps
    # would otherwise be the ones that meet a new view -- and pay the allocator's hipMallocs (round 3's driver line, W = 5:
    # one 80 ms step in 20).  A training run is in this state after its first epoch.
    # Round 5: SIX cycles instead of one.  Measured on MI355X (K = 20 steps between synchronisations, same process): the first
    # region after 13 untimed steps takes 538-548 us/step, after 29 steps 530-532, after 61 steps 525 = what every later region
    # takes -- the device needs ~25 ms of this workload (clocks, caches, the allocator's block list) before a 10 ms region
    # measures the steady state.  Nothing is skipped: these are full untimed steps, like the W warm-up steps that follow.
    # Python's cyclic garbage collector is parked from here to the end of the timed region: a generation-2 pass over the
    # interpreter's heap (torch is imported: millions of objects) takes milliseconds and comes about once in ~200 steps of this
    # loop -- one of them inside a 10 ms region is a 15-40 % error (seen in round 5: 0.584 ms wall mean beside a 0.503 median).
    # Collected HERE, in front of the untimed steps: a pause of the host between the warm-up and the timed region lets the
    # device fall idle, and the first steps behind ~50 ms of idleness run 10-25 % slower (measured: 0.572-0.581 instead of 0.51).
    import gc
    gc.collect()
    if os.environ.get("BENCH_KEEP_GC") != "1":
        gc.disable()
    for _ in range(PRE_CYCLES):
        for i in range(len(cams)):
            one_step(i)
    for i in range(args.warmup):
        one_step(i)
    fence()
    # Timed region (task contract): exactly K steps between barrier + synchronize on both sides, max over ranks -> `value`.
    # `ms_per_step` is that wall clock / K, the estimator `value` uses.
    # (Round 5: the per-step event marks moved OUT of the timed region into a pass of their own -- an event record is a
    # barrier packet in the stream, ~5-10 us of GPU time per step at this step length (found when the library's own
    # mid-forward event went: rocprofv3 kernel trace, profiles/r05_*); the timed region now holds the K steps and nothing else.)
    ExchangeStats.reset(False)
    rz.host_sync_counters(reset=True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    fence()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    gc.enable()
    gc.collect()   # (now, not as a surprise in the middle of the next pass)
    host_sync = rz.host_sync_counters()
    # second pass over the same K steps, each bracketed by HIP events on the launch stream (median / min / max of a step) and,
    # for N > 1, with the exchange instrumentation on
    for _ in range(3):   # the collector's pause above left the device idle: three untimed cycles before the events pass
        for i in range(len(cams)):
            one_step(i)
    ExchangeStats.reset(world > 1 or args.force_dp_path)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    stream = torch.cuda.current_stream(dev)
    for i in range(args.steps):
        marks[i].record(stream)
        one_step(args.warmup + i)
        ExchangeStats.end_step()
    marks[args.steps].record(stream)
    fence()
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    med = torch.tensor([median(step_ms)], device=dev, dtype=torch.float64)
    # The same K steps with EVERY forward waiting for its instance count on the host (the plain facade's default, and what [EXT]
    # does): the headline above assumes that every capacity / list-length promise holds (static splats, cameras rendered
    # before: a 100 % hit rate); a loop that meets new counts or first visits pays this figure for those steps.
    prev_async = rz.set_async_forward(False)
    for i in range(len(cams)):
        one_step(i)
    fence()
    t0s = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    fence()
    ms_sync_forward = (time.perf_counter() - t0s) / args.steps * 1e3
    rz.set_async_forward(prev_async)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(med, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # `ms_per_step` = wall clock of the K timed steps / K -- the estimator `value` uses and the task contract names; the median
    # of the per-step HIP-event times rides beside it (ms_per_step_median: robust against one slow step)
    ms_per_step = elapsed / args.steps * 1e3
    ms_median = float(med.item())
    value = world * N * H * W * args.steps / elapsed
    exchange = ExchangeStats.summary() if ExchangeStats.enabled else None
    ExchangeStats.reset(False)

    def progress(msg):
        if rank == 0:
            print(f"[bench] {msg} (+{time.perf_counter() - t_start:.1f} s)", file=sys.stderr, flush=True)

    progress(f"timed {args.steps} steps: wall-clock mean {ms_per_step:.4f} ms/step, median {ms_median:.4f}")
    # ---- per-stage durations with HIP events on the launch stream (same steps, same inputs) ----
    stage_ms = staged_pass(lib, args.steps, lambda i: one_step(args.warmup + i, record=True))
    R = stats["R"] / max(stats["n"], 1)
    vis = stats["vis"] / max(stats["n"], 1)
    b_alg = pipeline_bytes(N, vis, R, H * W, c_in)
    wl = {"splats": N, "width": W, "height": H, "color": args.color, "sh_degree": args.sh_degree, "mean_scale": args.mean_scale}
    tj, traffic_why = recorded_counters("traffic.json", wl)   # PMC bytes of this workload and these kernel sources only
    dom = max(stage_ms, key=stage_ms.get)
    traffic = tj.get(dom) if tj else None
    if traffic is not None and traffic_why is None:
        # the guide calibrates the x 2 FETCH_SIZE correction on wide coalesced streaming reads; the blend
        # kernels' reads are 64-byte record gathers, for which the corrected figure is an UPPER bound of the bytes moved
        traffic_why = ("bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate rocprofv3 --pmc passes (profiles/traffic.json); "
                       "the x 2 read correction of MI355X_MICROARCH.md is calibrated for wide coalesced streaming reads -- for the "
                       "gather-dominated blend kernels it makes this an upper bound")

    out = {
        "metric": "splats*px rasterized/sec (fwd+bwd)", "value": value, "unit": "splat*px/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_step_median": ms_median,
        "ms_per_step_wall_mean": ms_per_step, "ms_per_step_min": min(step_ms), "ms_per_step_max": max(step_ms),
        "ms_per_step_sync_forward": ms_sync_forward,
        "timing": "value = work / wall-clock of the K timed steps (barrier + synchronize on both sides, max over ranks; nothing but the "
                  "steps inside); ms_per_step = that wall-clock / K (= ms_per_step_wall_mean); ms_per_step_median / _min / _max = "
                  "per-step HIP-event times on the launch stream in a SECOND pass over the same steps (an event record per step "
                  "costs the stream several microseconds: they read slightly higher than ms_per_step).  The timed step launches its "
                  "forward without the mid-forward host wait (sr_forward_async) and redeems the ticket before the backward: "
                  "value assumes a 100 % hit rate of the capacity / list-length promises (static splats, cameras rendered before); "
                  "ms_per_step_sync_forward = the same K steps with every forward waiting on the host (the plain facade's default).  "
                  "The splat parameters are the same tensors in every step (synthetic inputs resident in HBM, no optimizer between the "
                  "steps), so part of the 192 MB SH tensor survives in the 256 MB Infinity Cache from step to step: an optimizer "
                  "update between two steps costs the forward preprocess ~3 us of this figure (NOTEBOOK.md section 9.10)",
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{N} synthetic splats (seed 1234), {W}x{H}, {'SH degree %d' % args.sh_degree if use_sh else 'precomputed colours'}, "
                               f"1 view per GPU per step, colour+depth+alpha outputs, fwd+bwd"
                               + ({"shard": ", SH coefficients sharded by splat range: RCCL all-to-all of colours and colour gradients "
                                            "+ all-reduce of the other per-splat gradients",
                                   "gather": ", RCCL all-gather of colour gradients + all-reduce of the other per-splat gradients",
                                   "allreduce": ", RCCL sum all-reduce of per-splat gradients"}[state["mode"]] if world > 1 else ""),
                   "splats": N, "width": W, "height": H, "views_per_step": world,
                   "visible_splats": vis, "tile_instances": R, "inputs": args.inputs, "dp_mode": state["mode"] if (world > 1 or args.force_dp_path) else None},
        "roofline": roofline_block(stage_ms, N, vis, R, H * W, c_in, ms_per_step, ms_median, traffic, traffic_why),
        "roofline_pipeline": {"b_alg_bytes": b_alg, "achieved": b_alg / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                              "frac": b_alg / (ms_per_step * 1e-3) / HBM_PEAK,
                              "frac_from_median": b_alg / (ms_median * 1e-3) / HBM_PEAK,
                              "frac_vs_measured_peak_6.29TBs": b_alg / (ms_per_step * 1e-3) / 6.29e12},
        "stage_ms": stage_ms, "stage_ms_sum": sum(stage_ms.values()),
        # host synchronisations of the forward inside the timed region (sr_debug_counters): forwards that waited for their
        # instance count (sr_forward: first render of a camera) vs forwards launched without waiting (sr_forward_async)
        "host_sync": dict(host_sync, async_forward="opt-in inside the step (checked_forward: async_forward() + resolve_pending())",
                          note="timed steps only; the cameras of the cycle were rendered before, so no forward waits MID-forward: each "
                               "step's ticket is redeemed before its backward (tickets_waited_for counts the redemptions that found "
                               "stage 1 still running)"),
        # algorithmic bytes of every stage / its measured duration, as a fraction of the 8 TB/s HBM peak
        "stage_hbm_frac": {k: (stage_bytes(k, N, vis, R, H * W, c_in) / (v * 1e-3) / HBM_PEAK if v > 0 else None) for k, v in stage_ms.items()},
    }
    progress("stage pass done")
    pairs_blended = None
    if rank == 0 and world == 1 and args.extra_workloads != "none":
        try:
            out["blend_work"] = blend_work_counters(N, W, H, args.mean_scale)
            pairs_blended = out["blend_work"]["pairs_blended"]
        except Exception as e:  # noqa: BLE001 -- a missing counting build must not cost the headline line
            out["blend_work"] = {"error": repr(e)}
        progress("blend_work done")
    out["roofline_valu"] = valu_roofline(wl, stage_ms, pairs_blended)
    if out["roofline_valu"] and out["roofline_valu"].get(dom):
        out["roofline"]["valu_issue_frac"] = out["roofline_valu"][dom].get("valu_issue_frac")
    if world > 1 or args.force_dp_path:
        # what a first run on a multi-GPU node needs to be diagnosed: which scheme ran over how many RCCL ranks, how long the
        # exchange window of a step is, how much of it is covered by local compute, what every GPU puts on the wire
        ex = exchange or {}
        out["exchange"] = {
            "dp_mode": state["mode"], "backend": args.backend if world > 1 else None,
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            "exchange_ms": ex.get("exchange_ms"), "overlap_ms": ex.get("overlap_ms"),
            "compute_ms": (ms_median - ex["exchange_ms"] + ex["overlap_ms"]) if ex else None,
            "exposed_exchange_ms": (ex["exchange_ms"] - ex["overlap_ms"]) if ex else None,
            "wire_bytes_per_gpu": ex.get("wire_bytes_per_gpu"),
            "wire_bytes_per_splat_per_gpu": (ex["wire_bytes_per_gpu"] / N) if ex else None,
            "predicted": None,
            "note": "rank 0's medians over the timed steps; exchange_ms = first collective issued -> last one complete (events on the "
                    "launch stream), overlap_ms = local compute inside that window (the SH-gradient rebuild), wire bytes by the ring "
                    "model (all-reduce 2 (G-1)/G S, all-gather / all-to-all (G-1)/G S)"}
    if world > 1:
        # measured bus bandwidth of the two collectives the exchange is made of, alone on an idle GPU (outside the timed region):
        # what the prediction below should be read against on THIS node
        out["exchange"]["collective_probe"] = collective_probe(N, world, dev)
        ex_ms = out["exchange"].get("exchange_ms")
        if ex_ms:
            out["exchange"]["window_bus_GBps"] = out["exchange"]["wire_bytes_per_gpu"] / (ex_ms * 1e-3) / 1e9
    if world > 1 or args.force_dp_path:
        # what this scheme should cost on an 8-GPU node by link arithmetic (view_parallel.predict_scaling), from this run's own
        # stage times: the first SCALE curve can be checked against it
        from splatfields_amd import view_parallel as vp
        single = stage_ms_sum = sum(stage_ms.values())
        rebuild = (exchange or {}).get("overlap_ms") or 0.09 * (N / 1e6)
        for gsz in sorted({world, 8} - {1}):
            pr = vp.predict_scaling(state["mode"], N, gsz, single, tail_ms=stage_ms["preprocess_backward"],
                                    slices=vp.GATHER_SLICES if state["mode"] == "gather" else 1,
                                    rebuild_ms=(rebuild * gsz / max(world, 1) if state["mode"] == "gather" else rebuild / gsz if state["mode"] == "shard" else 0.0))
            pr["compute_ms_source"] = "sum of this run's per-stage HIP-event times (one view, fwd+bwd, no exchange)"
            out["exchange"].setdefault("predicted_by_world", {})[str(gsz)] = pr
        out["exchange"]["predicted"] = out["exchange"]["predicted_by_world"].get("8")
    if rank == 0 and world == 1 and args.extra_workloads != "none":
        # SURVEY.md 8d "both colour paths" + denser tile lists (trained scenes sit at 5-15 instances per splat)
        extra = [("headline, precomputed colours", N, W, H, False, args.mean_scale),
                 ("dense: 300 k splats, mean scale 0.02", 300_000, 800, 800, True, 0.02),
                 ("dense: 100 k splats, mean scale 0.05", 100_000, 800, 800, True, 0.05)]
        out["other_workloads"] = []
        del params, sp     # the headline's tensors: every side workload starts from a drained device and an empty allocator cache
        for name, n_, w_, h_, sh_, ms_ in extra:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            # 16 warmup steps = two cycles of the 8 views: a buffer size learned from the last view of the first cycle is allocated in
            # the second one, not in the first timed step (one ~80 ms hipMalloc pair in 20 steps made the mean 4.3 ms beside a 0.53 ms median)
            # (round 5: + 32 -- six cycles in all, as for the headline: a 10 ms region needs ~25 ms of the workload in front of it)
            r = time_plain_workload(n_, w_, h_, sh_, ms_, args.sh_degree, 20, 16 + 32, dev)
            r["name"] = name
            out["other_workloads"].append(r)
            progress(f"workload '{name}' done")
        try:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            out["other_workloads"].append(time_reference_pattern(N, W, H, args.sh_degree, 20, 24, dev))
        except Exception as e:  # noqa: BLE001 -- a side measurement must not cost the headline line
            out["other_workloads"].append({"name": "reference call pattern, unmodified render()", "error": repr(e)[:300]})
        progress("reference call pattern done")
    if rank == 0 and world == 1 and args.extra_workloads != "none":
        try:
            out["deform_network"] = deform_network_probe()
        except Exception as e:  # noqa: BLE001 -- a side measurement must not cost the headline line
            out["deform_network"] = {"error": repr(e)[:300]}
        progress("deform network probe done")
    if rank == 0 and world == 1 and args.cpu_baseline != "none":
        from oracle.cpu_baseline import run_cpu_baseline, run_torch_oracle_config0  # the oracle: only the timed CPU baseline
        out["cpu_baseline_torch_config0"] = run_torch_oracle_config0()
        progress("torch oracle config 0 done")
        out["cpu_baseline"] = run_cpu_baseline(N, H, W, use_sh, args.sh_degree, args.cpu_seconds)
        progress("C oracle baseline done")
        # ---- parity of the measured path, on the measured workload, in the same run (north star: <= 1e-4 image error) ----
        # view 0 of the cycle, whole image, all gradients; checker only, after the timed region (two more oracle runs, with the
        # fragile-pixel analysis, in float and in double: a few seconds)
        try:
            out["parity"] = headline_parity(0, N, H, W, use_sh, args.sh_degree, args.mean_scale, dev)
        except Exception as e:  # noqa: BLE001 -- a side measurement must not cost the headline line
            out["parity"] = {"error": repr(e)[:300]}
        progress("parity done")
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
