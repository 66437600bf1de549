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

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md:35 (6.29e12 measured-achievable)


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

    def step(i, record=False):
        cam = cams[i % len(cams)]
        rs = GaussianRasterizationSettings(
            image_height=height, image_width=width, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
            scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, sh_degree=sh_degree,
            campos=cam.camera_center, prefiltered=False, debug=False)
        for p in params.values():
            p.grad = None
        color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
            means3D=params["means3D"], means2D=torch.zeros_like(params["means3D"], requires_grad=True),
            opacities=params["opacities"], shs=params["shs"] if use_sh else None,
            colors_precomp=None if use_sh else params["colors_precomp"], scales=params["scales"], rotations=params["rotations"])
        torch.autograd.backward((color, depth, alpha), (gi, gd, ga))
        if record:
            vis[0] = float((radii > 0).sum().item())

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    ms_per_step = (time.perf_counter() - t0) / steps * 1e3
    lib.sr_profile_enable(1)
    for i in range(steps):
        step(warmup + i, record=(i == steps - 1))
    torch.cuda.synchronize()
    ms = (C.c_double * _lib.PROFILE_STAGES)()
    cnt = (C.c_longlong * _lib.PROFILE_STAGES)()
    lib.sr_profile_collect(ms, cnt)
    lib.sr_profile_enable(0)
    stage_ms = {lib.sr_profile_stage_name(i).decode(): (ms[i] / max(cnt[i], 1)) for i in range(_lib.PROFILE_STAGES)}
    R = float(rz.LAST_INSTANCES)
    c_in = 12 * 16 if use_sh else 12
    b_alg = pipeline_bytes(n, vis[0], R, height * width, c_in)
    binning = stage_ms["scan"] + stage_ms["emit"] + stage_ms["sort_tiles"]
    blend = stage_ms["render_forward"] + stage_ms["render_backward"]
    return {"splats": n, "width": width, "height": height, "mean_scale": mean_scale,
            "color": "sh%d" % sh_degree if use_sh else "precomp", "steps": steps, "ms_per_step": ms_per_step,
            "value": n * height * width / (ms_per_step * 1e-3), "tile_instances": R, "instances_per_splat": R / n,
            "visible_splats": vis[0], "stage_ms": stage_ms, "binning_over_blend": binning / blend if blend > 0 else None,
            "sort_keys_per_s": R / (stage_ms["sort_tiles"] * 1e-3) if stage_ms["sort_tiles"] > 0 else None,
            "roofline_pipeline_frac": b_alg / (ms_per_step * 1e-3) / HBM_PEAK}


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
                    "last contributor; counted on one fwd+bwd of view 1 by the -DSR_BWD_STATS build (tools/bwd_stats.py)"}


def valu_roofline(workload: dict, stage_ms: dict):
    """Issue-side roofline of the two blend kernels from the committed SQ counter passes (profiles/pmc_sq.json, rocprofv3
    --pmc, per-launch averages): wave-level VALU / SALU / LDS / MFMA instruction counts and the kernel cycles per wave-VALU
    instruction per SIMD (kernel cycles = SQ_BUSY_CYCLES / 32 shader engines).  `frac` relates that to the densest issue rate
    measured in this pipeline (3.7 cycles per instruction per SIMD: the forward blend and round 1's backward blend, whose
    VALU pipes are saturated).  The counters are a recorded measurement of this workload; the launch duration next to them
    is the live one."""
    path = os.path.join(ROOT, "profiles", "pmc_sq.json")
    if not os.path.exists(path):
        return None
    try:
        pj = json.load(open(path))
    except Exception:
        return None
    if pj.get("_workload") != workload:
        return None
    out = {"source": pj.get("_source"), "simds": 1024, "issue_bound_cycles_per_valu_inst_per_simd": 3.7}
    for stage in ("render_forward", "render_backward"):
        c = pj.get(stage)
        if not c:
            continue
        cycles = c["SQ_BUSY_CYCLES"] / 32.0            # summed over the 32 shader engines
        out[stage] = {"wave_valu_insts_per_launch": c["SQ_INSTS_VALU"], "wave_salu_insts_per_launch": c["SQ_INSTS_SALU"],
                      "lds_insts_per_launch": c.get("SQ_INSTS_LDS"), "mfma_insts_per_launch": c.get("SQ_INSTS_MFMA"),
                      "kernel_cycles": cycles, "cycles_per_valu_inst_per_simd": cycles * 1024.0 / c["SQ_INSTS_VALU"],
                      "frac": min(1.0, 3.7 / (cycles * 1024.0 / c["SQ_INSTS_VALU"])),
                      "live_avg_launch_ms": stage_ms.get(stage)}
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
            "fused_training_max_rel_grad_diff_parameters", "fused_training_xyz_grad_median_rel_diff"]
    res = {k: d[k] for k in keep if k in d}
    res["name"] = "4-D config: deform network (stand-in of the reference's shapes), PyTorch-ROCm vs fused MLP kernels"
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
        means2D = torch.zeros_like(params["means3D"], requires_grad=True)
        if raw_split:
            color, radii, depth, alpha = GaussianRasterizer(rs).forward_raw(
                means3D=params["means3D"], means2D=means2D, opacity_logits=params["opacity"], shs=params["f_dc"],
                shs_rest=params["f_rest"], log_scales=params["scaling"], quaternions=params["rotation"])
        else:
            color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
                means3D=params["means3D"], means2D=means2D, opacities=params["opacities"],
                shs=params["shs"] if use_sh else None, colors_precomp=None if use_sh else params["colors_precomp"],
                scales=params["scales"], rotations=params["rotations"])
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
    for i in range(args.warmup):
        one_step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
    fence()
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * N * H * W * args.steps / elapsed

    def progress(msg):
        if rank == 0:
            print(f"[bench] {msg} (+{time.perf_counter() - t_start:.1f} s)", file=sys.stderr, flush=True)

    progress(f"timed {args.steps} steps: {ms_per_step:.4f} ms/step")
    # ---- per-stage durations with HIP events on the launch stream (same steps, same inputs) ----
    lib.sr_profile_enable(1)
    for i in range(args.steps):
        one_step(args.warmup + i, record=True)
    torch.cuda.synchronize()
    ms = (C.c_double * _lib.PROFILE_STAGES)()
    cnt = (C.c_longlong * _lib.PROFILE_STAGES)()
    lib.sr_profile_collect(ms, cnt)
    lib.sr_profile_enable(0)
    stage_ms = {lib.sr_profile_stage_name(i).decode(): (ms[i] / max(cnt[i], 1)) for i in range(_lib.PROFILE_STAGES)}
    R = stats["R"] / max(stats["n"], 1)
    vis = stats["vis"] / max(stats["n"], 1)
    dom = max(stage_ms, key=stage_ms.get)
    dom_bytes = stage_bytes(dom, N, vis, R, H * W, c_in)
    dom_bw = dom_bytes / (stage_ms[dom] * 1e-3)
    b_alg = pipeline_bytes(N, vis, R, H * W, c_in)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            same = tj.get("_workload") == {"splats": N, "width": W, "height": H, "color": args.color, "sh_degree": args.sh_degree}
            traffic = tj.get(dom) if same else None  # PMC bytes were collected for that workload only
        except Exception:
            traffic = None

    out = {
        "metric": "splats*px rasterized/sec (fwd+bwd)", "value": value, "unit": "splat*px/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{N} synthetic splats (seed 1234), {W}x{H}, {'SH degree %d' % args.sh_degree if use_sh else 'precomputed colours'}, "
                               f"1 view per GPU per step, colour+depth+alpha outputs, fwd+bwd"
                               + ({"shard": ", SH coefficients sharded by splat range: RCCL all-to-all of colours and colour gradients "
                                            "+ all-reduce of the other per-splat gradients",
                                   "gather": ", RCCL all-gather of colour gradients + all-reduce of the other per-splat gradients",
                                   "allreduce": ", RCCL sum all-reduce of per-splat gradients"}[state["mode"]] if world > 1 else ""),
                   "splats": N, "width": W, "height": H, "views_per_step": world,
                   "visible_splats": vis, "tile_instances": R, "inputs": args.inputs, "dp_mode": state["mode"] if (world > 1 or args.force_dp_path) else None},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": dom_bw / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": dom_bw / HBM_PEAK, "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes,
                     "avg_launch_ms": stage_ms[dom],
                     "note": ("the two blend kernels are VALU-issue-bound, not HBM-bound (rocprofv3 SQ counters in profiles/: 4.6-4.9 "
                              "launch cycles per wave-level VALU instruction per SIMD); `traffic` is L2<->fabric bytes incl. requests "
                              "served by the 256 MB infinity cache; the HBM-bound stages are preprocess / preprocess_backward "
                              "(DESIGN.md section 5)")},
        "roofline_pipeline": {"b_alg_bytes": b_alg, "achieved": b_alg / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                              "frac": b_alg / (ms_per_step * 1e-3) / HBM_PEAK,
                              "frac_vs_measured_peak_6.29TBs": b_alg / (ms_per_step * 1e-3) / 6.29e12},
        "stage_ms": stage_ms,
        # algorithmic bytes of every stage / its measured duration, as a fraction of the 8 TB/s HBM peak
        "stage_hbm_frac": {k: (stage_bytes(k, N, vis, R, H * W, c_in) / (v * 1e-3) / HBM_PEAK if v > 0 else None) for k, v in stage_ms.items()},
    }
    progress("stage pass done")
    wl = {"splats": N, "width": W, "height": H, "color": args.color, "sh_degree": args.sh_degree, "mean_scale": args.mean_scale}
    out["roofline_valu"] = valu_roofline(wl, stage_ms)
    out["roofline"]["note"] = ("the blend kernels are VALU-issue-bound, not HBM-bound: see roofline_valu (SQ counters) and blend_work "
                               "(pairs evaluated / blended); `traffic` is L2<->fabric bytes incl. requests served by the 256 MB "
                               "infinity cache; the HBM-bound stages are preprocess / preprocess_backward (DESIGN.md section 5)")
    if rank == 0 and world == 1 and args.extra_workloads != "none":
        try:
            out["blend_work"] = blend_work_counters(N, W, H, args.mean_scale)
        except Exception as e:  # noqa: BLE001 -- a missing counting build must not cost the headline line
            out["blend_work"] = {"error": repr(e)}
        progress("blend_work done")
    if rank == 0 and world == 1 and args.extra_workloads != "none":
        # SURVEY.md 8d "both colour paths" + denser tile lists (trained scenes sit at 5-15 instances per splat)
        extra = [("headline, precomputed colours", N, W, H, False, args.mean_scale),
                 ("dense: 300 k splats, mean scale 0.02", 300_000, 800, 800, True, 0.02),
                 ("dense: 100 k splats, mean scale 0.05", 100_000, 800, 800, True, 0.05)]
        out["other_workloads"] = []
        for name, n_, w_, h_, sh_, ms_ in extra:
            r = time_plain_workload(n_, w_, h_, sh_, ms_, args.sh_degree, 20, 8, dev)
            r["name"] = name
            out["other_workloads"].append(r)
            progress(f"workload '{name}' done")
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
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
