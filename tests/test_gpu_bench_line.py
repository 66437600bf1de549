"""The JSON line of bench.py, as the driver will see it: schema of the 1-GPU line (median timing, roofline blocks) and of the
N > 1 line (exchange diagnostics), the latter through two gloo ranks sharing the one GPU of the test box
(`--backend gloo --single-device`): the control flow, the step functions and the instrumentation are the ones a RCCL run uses."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SMALL = ["--splats", "20000", "--width", "256", "--height", "192", "--steps", "4", "--warmup", "2", "--cpu-baseline", "none",
         "--extra-workloads", "none"]


def _last_json(text):
    return json.loads([l for l in text.strip().splitlines() if l.startswith("{")][-1])


def test_single_gpu_line_has_median_timing_and_roofline_blocks(hip_device):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "ms_per_step_wall_mean",
              "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "roofline_pipeline", "stage_ms"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    # ms_per_step and value come from the same wall-clock bracket (the task contract's K timed steps)
    assert d["ms_per_step"] == d["ms_per_step_wall_mean"]
    assert abs(d["value"] - 20000 * 256 * 192 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["kernel"] in d["stage_ms"] and rf["bound"] == {"preprocess": "hbm", "preprocess_backward": "hbm", "render_forward": "valu",
                                                            "render_backward": "valu", "scan": "latency", "emit": "latency",
                                                            "sort_tiles": "valu"}[rf["kernel"]]
    assert rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] is None   # counters are recorded for the headline workload only
    # the north star's figure lives inside `roofline` (what the driver records), computed from ms_per_step
    assert abs(rf["pipeline_frac"] - rf["pipeline_b_alg_bytes"] / (d["ms_per_step"] * 1e-3) / 8e12) < 1e-12
    assert abs(rf["pipeline_frac"] - d["roofline_pipeline"]["frac"]) < 1e-12 and rf["pipeline_frac_vs_6.29"] > rf["pipeline_frac"]
    assert set(rf["stage_bound"]) == set(d["stage_ms"])
    assert "exchange" not in d


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["gather", "allreduce", "shard"])
def test_two_rank_line_carries_the_exchange_diagnostics(hip_device, mode):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
           "--dp-mode", mode] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["views_per_step"] == 2
    ex = d["exchange"]
    for k in ("dp_mode", "rccl_ranks", "exchange_ms", "overlap_ms", "compute_ms", "exposed_exchange_ms", "wire_bytes_per_gpu",
              "wire_bytes_per_splat_per_gpu"):
        assert k in ex, k
    assert ex["dp_mode"] == mode and ex["rccl_ranks"] == 2
    # the link-arithmetic prediction for an 8-GPU node rides in the same block (view_parallel.predict_scaling)
    pr = ex["predicted"]
    for k in ("wire_bytes_per_gpu", "links", "link_GBps", "compute_ms", "expected_exposed_exchange_ms", "expected_speedup", "link_peak",
              "one_way_peak", "rccl_typical", "model"):
        assert k in pr, k
    assert pr["world"] == 8 and pr["links"] == 7 and pr["dp_mode"] == mode
    assert abs(pr["link_peak"]["exchange_ms"] - pr["wire_bytes_per_gpu"] / (7 * 153e9) * 1e3) < 1e-9
    assert 1.0 < pr["rccl_typical"]["speedup"] <= pr["one_way_peak"]["speedup"] <= pr["link_peak"]["speedup"] <= 8.0
    assert "2" in ex["predicted_by_world"]
    # the measured bus bandwidth of the exchange's two collectives rides beside the prediction
    probe = ex["collective_probe"]
    assert probe["all_gather_colour_gradients"]["bus_GBps"] > 0 and probe["all_reduce_geometric_gradients"]["bus_GBps"] > 0
    assert probe["all_gather_colour_gradients"]["wire_bytes_per_gpu"] == 0.5 * 2 * 20000 * 12
    assert ex["window_bus_GBps"] > 0
    assert ex["exchange_ms"] > 0.0 and 0.0 <= ex["overlap_ms"] <= ex["exchange_ms"]
    assert abs(ex["compute_ms"] - (d["ms_per_step_median"] - ex["exchange_ms"] + ex["overlap_ms"])) < 1e-9
    # ring model at 2 ranks: all-reduce of S moves S per GPU, all-gather / all-to-all of S in total S / 2
    n, geo = 20000, 44.0   # means3D 12 + scales 12 + rotations 16 + opacity 4 bytes per splat
    # gather: all-gather of 2 x 12 B/splat -> 12; shard: two all-to-alls of 12 B/splat -> 6 + 6; plain: 236 B/splat in one buffer
    want = {"gather": n * (12 + geo), "allreduce": n * (192 + geo), "shard": n * (6 + 6 + geo)}[mode]
    assert abs(ex["wire_bytes_per_gpu"] - want) <= 0.02 * want, (ex["wire_bytes_per_gpu"], want)


def test_bench_launches_its_own_ranks_when_started_without_a_launcher(hip_device):
    """`python bench.py --gpus 2` with no torch.distributed.run around it and no WORLD_SIZE in the environment must start two
    ranks itself and report n_gpus == 2 (round 4's bench silently ran one rank and printed n_gpus: 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["exchange"]["rccl_ranks"] == 2 and d["config"]["views_per_step"] == 2
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1   # ONE line, from rank 0
    # the timed steps render cameras that were rendered before: no forward waits on the host
    assert d["host_sync"]["forward_host_waits"] == 0 and d["host_sync"]["async_forwards"] >= 4
