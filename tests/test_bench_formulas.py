"""bench.py's algorithmic-byte model -- pure arithmetic, no GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_pipeline_bytes_matches_the_worked_example_of_the_survey():
    # N = V_vis = 1 M, R = 4 M, 800x800 -> 1.03 GB (precomputed colours) / 1.57 GB (SH degree 3)
    n, r, hw = 1_000_000, 4_000_000, 800 * 800
    assert abs(bench.pipeline_bytes(n, n, r, hw, 12) / 1e9 - 1.03) < 0.01
    assert abs(bench.pipeline_bytes(n, n, r, hw, 192) / 1e9 - 1.57) < 0.01
    # A = 3 (44 + C_in) + 16 per splat
    assert bench.pipeline_bytes(1, 0, 0, 0, 12) == 184 and bench.pipeline_bytes(1, 0, 0, 0, 192) == 724


def test_stage_bytes_are_positive_and_scale_with_their_units():
    for stage in ("preprocess", "scan", "emit", "sort_tiles", "render_forward", "render_backward", "preprocess_backward"):
        a = bench.stage_bytes(stage, 1000, 900, 2500, 64 * 64, 192)
        b = bench.stage_bytes(stage, 2000, 1800, 5000, 2 * 64 * 64, 192)
        assert a > 0 and abs(b / a - 2.0) < 1e-9, stage
    # the blend kernels move 44 / 88 bytes per instance + 28 bytes per pixel
    assert bench.stage_bytes("render_forward", 0, 0, 10, 0, 12) == 440 and bench.stage_bytes("render_backward", 0, 0, 10, 0, 12) == 880
    assert bench.stage_bytes("render_forward", 0, 0, 0, 10, 12) == 280


def test_roofline_valu_is_reproducible_from_the_committed_counters():
    """bench.py's issue-side roofline comes from profiles/pmc_sq.json (rocprofv3 SQ counter passes): pure arithmetic."""
    import json
    wl = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json")))["_workload"]
    r = bench.valu_roofline(wl, {"render_forward": 0.12, "render_backward": 0.28})
    assert r is not None and r["simds"] == 1024
    for stage in ("render_forward", "render_backward"):
        s = r[stage]
        assert s["wave_valu_insts_per_launch"] > 1e7 and s["kernel_cycles"] > 1e5
        assert abs(s["cycles_per_valu_inst_per_simd"] - s["kernel_cycles"] * 1024 / s["wave_valu_insts_per_launch"]) < 1e-9
        assert 0.0 < s["frac"] <= 1.0
    # the entry-per-lane backward issues fewer instructions than the forward has per launch x 2, at a lower issue rate
    assert r["render_backward"]["mfma_insts_per_launch"] > 0 and r["render_forward"]["mfma_insts_per_launch"] == 0
    # another workload has no recorded counters
    assert bench.valu_roofline(dict(wl, splats=123), {}) is None
