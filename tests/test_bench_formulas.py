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


def test_roofline_block_carries_the_pipeline_figure_and_the_real_bound():
    """`roofline` (the object the driver records): dominant stage in HBM terms as the contract defines it, what bounds that stage,
    and the north star's pipeline figure B_alg / ms_per_step inside the same object."""
    stage_ms = {"preprocess": 0.07, "scan": 0.05, "emit": 0.03, "sort_tiles": 0.04, "render_forward": 0.11, "render_backward": 0.24,
                "preprocess_backward": 0.09}
    n, vis, R, hw, c_in = 1_000_000, 960_000.0, 2_290_000.0, 640_000, 192
    rf = bench.roofline_block(stage_ms, n, vis, R, hw, c_in, ms_per_step=0.64, ms_median=0.63, traffic=5.0e8, traffic_why=None,
                              valu={"render_backward": {"valu_issue_frac": 0.3}})
    assert rf["kernel"] == "render_backward" and rf["bound"] == "valu" and rf["valu_issue_frac"] == 0.3 and rf["traffic"] == 5.0e8
    want = bench.stage_bytes("render_backward", n, vis, R, hw, c_in) / 0.24e-3
    assert abs(rf["achieved"] - want / 1e9) < 1e-9 and abs(rf["frac"] - want / 8e12) < 1e-12 and rf["peak"] == 8000.0
    b_alg = bench.pipeline_bytes(n, vis, R, hw, c_in)
    assert rf["pipeline_b_alg_bytes"] == b_alg
    assert abs(rf["pipeline_frac"] - b_alg / 0.64e-3 / 8e12) < 1e-15 and abs(rf["pipeline_frac_vs_6.29"] - b_alg / 0.64e-3 / 6.29e12) < 1e-15
    assert abs(rf["pipeline_frac_from_median"] - b_alg / 0.63e-3 / 8e12) < 1e-15
    assert rf["stage_bound"]["preprocess"] == "hbm" and rf["stage_bound"]["emit"] == "latency" and set(rf["stage_bound"]) == set(stage_ms)
    # an HBM-bound stage dominating (tiny images, huge clouds) is labelled as such
    rf2 = bench.roofline_block(dict(stage_ms, preprocess_backward=0.5), n, vis, R, hw, c_in, 1.0, 1.0)
    assert rf2["kernel"] == "preprocess_backward" and rf2["bound"] == "hbm" and "valu_issue_frac" not in rf2


def test_roofline_valu_is_reproducible_from_the_committed_counters():
    """bench.py's issue-side roofline comes from profiles/pmc_sq.json (rocprofv3 SQ counter passes): pure arithmetic against
    hardware ceilings (2 cycles per wave-level VALU instruction per SIMD-32, 157.3 TFLOP/s fp32), nothing self-referential."""
    import json
    pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json")))
    wl = dict(pj["_workload"])
    stage_ms = {"render_forward": 0.12, "render_backward": 0.28}
    r = bench.valu_roofline(wl, stage_ms, pairs_blended=36.9e6, check_hash=False)
    assert r is not None and r["simds"] == 1024 and r["valu_issue_cycles_per_wave_inst"] == 2.0
    for stage in ("render_forward", "render_backward"):
        s, c = r[stage], pj[stage]
        cycles = c["SQ_BUSY_CYCLES"] / 32.0
        assert abs(s["kernel_cycles"] - cycles) < 1e-6
        assert abs(s["valu_issue_frac"] - c["SQ_INSTS_VALU"] * 2.0 / 1024.0 / cycles) < 1e-12 and s["frac"] == s["valu_issue_frac"]
        assert 0.0 < s["valu_issue_frac"] < 1.0
        sh = s["wave_cycle_shares"]
        assert abs(sh["parked"] - c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) < 1e-12
        assert 0.9 < sh["parked"] + sh["issue_stalled"] + sh["issuing"] < 1.1   # the three buckets partition the wave cycles
        flop = {"render_forward": 30.0, "render_backward": 90.0}[stage]
        assert abs(s["useful_flop_frac"] - 36.9e6 * flop / (stage_ms[stage] * 1e-3) / 157.3e12) < 1e-12
    # since round 4 neither blend kernel issues an MFMA (fp32 MFMAs and VALU instructions do not overlap on a gfx950 SIMD)
    assert r["render_backward"]["mfma_insts_per_launch"] == 0 and r["render_forward"]["mfma_insts_per_launch"] == 0
    # another workload has no recorded counters
    assert bench.valu_roofline(dict(wl, splats=123), {}, check_hash=False) is None


def test_recorded_counters_are_refused_when_the_kernel_sources_changed(tmp_path, monkeypatch):
    """profiles/traffic.json and pmc_sq.json carry the hash of the kernel sources they were measured on (build.source_hash);
    bench.py refuses them for other sources instead of quoting stale numbers."""
    import json
    from splatfields_amd.build import source_hash
    wl = {"splats": 1000, "width": 64, "height": 64, "color": "sh", "sh_degree": 3, "mean_scale": None}
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    good = {"_workload": dict(wl), "_source_hash": source_hash(), "render_backward": 123.0}
    (prof / "traffic.json").write_text(json.dumps(good))
    pj, why = bench.recorded_counters("traffic.json", wl)
    assert why is None and pj["render_backward"] == 123.0
    (prof / "traffic.json").write_text(json.dumps(dict(good, _source_hash="0123456789abcdef")))
    pj, why = bench.recorded_counters("traffic.json", wl)
    assert pj is None and "sources changed" in why
    pj, why = bench.recorded_counters("traffic.json", dict(wl, splats=7))
    assert pj is None and "another workload" in why
    assert bench.recorded_counters("missing.json", wl) == (None, "no recorded counters")
    # a stale SQ file turns into an explicit marker, not into numbers
    (prof / "pmc_sq.json").write_text(json.dumps({"_workload": dict(wl), "_source_hash": "0" * 16, "render_forward": {}}))
    assert "stale" in bench.valu_roofline(wl, {})


def test_median_of_per_step_times():
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5 and bench.median([]) == 0.0
    # one slow step (a box hiccup) moves the mean, not the median
    xs = [0.70] * 19 + [1.40]
    assert bench.median(xs) == 0.70 and sum(xs) / len(xs) > 0.73


def test_source_hash_ignores_comments_and_layout(tmp_path, monkeypatch):
    """The stamp that ties recorded counters to the kernels (build.source_hash) is taken over the code only: editing a comment or
    re-indenting must not invalidate a profile, changing a token must."""
    from splatfields_amd import build
    a = "int f(int x) {\n    // add one\n    return x + 1;   /* done */\n}\nconst char* s = \"// not a comment\";\n"
    b = "int f(int x) {\n\n  // a different remark\n  return x + 1;\n}\nconst char* s = \"// not a comment\";\n"
    c = a.replace("x + 1", "x + 2")
    assert build.strip_comments(a) == build.strip_comments(b) != build.strip_comments(c)
    assert "// not a comment" in build.strip_comments(a)
    src = tmp_path / "csrc"
    src.mkdir()
    monkeypatch.setattr(build, "CSRC", src)
    monkeypatch.setattr(build, "SOURCES", ["k.hip"])
    monkeypatch.setattr(build, "HEADERS", [])
    hashes = []
    for text in (a, b, c):
        (src / "k.hip").write_text(text)
        hashes.append(build.source_hash())
    assert hashes[0] == hashes[1] != hashes[2]


def test_rank_count_mismatch_is_a_hard_error():
    """--gpus N must be the number of ranks that run: a launcher that started another number is refused before anything is timed
    (checked ahead of the device probe, so this runs without a GPU)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for gpus, ws in (("2", "1"), ("1", "2"), ("4", "2")):
        env = dict(os.environ, WORLD_SIZE=ws, RANK="0", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", gpus], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode != 0 and f"--gpus {gpus} but WORLD_SIZE={ws}" in r.stderr, (gpus, ws, r.stderr[-500:])
