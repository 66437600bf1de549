"""Pure arithmetic of the view-parallel step: the wire-byte model of the three exchange schemes, the link-arithmetic prediction
bench.py reports for an 8-GPU node (`exchange.predicted`), and the row ranges of the sliced backward -- no GPU, no process group."""
import pytest

from splatfields_amd.rasterizer import slice_ranges
from splatfields_amd.view_parallel import GATHER_SLICES, XGMI_LINK_GBPS, XGMI_LINKS, predict_scaling, wire_bytes_per_gpu


def test_wire_bytes_match_the_design_table():
    # bytes per splat on the wire of each GPU, 8 ranks, SH degree 3 stored (16 coefficients)
    n = 1_000_000
    assert wire_bytes_per_gpu("allreduce", n, 8) / n == pytest.approx(2 * 7 / 8 * 236)          # 413
    assert wire_bytes_per_gpu("gather", n, 8) / n == pytest.approx(7 / 8 * 96 + 2 * 7 / 8 * 44)  # 161
    assert wire_bytes_per_gpu("shard", n, 8) / n == pytest.approx(2 * 7 / 8 * 12 + 2 * 7 / 8 * 44)  # 98
    # two ranks (the gloo tests of bench.py's line): 56 / 236 / 56 B per splat
    assert wire_bytes_per_gpu("gather", n, 2) / n == pytest.approx(0.5 * 24 + 44)
    assert wire_bytes_per_gpu("allreduce", n, 2) / n == pytest.approx(236)
    assert wire_bytes_per_gpu("gather", n, 1) == 0


def test_prediction_follows_the_stated_timeline():
    n, c, tail, reb = 1_000_000, 0.64, 0.094, 0.09
    p = predict_scaling("gather", n, 8, c, tail_ms=tail, slices=4, rebuild_ms=reb)
    ex = 161e6 / (XGMI_LINKS * XGMI_LINK_GBPS * 1e9) * 1e3
    assert p["link_peak"]["exchange_ms"] == pytest.approx(ex)
    exposed = max(reb, ex - tail * 0.75 + reb / 4)
    assert p["link_peak"]["exposed_exchange_ms"] == pytest.approx(exposed) == pytest.approx(p["expected_exposed_exchange_ms"])
    assert p["link_peak"]["speedup"] == pytest.approx(8 * c / (c + exposed)) == pytest.approx(p["expected_speedup"])
    # the sliced exchange is what takes the default scheme past 6 x at the link peak; unsliced it stays below
    assert p["expected_speedup"] >= 6.0
    assert predict_scaling("gather", n, 8, c, tail_ms=tail, slices=1, rebuild_ms=reb)["expected_speedup"] < 6.0
    # slower wires, smaller speed-up; the plain all-reduce is the worst scheme, 1 GPU predicts no exchange
    assert p["rccl_typical"]["speedup"] < p["one_way_peak"]["speedup"] < p["link_peak"]["speedup"] <= 8.0
    assert predict_scaling("allreduce", n, 8, c)["expected_speedup"] < p["expected_speedup"]
    one = predict_scaling("gather", n, 1, c, tail_ms=tail, slices=4, rebuild_ms=reb)
    assert one["link_peak"]["exchange_ms"] == 0 and one["expected_speedup"] == pytest.approx(1.0)
    assert GATHER_SLICES >= 2


def test_slice_ranges_tile_the_cloud_in_granules_of_256():
    for n, k in ((1_000_000, 4), (6000, 4), (6000, 3), (100, 4), (1024, 4), (1025, 1), (257, 8), (300_001, 5)):
        r = slice_ranges(n, k)
        assert r[0][0] == 0 and r[-1][1] == n and len(r) <= max(k, 1)
        assert all(lo % 256 == 0 and lo < hi for lo, hi in r)
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert slice_ranges(0, 4) == []
    assert slice_ranges(1_000_000, 4)[0] == (0, 250_112)
