"""BASELINE.json's full size (1 M splats, 800x800): whole images and all gradients against the fp32 C oracle (OpenMP on the
box's host cores), plus size-independent properties: determinism, linearity of the backward, splat-order permutation
invariance, and a 6 M-splat run."""
import pytest
import torch

from tests.helpers import make_scene, run_hip

pytestmark = pytest.mark.gpu

# Share of pixels that may exceed 1e-4 relative error (two fp32 evaluations of alpha >= 1/255 / T >= 1e-4 flip on a few pixels):
# observed on MI355X at the headline size 2e-5 .. 9e-5 per image (gpurun_out/parity_observed.jsonl, bench.py's `parity` block);
# the bound is ~10 x that (rounds 1-4 allowed 5e-3).
IMAGE_FLIP_SHARE = 1e-3


@pytest.fixture(scope="module")
def scene():
    return make_scene(1_000_000, 800, 800)


def test_bitwise_determinism_and_linearity(hip_device, scene):
    sp, cam, st, grads = scene
    o1, g1 = run_hip(sp, st, grads, hip_device)
    o2, g2 = run_hip(sp, st, grads, hip_device)
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k  # no float atomics anywhere: bit-reproducible
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    # backward is linear in the upstream gradient; scaling by 2 is exact in binary floating point
    _, g3 = run_hip(sp, st, tuple(2.0 * g for g in grads), hip_device)
    for k in g1:
        assert torch.equal(2.0 * g1[k], g3[k]), k
    assert (o1["alpha"] >= 0).all() and (o1["alpha"] <= 1).all() and torch.isfinite(o1["color"]).all()
    assert all(torch.isfinite(v).all() for v in g1.values())


def test_permuting_the_splats_permutes_the_results(hip_device, scene):
    sp, cam, st, grads = scene
    o1, g1 = run_hip(sp, st, grads, hip_device)
    perm = torch.randperm(1_000_000, generator=torch.Generator().manual_seed(5))
    spp = {k: v[perm] for k, v in sp.items()}
    o2, g2 = run_hip(spp, st, grads, hip_device)
    assert torch.equal(o1["radii"][perm], o2["radii"])
    # depth ties between distinct splats are broken by index, so a handful of pixels may reorder
    same = (o1["color"] == o2["color"]).float().mean().item()
    assert same > 0.999  # ~1.6k entries per tile drawn from ~8M distinct fp32 depths: a few hundred ties
    assert torch.allclose(o1["color"], o2["color"], atol=2e-2)
    a, b = g1["opacities"][perm], g2["opacities"]
    assert (a == b).float().mean().item() > 0.99
    assert torch.allclose(a, b, atol=1e-2 * a.abs().max().item())


def test_window_against_c_oracle(hip_device, scene):
    """All 1 M splats binned by the C oracle, a 5x5-tile window blended: compares the HIP image there."""
    from oracle import c_oracle
    sp, cam, st, grads = scene
    o1, _ = run_hip(sp, st, grads, hip_device)
    win = (22, 22, 27, 27)
    ref, _, _ = c_oracle.rasterize(sp, st, use_sh=True, tile_window=win, threads=8)
    sl = (slice(None), slice(win[1] * 16, win[3] * 16), slice(win[0] * 16, win[2] * 16))
    for k in ("color", "depth", "alpha"):
        a, b = o1[k][sl].double(), ref[k][sl].double()
        rel = ((a - b).abs() / b.abs().clamp_min(1e-3))
        assert rel.median().item() < 1e-5
        assert (rel > 1e-4).float().mean().item() < IMAGE_FLIP_SHARE, k  # both fp32: a few pixels flip a threshold decision
        assert (a - b).abs().max().item() <= 2e-2 * max(1.0, b.abs().max().item()), k
    assert torch.equal(o1["radii"], ref["radii"])


@pytest.mark.parametrize("view", range(8))
def test_headline_config_images_and_all_gradients_against_c_oracle(hip_device, view):
    """BASELINE.json's metric configuration (1 M splats, 800x800, SH degree 3), EVERY view bench.py cycles through and BOTH
    colour paths: the WHOLE image and ALL six gradient tensors against the C oracle in double and in float (OpenMP on the
    box's host cores: a few seconds each), with the oracle's account of what two fp32 evaluations may differ in
    (tests/helpers.py: assert_parity_explained) -- nothing unexplained:
      * non-fragile pixels <= 1e-4 relative (+ the oracle's bound for float rounding of the splats' stored centres),
      * gradient elements beyond 1e-3 of the tensor's maximum only on splats blended into a fragile pixel,
      * radii equal except where the ceil's argument is within rounding of an integer."""
    import os
    from tests.helpers import assert_parity_explained
    if (os.cpu_count() or 1) < 32 and view not in (0, 7):
        # four oracle passes per view at 1 M splats take seconds on the GPU boxes' 256 host threads and minutes on a handful
        pytest.skip("host with < 32 hardware threads: the full-size oracle comparison runs for views 0 and 7 only")
    sp, cam, st, grads = make_scene(1_000_000, 800, 800, view=view)
    for use_sh in (True, False):
        out, g = run_hip(sp, st, grads, hip_device, use_sh=use_sh)
        assert_parity_explained(out, g, sp, st, grads, use_sh=use_sh, tag=f"headline {'sh' if use_sh else 'precomputed colours'} view {view}")


def test_six_million_splats_chunked_count_matrix(hip_device):
    """Beyond 262144 splats a count-matrix chunk spans several 256-splat sub-batches; at 6 M splats it spans 23, and the
    per-splat arrays pass 1 GB: sizes, 64-bit indexing and the chunk arithmetic at scale.  Properties only (the oracles
    would need minutes): bit-reproducible, finite, alpha in [0,1], every splat's radius consistent with a second view of
    the same data, gradients of culled splats exactly zero."""
    n = 6_000_000
    sp, cam, st, grads = make_scene(n, 800, 800, seed=77)
    o1, g1 = run_hip(sp, st, grads, hip_device)
    o2, g2 = run_hip(sp, st, grads, hip_device)
    for k in o1:
        assert torch.equal(o1[k], o2[k]), k
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    assert (o1["alpha"] >= 0).all() and (o1["alpha"] <= 1).all() and torch.isfinite(o1["color"]).all()
    assert all(torch.isfinite(v).all() for v in g1.values())
    culled = o1["radii"] == 0
    assert culled.any() and (~culled).sum() > 0.9 * n
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "means2D"):
        assert (g1[k][culled] == 0).all(), k
    assert g1["shs"].abs().sum() > 0 and g1["means3D"].abs().sum() > 0
