"""distCUDA2 counterpart against an exact CPU k-NN (scipy cKDTree): the reference calls it once at
initialisation (scene/gaussian_model.py:105) to get the mean squared distance to the 3 nearest neighbours."""
import numpy as np
import pytest
import torch
from scipy.spatial import cKDTree

pytestmark = pytest.mark.gpu


def oracle_mean_dist2(pts: np.ndarray) -> np.ndarray:
    n = len(pts)
    if n == 0:
        return np.zeros(0, np.float32)
    k = min(4, n)
    d, idx = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=k)
    d = d.reshape(n, k); idx = idx.reshape(n, k)
    out = np.zeros(n)
    for i in range(n) if n < 8 else []:
        others = [dd for dd, j in zip(d[i], idx[i]) if j != i][:3]
        out[i] = sum(x * x for x in others) / 3.0
    if n >= 8:
        # self is one of the zero-distance hits; drop exactly one zero per row (duplicates keep theirs)
        self_pos = np.argmax(idx == np.arange(n)[:, None], axis=1)
        has_self = (idx == np.arange(n)[:, None]).any(axis=1)
        d2 = d ** 2
        d2[np.arange(n), np.where(has_self, self_pos, k - 1)] = 0.0
        out = d2.sum(axis=1) / 3.0
    return out


@pytest.mark.parametrize("n,kind", [(1, "uniform"), (2, "uniform"), (3, "uniform"), (5, "uniform"), (1000, "uniform"),
                                     (20000, "clustered"), (300000, "uniform"), (5000, "duplicates"), (4000, "planar")])
def test_matches_exact_knn(hip_device, n, kind):
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    pts = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    if kind == "clustered":
        centres = rng.uniform(-1, 1, size=(20, 3))
        pts = (centres[rng.integers(0, 20, n)] + 0.01 * rng.standard_normal((n, 3))).astype(np.float32)
        pts[:50] = rng.uniform(-30, 30, size=(50, 3))  # far outliers stretch the grid
    if kind == "duplicates":
        pts[1000:2000] = pts[:1000]
    if kind == "planar":
        pts[:, 2] = 0.25
    got = distCUDA2(torch.from_numpy(pts).to(hip_device)).cpu().numpy()
    ref = oracle_mean_dist2(pts)
    assert got.shape == (n,)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-12)


def test_initial_scales_as_the_reference_computes_them(hip_device):
    """scene/gaussian_model.py:105-109: scales = log(sqrt(clamp_min(distCUDA2(points), 1e-7)))"""
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(10000, 3, generator=g).to(hip_device)
    dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
    assert scales.shape == (10000, 3) and torch.isfinite(scales).all()
    assert abs(torch.exp(scales).mean().item() - 0.045) < 0.02  # ~ N^(-1/3) spacing of a unit cube
