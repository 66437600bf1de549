"""The two CPU oracles check each other, the committed golden scenes pin both, and the fp64 oracle's
autograd gradients are verified against finite differences."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import torch_oracle as O
from tests.helpers import grad_error, image_errors, make_scene

G = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    z = np.load(os.path.join(G, name + ".npz"))
    sp = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("in_")}
    H, W, deg, use_sh = [int(x) for x in z["meta"]]
    st = O.OracleSettings(H, W, float(z["tanfov"][0]), float(z["tanfov"][1]), torch.tensor(z["bg"]), float(z["scale_modifier"]),
                          torch.tensor(z["viewmatrix"]), torch.tensor(z["projmatrix"]), deg, torch.tensor(z["campos"]))
    grads = tuple(torch.tensor(z[k]) for k in ("g_img", "g_depth", "g_alpha"))
    return z, sp, st, grads, bool(use_sh)


@pytest.mark.parametrize("name", ["tiny_sh3", "tiny_rgb", "tiny_sh1_mod"])
def test_torch_oracle_reproduces_golden(name):
    z, sp, st, grads, use_sh = load_golden(name)
    out, gr = O.fwd_bwd(sp, st, *grads, use_sh=use_sh, dtype=torch.float64)
    np.testing.assert_allclose(out.color.detach().numpy(), z["out_color"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out.depth.detach().numpy(), z["out_depth"], rtol=1e-10, atol=1e-12)
    np.testing.assert_array_equal(out.radii.numpy(), z["out_radii"])
    assert out.num_rendered == int(z["num_rendered"])
    for k, v in gr.items():
        np.testing.assert_allclose(v.numpy(), z["grad_" + k], rtol=1e-8, atol=1e-14)


@pytest.mark.parametrize("name", ["tiny_sh3", "tiny_rgb", "tiny_sh1_mod"])
def test_c_oracle_matches_golden(name):
    z, sp, st, grads, use_sh = load_golden(name)
    out, gr, nr = c_oracle.rasterize(sp, st, use_sh=use_sh, g_img=grads[0], g_depth=grads[1], g_alpha=grads[2])
    fragile = torch.tensor(z["fragile"])
    assert nr == int(z["num_rendered"])
    assert torch.equal(out["radii"], torch.tensor(z["out_radii"]))
    for k in ("color", "depth", "alpha"):
        robust, frag = image_errors(out[k], torch.tensor(z["out_" + k]), fragile)
        assert robust < 1e-4 and frag < 2e-2, (k, robust, frag)
    for k, v in gr.items():
        assert grad_error(v, torch.tensor(z["grad_" + k])) < 2e-3, k


def test_c_oracle_vs_torch_oracle_10k_config0():
    """BASELINE.json configs[0]: 10k random Gaussians, 1 camera 256x256 (CPU plumbing case)."""
    sp, cam, st, grads = make_scene(10000, 256, 256)
    out, gr, nr = c_oracle.rasterize(sp, st, use_sh=True, g_img=grads[0], g_depth=grads[1], g_alpha=grads[2])
    ref, gref = O.fwd_bwd(sp, st, *grads, use_sh=True, dtype=torch.float64)
    assert nr == ref.num_rendered
    assert int((out["radii"] != ref.radii).sum()) == 0
    for k, r in (("color", ref.color), ("depth", ref.depth), ("alpha", ref.alpha)):
        robust, frag = image_errors(out[k], r.detach(), ref.fragile)
        assert robust < 1e-4 and frag < 2e-2, (k, robust, frag)
    for k in gr:
        assert grad_error(gr[k], gref[k]) < 5e-3, k


def test_c_oracle_thread_count_and_tile_window():
    sp, cam, st, grads = make_scene(3000, 96, 80, mean_scale=0.03)
    full, _, _ = c_oracle.rasterize(sp, st, use_sh=True, threads=1)
    par, _, _ = c_oracle.rasterize(sp, st, use_sh=True, threads=4)
    assert torch.equal(full["color"], par["color"])  # forward is deterministic regardless of threads
    win, _, _ = c_oracle.rasterize(sp, st, use_sh=True, tile_window=(1, 2, 4, 4), threads=2)
    assert torch.equal(win["color"][:, 32:64, 16:64], full["color"][:, 32:64, 16:64])


def test_autograd_gradients_against_finite_differences():
    """fp64 central differences on a tiny scene with large margins from every threshold."""
    sp, cam, st, grads = make_scene(12, 24, 24, seed=21, mean_scale=0.25, sh_degree=2)
    sp = {k: v.to(torch.float64) for k, v in sp.items()}
    sp["opacities"] = sp["opacities"] * 0.5 + 0.2
    gi, gd, ga = [g.to(torch.float64) * 1e3 for g in grads]

    def loss_of(d):
        out = O.rasterize(d["means3D"], None, d["opacities"], shs=d["shs"], scales=d["scales"], rotations=d["rotations"], settings=st)
        return (out.color * gi).sum() + (out.depth * gd).sum() + (out.alpha * ga).sum(), out

    base, out0 = loss_of(sp)
    _, gr = O.fwd_bwd(sp, st, gi, gd, ga, use_sh=True, dtype=torch.float64)
    rng = np.random.default_rng(0)
    checked = 0
    for name in ("means3D", "scales", "rotations", "opacities", "shs"):
        flat = sp[name].reshape(-1)
        for idx in rng.choice(flat.numel(), size=min(6, flat.numel()), replace=False):
            eps = 1e-6
            plus = {k: v.clone() for k, v in sp.items()}; minus = {k: v.clone() for k, v in sp.items()}
            plus[name].reshape(-1)[idx] += eps; minus[name].reshape(-1)[idx] -= eps
            lp, op = loss_of(plus); lm, om = loss_of(minus)
            if not (torch.equal(op.n_contrib, out0.n_contrib) and torch.equal(om.n_contrib, out0.n_contrib)
                    and torch.equal(op.radii, out0.radii) and torch.equal(om.radii, out0.radii)):
                continue  # a discrete decision flipped inside the stencil: not differentiable there
            fd = ((lp - lm) / (2 * eps)).item()
            an = gr[name].reshape(-1)[idx].item()
            # alpha clamp pass-through and the fov clamp are the only deliberate deviations; neither is active here
            assert abs(fd - an) <= 1e-5 * max(1.0, abs(fd)), (name, int(idx), fd, an)
            checked += 1
    assert checked >= 20


def test_means2d_gradient_is_ndc_scaled():
    """dL/dmeans2D == dL/d(pixel mean) * (0.5 W, 0.5 H): the convention densification relies on
    (reference scene/gaussian_model.py:427-438, arguments/__init__.py:159)."""
    sp, cam, st, grads = make_scene(40, 48, 32, seed=4, mean_scale=0.1)
    sp = {k: v.to(torch.float64) for k, v in sp.items()}
    m2 = torch.zeros(40, 3, dtype=torch.float64, requires_grad=True)
    out = O.rasterize(sp["means3D"], m2, sp["opacities"], shs=sp["shs"], scales=sp["scales"], rotations=sp["rotations"], settings=st)
    pix = out.pre.pix
    loss = (out.color * grads[0].to(torch.float64)).sum()
    g_m2, g_pix = torch.autograd.grad(loss, [m2, pix])
    assert torch.allclose(g_m2[:, 0], g_pix[:, 0] * 0.5 * 48) and torch.allclose(g_m2[:, 1], g_pix[:, 1] * 0.5 * 32)
    assert (g_m2[:, 2] == 0).all()


def test_c_oracle_in_double_equals_the_fp64_torch_oracle():
    """Round 6: the C restatement exists in float (the published arithmetic) and in double (-DREF_REAL=double, the reference of
    the full-size comparisons).  In double it must reproduce the independent fp64 autograd oracle to the rounding of its float
    outputs -- images, radii, instance count and every gradient (explicit backward vs autograd)."""
    sp, cam, st, grads = make_scene(4000, 144, 112, mean_scale=0.05)
    out, gr, nr = c_oracle.rasterize(sp, st, use_sh=True, g_img=grads[0], g_depth=grads[1], g_alpha=grads[2], precision="fp64", threads=4)
    ref, gref = O.fwd_bwd(sp, st, *grads, use_sh=True, dtype=torch.float64)
    assert nr == ref.num_rendered and int((out["radii"] != ref.radii).sum()) == 0
    for k, r in (("color", ref.color), ("depth", ref.depth), ("alpha", ref.alpha)):
        robust, frag = image_errors(out[k], r.detach(), ref.fragile)
        assert robust < 2e-6 and frag < 2e-2, (k, robust, frag)     # float outputs of a double computation
    for k in gr:
        assert grad_error(gr[k], gref[k]) < 2e-6, (k, grad_error(gr[k], gref[k]))
    # precomputed-colour path
    out2, gr2, _ = c_oracle.rasterize(sp, st, use_sh=False, g_img=grads[0], g_depth=grads[1], g_alpha=grads[2], precision="fp64", threads=4)
    ref2, gref2 = O.fwd_bwd(sp, st, *grads, use_sh=False, dtype=torch.float64)
    for k in gr2:
        assert grad_error(gr2[k], gref2[k]) < 2e-6, k


def test_c_oracle_fragile_accounting_explains_float_vs_double():
    """The C oracle's account of what two fp32 evaluations may differ in (oracle/raster_ref.c: fragile pixels, flagged splats,
    rectangle decisions, the conditioning bound) applied to ITS OWN float build against its double build -- the float oracle
    is an fp32 evaluation like the HIP path: nothing may stay unexplained, the fragile mask must contain the torch oracle's,
    and the analysis must not change the float build's results."""
    from oracle import parity as P
    sp, cam, st, grads = make_scene(20000, 256, 256)
    kw = dict(use_sh=True, g_img=grads[0], g_depth=grads[1], g_alpha=grads[2], threads=8)
    o32, g32, _ = c_oracle.rasterize(sp, st, **kw)
    o32f, g32f, _ = c_oracle.rasterize(sp, st, fragile=True, **kw)
    for k in ("color", "depth", "alpha", "radii"):
        assert torch.equal(o32[k], o32f[k]), k                          # the analysis is read-only
    r64, rg64, _ = c_oracle.rasterize(sp, st, precision="fp64", fragile=True, xy_ulps=4.0, **kw)
    fig = P.compare_flagged(o32, g32, r64, rg64, oracle_precision="fp64")
    assert fig["unexplained"] == 0, fig
    assert fig["radii"]["unexplained"] == 0
    assert 0.0 < fig["images"]["color"]["fragile_share"] < 0.05          # a small scene: few near-threshold pixels
    assert 0.0 < fig["flagged_splat_share"] < 0.6
    assert fig["gradient_max_unflagged"] < 1e-3
    ref = O.rasterize(sp["means3D"].double(), None, sp["opacities"].double(), shs=sp["shs"].double(), scales=sp["scales"].double(),
                      rotations=sp["rotations"].double(), settings=st)
    assert not (ref.fragile & ~r64["fragile"]).any()                     # same margins + the depth-tie and rectangle rules
    cb = r64["cond_bound"]
    assert cb.shape == (5, 256, 256) and (cb >= 0).all() and float(cb.max()) < 1e-3 and float(cb.mean()) < 2e-5
    assert r64["splat_flag"].dtype == torch.bool and r64["radius_raw"].shape == (20000,)
    vis = o32["radii"] > 0
    assert torch.equal(torch.ceil(r64["radius_raw"][vis]).to(torch.int32), r64["radii"][vis])
