"""Device-side densify / prune (splatfields_amd/densify.py, csrc/densify.hip) against the REFERENCE ITSELF: the fixtures
tests/golden/densify_*.npz hold the state of the reference's own GaussianModel (scene/gaussian_model.py) before and after its
densify_and_prune (:411-425) ran on CPU in the build container (tests/golden/make_golden.py: densify_cases), together with the
unit normal samples its torch.normal call drew.  Row order, copied rows and Adam moments must be bit-exact; resampled positions
and rescaled log-scales within fp32 round-off.  A property test covers a large cloud without any restatement of the reference."""
import math
import os

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
        "rotation": "_rotation"}


class Holder:
    """The attributes of the reference's GaussianModel that densify_and_prune touches, filled from arrays."""

    def __init__(self, params, moments, steps, accum, denom, radii, percent_dense, dev):
        self.percent_dense = float(percent_dense)
        groups = []
        for name, attr in ATTR.items():
            p = nn.Parameter(torch.as_tensor(params[name]).clone().to(dev))
            setattr(self, attr, p)
            groups.append({"params": [p], "lr": 1e-3, "name": name})
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        for grp in groups:
            name = grp["name"]
            self.optimizer.state[grp["params"][0]] = {
                "step": torch.tensor(float(steps[name])), "exp_avg": torch.as_tensor(moments[name][0]).clone().to(dev),
                "exp_avg_sq": torch.as_tensor(moments[name][1]).clone().to(dev)}
        self.xyz_gradient_accum = torch.as_tensor(accum).clone().to(dev)
        self.denom = torch.as_tensor(denom).clone().to(dev)
        self.max_radii2D = torch.as_tensor(radii).clone().to(dev)


@pytest.mark.parametrize("case", ["aniso_screen", "aniso_noscreen", "isotropic_screen", "tiny"])
def test_densify_and_prune_equals_the_reference_class(hip_device, case):
    from splatfields_amd.densify import densify_and_prune
    z = np.load(os.path.join(GOLDEN, f"densify_{case}.npz"))
    dev = hip_device
    h = Holder({k: z["in:" + k] for k in ATTR}, {k: (z["in_m:" + k], z["in_v:" + k]) for k in ATTR}, {k: z["step:" + k] for k in ATTR},
               z["in:accum"], z["in:denom"], z["in:radii"], z["percent_dense"], dev)
    screen = float(z["max_screen_size"]) or None
    unit = torch.as_tensor(z["unit_normals"]).to(dev)
    counts = densify_and_prune(h, float(z["max_grad"]), float(z["min_opacity"]), float(z["extent"]), screen, unit_normals=unit)
    n_in, n_out = z["in:xyz"].shape[0], z["out:xyz"].shape[0]
    assert counts["total"] == n_out == counts["kept"] + counts["clones"] + counts["children"]
    assert counts["children"] <= 2 * int(z["split_mask"].sum())            # children can still be pruned (opacity, world size)
    if n_in >= 700:
        assert counts["clones"] > 0 and counts["children"] > 0 and counts["kept"] < n_in   # every branch is exercised
    k = counts["kept"] + counts["clones"]
    for name, attr in ATTR.items():
        ours, ref = getattr(h, attr).detach().cpu(), torch.as_tensor(z["out:" + name])
        assert ours.shape == ref.shape, name
        if name in ("xyz", "scaling"):
            assert torch.equal(ours[:k], ref[:k]), name                      # untouched rows: bit-exact
            assert torch.allclose(ours, ref, rtol=1e-5, atol=1e-6), name     # resampled / rescaled rows of the children
        else:
            assert torch.equal(ours, ref), name                              # pure row moves: bit-exact
        st = h.optimizer.state[getattr(h, attr)]
        assert torch.equal(st["exp_avg"].cpu(), torch.as_tensor(z["out_m:" + name])), name
        assert torch.equal(st["exp_avg_sq"].cpu(), torch.as_tensor(z["out_v:" + name])), name
        assert float(st["step"]) == float(z["step:" + name])
    for grp in h.optimizer.param_groups:
        assert grp["params"][0] is getattr(h, ATTR[grp["name"]])
    assert tuple(h.xyz_gradient_accum.shape) == z["out:accum"].shape and not h.xyz_gradient_accum.any()
    assert tuple(h.denom.shape) == z["out:denom"].shape and not h.denom.any()
    assert tuple(h.max_radii2D.shape) == z["out:radii"].shape and not h.max_radii2D.any()   # the reference's postfix zeroes them
    # the optimizer keeps working on the new parameters
    for grp in h.optimizer.param_groups:
        grp["params"][0].grad = torch.ones_like(grp["params"][0])
    h.optimizer.step()


def test_large_cloud_properties(hip_device):
    """300 k splats: provenance of every output row through a tag column (no restatement of the reference's sequence)."""
    from splatfields_amd.densify import densify_and_prune_tensors
    dev, n = hip_device, 300_000
    g = torch.Generator().manual_seed(n)
    r = lambda *s: torch.randn(*s, generator=g)
    f_dc = r(n, 1, 3)
    f_dc[:, 0, 0] = torch.arange(n, dtype=torch.float32)             # row i carries its index
    params = {"xyz": r(n, 3), "f_dc": f_dc, "f_rest": r(n, 3, 3) * 0.1, "opacity": r(n, 1) * 2.5,
              "scaling": math.log(0.03) + 1.2 * r(n, 3), "rotation": r(n, 4)}
    params = {k: v.to(dev) for k, v in params.items()}
    denom = torch.randint(0, 6, (n, 1), generator=g).float()
    accum = torch.rand(n, 1, generator=g) * 0.01
    accum[denom == 0] = 0.0
    radii = torch.rand(n, generator=g) * 40
    args = (accum.to(dev), denom.to(dev), radii.to(dev), 0.0035, 0.1, 4.0, 20.0)
    unit = torch.randn(2, n, 3, generator=g).to(dev)
    out1, _, c = densify_and_prune_tensors(params, None, *args, unit_normals=unit)
    out2, _, c2 = densify_and_prune_tensors(params, None, *args, unit_normals=unit)
    assert c == c2 and all(torch.equal(out1[k], out2[k]) for k in out1)          # deterministic
    kept, clones, children = c["kept"], c["clones"], c["children"]
    assert c["total"] == kept + clones + children == out1["xyz"].shape[0]
    assert kept > 0 and clones > 0 and children > 0
    t = out1["f_dc"][:, 0, 0].cpu().long()
    tk, tc, tch = t[:kept], t[kept:kept + clones], t[kept + clones:]
    for blk in (tk, tc):
        assert (blk[1:] > blk[:-1]).all()                                         # ascending source order inside a block
    first = int((tch[1:] < tch[:-1]).nonzero()[0]) + 1 if (tch[1:] < tch[:-1]).any() else tch.numel()
    assert (tch[:first][1:] > tch[:first][:-1]).all() and (tch[first:][1:] > tch[first:][:-1]).all()   # first children, then second children
    assert np.intersect1d(tk.numpy(), tch.numpy()).size == 0                      # a split parent is removed
    src = params["scaling"].cpu()
    assert torch.allclose(out1["scaling"][kept + clones:].cpu(), torch.log(torch.exp(src[tch]) / 1.6), rtol=1e-5, atol=1e-6)
    for name in ("f_rest", "opacity", "rotation"):                                # copied rows are copies
        assert torch.equal(out1[name].cpu(), params[name].cpu()[t])
    assert torch.equal(out1["xyz"][:kept + clones].cpu(), params["xyz"].cpu()[t[:kept + clones]])
