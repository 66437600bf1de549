"""Device-side densify / prune (splatfields_amd/densify.py, csrc/densify.hip) against a line-by-line PyTorch restatement of
reference scene/gaussian_model.py:272-425 run on the same inputs with the same normal samples: row order, copied rows and
Adam moments bit-exact; resampled positions and rescaled log-scales within fp32 round-off."""
import math

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def build_rotation(r):   # reference utils/general_utils.py:138-159
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


class RefModel:
    """The part of the reference's GaussianModel the densification touches, restated with torch ops in the reference's order."""

    def __init__(self, p, dev, isotropic=False):
        self.use_isotropic = isotropic
        self.percent_dense = 0.01
        self._xyz = nn.Parameter(p["xyz"].clone().to(dev)); self._features_dc = nn.Parameter(p["f_dc"].clone().to(dev))
        self._features_rest = nn.Parameter(p["f_rest"].clone().to(dev)); self._opacity = nn.Parameter(p["opacity"].clone().to(dev))
        self._scaling = nn.Parameter(p["scaling"].clone().to(dev)); self._rotation = nn.Parameter(p["rotation"].clone().to(dev))
        groups = [{"params": [self._xyz], "lr": 1e-3, "name": "xyz"}, {"params": [self._features_dc], "lr": 1e-3, "name": "f_dc"},
                  {"params": [self._features_rest], "lr": 1e-3, "name": "f_rest"}, {"params": [self._opacity], "lr": 1e-3, "name": "opacity"},
                  {"params": [self._scaling], "lr": 1e-3, "name": "scaling"}, {"params": [self._rotation], "lr": 1e-3, "name": "rotation"}]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        n = p["xyz"].shape[0]
        self.xyz_gradient_accum = p["accum"].clone().to(dev); self.denom = p["denom"].clone().to(dev)
        self.max_radii2D = p["radii"].clone().to(dev)
        g = torch.Generator().manual_seed(7)
        for grp in groups:   # one Adam step so that the moments exist and are non-trivial
            grp["params"][0].grad = torch.randn(grp["params"][0].shape, generator=g).to(dev)
        self.optimizer.step()
        assert n == self._xyz.shape[0]

    @property
    def get_scaling(self):
        s = torch.exp(self._scaling)
        return s.repeat(1, 3) if self.use_isotropic else s

    # --- reference scene/gaussian_model.py:272-425, restated ---
    def _prune_optimizer(self, mask):
        out = {}
        for group in self.optimizer.param_groups:
            st = self.optimizer.state.get(group["params"][0], None)
            if st is not None:
                st["exp_avg"] = st["exp_avg"][mask]; st["exp_avg_sq"] = st["exp_avg_sq"][mask]
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
                self.optimizer.state[group["params"][0]] = st
            else:
                group["params"][0] = nn.Parameter(group["params"][0][mask].requires_grad_(True))
            out[group["name"]] = group["params"][0]
        return out

    def _assign(self, t):
        self._xyz, self._features_dc, self._features_rest = t["xyz"], t["f_dc"], t["f_rest"]
        self._opacity, self._scaling, self._rotation = t["opacity"], t["scaling"], t["rotation"]

    def prune_points(self, mask):
        valid = ~mask
        self._assign(self._prune_optimizer(valid))
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid]; self.denom = self.denom[valid]; self.max_radii2D = self.max_radii2D[valid]

    def densification_postfix(self, d):
        out = {}
        for group in self.optimizer.param_groups:
            ext = d[group["name"]]
            if group["name"] == "scaling" and self.use_isotropic:
                ext = ext[:, :1]
            st = self.optimizer.state.get(group["params"][0], None)
            if st is not None:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
                self.optimizer.state[group["params"][0]] = st
            else:
                group["params"][0] = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
            out[group["name"]] = group["params"][0]
        self._assign(out)
        m, dev = self._xyz.shape[0], self._xyz.device
        self.xyz_gradient_accum = torch.zeros((m, 1), device=dev); self.denom = torch.zeros((m, 1), device=dev)
        self.max_radii2D = torch.zeros((m,), device=dev)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, unit):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        n0 = self._xyz.shape[0]
        # clone
        sel = (torch.norm(grads, dim=-1) >= max_grad) & (self.get_scaling.max(dim=1).values <= self.percent_dense * extent)
        self.densification_postfix({"xyz": self._xyz[sel], "f_dc": self._features_dc[sel], "f_rest": self._features_rest[sel],
                                    "opacity": self._opacity[sel], "scaling": self._scaling[sel], "rotation": self._rotation[sel]})
        # split (N = 2)
        n_init = self._xyz.shape[0]
        padded = torch.zeros(n_init, device=self._xyz.device)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= max_grad) & (self.get_scaling.max(dim=1).values > self.percent_dense * extent)
        stds = self.get_scaling[sel].repeat(2, 1)
        idx = torch.nonzero(sel).reshape(-1)
        assert (idx < n0).all()
        samples = torch.cat([unit[0][idx], unit[1][idx]], dim=0) * stds     # = torch.normal(0, stds) with these unit samples
        rots = build_rotation(self._rotation[sel]).repeat(2, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self._xyz[sel].repeat(2, 1)
        new_scaling = torch.log(self.get_scaling[sel].repeat(2, 1) / (0.8 * 2))
        self.densification_postfix({"xyz": new_xyz, "f_dc": self._features_dc[sel].repeat(2, 1, 1),
                                    "f_rest": self._features_rest[sel].repeat(2, 1, 1), "opacity": self._opacity[sel].repeat(2, 1),
                                    "scaling": new_scaling, "rotation": self._rotation[sel].repeat(2, 1)})
        self.prune_points(torch.cat((sel, torch.zeros(2 * int(sel.sum()), device=sel.device, dtype=torch.bool))))
        # prune
        prune = (torch.sigmoid(self._opacity) < min_opacity).squeeze()
        if max_screen_size:
            prune = prune | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune)


def make_inputs(n, isotropic, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    scale_cols = 1 if isotropic else 3
    p = {"xyz": r(n, 3), "f_dc": r(n, 1, 3), "f_rest": r(n, 15, 3) * 0.1, "opacity": r(n, 1) * 2.5,
         "scaling": math.log(0.03) + 1.2 * r(n, scale_cols), "rotation": r(n, 4),
         "accum": torch.rand(n, 1, generator=g) * 0.01, "denom": torch.randint(0, 6, (n, 1), generator=g).float(),
         "radii": torch.rand(n, generator=g) * 40}
    p["accum"][p["denom"] == 0] = 0.0   # 0 / 0 -> NaN -> 0 as in the reference
    return p


@pytest.mark.parametrize("n,isotropic,screen", [(20000, False, 20), (20000, False, None), (7000, True, 20), (300000, False, 20), (5, False, None)])
def test_densify_and_prune_equals_the_reference_sequence(hip_device, n, isotropic, screen):
    from splatfields_amd.densify import densify_and_prune
    dev = hip_device
    p = make_inputs(n, isotropic, seed=n)
    unit = torch.randn(2, n, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    ref, ours = RefModel(p, dev, isotropic), RefModel(p, dev, isotropic)
    max_grad, min_opacity, extent = 0.0035, 0.1, 4.0
    ref.densify_and_prune(max_grad, min_opacity, extent, screen, unit)
    counts = densify_and_prune(ours, max_grad, min_opacity, extent, screen, unit_normals=unit)
    assert counts["total"] == ref._xyz.shape[0]
    if n >= 7000:
        assert counts["clones"] > 0 and counts["children"] > 0 and counts["kept"] < n   # every branch is exercised
    for name in ("_features_dc", "_features_rest", "_opacity", "_rotation"):
        assert torch.equal(getattr(ours, name).detach(), getattr(ref, name).detach()), name      # pure row moves: bit-exact
    for name in ("_xyz", "_scaling"):
        a, b = getattr(ours, name).detach(), getattr(ref, name).detach()
        assert a.shape == b.shape
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), name                                   # resampled / rescaled rows
        k = counts["kept"] + counts["clones"]
        assert torch.equal(a[:k], b[:k])                                                          # untouched rows: bit-exact
    for go, gr in zip(ours.optimizer.param_groups, ref.optimizer.param_groups):
        so, sr_ = ours.optimizer.state[go["params"][0]], ref.optimizer.state[gr["params"][0]]
        assert go["params"][0] is getattr(ours, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest",
                                                  "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}[go["name"]])
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(so[key], sr_[key]), (go["name"], key)
        assert so["step"] == sr_["step"]
    assert ours.xyz_gradient_accum.shape == ref.xyz_gradient_accum.shape and not ours.xyz_gradient_accum.any()
    assert ours.denom.shape == ref.denom.shape and ours.max_radii2D.shape == ref.max_radii2D.shape
    # the optimizer keeps working on the new parameters
    for grp in ours.optimizer.param_groups:
        grp["params"][0].grad = torch.ones_like(grp["params"][0])
    ours.optimizer.step()
