"""End-to-end use of the boundary the way reference train.py:169-322 does: render views through `render()`, L1 loss
on colour plus the mask loss on the alpha image, backward, Adam step on the activated-parameter pre-images.  The loss
must go down -- the gradients coming out of the HIP backward are usable for optimisation, not just numerically close."""
import types

import pytest
import torch

from splatfields_amd.render import render
from splatfields_amd.synthetic import make_camera, make_splats

pytestmark = pytest.mark.gpu


def test_fitting_a_target_view_reduces_the_loss(hip_device):
    dev = hip_device
    torch.manual_seed(0)
    n, W, H = 4000, 160, 128
    target_sp = make_splats(n, seed=21, mean_scale=0.05, device=dev)
    pipe = types.SimpleNamespace(debug=False)
    bg = torch.ones(3, device=dev)
    cams = [make_camera(k, W, H, device=dev) for k in (0, 2, 5)]

    def pack(sp, deg=1):
        return {"means3D": sp["means3D"], "active_sh_degree": deg, "gaussian_opacity": sp["opacities"],
                "gaussian_features": sp["shs"], "gaussian_scales": sp["scales"], "gaussian_rotations": sp["rotations"]}

    with torch.no_grad():
        targets = [(render(c, pack(target_sp), pipe, bg)["render"], render(c, pack(target_sp), pipe, bg)["opacity"]) for c in cams]

    # learnable pre-activation parameters, as scene/gaussian_model.py:64-86 activates them
    xyz = (target_sp["means3D"] + 0.02 * torch.randn(n, 3, device=dev)).requires_grad_(True)
    log_scale = torch.log(target_sp["scales"] * 1.3).requires_grad_(True)
    rot = target_sp["rotations"].clone().requires_grad_(True)
    opacity_logit = torch.logit(target_sp["opacities"].clamp(0.05, 0.95) * 0.8).requires_grad_(True)
    shs = (target_sp["shs"] + 0.2 * torch.randn_like(target_sp["shs"])).requires_grad_(True)
    opt = torch.optim.Adam([{"params": [xyz], "lr": 2e-4}, {"params": [log_scale], "lr": 5e-3}, {"params": [rot], "lr": 1e-3},
                            {"params": [opacity_logit], "lr": 2e-2}, {"params": [shs], "lr": 5e-3}])
    losses = []
    for it in range(60):
        opt.zero_grad(set_to_none=True)
        total = 0.0
        for cam, (img_t, alpha_t) in zip(cams, targets):  # view loop of train.py:169, mean of the losses (:242)
            gd = {"means3D": xyz, "active_sh_degree": 1, "gaussian_opacity": torch.sigmoid(opacity_logit),
                  "gaussian_features": shs, "gaussian_scales": torch.exp(log_scale),
                  "gaussian_rotations": torch.nn.functional.normalize(rot)}
            pkg = render(cam, gd, pipe, bg)
            total = total + (pkg["render"] - img_t).abs().mean() + 0.1 * (pkg["opacity"] - alpha_t).abs().mean()
        loss = total / len(cams)
        loss.backward()
        assert pkg["viewspace_points"].grad is not None  # densification statistics input (train.py:307)
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    assert all(torch.isfinite(p).all() for p in (xyz, log_scale, rot, opacity_logit, shs))


def test_fitting_through_the_fused_model_path_with_densification_stats(hip_device):
    """The same optimisation through `render_model` (raw parameters, two SH tensors: the accessors run inside the kernels)
    with the fused per-view densification bookkeeping of train.py:280-286 / gaussian_model.py:427-431."""
    from splatfields_amd.densify_stats import densification_stats
    from splatfields_amd.render import render_model
    dev = hip_device
    torch.manual_seed(0)
    n, W, H = 4000, 160, 128
    target_sp = make_splats(n, seed=21, mean_scale=0.05, device=dev)
    pipe = types.SimpleNamespace(debug=False)
    bg = torch.ones(3, device=dev)
    cams = [make_camera(k, W, H, device=dev) for k in (0, 2, 5)]
    pack = {"means3D": target_sp["means3D"], "active_sh_degree": 1, "gaussian_opacity": target_sp["opacities"],
            "gaussian_features": target_sp["shs"], "gaussian_scales": target_sp["scales"], "gaussian_rotations": target_sp["rotations"]}
    with torch.no_grad():
        targets = [render(c, pack, pipe, bg) for c in cams]
    model = types.SimpleNamespace(
        active_sh_degree=1,
        _xyz=(target_sp["means3D"] + 0.02 * torch.randn(n, 3, device=dev)).requires_grad_(True),
        _scaling=torch.log(target_sp["scales"] * 1.3).requires_grad_(True),
        _rotation=(target_sp["rotations"] * 1.5).requires_grad_(True),
        _opacity=torch.logit(target_sp["opacities"].clamp(0.05, 0.95) * 0.8).requires_grad_(True),
        _features_dc=(target_sp["shs"][:, :1] + 0.2 * torch.randn(n, 1, 3, device=dev)).requires_grad_(True),
        _features_rest=(target_sp["shs"][:, 1:] + 0.2 * torch.randn(n, 15, 3, device=dev)).requires_grad_(True))
    opt = torch.optim.Adam([{"params": [model._xyz], "lr": 2e-4}, {"params": [model._scaling], "lr": 5e-3},
                            {"params": [model._rotation], "lr": 1e-3}, {"params": [model._opacity], "lr": 2e-2},
                            {"params": [model._features_dc], "lr": 5e-3}, {"params": [model._features_rest], "lr": 5e-3 / 20}])
    accum, denom, max_radii = torch.zeros(n, 1, device=dev), torch.zeros(n, 1, device=dev), torch.zeros(n, device=dev)
    losses = []
    for it in range(60):
        opt.zero_grad(set_to_none=True)
        total, pkgs = 0.0, []
        for cam, t in zip(cams, targets):
            pkg = render_model(cam, model, pipe, bg)
            total = total + (pkg["render"] - t["render"]).abs().mean() + 0.1 * (pkg["opacity"] - t["opacity"]).abs().mean()
            pkgs.append(pkg)
        loss = total / len(cams)
        loss.backward()
        with torch.no_grad():
            densification_stats(pkgs[-1]["viewspace_points"].grad, pkgs[-1]["radii"], accum, denom, max_radii)  # last view wins
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    vis = pkgs[-1]["radii"] > 0
    # visible in the last iteration: counted at least then (and at most once per iteration), radius recorded
    assert ((denom[vis] >= 1) & (denom[vis] <= 60)).all() and (max_radii[vis] > 0).all() and (accum >= 0).all()
    assert denom.max().item() == 60.0 and (accum[vis].sum() > 0)
    assert all(torch.isfinite(p).all() for p in (model._xyz, model._scaling, model._rotation, model._opacity,
                                                  model._features_dc, model._features_rest))
