"""splatfields_amd.deform_field.SplatFields against fixtures made by running the reference's SplatFields (utils/time_utils.py:305-508)
on CPU without plane features -- tests/golden/make_golden.py: splatfields_cases.  Every output of forward(xyz, t) and the gradients
of sum(output * probe) w.r.t. the positions and every parameter.

* CPU: the network's host side (six GeneralMLPs wired as in the reference, time embedding, FlowHead incl. the SE(3) exponential,
  view-dependent colour head, parameter names = the reference's state dict) with the fused MLP op replaced by its formula;
* GPU: the same through the HIP kernels.
"""
import glob
import os

import numpy as np
import pytest
import torch

from test_general_mlp import GOLDEN, formula


def formula_points(xyz, feat, multires, weights, biases, skips=(), negative_slope=0.01, _shape=None, time=None, time_multires=0):
    """what fused_general_mlp_points computes, in plain PyTorch: the reference's input assembly (utils/time_utils.py:9-57,
    178-181: positional encoding of the positions, features behind it, the time embedding last) + the layer formula"""
    from splatfields_amd.general_mlp import positional_encoding
    parts = [positional_encoding(xyz, multires)] + ([feat] if feat is not None else []) + \
        ([positional_encoding(time.reshape(-1, 1), time_multires)] if time is not None else [])
    return formula(torch.cat(parts, dim=-1) if len(parts) > 1 else parts[0], weights, biases, skips, negative_slope)

CASES = sorted(os.path.basename(p)[len("splatfields_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "splatfields_*.npz")))


def case_config(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SPLATFIELDS_CASES[name]


def run_case(name, device, out_tol, grad_tol):
    from splatfields_amd.deform_field import SplatFields
    data = np.load(os.path.join(GOLDEN, f"splatfields_{name}.npz"))
    n_frames, kwargs, _ = case_config(name)
    net = SplatFields(radius=None, n_frames=n_frames, **kwargs)
    state = {k[len("param:"):]: torch.from_numpy(data[k]) for k in data.files if k.startswith("param:")}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in state.items()}
    net.load_state_dict(state, strict=True)
    net = net.to(device)
    xyz = torch.from_numpy(data["xyz"]).to(device).requires_grad_()
    out = net(xyz, torch.from_numpy(data["t"]).to(device))
    if "rgb_fnc" in out:
        out = dict(out, rgb=out["rgb_fnc"](torch.from_numpy(data["viewdir"]).to(device)))
        del out["rgb_fnc"]
    expected = sorted(k[len("out:"):] for k in data.files if k.startswith("out:"))
    assert sorted(k for k, v in out.items() if v is not None) == expected
    loss = 0.0
    for k in expected:
        ref = torch.from_numpy(data["out:" + k])
        assert out[k].shape == ref.shape, k
        assert (out[k].detach().cpu() - ref).abs().max().item() <= out_tol * max(ref.abs().max().item(), 1e-6), k
        loss = loss + (out[k] * torch.from_numpy(data["probe:" + k]).to(device)).sum()
    loss.backward()
    pairs = [("xyz", xyz.grad, data["grad_xyz"])]
    for k, p in net.named_parameters():
        pairs.append((k, p.grad if p.grad is not None else torch.zeros_like(p), data["grad:" + k]))
    for k, got, want in pairs:
        want = torch.from_numpy(want)
        err = (got.detach().cpu() - want).abs().max().item()
        assert err <= grad_tol * want.abs().max().item() + 1e-8, (k, err, want.abs().max().item())


def test_fixtures_present():
    assert set(CASES) >= {"dynamic_se3", "dynamic_dct", "static_viewdep"}


@pytest.mark.parametrize("name", CASES)
def test_host_side_matches_the_reference_network(monkeypatch, name):
    from splatfields_amd import general_mlp
    monkeypatch.setattr(general_mlp, "fused_general_mlp", formula)
    run_case(name, torch.device("cpu"), 2e-5, 2e-4)


def test_se3_transform_is_a_rigid_motion_for_unit_axes():
    """known answers for the written-out exponential: unit axis -> orthonormal R with det 1; rotation about z by theta; pure
    translation direction for the screw's linear part at small angles."""
    from splatfields_amd.deform_field import se3_transform
    g = torch.Generator().manual_seed(0)
    w = torch.nn.functional.normalize(torch.randn(64, 3, generator=g, dtype=torch.float64), dim=-1)
    v = torch.randn(64, 3, generator=g, dtype=torch.float64)
    theta = torch.rand(64, 1, generator=g, dtype=torch.float64) * 3
    T = se3_transform(w, v, theta)
    R = T[:, :3, :3]
    assert (R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max() < 1e-12
    assert (torch.linalg.det(R) - 1).abs().max() < 1e-12
    assert (T[:, 3] - torch.tensor([0, 0, 0, 1.0], dtype=torch.float64)).abs().max() == 0
    z = torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64)
    Tz = se3_transform(z, torch.zeros(1, 3, dtype=torch.float64), torch.tensor([[0.5]], dtype=torch.float64))
    c, s = np.cos(0.5), np.sin(0.5)
    assert torch.allclose(Tz[0, :3, :3], torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=torch.float64), atol=1e-15)
    # (w * theta) is the rotation vector: compare with the matrix exponential of the twist
    twist = torch.zeros(64, 4, 4, dtype=torch.float64)
    twist[:, 0, 1], twist[:, 0, 2], twist[:, 1, 0] = -w[:, 2], w[:, 1], w[:, 2]
    twist[:, 1, 2], twist[:, 2, 0], twist[:, 2, 1] = -w[:, 0], -w[:, 1], w[:, 0]
    twist[:, :3, 3] = v
    assert (torch.linalg.matrix_exp(twist * theta[:, :, None]) - T).abs().max() < 1e-10


def test_encoder_contract():
    from splatfields_amd.deform_field import SplatFields
    default = SplatFields(n_frames=0, encoder_args={"noise_res": 2})   # the reference's default encoder_type: the tri-plane sampler
    assert default.feat_dim == 48 and tuple(default.encoder.planes.shape) == (3, 16, 32, 32)
    assert "encoder.planes" in default.state_dict() and default.mlp_deform.d_in == 3 * 13 + 48
    assert SplatFields(n_frames=0, encoder_type="none").feat_dim == 0    # outside the reference's list: no plane features

    class Planes(torch.nn.Module):                               # anything with the encoder's interface can be plugged in
        out_dim = 6

        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.randn(6, 3))

        def forward(self, x):
            return x @ self.table.t()

    net = SplatFields(n_frames=4, encoder=Planes(), composition_rank=2)
    names = set(net.state_dict())
    assert "encoder.table" in names and "mlp_refine_feat.0.weight" in names and "mlp_flow_head.branch_w.weight" in names
    assert net.mlp_deform.d_in == 3 * 13 + 6 + 7 and net.mlp_opacity.d_in == 3 * 7 + 6 + 7
    assert "mlp_deform.net.2.weights_t" in names and "mlp_deform.net.1.weights_t" not in names and "mlp_deform.net.7.weights_t" not in names


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_fused_network_matches_the_reference_network(hip_device, name):
    run_case(name, hip_device, 5e-5, 5e-4)


@pytest.mark.gpu
def test_network_into_rasterizer_one_backward(hip_device, monkeypatch):
    """configs[4]'s step on one GPU (reference train.py:62-101, 169-252): positions + time -> SplatFields -> rasterizer on
    precomputed colours, one backward through both.  The network once on the fused kernels and once with the fused op replaced
    by its PyTorch formula; images and the gradients that reach the positions and the network's parameters agree."""
    import math
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from splatfields_amd import general_mlp
    from splatfields_amd.deform_field import SplatFields
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    dev = hip_device
    n, W, H = 3000, 128, 96
    torch.manual_seed(11)
    net = SplatFields(radius=None, n_frames=6, encoder_type="none", composition_rank=2).to(dev)
    sp = make_splats(n, seed=21, device=dev)
    xyz = sp["means3D"].clone().requires_grad_(True)
    t = torch.full((n, 1), 0.4, device=dev)
    cam = make_camera(2, W, H, device=dev)
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False)

    def step():
        net.zero_grad(set_to_none=True)
        xyz.grad = None
        out = net(xyz, t)
        scales = sp["scales"] + 0.01 * out["scales"]                 # train.py:74: the network's scale output is a residual
        color, radii, depth = GaussianRasterizer(rs)(means3D=out["means3D"], means2D=torch.zeros_like(xyz), opacities=out["opacity"],
                                                     colors_precomp=out["rgb"], scales=scales, rotations=out["rotations"])
        ((color * gi).sum() + (depth * gd).sum()).backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        grads["xyz"] = xyz.grad.clone()
        return color.detach(), int((radii > 0).sum()), grads

    img_a, vis_a, ga_ = step()
    monkeypatch.setattr(general_mlp, "fused_general_mlp", formula)
    monkeypatch.setattr(general_mlp, "fused_general_mlp_points", formula_points)   # the path device tensors take
    img_b, vis_b, gb_ = step()
    assert vis_a > n // 4 and abs(vis_a - vis_b) <= 2
    assert (img_a - img_b).abs().max().item() <= 1e-3
    assert set(ga_) == set(gb_) and "mlp_flow_head.branch_w.weight" in ga_ and "mlp_deform.net.3.matrix_t" in ga_
    for k in ga_:
        num, den = (ga_[k] - gb_[k]).norm().item(), gb_[k].norm().item()
        assert num <= 1e-2 * den + 1e-12, (k, num, den)      # wiring test; precision is pinned by the fixture tests above


@pytest.mark.gpu
def test_config5_full_size_step_is_deterministic_and_matches_the_formula_path(hip_device, monkeypatch):
    """BASELINE.json configs[4] at FULL size on one GPU (reference train.py:62-101, utils/time_utils.py:467-508, run_owlii.sh:7):
    100 k points at one of 50 frames -> the product `SplatFields` (tri-plane sampler + refine MLP, six ResField MLPs with
    composition rank > 0, flow head) -> the rasterizer at 800x800 on precomputed colours (train.py:80-81, scales as a residual,
    :74) -> one backward through both.
      * deterministic: the step run twice gives bit-identical images, position gradients and parameter gradients (no
        floating-point atomic anywhere: slot reduction, tri-plane scatter in 64-bit fixed point, slab sums in fixed order);
      * the fused MLP kernels agree with their PyTorch formula inside the same step: images, and the gradient of every one of
        the network's parameter tensors."""
    import math
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from splatfields_amd import general_mlp
    from splatfields_amd.deform_field import SplatFields
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    dev = hip_device
    n, W, H = 100_000, 800, 800
    torch.manual_seed(5)
    net = SplatFields(radius=None, n_frames=50, composition_rank=10, flow_model="offset", encoder_args={"noise_res": 20}).to(dev)
    assert net.feat_dim == 48 and tuple(net.encoder.planes.shape) == (3, 16, 320, 320)
    with torch.no_grad():
        net.encoder.planes.normal_(0.0, 0.3)                 # a free-plane sampler starts at zero features: give the lookup something to do
    sp = make_splats(n, seed=23, device=dev)
    xyz = sp["means3D"].clone().requires_grad_(True)
    t = torch.full((n, 1), 17.0 / 49.0, device=dev)          # frame 17 of 50
    cam = make_camera(3, W, H, device=dev)
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False)
    params = dict(net.named_parameters())

    def step():
        for p in params.values():
            p.grad = None
        xyz.grad = None
        out = net(xyz, t)
        scales = sp["scales"] + 0.01 * out["scales"]
        color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
            means3D=out["means3D"], means2D=torch.zeros_like(xyz, requires_grad=True), opacities=out["opacity"], colors_precomp=out["rgb"],
            scales=scales, rotations=out["rotations"])
        torch.autograd.backward((color, depth, alpha), (gi, gd, ga))
        grads = {k: p.grad.clone() for k, p in params.items() if p.grad is not None}
        grads["xyz"] = xyz.grad.clone()
        return color.detach().clone(), depth.detach().clone(), int((radii > 0).sum()), grads

    img_a, dep_a, vis_a, g_a = step()
    img_b, dep_b, vis_b, g_b = step()
    assert vis_a > n // 2 and vis_a == vis_b
    assert torch.equal(img_a, img_b) and torch.equal(dep_a, dep_b)
    assert set(g_a) == set(g_b) and len(g_a) > 60
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), k
    for must in ("encoder.planes", "mlp_refine_feat.0.weight", "mlp_deform.net.3.matrix_t", "mlp_deform.net.3.weights_t", "mlp_rgb.net.0.weight",
                 "mlp_flow_head.gaussian_warp.weight", "mlp_scale.net.2.matrix_t", "mlp_opacity.net.5.bias", "mlp_rotation.net.4.weight"):
        assert must in g_a and torch.isfinite(g_a[must]).all() and g_a[must].abs().max() > 0, must
    monkeypatch.setattr(general_mlp, "fused_general_mlp", formula)
    monkeypatch.setattr(general_mlp, "fused_general_mlp_points", formula_points)
    img_f, dep_f, vis_f, g_f = step()
    assert abs(vis_f - vis_a) <= 20
    assert (img_f - img_a).abs().max().item() <= 2e-3 and (img_f - img_a).abs().mean().item() <= 2e-5
    assert set(g_f) == set(g_a)
    for k in g_a:
        num, den = (g_a[k] - g_f[k]).norm().item(), g_f[k].norm().item()
        assert num <= 1e-2 * den + 1e-12, (k, num, den)   # leaky-ReLU units within rounding of 0 take the other branch


def test_learning_rate_schedule_known_values():
    """values of the reference's get_expon_lr_func (utils/general_utils.py:86-119), evaluated by importing it in the build container"""
    from splatfields_amd.deform_field import expon_lr
    a = dict(lr_init=0.00016 * 5, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=40000)
    want_a = {0: 0.0008000000000000003, 1: 0.0007998757174928709, 1000: 0.0006848819757262796, 20000: 3.577708763999665e-05,
              40000: 1.5999999999999995e-06, 50000: 1.5999999999999995e-06, -1: 0.0}
    for step, want in want_a.items():
        assert expon_lr(step, **a) == pytest.approx(want, rel=1e-12, abs=0)
    b = dict(lr_init=0.01, lr_final=0.0001, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=3000)
    want_b = {0: 0.0010000000000000005, 100: 0.0032430793766233938, 500: 0.004641588833612781, 1500: 0.0010000000000000002,
              3000: 0.00010000000000000009}
    for step, want in want_b.items():
        assert expon_lr(step, **b) == pytest.approx(want, rel=1e-12)
    assert expon_lr(10, 0.0, 0.0) == 0.0


def test_model_wrapper_optimizer_schedule_and_checkpoints(tmp_path):
    """SplatFieldsModel on CPU tensors (no forward): Adam group as the reference builds it, schedule applied to the group,
    deform/iteration_<n>/deform.pth layout, load of the latest iteration."""
    import types
    from splatfields_amd.deform_field import SplatFieldsModel
    hyper = types.SimpleNamespace(encoder_type="none", composition_rank=1, deform_d=2, rgb_d=2, flow_d=2, scale_d=2, opacity_d=2,
                                  rotation_d=2, deform_skips=[0], rgb_skips=[0], flow_skips=[0], n_frames=3)
    kwargs = dict(hyper.__dict__)
    n_frames = kwargs.pop("n_frames")
    model = SplatFieldsModel(types.SimpleNamespace(n_frames=n_frames, **kwargs), radius=None, device="cpu")
    train = types.SimpleNamespace(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, deform_lr_max_steps=40000)
    model.train_setting(train)
    group = model.optimizer.param_groups[0]
    assert group["name"] == "deform" and group["eps"] == 1e-15 and len(group["params"]) == len(list(model.deform.parameters()))
    assert model.update_learning_rate(1000) == pytest.approx(0.0006848819757262796, rel=1e-12) and group["lr"] == model.update_learning_rate(1000)
    model.save_weights(str(tmp_path), 7)
    with torch.no_grad():
        first = next(model.deform.parameters())
        saved = first.clone()
        first.add_(1.0)
    model.save_weights(str(tmp_path), 12)
    assert (tmp_path / "deform" / "iteration_7" / "deform.pth").exists()
    model.load_weights(str(tmp_path), 7)
    assert torch.equal(next(model.deform.parameters()), saved)
    model.load_weights(str(tmp_path))                               # latest = 12
    assert torch.equal(next(model.deform.parameters()), saved + 1.0)


def test_reference_generator_checkpoint_is_refused_loudly():
    """A reference DEFAULT-config `deform.pth` carries the weights of its time-conditioned plane generators (`encoder.subs.*`,
    reference scene/tripFields.py:383-428).  The decoder-free sampler built here has no place for them: loading must say so
    instead of failing on a key listing (or, non-strict, silently training free planes); unknown encoder_args are reported."""
    import warnings
    from splatfields_amd.deform_field import SplatFields
    net = SplatFields(n_frames=0, encoder_args={"noise_res": 1})
    state = {k: v.clone() for k, v in net.state_dict().items() if k != "encoder.planes"}
    state["encoder.subs.0.net.decoder.conv_in.weight"] = torch.zeros(4, 4, 3, 3)
    with pytest.raises(RuntimeError, match="GENERATOR"):
        net.load_state_dict(state)
    with pytest.raises(RuntimeError, match="GENERATOR"):
        net.load_state_dict(state, strict=False)
    net.load_state_dict(net.state_dict())                      # its own checkpoints still load
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        SplatFields(n_frames=0, encoder_args={"noise_res": 1, "layer_kwargs": {"x": 1}, "contract_ngp": True})
    assert any("contract_ngp" in str(x.message) and "layer_kwargs" in str(x.message) for x in w)


def test_resfield_frame_index_is_bounds_checked_on_the_host():
    """reference utils/resfields.py: `mat[frame_id]` raises IndexError outside the capacity; so does a host-side frame index here
    (a device-side one cannot be checked without a synchronisation: the kernel poisons the composed weights with NaN instead)."""
    from splatfields_amd.general_mlp import GeneralMLP, compose_resfield_weights
    mlp = GeneralMLP(in_features=3, out_features=3, hidden_features=16, num_hidden_layers=3, skips=(), multires=0, act="leaky_relu",
                     composition_rank=2, n_frames=5)
    layers = list(mlp.net)
    assert any(l.has_residual for l in layers)
    compose_resfield_weights(layers, 4)
    compose_resfield_weights(layers, -1)
    for bad in (5, -6, 100):
        with pytest.raises(IndexError):
            compose_resfield_weights(layers, bad)
