"""splatfields_amd.general_mlp.GeneralMLP against fixtures made by running the reference's GeneralMLP (utils/time_utils.py:123-191
with the ResField layers of utils/resfields.py) on CPU -- tests/golden/make_golden.py: general_mlp_cases.  Each fixture holds the
reference module's state dict, inputs, frame id, output and the gradients of sum(output * probe) w.r.t. inputs and parameters.

* CPU: the module's host side (parameter names / shapes = the reference's state dict, positional-encoding order, skip numbering,
  per-frame ResField composition, output activations) with the fused op replaced by its PyTorch formula;
* GPU: the same comparison through the HIP kernels, float32: outputs <= 2e-5, gradients <= 2e-4 of each tensor's largest entry.
"""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[len("general_mlp_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "general_mlp_*.npz")))


def constructor_kwargs(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                      # definitions only; nothing is generated, /root/reference is not touched
    return mod.GENERAL_MLP_CASES[name][0]


def formula(h_in, weights, biases, skips=(), negative_slope=0.01, _shape=None):
    """what fused_general_mlp computes, in plain PyTorch (reference utils/time_utils.py:178-188)"""
    h = h_in
    for i, (W, b) in enumerate(zip(weights, biases)):
        h = F.leaky_relu(F.linear(h, W, b), negative_slope)
        if i in set(skips) and i != len(weights) - 1:
            h = torch.cat([h_in, h], dim=-1)
    return h


def run_case(name, device, out_tol, grad_tol):
    from splatfields_amd.general_mlp import GeneralMLP
    data = np.load(os.path.join(GOLDEN, f"general_mlp_{name}.npz"))
    net = GeneralMLP(**constructor_kwargs(name))
    state = {k[len("param:"):]: torch.from_numpy(data[k]) for k in data.files if k.startswith("param:")}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in state.items()}
    net.load_state_dict(state, strict=True)
    net = net.to(device)
    xyz = torch.from_numpy(data["xyz"]).to(device).requires_grad_()
    feat = torch.from_numpy(data["feat"]).to(device).requires_grad_() if "feat" in data.files else None
    frame = int(data["frame_id"])
    frame_id = None if frame < 0 else torch.tensor(frame, device=device)
    out = net(xyz, feat, frame_id=frame_id)
    (out * torch.from_numpy(data["probe"]).to(device)).sum().backward()
    ref_out = torch.from_numpy(data["out"])
    assert out.shape == ref_out.shape
    assert (out.detach().cpu() - ref_out).abs().max().item() <= out_tol * max(ref_out.abs().max().item(), 1e-6)
    pairs = [("xyz", xyz.grad, data["grad_xyz"])]
    if feat is not None:
        pairs.append(("feat", feat.grad, data["grad_feat"]))
    for k, p in net.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        pairs.append((k, g, data["grad:" + k]))
    for k, got, want in pairs:
        want = torch.from_numpy(want)
        err = (got.detach().cpu() - want).abs().max().item()
        assert err <= grad_tol * want.abs().max().item() + 1e-9, (k, err, want.abs().max().item())
    return net, frame


def test_fixtures_present():
    assert set(CASES) >= {"scale", "opacity", "rotation", "deform", "static_rgb", "no_features"}


@pytest.mark.parametrize("name", CASES)
def test_host_side_matches_the_reference_module(monkeypatch, name):
    from splatfields_amd import general_mlp
    monkeypatch.setattr(general_mlp, "fused_general_mlp", formula)
    net, frame = run_case(name, torch.device("cpu"), 1e-5, 1e-4)
    # only the current frame's coefficients receive a gradient; the reference composes (and differentiates) every frame's matrix
    for k, p in net.named_parameters():
        if k.endswith("weights_t"):
            rows = p.grad.abs().sum(dim=1)
            assert rows[frame] > 0 and rows.sum() == rows[frame]


def test_unsupported_configurations_fail_loudly():
    from splatfields_amd.general_mlp import GeneralMLP
    with pytest.raises(NotImplementedError):
        GeneralMLP(act="softplus")
    with pytest.raises(KeyError):
        GeneralMLP(act="relu", out_activation="gelu")
    net = GeneralMLP(in_features=3, out_features=3, hidden_features=64, num_hidden_layers=2, skips=[], multires=2, act="leaky_relu",
                     composition_rank=2, n_frames=4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        net(torch.zeros(5, 3), frame_id=1)
    with pytest.raises(ValueError, match="frame_id"):
        net(torch.zeros(5, 3))
    wide = GeneralMLP(in_features=3, out_features=3, hidden_features=96, num_hidden_layers=2, skips=[], multires=2, act="leaky_relu",
                      composition_rank=0, n_frames=0)
    with pytest.raises(ValueError, match="hidden widths"):
        wide(torch.zeros(5, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_fused_module_matches_the_reference_module(hip_device, name):
    run_case(name, hip_device, 2e-5, 2e-4)


@pytest.mark.gpu
def test_reference_checkpoint_round_trip(hip_device, tmp_path):
    """state_dict written by one module loads into another (the reference's deform.pth layout) and gives identical outputs."""
    from splatfields_amd.general_mlp import GeneralMLP
    kw = dict(in_features=3 + 8 + 7, out_features=3, hidden_features=128, num_hidden_layers=6, skips=[3], multires=6,
              out_activation="none", act="leaky_relu", composition_rank=4, n_frames=12)
    a, b = GeneralMLP(**kw).to(hip_device), GeneralMLP(**kw).to(hip_device)
    torch.save(a.state_dict(), tmp_path / "deform.pth")
    b.load_state_dict(torch.load(tmp_path / "deform.pth"))
    xyz, feat = torch.rand(3000, 3, device=hip_device), torch.randn(3000, 15, device=hip_device)
    with torch.no_grad():
        for frame in (0, 7, 11):
            assert torch.equal(a(xyz, feat, frame_id=frame), b(xyz, feat, frame_id=torch.tensor(frame, device=hip_device)))
        assert not torch.equal(a(xyz, feat, frame_id=0), a(xyz, feat, frame_id=1))
        # negative indices wrap as the reference's `mat[frame_id]` does -- host ints and device tensors alike
        assert torch.equal(a(xyz, feat, frame_id=-1), a(xyz, feat, frame_id=11))
        assert torch.equal(a(xyz, feat, frame_id=torch.tensor(-1, device=hip_device)), a(xyz, feat, frame_id=11))
        assert torch.equal(a(xyz, feat, frame_id=torch.tensor(-12, device=hip_device)), a(xyz, feat, frame_id=0))
        # a device-side index outside [-capacity, capacity) cannot raise from a kernel: the weights are poisoned instead
        assert torch.isnan(a(xyz, feat, frame_id=torch.tensor(-13, device=hip_device))).all()
        assert torch.isnan(a(xyz, feat, frame_id=torch.tensor(12, device=hip_device))).all()


def test_normalize_activation_matches_torch():
    """the 'normalize' output activation (rotations, reference utils/time_utils.py:174) has its own compact backward: same values
    and gradients as F.normalize, also at and below the eps clamp"""
    import torch.nn.functional as F
    from splatfields_amd.general_mlp import _Normalize
    torch.manual_seed(3)
    x = torch.randn(500, 4, dtype=torch.float64)
    x[3] = 0.0
    x[4] = 1e-14
    x.requires_grad_(True)
    g = torch.randn(500, 4, dtype=torch.float64)
    a = _Normalize.apply(x)
    ga, = torch.autograd.grad(a, x, g)
    b = F.normalize(x, dim=-1)
    gb, = torch.autograd.grad(b, x, g)
    assert torch.equal(a, b) and (ga - gb).abs().max().item() <= 1e-12 * gb.abs().max().item()
