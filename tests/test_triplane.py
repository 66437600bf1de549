"""Tri-plane encoder (splatfields_amd/triplane.py, csrc/triplane.hip) against the reference's own VarTriPlaneEncoder.forward
(scene/tripFields.py:430-436), whose outputs and gradients on small planes are the fixtures tests/golden/triplane_*.npz
(tests/golden/make_golden.py: triplane_cases), and against torch's grid_sample at the reference's plane size."""
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fixtures_hold_what_the_reference_forward_computes():
    """CPU: the fixture is self-consistent with torch's grid_sample called the way the reference calls it."""
    z = np.load(os.path.join(GOLDEN, "triplane_c8_20x28.npz"))
    planes, pts = torch.as_tensor(z["planes"]), torch.as_tensor(z["pts"])
    assert z["axis"].tolist() == [[0, 1], [1, 2], [2, 0]]
    coord = torch.stack([pts[..., ax] for ax in z["axis"].tolist()])
    feat = torch.nn.functional.grid_sample(planes, coord, align_corners=False).permute(2, 3, 0, 1).reshape(1, pts.shape[1], -1)
    assert torch.allclose(feat, torch.as_tensor(z["out"]), atol=1e-6)
    assert ((pts.abs() > 1).any(-1)).float().mean() > 0.1      # the zero-padding branch is in the fixture


def test_sampler_interface_and_argument_checks():
    from splatfields_amd.triplane import TriPlaneSampler, triplane_lookup
    s = TriPlaneSampler(out_ch=8, resolution=12)
    assert s.out_dim == 24 and s.n_planes == 3 and s.axis == [[0, 1], [1, 2], [2, 0]] and tuple(s.planes.shape) == (3, 8, 12, 12)
    assert TriPlaneSampler(out_ch=8, resolution=12, fuse_mode="add").out_dim == 8
    assert [k for k, _ in s.named_parameters()] == ["planes"]
    gen = torch.nn.Conv2d(1, 1, 1)
    assert "plane_source.weight" in dict(TriPlaneSampler(out_ch=8, plane_source=gen).named_parameters())   # a generator trains with the net
    with pytest.raises(RuntimeError, match="no CPU path"):
        triplane_lookup(torch.zeros(3, 8, 4, 4), torch.zeros(5, 3))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["c16_24x24", "c8_20x28"])
def test_lookup_equals_the_reference_encoder(hip_device, case):
    from splatfields_amd.triplane import TriPlaneSampler
    z = np.load(os.path.join(GOLDEN, f"triplane_{case}.npz"))
    dev = hip_device

    class Source(torch.nn.Module):                       # the generator half: hands out the fixture's planes
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.as_tensor(z["planes"]).clone())

        def get_planes(self, frame_id=None):
            return self.p

    src = Source().to(dev)
    enc = TriPlaneSampler(out_ch=z["planes"].shape[1], plane_source=src)
    pts = torch.as_tensor(z["pts"]).to(dev).requires_grad_(True)
    out = enc(pts)                                        # [1, N, 3 C] like the reference's encoder(x[None])
    ref = torch.as_tensor(z["out"])
    assert tuple(out.shape) == ref.shape
    assert (out.detach().cpu() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    (out * torch.as_tensor(z["probe"]).to(dev)).sum().backward()
    for got, want, name in ((src.p.grad, z["grad_planes"], "planes"), (pts.grad, z["grad_pts"], "points")):
        want = torch.as_tensor(want)
        err = (got.detach().cpu() - want).abs().max().item() / want.abs().max().item()
        assert err <= 5e-6, (name, err)


@pytest.mark.gpu
def test_reference_plane_size_against_torch_and_determinism(hip_device):
    """16 x 320 x 320 planes (the reference's Tensorial2D output), 100 k points: values and both gradients against torch's
    grid_sample on the same device; the plane gradient is bit-identical from run to run (fixed-point integer accumulation)."""
    from splatfields_amd.triplane import triplane_lookup
    dev = hip_device
    g = torch.Generator().manual_seed(5)
    planes = torch.randn(3, 16, 320, 320, generator=g).to(dev).requires_grad_(True)
    pts = (torch.rand(100_000, 3, generator=g) * 2.2 - 1.1).to(dev).requires_grad_(True)
    probe = torch.randn(100_000, 48, generator=g).to(dev)

    def run(fn):
        planes.grad = pts.grad = None
        out = fn()
        (out * probe).sum().backward()
        return out.detach(), planes.grad.clone(), pts.grad.clone()

    def torch_ref():
        coord = torch.stack([pts[None][..., ax] for ax in ([0, 1], [1, 2], [2, 0])])
        return torch.nn.functional.grid_sample(planes, coord, align_corners=False).permute(2, 3, 0, 1).reshape(-1, 48)

    o1, gp1, gx1 = run(lambda: triplane_lookup(planes, pts))
    o2, gp2, gx2 = run(lambda: triplane_lookup(planes, pts))
    ot, gpt, gxt = run(torch_ref)
    assert torch.equal(o1, o2) and torch.equal(gp1, gp2) and torch.equal(gx1, gx2)       # bit-reproducible
    assert (o1 - ot).abs().max().item() <= 2e-6 * ot.abs().max().item()
    assert (gp1 - gpt).abs().max().item() <= 1e-5 * gpt.abs().max().item()               # torch sums with float atomics
    assert (gx1 - gxt).abs().max().item() <= 1e-4 * gxt.abs().max().item()
    # degenerate inputs: no points, all points far outside, zero upstream gradient
    assert tuple(triplane_lookup(planes, pts[:0]).shape) == (0, 48)
    far = torch.full((7, 3), 5.0, device=dev, requires_grad=True)
    out = triplane_lookup(planes, far)
    assert not out.any()
    planes.grad = None
    (out * 0.0).sum().backward()
    assert not planes.grad.any() and not far.grad.any()


@pytest.mark.gpu
def test_default_splatfields_constructs_and_trains(hip_device):
    """The reference's default network (`encoder_type='VarTriPlaneEncoder'`, utils/time_utils.py:313-334) builds and runs
    forward + backward end to end: tri-plane lookup -> refine MLP -> the six fused GeneralMLPs."""
    from splatfields_amd.deform_field import SplatFields
    dev = hip_device
    torch.manual_seed(0)
    net = SplatFields(n_frames=4, composition_rank=2, encoder_args={"out_ch": 16, "noise_res": 4}).to(dev)
    assert net.feat_dim == 48 and tuple(net.encoder.planes.shape) == (3, 16, 64, 64)
    assert "encoder.planes" in dict(net.named_parameters())
    xyz = (torch.rand(3000, 3, device=dev) * 1.6 - 0.8).requires_grad_(True)
    out = net(xyz, torch.full((3000, 1), 0.4, device=dev))
    loss = sum(v.sum() for k, v in out.items() if torch.is_tensor(v) and k != "flow")
    loss.backward()
    assert torch.isfinite(net.encoder.planes.grad).all() and net.encoder.planes.grad.abs().sum() > 0
    assert torch.isfinite(xyz.grad).all() and xyz.grad.abs().sum() > 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.mlp_deform.parameters())
