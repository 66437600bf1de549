"""Fused MLP forward (splatfields_amd/fused_mlp.py -> sr_mlp_forward) against a PyTorch restatement of the reference's
`GeneralMLP.forward` (utils/time_utils.py:178-191): fp32, <= 2e-5 relative to the output's magnitude."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def general_mlp_reference(h_in, weights, biases, skips, slope):
    h = h_in
    for i, (W, b) in enumerate(zip(weights, biases)):
        h = F.leaky_relu(F.linear(h, W, b), slope)                  # act after every layer, the last one included
        if i in skips and i != len(weights) - 1:
            h = torch.cat([h_in, h], dim=-1)                        # the input goes IN FRONT of the hidden state
    return h


def make_net(d_in, hidden, n_hidden, skips, out, dev, seed):
    g = torch.Generator().manual_seed(seed)
    dims_in = [d_in] + [hidden + (d_in if (j - 1) in skips else 0) for j in range(1, n_hidden + 2)]
    dims_out = [hidden] * (n_hidden + 1) + [out]
    weights = [(torch.randn(o, i, generator=g) / i ** 0.5).to(dev) for i, o in zip(dims_in, dims_out)]
    biases = [(0.1 * torch.randn(o, generator=g)).to(dev) for o in dims_out]
    return weights, biases


# (d_in, hidden, hidden layers, skips, out): the six networks of reference utils/time_utils.py:343-447 with a 48-channel
# tri-plane feature and a 7-channel time embedding -- deform / rgb 128x6 skip 3, flow 128x6 -> 128, scale / opacity 64x4
# skip 2, rotation 64x3 without an effective skip -- plus odd sizes
CASES = [(94, 128, 6, [3], 3), (94, 128, 6, [3], 128), (82, 64, 4, [2], 3), (76, 64, 4, [2], 1), (76, 64, 3, [20], 4),
         (33, 128, 1, [], 5), (130, 64, 2, [0, 1], 16)]


@pytest.mark.parametrize("d_in,hidden,n_hidden,skips,out", CASES)
@pytest.mark.parametrize("n", [1000, 4096])
def test_fused_forward_matches_the_reference_formula(hip_device, d_in, hidden, n_hidden, skips, out, n):
    from splatfields_amd.fused_mlp import FusedGeneralMLP
    dev = hip_device
    weights, biases = make_net(d_in, hidden, n_hidden, skips, out, dev, seed=d_in + hidden + out)
    h_in = torch.randn(n, d_in, generator=torch.Generator().manual_seed(n)).to(dev)
    net = FusedGeneralMLP(weights, biases, d_in, skips, negative_slope=0.01)
    y = net(h_in)
    ref = general_mlp_reference(h_in.double(), [w.double() for w in weights], [b.double() for b in biases], set(skips), 0.01)
    assert y.shape == ref.shape
    err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-5, err
    # weights updated in place (a ResField frame change) are re-packed
    weights[0].mul_(0.5)
    ref2 = general_mlp_reference(h_in.double(), [w.double() for w in weights], [b.double() for b in biases], set(skips), 0.01)
    assert (net(h_in).double() - ref2).abs().max().item() / ref2.abs().max().item() <= 2e-5


def test_fused_forward_timing_vs_pytorch(hip_device):
    """100 k points through the deform-sized network: the fused kernel must beat the layer-by-layer PyTorch forward (printed)."""
    import time
    from splatfields_amd.fused_mlp import FusedGeneralMLP
    dev = hip_device
    weights, biases = make_net(94, 128, 6, [3], 3, dev, seed=1)
    h_in = torch.randn(100_000, 94, device=dev)
    net = FusedGeneralMLP(weights, biases, 94, [3])

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 20 * 1e3

    with torch.no_grad():
        t_fused = timed(lambda: net(h_in))
        t_torch = timed(lambda: general_mlp_reference(h_in, weights, biases, {3}, 0.01))
    print(f"\nfused MLP forward, 100k points, 94 -> 128 x 7 -> 3: {t_fused:.3f} ms; PyTorch-ROCm: {t_torch:.3f} ms")
    assert t_fused < t_torch
