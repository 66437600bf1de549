"""Fused MLP (splatfields_amd/fused_mlp.py -> sr_mlp_pack / sr_mlp_chain) against a PyTorch restatement of the reference's
`GeneralMLP.forward` (utils/time_utils.py:178-191): forward fp32 <= 2e-5 relative to the output's magnitude; gradients of
h_in, every weight and bias against float64 autograd of the same formula, <= 1e-4 of each tensor's largest entry."""
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


def test_device_packer_matches_the_pytorch_statement(hip_device):
    """sr_mlp_pack == pack_layer_weight, plain and transposed blocks, with padding of rows, both column blocks and the bias."""
    import ctypes as C
    from splatfields_amd import _lib
    from splatfields_amd.fused_mlp import pack_layer_weight
    lib, dev = _lib.load(), hip_device
    g = torch.Generator().manual_seed(5)
    W = torch.randn(70, 94 + 128, generator=g).to(dev)          # [rows, input block | hidden block]
    bias = torch.randn(70, generator=g).to(dev)
    cases = [
        # (transposed, row0, n_rows, n_mem, mem_pad, mem_col0, n_reg, reg_width, reg_col0, out_tiles, reference matrix)
        (0, 0, 70, 94, 96, 0, 128, 128, 94, 5, W),
        (0, 3, 40, 0, 0, 0, 100, 128, 10, 3, W[3:43, 10:110]),
        (1, 94, 128, 70, 96, 0, 0, 0, 0, 8, W[:, 94:].t()),      # W[:, hidden block]^T: rows = hidden inputs, columns = outputs
        (1, 16, 32, 0, 0, 0, 70, 96, 0, 2, W[:, 16:48].t()),
    ]
    jobs = (_lib.SrMlpPackJob * len(cases))()
    outs, bouts = [], []
    for j, (tr, row0, rows, n_mem, mem_pad, mc0, n_reg, rw, rc0, ot, _) in enumerate(cases):
        outs.append(torch.full((16 * ot * (mem_pad + rw),), float("nan"), device=dev))
        bouts.append(torch.full((16 * ot,), float("nan"), device=dev))
        jobs[j] = _lib.SrMlpPackJob(w=W.data_ptr(), bias_src=bias.data_ptr(), dst=outs[j].data_ptr(), bias_dst=bouts[j].data_ptr(),
                                    ld=W.stride(0), transposed=tr, row0=row0, n_rows=rows, n_mem=n_mem, mem_pad=mem_pad, mem_col0=mc0,
                                    n_reg=n_reg, reg_width=rw, reg_col0=rc0, out_tiles=ot, n_bias=min(70, 16 * ot))
    _lib.check(lib.sr_mlp_pack(len(cases), jobs, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    torch.cuda.synchronize()
    for j, (tr, row0, rows, n_mem, mem_pad, mc0, n_reg, rw, rc0, ot, ref) in enumerate(cases):
        want = pack_layer_weight(ref.contiguous(), n_mem, mem_pad, rw, ot)
        assert torch.equal(outs[j], want), j
        nb = min(70, 16 * ot)
        assert torch.equal(bouts[j][:nb], bias[:nb]) and (bouts[j][nb:] == 0).all()
    bad = (_lib.SrMlpPackJob * 1)()
    bad[0] = jobs[0]
    bad[0].mem_pad = 90                                           # not a multiple of 32
    assert lib.sr_mlp_pack(1, bad, None) != 0


GRAD_CASES = [(94, 128, 6, [3], 3), (82, 64, 4, [2], 3), (76, 64, 3, [20], 4), (33, 128, 1, [], 5), (130, 64, 2, [0, 1], 16),
              (94, 128, 6, [3, 6], 128)]


@pytest.mark.parametrize("d_in,hidden,n_hidden,skips,out", GRAD_CASES)
@pytest.mark.parametrize("n,slope", [(777, 0.01), (2048, 0.2)])
def test_fused_gradients_match_float64_autograd(hip_device, d_in, hidden, n_hidden, skips, out, n, slope):
    from splatfields_amd.fused_mlp import fused_general_mlp
    dev = hip_device
    weights, biases = make_net(d_in, hidden, n_hidden, skips, out, dev, seed=7 + d_in + out)
    g = torch.Generator().manual_seed(n)
    h_in = torch.randn(n, d_in, generator=g).to(dev)
    dY = torch.randn(n, out, generator=g).to(dev)

    def leaves(dtype):
        def leaf(t):
            return t.detach().clone().to(dtype).requires_grad_()
        return leaf(h_in), [leaf(w) for w in weights], [leaf(b) for b in biases]

    x, ws, bs = leaves(torch.float32)
    y = fused_general_mlp(x, ws, bs, skips=skips, negative_slope=slope)
    y.backward(dY)
    xr, wr, br = leaves(torch.float64)
    yr = general_mlp_reference(xr, wr, br, set(skips), slope)
    yr.backward(dY.double())
    assert (y.double() - yr).abs().max().item() <= 2e-5 * yr.abs().max().item()
    for name, got, ref in [("h_in", x, xr)] + [(f"W{j}", a, b) for j, (a, b) in enumerate(zip(ws, wr))] + \
            [(f"b{j}", a, b) for j, (a, b) in enumerate(zip(bs, br))]:
        assert got.grad is not None and got.grad.shape == ref.grad.shape, name
        scale = ref.grad.abs().max().item()
        err = (got.grad.double() - ref.grad).abs().max().item()
        assert err <= 1e-4 * scale + 1e-12, (name, err, scale)


def test_fused_gradients_flow_through_composed_weights(hip_device):
    """ResField use: the effective weight W + (coefficients @ basis) is an autograd expression; dL/dW reaches both factors.
    Gradients only for the tensors that ask for them; a second backward through the same graph is refused by autograd, a second
    forward+backward gives the same numbers (deterministic kernels)."""
    from splatfields_amd.fused_mlp import fused_general_mlp
    dev = hip_device
    weights, biases = make_net(94, 128, 3, [0], 3, dev, seed=3)
    g = torch.Generator().manual_seed(1)
    coeff = torch.randn(1, 10, generator=g).to(dev).requires_grad_()
    basis = (0.01 * torch.randn(10, 128 * 128, generator=g)).to(dev).requires_grad_()
    h_in = torch.randn(1500, 94, generator=g).to(dev)                    # no gradient wanted for the input
    W1 = weights[1].clone().requires_grad_()

    def run(fused):
        for t in (coeff, basis, W1):
            t.grad = None
        ws = list(weights)
        ws[1] = W1
        ws[2] = weights[2] + (coeff @ basis).view(128, 128)
        if fused:
            y = fused_general_mlp(h_in, ws, biases, skips=[0])
        else:
            y = general_mlp_reference(h_in, ws, biases, {0}, 0.01)
        (y ** 2).sum().backward()
        return y.detach(), coeff.grad.clone(), basis.grad.clone(), W1.grad.clone()

    a, b = run(True), run(False)
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() <= 2e-4 * v.abs().max().item()
    a2 = run(True)
    for u, v in zip(a, a2):
        assert torch.equal(u, v)
    assert h_in.grad is None and weights[0].grad is None


def test_fused_training_step_timing_vs_pytorch(hip_device):
    """forward + backward of the deform-sized network on 100 k points, all gradients: fused vs layer-by-layer autograd (printed)."""
    import time
    from splatfields_amd.fused_mlp import fused_general_mlp
    dev = hip_device
    weights, biases = make_net(94, 128, 6, [3], 3, dev, seed=1)
    for t in weights + biases:
        t.requires_grad_()
    h_in = torch.randn(100_000, 94, device=dev).requires_grad_()
    dY = torch.randn(100_000, 3, device=dev)

    def step(fused):
        y = fused_general_mlp(h_in, weights, biases, skips=[3]) if fused else general_mlp_reference(h_in, weights, biases, {3}, 0.01)
        torch.autograd.grad(y, [h_in] + weights + biases, dY)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3

    t_fused, t_torch = timed(lambda: step(True)), timed(lambda: step(False))
    print(f"\nfused MLP forward+backward, 100k points, 94 -> 128 x 7 -> 3: {t_fused:.3f} ms; PyTorch-ROCm autograd: {t_torch:.3f} ms")
    assert t_fused < t_torch


@pytest.mark.parametrize("n", [50, 1000, 100_003])
def test_weight_grad_kernel_matches_float64_products(hip_device, n):
    """sr_mlp_weight_grad on its own: several jobs of different shapes in one launch, ragged point counts, padding columns
    holding NaN (must not reach the results), bias sums, bit-identical repeats."""
    import ctypes as C
    from splatfields_amd import _lib
    lib, dev = _lib.load(), hip_device
    g = torch.Generator().manual_seed(n)
    dz_a = torch.randn(n, 128, generator=g).to(dev)
    dz_b = torch.zeros(n, 32, device=dev)
    dz_b[:, :3] = torch.randn(n, 3, generator=g).to(dev)
    dz_b[:, 3:] = float("nan")                                     # beyond m: dropped
    x_in = torch.randn(n, 96, generator=g).to(dev)
    x_in[:, 94:] = float("nan")                                    # beyond k: dropped
    x_h = torch.randn(n, 128, generator=g).to(dev)
    x_wide = torch.randn(n, 200, generator=g).to(dev)
    dw1, db1 = torch.full((128, 94 + 128), 7.0, device=dev), torch.full((128,), 7.0, device=dev)
    dw2, db2 = torch.full((3, 128), 7.0, device=dev), torch.full((3,), 7.0, device=dev)
    dw3 = torch.full((128, 200), 7.0, device=dev)
    J = _lib.SrMlpGradJob
    jobs = (J * 4)(J(dz=dz_a.data_ptr(), x=x_in.data_ptr(), dw=dw1.data_ptr(), db=db1.data_ptr(), dz_row=128, m=128, x_row=96, k=94, dw_row=222, dw_col0=0),
                   J(dz=dz_a.data_ptr(), x=x_h.data_ptr(), dw=dw1.data_ptr(), db=None, dz_row=128, m=128, x_row=128, k=128, dw_row=222, dw_col0=94),
                   J(dz=dz_b.data_ptr(), x=x_h.data_ptr(), dw=dw2.data_ptr(), db=db2.data_ptr(), dz_row=32, m=3, x_row=128, k=128, dw_row=128, dw_col0=0),
                   J(dz=dz_a.data_ptr(), x=x_wide.data_ptr(), dw=dw3.data_ptr(), db=None, dz_row=128, m=128, x_row=200, k=200, dw_row=200, dw_col0=0))
    nbytes = lib.sr_mlp_weight_grad_workspace(n, 4, jobs)
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def run():
        _lib.check(lib.sr_mlp_weight_grad(n, 4, jobs, C.c_void_p(ws.data_ptr()), nbytes, stream))
        torch.cuda.synchronize()
        return [t.clone() for t in (dw1, db1, dw2, db2, dw3)]

    got = run()
    a64, b64 = dz_a.double(), dz_b[:, :3].double()
    want = [torch.cat([a64.t() @ x_in[:, :94].double(), a64.t() @ x_h.double()], 1), a64.sum(0), b64.t() @ x_h.double(), b64.sum(0),
            a64.t() @ x_wide.double()]
    for u, v in zip(got, want):
        assert torch.isfinite(u).all()
        assert (u.double() - v).abs().max().item() <= 2e-5 * v.abs().max().item() + 1e-6
    for u, v in zip(got, run()):
        assert torch.equal(u, v)
    assert lib.sr_mlp_weight_grad(n, 4, jobs, C.c_void_p(ws.data_ptr()), nbytes - 16, stream) != 0     # workspace too small


def test_resfield_composition_kernel(hip_device):
    """All ResField layers of a network composed by one launch (sr_resfield_compose) and differentiated by sr_resfield_backward,
    against the PyTorch statement `W + (weights_t[frame] @ matrix_t).view_as(W)` (reference utils/resfields.py:229,294-300) at the
    reference's rank 40 / 100 frames and its layer shapes; bit-reproducible."""
    from splatfields_amd.general_mlp import ResFieldLinear, compose_resfield_weights
    dev = hip_device
    torch.manual_seed(3)
    layers = [ResFieldLinear(94, 128), ResFieldLinear(128, 128, 40, 100), ResFieldLinear(222, 128, 40, 100), ResFieldLinear(146, 64, 7, 100),
              ResFieldLinear(128, 3)]
    layers = [l.to(dev) for l in layers]
    frame = torch.tensor(37, device=dev)
    probes = [torch.randn_like(l.weight) for l in layers]

    def grads(fn):
        for l in layers:
            for p in l.parameters():
                p.grad = None
        ws = fn()
        sum((w * p).sum() for w, p in zip(ws, probes)).backward()
        out = [w.detach().clone() for w in ws]
        g = {f"{i}.{k}": p.grad.clone() for i, l in enumerate(layers) for k, p in l.named_parameters() if p.grad is not None}
        return out, g

    w_ref, g_ref = grads(lambda: [l.effective(frame) for l in layers])
    w1, g1 = grads(lambda: compose_resfield_weights(layers, frame))
    w2, g2 = grads(lambda: compose_resfield_weights(layers, 37))          # a Python int works too
    assert compose_resfield_weights(layers, frame)[0] is layers[0].weight   # layers without a residual pass through untouched
    for a, b, c in zip(w_ref, w1, w2):
        assert torch.equal(b, c) and (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())
    assert set(g1) == set(g_ref)
    for k in g_ref:
        assert torch.equal(g1[k], g2[k]), k
        err = (g1[k] - g_ref[k]).abs().max().item() / max(g_ref[k].abs().max().item(), 1e-12)
        assert err <= 2e-5, (k, err)
    assert (g1["1.weights_t"][torch.arange(100, device=dev) != 37] == 0).all()    # only the frame's row receives a gradient


@pytest.mark.parametrize("n,k,m", [(100_000, 48, 48), (5000, 128, 3), (777, 128, 12), (64, 48, 48)])
def test_point_linear_gradients_match_float64(hip_device, n, k, m):
    """PointLinear (tri-plane refine MLP, flow head: reference utils/time_utils.py:331-333, :78-113) = nn.Linear whose weight and
    bias gradients come from the slab kernel: output identical to F.linear, gradients against float64 autograd <= 1e-5 of each
    tensor's largest entry, and bit-reproducible."""
    from splatfields_amd.fused_mlp import PointLinear
    dev = hip_device
    torch.manual_seed(n + k + m)
    lin = PointLinear(k, m).to(dev)
    x = torch.randn(n, k, device=dev, requires_grad=True)
    g = torch.randn(n, m, device=dev)
    y = lin(x)
    assert torch.equal(y, F.linear(x, lin.weight, lin.bias))
    y.backward(g)
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x64 = x.detach().double().requires_grad_(True)
    W64, b64 = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
    F.linear(x64, W64, b64).backward(g.double())
    for a, b in zip(got, (x64.grad, W64.grad, b64.grad)):
        assert ((a.double() - b).abs().max() / b.abs().max()).item() < 1e-5
    x.grad = None
    lin.zero_grad(set_to_none=True)
    lin(x).backward(g)
    assert torch.equal(lin.weight.grad, got[1]) and torch.equal(lin.bias.grad, got[2])
    with torch.no_grad():                       # nothing to save, the plain library call
        assert torch.equal(lin(x), y)


@pytest.mark.parametrize("d_in,hidden,n_hidden,skips,out", [(94, 128, 6, [3], 3), (82, 64, 4, [2], 3)])
@pytest.mark.parametrize("slope", [0.01, 0.0])
def test_sign_bits_carry_the_activation_derivative(hip_device, d_in, hidden, n_hidden, skips, out, slope):
    """The forward writes one bit per saved activation (SrMlpOp.sign_store, include/splatraster.h); the backward chain reads
    leaky'(.) off those bits instead of the activations: same words as `acts > 0`, and bit-identical dZ / dL/dx0 to the chain
    that masks with the float activations."""
    from splatfields_amd import fused_mlp as fm
    dev = hip_device
    n = 3000
    weights, biases = make_net(d_in, hidden, n_hidden, skips, out, dev, 11)
    g = torch.Generator().manual_seed(5)
    shape = fm._Shape(weights, d_in, skips)
    x0 = F.pad(torch.randn(n, d_in, generator=g), (0, shape.mem_pad - d_in)).to(dev).contiguous()
    dY = torch.randn(n, out, generator=g).to(dev)
    y, acts, signs = fm._forward(shape, x0, weights, biases, slope, True)
    c = torch.arange(hidden, device=dev)
    word, bit = (c % 16) // 4, 4 * (c // 16) + c % 4
    assert torch.equal(((signs[:, :, word] >> bit) & 1).bool(), acts > 0)
    a = fm._backward(shape, x0, acts, y, dY, weights, slope, True, signs)
    b = fm._backward(shape, x0, acts, y, dY, weights, slope, True)
    assert all(torch.equal(u, v) for u, v in zip(a, b))


def test_time_embedding_inside_the_input_kernel(hip_device):
    """`fused_general_mlp_points(..., time=t, time_multires=3)` == the same network fed `cat([features, positional_encoding(t, 3)])`
    (what the reference passes: utils/time_utils.py:455-456), output and every gradient; and sr_mlp_top_gradient / the un-cleared
    dL/dx0 leave the padding columns zero."""
    from splatfields_amd import fused_mlp as fm
    from splatfields_amd.general_mlp import positional_encoding
    dev = hip_device
    n, F_, L, TL = 5000, 48, 6, 3
    d_in = 3 * (1 + 2 * L) + F_ + 1 + 2 * TL
    weights, biases = make_net(d_in, 128, 6, [3], 3, dev, 3)
    for t_ in weights + biases:
        t_.requires_grad_(True)
    g = torch.Generator().manual_seed(9)
    xyz = torch.randn(n, 3, generator=g).to(dev).requires_grad_(True)
    feat = torch.randn(n, F_, generator=g).to(dev).requires_grad_(True)
    t = torch.rand(n, 1, generator=g).to(dev)
    dY = torch.randn(n, 3, generator=g).to(dev)
    leaves = [xyz, feat] + weights + biases
    a = fm.fused_general_mlp_points(xyz, feat, L, weights, biases, skips=[3], time=t, time_multires=TL)
    ga = torch.autograd.grad(a, leaves, dY)
    b = fm.fused_general_mlp_points(xyz, torch.cat([feat, positional_encoding(t, TL)], dim=-1), L, weights, biases, skips=[3])
    gb = torch.autograd.grad(b, leaves, dY)
    assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()
    for u, v in zip(ga, gb):
        assert (u - v).abs().max().item() <= 1e-5 * v.abs().max().item()
    shape = fm._Shape(weights, d_in, [3])
    x0 = F.pad(torch.randn(n, d_in, generator=g), (0, shape.mem_pad - d_in)).to(dev).contiguous()
    wd, bd = [w.detach() for w in weights], [b_.detach() for b_ in biases]
    y, acts, signs = fm._forward(shape, x0, wd, bd, 0.01, True)
    dx0, G, dz = fm._backward(shape, x0, acts, y, dY, wd, 0.01, True, signs)
    assert (dx0[:, d_in:] == 0).all() and (G[:, 3:] == 0).all()
    assert torch.equal(G[:, :3], dY * torch.where(y > 0, 1.0, 0.01))
