"""Where the time of one fused GeneralMLP training step goes (100 k points, the deform-sized network): the pieces of
splatfields_amd/fused_mlp.py timed one by one next to PyTorch-ROCm autograd.  Prints one JSON line.  GPU only."""
import argparse
import json
import sys
import os

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from splatfields_amd import fused_mlp as fm  # noqa: E402


def timed(fn, steps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--d-in", type=int, default=94)
    ap.add_argument("--out", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    L, H, D = a.layers, a.hidden, a.d_in
    skips = [3] if L > 5 else []
    dims_in = [D] + [H + (D if (j - 1) in skips else 0) for j in range(1, L)]
    dims_out = [H] * (L - 1) + [a.out]
    weights = [(torch.randn(o, i, generator=g) / i ** 0.5).to(dev).requires_grad_() for i, o in zip(dims_in, dims_out)]
    biases = [(0.1 * torch.randn(o, generator=g)).to(dev).requires_grad_() for o in dims_out]
    h_in = torch.randn(a.points, D, generator=g).to(dev).requires_grad_()
    dY = torch.randn(a.points, a.out, generator=g).to(dev)
    shape = fm._Shape(weights, D, skips)
    wd, bd = [w.detach() for w in weights], [b.detach() for b in biases]
    x0 = F.pad(h_in.detach(), (0, shape.mem_pad - D)).contiguous()
    res = {"points": a.points, "hidden": H, "layers": L}

    def ref(x):
        h = x
        for i, (W, b) in enumerate(zip(weights, biases)):
            h = F.leaky_relu(F.linear(h, W, b), 0.01)
            if i in skips and i != L - 1:
                h = torch.cat([x, h], dim=-1)
        return h

    res["torch_fwd_bwd_ms"] = timed(lambda: torch.autograd.grad(ref(h_in), [h_in] + weights + biases, dY))
    res["fused_fwd_bwd_ms"] = timed(lambda: torch.autograd.grad(fm.fused_general_mlp(h_in, weights, biases, skips=skips), [h_in] + weights + biases, dY))
    with torch.no_grad():
        res["torch_fwd_ms"] = timed(lambda: ref(h_in))
        res["fused_fwd_nosave_ms"] = timed(lambda: fm._forward(shape, x0, wd, bd, 0.01, save=False))
        res["fused_fwd_save_ms"] = timed(lambda: fm._forward(shape, x0, wd, bd, 0.01, save=True))
        y, acts, signs = fm._forward(shape, x0, wd, bd, 0.01, save=True)
        res["fused_bwd_chain_ms"] = timed(lambda: fm._backward(shape, x0, acts, y, dY, wd, 0.01, True, signs))
        res["fused_bwd_chain_float_masks_ms"] = timed(lambda: fm._backward(shape, x0, acts, y, dY, wd, 0.01, True))
        res["fused_bwd_chain_noinput_ms"] = timed(lambda: fm._backward(shape, x0, acts, y, dY, wd, 0.01, False, signs))
        dx0, gz, dz = fm._backward(shape, x0, acts, y, dY, wd, 0.01, True, signs)

        def dws():
            out = []
            for j in range(L):
                dZt = (gz[:, :a.out] if j == L - 1 else dz[j]).t()
                if j == 0:
                    out.append((dZt @ x0)[:, :D])
                elif shape.reads_input[j]:
                    out.append(torch.cat([(dZt @ x0)[:, :D], dZt @ acts[j - 1]], dim=1))
                else:
                    out.append(dZt @ acts[j - 1])
            return out
        res["dW_gemms_ms"] = timed(dws)
        res["fused_weight_grads_ms"] = timed(lambda: fm._weight_grads(shape, x0, acts, gz, dz, wd))
        res["db_sums_ms"] = timed(lambda: [(gz[:, :a.out] if j == L - 1 else dz[j]).sum(0) for j in range(L)])
        res["one_dW_gemm_128x128_ms"] = timed(lambda: dz[1].t() @ acts[0])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
