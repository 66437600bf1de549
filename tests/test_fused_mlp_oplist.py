"""Host logic of the fused MLP (splatfields_amd/fused_mlp.py) without a GPU: the pack jobs and op lists it hands to
sr_mlp_pack / sr_mlp_chain are interpreted here with plain PyTorch on CPU tensors (an op = one matrix applied to
[memory channels | running state] + epilogue, include/splatraster.h), and the forward, dL/dh_in and the dZ of every layer
they produce are compared with float64 autograd of the reference's GeneralMLP formula (utils/time_utils.py:178-191).  The
kernels themselves are covered by tests/test_gpu_fused_mlp.py."""
import contextlib
import ctypes

import pytest
import torch
import torch.nn.functional as F


def view(ptr, rows, row, cols):
    n = (rows - 1) * row + cols
    arr = (ctypes.c_float * n).from_address(ptr)
    t = torch.frombuffer(arr, dtype=torch.float32)
    return torch.as_strided(t, (rows, cols), (row, 1))

def bit_view(ptr, rows):
    arr = (ctypes.c_int32 * (4 * rows)).from_address(ptr)
    return torch.frombuffer(arr, dtype=torch.int32).view(rows, 4)


def channel_bit_index(channels):
    """SrMlpOp.sign_store / mask_bits (include/splatraster.h): channel 16 t + 4 k + i  <->  word k, bit 4 t + i"""
    c = torch.arange(channels)
    return (c % 16) // 4, 4 * (c // 16) + c % 4


def job_matrix(job):
    K = job["n_mem"] + job["n_reg"]
    ld = job["ld"]
    R = 16 * job["out_tiles"]
    A = torch.zeros(R, job["mem_pad"] + job["reg_width"])
    nr = job["n_rows"]
    def col(c): return c
    if job["transposed"]:
        # element = W[col][row0+r]
        ncols_src = max(job["n_mem"], job["reg_col0"] + job["n_reg"])
        W = view(job["w"], ncols_src, ld, job["row0"] + nr)
        if job["n_mem"]: A[:nr, :job["n_mem"]] = W[:job["n_mem"], job["row0"]:job["row0"]+nr].t()
        if job["n_reg"]: A[:nr, job["mem_pad"]:job["mem_pad"]+job["n_reg"]] = W[job["reg_col0"]:job["reg_col0"]+job["n_reg"], job["row0"]:job["row0"]+nr].t()
    else:
        W = view(job["w"], job["row0"] + nr, ld, job["reg_col0"] + job["n_reg"] if job["n_reg"] else job["n_mem"])
        if job["n_mem"]: A[:nr, :job["n_mem"]] = W[job["row0"]:job["row0"]+nr, :job["n_mem"]]
        if job["n_reg"]: A[:nr, job["mem_pad"]:job["mem_pad"]+job["n_reg"]] = W[job["row0"]:job["row0"]+nr, job["reg_col0"]:job["reg_col0"]+job["n_reg"]]
    return A

def interpret(self, lib, ptrs, device, n, ht, slope, stream):
    """stands in for _Plan.run: the plan's pointer fields are symbolic (name, index, byte offset) and are resolved here"""
    def resolved(d, keys):
        return {k: (ptrs[v[0]][v[1]] + v[2] if (k in keys and v is not None) else v) for k, v in d.items()}
    state = torch.zeros(n, 16 * ht)
    for job, op in zip(self.jobs, self.ops):
        job, op = resolved(job, ("w", "bias")), resolved(op, ("src", "mask", "store", "sign_store", "mask_bits"))
        A = job_matrix(job)
        mem_t, reg_t = job["mem_pad"], job["reg_width"]
        parts = []
        if mem_t: parts.append(view(op["src"], n, op["src_row"], mem_t))
        if reg_t: parts.append(state[:, :reg_t])
        X = torch.cat(parts, 1)
        acc = X @ A.t()
        if job.get("bias"):
            b = view(job["bias"], 1, job["n_bias"], job["n_bias"])[0]
            acc[:, :job["n_bias"]] += b
        ep = op["epilogue"]
        if ep == 1: acc = torch.maximum(acc, slope * acc)
        elif ep == 2:
            if op.get("mask_bits"):
                word, bit = channel_bit_index(acc.shape[1])
                m = (bit_view(op["mask_bits"], n)[:, word] >> bit) & 1
            else:
                m = view(op["mask"], n, op["mask_row"], acc.shape[1])
            acc = acc * torch.where(m > 0, 1.0, slope)
        if op.get("sign_store"):
            word, bit = channel_bit_index(acc.shape[1])
            words = torch.zeros(n, 4, dtype=torch.int64)
            words.index_add_(1, word, (acc > 0).to(torch.int64) << bit)
            bit_view(op["sign_store"], n).copy_(words.to(torch.int32))
        if op.get("store"):
            dst = view(op["store"], n, op["store_row"], op["store_channels"])
            if op.get("store_accumulate"): dst += acc[:, :op["store_channels"]]
            else: dst.copy_(acc[:, :op["store_channels"]])
        if not op.get("keep_state"):
            state = torch.zeros(n, 16 * ht); state[:, :acc.shape[1]] = acc


def reference(h_in, ws, bs, skips, slope):
    h = h_in
    for i, (W, b) in enumerate(zip(ws, bs)):
        h = F.leaky_relu(F.linear(h, W, b), slope)
        if i in skips and i != len(ws) - 1:
            h = torch.cat([h_in, h], -1)
    return h


@pytest.mark.parametrize("d_in,H,nh,skips,out", [(94, 128, 6, [3], 3), (82, 64, 4, [2], 3), (130, 64, 2, [0, 1], 16),
                                                  (94, 128, 6, [3, 6], 128), (33, 128, 1, [], 5)])
def test_op_lists_compute_the_network_and_its_gradients(monkeypatch, d_in, H, nh, skips, out):
    from splatfields_amd import fused_mlp as fm, _lib

    class _Stream:
        cuda_stream = 0
    monkeypatch.setattr(_lib, "load", lambda: None)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: _Stream())
    monkeypatch.setattr(fm._Plan, "run", interpret)
    # sr_mlp_top_gradient (include/splatraster.h): dY * leaky'(y), zero-padded to the row the backward chain reads
    monkeypatch.setattr(fm, "_top_gradient", lambda lib, shape, y, dY, slope, G: G.copy_(
        F.pad(dY * torch.where(y > 0, 1.0, slope), (0, shape.out_pad - shape.out_features))))
    g = torch.Generator().manual_seed(1)
    L = nh + 2
    dims_in = [d_in] + [H + (d_in if (j - 1) in skips else 0) for j in range(1, L)]
    dims_out = [H] * (L - 1) + [out]
    ws = [torch.randn(o, i, generator=g) / i ** 0.5 for i, o in zip(dims_in, dims_out)]
    bs = [0.1 * torch.randn(o, generator=g) for o in dims_out]
    n = 37
    h_in, dY = torch.randn(n, d_in, generator=g), torch.randn(n, out, generator=g)
    shape = fm._Shape(ws, d_in, skips)
    x0 = F.pad(h_in, (0, shape.mem_pad - d_in)).contiguous()
    y, acts, signs = fm._forward(shape, x0, ws, bs, 0.05, True)
    dx0, G, dz = fm._backward(shape, x0, acts, y, dY, ws, 0.05, True, signs)
    dx0_f, G_f, dz_f = fm._backward(shape, x0, acts, y, dY, ws, 0.05, True)        # leaky' read off the activations themselves
    assert torch.equal(dx0, dx0_f) and torch.equal(dz, dz_f)
    hr = h_in.double().requires_grad_()
    wr, br = [w.double().requires_grad_() for w in ws], [b.double().requires_grad_() for b in bs]
    yr = reference(hr, wr, br, set(skips), 0.05)
    yr.backward(dY.double())
    assert (y - yr).abs().max().item() <= 1e-5
    assert (dx0[:, :d_in] - hr.grad).abs().max().item() <= 1e-5 and (dx0[:, d_in:] == 0).all()
    for j in range(L):                                    # the contraction sr_mlp_weight_grad performs on these buffers
        dZ = G[:, :out] if j == L - 1 else dz[j]
        segments = ([x0[:, :d_in]] if shape.reads_input[j] else []) + ([acts[j - 1]] if j > 0 else [])
        assert (dZ.t() @ torch.cat(segments, 1) - wr[j].grad).abs().max().item() <= 5e-5
        assert (dZ.sum(0) - br[j].grad).abs().max().item() <= 5e-5
    assert (G[:, out:] == 0).all()


def test_shape_validation():
    from splatfields_amd.fused_mlp import _Shape
    w = [torch.zeros(128, 94), torch.zeros(128, 128), torch.zeros(3, 128)]
    s = _Shape(w, 94, [5])                                 # a skip index past the hidden layers is ignored, as in the reference
    assert s.reads_input == [True, False, False] and s.mem_pad == 96 and s.out_pad == 32 and s.input_groups == [(0, 6)]
    with pytest.raises(ValueError):
        _Shape([torch.zeros(100, 94), torch.zeros(3, 100)], 94, [])          # hidden width
    with pytest.raises(ValueError):
        _Shape(w, 94, [0])                                                   # layer 1 would need 94 + 128 inputs
    with pytest.raises(ValueError):
        _Shape(w[:1], 94, [])


def test_argument_validation_happens_before_any_launch(monkeypatch):
    """bad arguments are refused by the host side (no GPU needed to see the message); CPU tensors are refused outright."""
    from splatfields_amd import fused_mlp as fm, _lib
    monkeypatch.setattr(_lib, "load", lambda: None)
    w = [torch.zeros(64, 20), torch.zeros(64, 64), torch.zeros(3, 64)]
    b = [torch.zeros(64), torch.zeros(64), torch.zeros(3)]
    with pytest.raises(RuntimeError, match="no CPU path"):
        fm.fused_general_mlp(torch.zeros(5, 20), w, b)


def test_packed_layout_is_the_documented_index_formula():
    """pack_layer_weight (the PyTorch statement the device packer is tested against) element by element against the layout
    csrc/mlp.hip documents: float ((((c MT + mt) 2 + tl) 64 + 16 k + m) 4 + i) = W'[16 mt + m][16 (2 c + tl) + 4 k + i]."""
    from splatfields_amd.fused_mlp import pack_layer_weight
    g = torch.Generator().manual_seed(2)
    M, n_mem, n_reg, mem_pad, reg_width, MT = 40, 21, 50, 32, 64, 3
    W = torch.randn(M, n_mem + n_reg, generator=g)
    packed = pack_layer_weight(W, n_mem, mem_pad, reg_width, MT)
    Wp = torch.zeros(16 * MT, mem_pad + reg_width)
    Wp[:M, :n_mem] = W[:, :n_mem]
    Wp[:M, mem_pad:mem_pad + n_reg] = W[:, n_mem:]
    assert packed.numel() == 16 * MT * (mem_pad + reg_width)
    for c in range((mem_pad + reg_width) // 32):
        for mt in range(MT):
            for tl in range(2):
                for lane in range(64):
                    k, m = lane >> 4, lane & 15
                    for i in range(4):
                        d = ((((c * MT + mt) * 2 + tl) * 64 + lane) * 4 + i)
                        assert packed[d] == Wp[16 * mt + m][16 * (2 * c + tl) + 4 * k + i]
