"""Fused MLPs of the SplatFields deform network, forward and backward (include/splatraster.h: sr_mlp_chain, sr_mlp_pack,
sr_mlp_weight_grad; csrc/mlp.hip).

`fused_general_mlp` evaluates one `GeneralMLP` of reference utils/time_utils.py:123-191 -- `h = act(layer_i(h))` for every
layer, `h = cat([h_in, h])` after the layers listed in `skips`, act = leaky ReLU -- for all points in one kernel
(activations in registers, exact fp32 MFMA), and is differentiable: the backward runs the activation-gradient chain
dZ_{l-1} = (W_l^T dZ_l) * act'(.) and dL/dh_in in one more kernel of the same shape, and the weight gradients
dW_l = dZ_l^T [h_in | h_{l-1}], db_l = sum dZ_l of ALL layers in one launch of a kernel that splits the points over the
chip (library GEMMs serialise this contraction: 2.7 of the 3.9 ms of a PyTorch-ROCm step of one 128 x 8 network).  It takes the
layers' *effective* weights as ordinary autograd tensors: for ResField layers (reference utils/resfields.py:378-405) the
caller composes `W + delta(frame)` first and autograd carries dW on to W and the delta factors.  `FusedGeneralMLP` is the
module-style wrapper.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import _lib


def pack_layer_weight(W: torch.Tensor, n_mem: int, mem_pad: int, reg_width: int, out_tiles: int) -> torch.Tensor:
    """PyTorch statement of the packing (the device packer sr_mlp_pack is tested against it).  W [M, n_mem + n_reg] (the
    reference's column order: network input first, hidden state after it) -> the packed K order of csrc/mlp.hip:
    float ((((c MT + mt) 2 + tl) 64 + 16 k + m) 4 + i) = W'[16 mt + m][16 (2 c + tl) + 4 k + i], where W' is W with the input
    block zero-padded to `mem_pad` columns, the hidden block to `reg_width`, and the rows to 16 `out_tiles`."""
    M, K = W.shape
    n_reg = K - n_mem
    Wp = W.new_zeros(16 * out_tiles, mem_pad + reg_width)
    if n_mem:
        Wp[:M, :n_mem] = W[:, :n_mem]
    if n_reg:
        Wp[:M, mem_pad:mem_pad + n_reg] = W[:, n_mem:]
    kt = (mem_pad + reg_width) // 16
    v = Wp.view(out_tiles, 16, kt // 2, 2, 4, 4)            # (mt, m, c, tl, k, i)
    return v.permute(2, 0, 3, 4, 1, 5).contiguous().reshape(-1)   # (c, mt, tl, k, m, i)


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """the tensor as contiguous float32 -- itself when it already is (the usual case: one attribute
    test instead of three tensor methods; ~50 of these per network and step are host time the GPU waits for)"""
    if t.dtype is torch.float32 and t.is_contiguous():
        return t          # only its data pointer is used; autograd.Function inputs may be saved as they are
    return t.detach().to(torch.float32).contiguous()


class _Shape:
    """Static description of one GeneralMLP: which layers read the network input, tile counts, validity."""

    def __init__(self, weights: Sequence[torch.Tensor], d_in: int, skips: Sequence[int]):
        self.n_layers = len(weights)
        if self.n_layers < 2:
            raise ValueError("a fused MLP has at least two layers")
        self.d_in = int(d_in)
        self.hidden = int(weights[0].shape[0])
        if self.hidden not in (64, 128):
            raise ValueError("fused MLPs support hidden widths 64 and 128")
        self.ht = self.hidden // 16
        self.skips = {int(s) for s in skips if 0 <= int(s) < self.n_layers - 1}
        self.out_features = int(weights[-1].shape[0])
        if self.out_features > self.hidden:
            raise ValueError("the output may not be wider than the hidden layers")
        self.out_tiles_last = (self.out_features + 15) // 16
        self.out_pad = (self.out_features + 31) // 32 * 32      # the top gradient is a memory input: two tiles per chunk
        self.mem_pad = (self.d_in + 31) // 32 * 32
        self.reads_input = [j == 0 or (j - 1) in self.skips for j in range(self.n_layers)]
        for j, W in enumerate(weights):
            want = self.d_in if j == 0 else self.hidden + (self.d_in if self.reads_input[j] else 0)
            rows = self.hidden if j < self.n_layers - 1 else self.out_features
            if tuple(W.shape) != (rows, want):
                raise ValueError(f"layer {j}: expected weight [{rows}, {want}], got {tuple(W.shape)}")
        # input-gradient ops write <= ht tiles each
        self.input_groups = [(g, min(self.ht, self.mem_pad // 16 - g)) for g in range(0, self.mem_pad // 16, self.ht)
                             if self.d_in - 16 * g > 0]
        self.layer_dims = [tuple(W.shape) for W in weights]
        self.plans: dict = {}     # descriptor arrays of the forward / backward / weight-gradient launches, built on first use


_PACK_PTRS, _OP_PTRS = ("w", "bias_src"), ("src", "mask", "store", "sign_store", "mask_bits")


class _Plan:
    """One pack launch + one chain launch of a fixed shape, with the ctypes descriptor arrays built ONCE: everything static
    (tile counts, strides, epilogues) is filled in when the plan is made; pointer fields are symbolic -- ("W", j, byte offset)
    = weights[j].data_ptr() + offset -- and are patched per call (building ~40 descriptor structs per network and step in
    Python cost ~0.6 ms of host time per network; patching ~60 pointers costs a few tens of microseconds)."""

    def __init__(self):
        self.jobs: List[dict] = []
        self.ops: List[dict] = []
        self.sizes: List[Tuple[int, int]] = []
        self._arrays = None
        self._lock = threading.Lock()

    def add(self, job: dict, op: dict, with_bias: bool):
        floats = 16 * job["out_tiles"] * (job["mem_pad"] + job["reg_width"])
        self.jobs.append(job)
        self.sizes.append((floats, 16 * job["out_tiles"] if with_bias else 0))
        self.ops.append(op)

    def _freeze(self):
        if len(self.jobs) > _lib.MLP_MAX_PACK_JOBS or len(self.ops) > _lib.MLP_MAX_OPS:
            raise ValueError("network too deep for one fused chain")
        n = len(self.jobs)
        jobs = (_lib.SrMlpPackJob * n)()
        ops = (_lib.SrMlpOp * n)()
        binds, at = [], 0     # (struct, field, symbol, index, byte offset)
        for j, (job, op, (wf, bf)) in enumerate(zip(self.jobs, self.ops, self.sizes)):
            J, O = jobs[j], ops[j]
            J.ld, J.transposed, J.row0, J.n_rows = job["ld"], job["transposed"], job["row0"], job["n_rows"]
            J.n_mem, J.mem_pad, J.mem_col0 = job["n_mem"], job["mem_pad"], 0
            J.n_reg, J.reg_width, J.reg_col0 = job["n_reg"], job["reg_width"], job["reg_col0"]
            J.out_tiles, J.n_bias = job["out_tiles"], job.get("n_bias", 0)
            O.out_tiles, O.mem_tiles, O.reg_tiles = job["out_tiles"], job["mem_pad"] // 16, job["reg_width"] // 16
            O.src_row, O.epilogue, O.mask_row = op.get("src_row", 0), op["epilogue"], op.get("mask_row", 0)
            O.store_row, O.store_channels = op.get("store_row", 0), op.get("store_channels", 0)
            O.store_accumulate, O.keep_state = op.get("store_accumulate", 0), op.get("keep_state", 0)
            binds.append((J, "dst", "buf", 0, 4 * at)); binds.append((O, "w_packed", "buf", 0, 4 * at))
            if bf:
                binds.append((J, "bias_dst", "buf", 0, 4 * (at + wf))); binds.append((O, "bias", "buf", 0, 4 * (at + wf)))
            at += wf + bf
            for f, key in (("w", "w"), ("bias_src", "bias")):
                if job.get(key) is not None:
                    binds.append((J, f) + tuple(job[key]))
            for f in _OP_PTRS:
                if op.get(f) is not None:
                    binds.append((O, f) + tuple(op[f]))
        self._arrays = (jobs, ops, binds, at)

    def run(self, lib, ptrs: dict, device, n_points: int, ht: int, slope: float, stream) -> torch.Tensor:
        with self._lock:      # the descriptor arrays are shared by every call with this shape
            if self._arrays is None:
                self._freeze()
            jobs, ops, binds, total = self._arrays
            buf = torch.empty(total, dtype=torch.float32, device=device)
            ptrs = dict(ptrs, buf=(buf.data_ptr(),))
            for obj, field, sym, idx, off in binds:
                setattr(obj, field, ptrs[sym][idx] + off)
            _lib.check(lib.sr_mlp_pack(len(jobs), jobs, C.c_void_p(stream)))
            _lib.check(lib.sr_mlp_chain(n_points, ht, len(ops), ops, slope, C.c_void_p(stream)))
        return buf      # alive until the caller drops it; the stream orders its reuse


def _forward_plan(shape: _Shape, save: bool) -> _Plan:
    key = ("fwd", save)
    if key not in shape.plans:
        H, L = shape.hidden, shape.n_layers
        b = _Plan()
        for j in range(L):
            last = j == L - 1
            rows, cols = shape.layer_dims[j]
            n_mem = shape.d_in if shape.reads_input[j] else 0
            job = dict(w=("W", j, 0), ld=cols, transposed=0, row0=0, n_rows=rows, n_mem=n_mem,
                       mem_pad=shape.mem_pad if n_mem else 0, n_reg=cols - n_mem, reg_width=0 if j == 0 else H, reg_col0=n_mem,
                       out_tiles=shape.out_tiles_last if last else shape.ht, bias=("B", j, 0), n_bias=rows)
            op = dict(epilogue=_lib.MLP_LEAKY, src=("x0", 0, 0) if n_mem else None, src_row=shape.mem_pad)
            if last:
                op.update(store=("y", 0, 0), store_row=shape.out_features, store_channels=shape.out_features)
            elif save:       # the activation itself (an operand of dW of the layer above) and its signs (leaky' for the backward chain)
                op.update(store=("acts", j, 0), store_row=H, store_channels=H, sign_store=("signs", j, 0))
            b.add(job, op, with_bias=True)
        shape.plans[key] = b
    return shape.plans[key]


def _forward(shape: _Shape, x0: torch.Tensor, weights, biases, slope: float, save: bool):
    """x0 [N, mem_pad] -> (y [N, out], acts [L-1, N, hidden] or None, signs [L-1, N, 4] int32 or None: one bit per activation,
    set where it is > 0, in the layout of SrMlpOp.sign_store)."""
    lib = _lib.load()
    dev, n, H, L = x0.device, x0.shape[0], shape.hidden, shape.n_layers
    y = torch.empty(n, shape.out_features, dtype=torch.float32, device=dev)
    acts = torch.empty(L - 1, n, H, dtype=torch.float32, device=dev) if save else None
    signs = torch.empty(L - 1, n, 4, dtype=torch.int32, device=dev) if save else None
    ptrs = {"W": [w.data_ptr() for w in weights], "B": [b_.data_ptr() for b_ in biases], "x0": (x0.data_ptr(),), "y": (y.data_ptr(),),
            "acts": [acts.data_ptr() + 4 * j * n * H for j in range(L - 1)] if save else (),
            "signs": [signs.data_ptr() + 16 * j * n for j in range(L - 1)] if save else ()}
    with torch.cuda.device(dev):
        _forward_plan(shape, save).run(lib, ptrs, dev, n, shape.ht, slope, torch.cuda.current_stream(dev).cuda_stream)
    return y, acts, signs


def _backward_plan(shape: _Shape, need_input: bool, bits: bool) -> _Plan:
    key = ("bwd", need_input, bits)
    if key not in shape.plans:
        H, L = shape.hidden, shape.n_layers
        b = _Plan()
        wrote_input = False        # dL/dx0 is not cleared beforehand: the first layer that feeds it (the topmost) stores, incl. zero padding
        for j in range(L - 1, -1, -1):
            ld = shape.layer_dims[j][1]
            top = j == L - 1
            n_mem_w = shape.d_in if shape.reads_input[j] else 0        # columns of W_j that multiply the network input
            # this layer's dZ is the op input: memory (G) for the top layer, the register state below it
            cols = dict(n_mem=shape.out_features, mem_pad=shape.out_pad, n_reg=0, reg_width=0, reg_col0=0) if top else \
                dict(n_mem=0, mem_pad=0, n_reg=H, reg_width=H, reg_col0=0)
            src = dict(src=("G", 0, 0), src_row=shape.out_pad) if top else {}
            if need_input and n_mem_w:
                for g0, cnt in shape.input_groups:                      # dL/dx0 (+)= W_j[:, input block]^T dZ_j, <= ht tiles at a time
                    rows = min(16 * cnt, shape.d_in - 16 * g0)
                    job = dict(w=("W", j, 0), ld=ld, transposed=1, row0=16 * g0, n_rows=rows, out_tiles=cnt, **cols)
                    # packed rows beyond `rows` are zero: storing all 16 cnt channels also writes the padding columns' zeros
                    op = dict(epilogue=_lib.MLP_NONE, keep_state=1, store=("dx0", 0, 4 * 16 * g0), store_row=shape.mem_pad,
                              store_channels=rows if wrote_input else 16 * cnt, store_accumulate=1 if wrote_input else 0, **src)
                    b.add(job, op, with_bias=False)
                wrote_input = True
            if j > 0:                                                   # dZ_{j-1} = (W_j[:, hidden block]^T dZ_j) * act'(h_{j-1})
                job = dict(w=("W", j, 0), ld=ld, transposed=1, row0=n_mem_w, n_rows=H, out_tiles=shape.ht, **cols)
                mask = dict(mask_bits=("signs", j - 1, 0)) if bits else dict(mask=("acts", j - 1, 0), mask_row=H)
                op = dict(epilogue=_lib.MLP_MASK, store=("dz", j - 1, 0), store_row=H, store_channels=H, **mask, **src)
                b.add(job, op, with_bias=False)
        shape.plans[key] = b
    return shape.plans[key]


def _top_gradient(lib, shape: _Shape, y, dY, slope: float, G):
    """G[:, :out] = dY * leaky'(y), zero padding behind: one launch (sr_mlp_top_gradient)"""
    dev = y.device
    with torch.cuda.device(dev):
        _lib.check(lib.sr_mlp_top_gradient(y.shape[0], shape.out_features, shape.out_pad, C.c_void_p(y.data_ptr()), C.c_void_p(dY.data_ptr()),
                                           slope, C.c_void_p(G.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


def _backward(shape: _Shape, x0, acts, y, dY, weights, slope: float, need_input: bool, signs=None):
    """-> (dL/dx0 [N, mem_pad] or None, G [N, out_pad] = dZ of the last layer (zero-padded), dz [L-1, N, hidden] = dZ of the others).
    leaky'(.) of the hidden layers is read off `signs` (16 bytes per point and layer) when given, else off `acts` (4 * hidden)."""
    lib = _lib.load()
    dev, n, H, L = x0.device, x0.shape[0], shape.hidden, shape.n_layers
    G = torch.empty(n, shape.out_pad, dtype=torch.float32, device=dev)
    dz = torch.empty(L - 1, n, H, dtype=torch.float32, device=dev)
    dx0 = torch.empty(n, shape.mem_pad, dtype=torch.float32, device=dev) if need_input else None   # the chain's first input op stores
    _top_gradient(lib, shape, y, dY, slope, G)
    ptrs = {"W": [w.data_ptr() for w in weights], "G": (G.data_ptr(),), "dx0": (dx0.data_ptr() if need_input else 0,),
            "acts": [acts.data_ptr() + 4 * j * n * H for j in range(L - 1)], "dz": [dz.data_ptr() + 4 * j * n * H for j in range(L - 1)],
            "signs": [signs.data_ptr() + 16 * j * n for j in range(L - 1)] if signs is not None else ()}
    with torch.cuda.device(dev):
        _backward_plan(shape, need_input, signs is not None).run(lib, ptrs, dev, n, shape.ht, slope, torch.cuda.current_stream(dev).cuda_stream)
    return dx0, G, dz


def _grad_jobs(shape: _Shape):
    """static part of the weight-gradient job list: (layer, dz symbol, dz_row, m, x symbol, x_row, k, dw_row, dw_col0, first)"""
    if "grad" not in shape.plans:
        H, L, out = shape.hidden, shape.n_layers, []
        for j in range(L):
            top = j == L - 1
            dz_sym, dz_row, m = (("G", 0), shape.out_pad, shape.out_features) if top else (("dz", j), H, H)
            col0, first = 0, True
            segments = []
            if shape.reads_input[j]:
                segments.append((("x0", 0), shape.mem_pad, shape.d_in))
            if j > 0:
                segments.append((("acts", j - 1), H, H))
            for x_sym, x_row, k in segments:
                out.append((j, dz_sym, dz_row, m, x_sym, x_row, k, shape.layer_dims[j][1], col0, first))
                col0 += k
                first = False
        if len(out) > _lib.MLP_MAX_GRAD_JOBS:
            raise ValueError("network too deep for one weight-gradient launch")
        arr = (_lib.SrMlpGradJob * len(out))()
        for i, (j, _, dz_row, m, _, x_row, k, dw_row, col0, _) in enumerate(out):
            arr[i].dz_row, arr[i].m, arr[i].x_row, arr[i].k, arr[i].dw_row, arr[i].dw_col0 = dz_row, m, x_row, k, dw_row, col0
        shape.plans["grad"] = (out, arr, threading.Lock())
    return shape.plans["grad"]


def _weight_grads(shape: _Shape, x0, acts, G, dz, weights):
    """dW_j = dZ_j^T [h_in | h_{j-1}], db_j = sum over points of dZ_j, all layers in one launch (sr_mlp_weight_grad).  The kernel
    addresses its operands with 32-bit byte offsets: larger point sets are cut into chunks whose partial gradients are added
    (a reference-sized network reaches that limit at ~2.4 M points)."""
    lib = _lib.load()
    dev, n, H, L = x0.device, x0.shape[0], shape.hidden, shape.n_layers
    jobs, arr, lock = _grad_jobs(shape)
    row_bytes = 4 * max(shape.mem_pad, shape.out_pad, H)
    n_max = max(64, ((1 << 31) - 1) // row_bytes // 64 * 64)
    total_W, total_b = None, None
    for lo in range(0, n, n_max):
        cnt = min(n_max, n - lo)
        dWs = [torch.empty_like(W) for W in weights]
        dbs = [torch.empty(W.shape[0], dtype=torch.float32, device=dev) for W in weights]
        base = {"G": (G.data_ptr() + 4 * lo * shape.out_pad,), "x0": (x0.data_ptr() + 4 * lo * shape.mem_pad,),
                "dz": [dz.data_ptr() + 4 * (j * n + lo) * H for j in range(L - 1)],
                "acts": [acts.data_ptr() + 4 * (j * n + lo) * H for j in range(L - 1)]}
        with lock:
            for i, (j, dz_sym, _, _, x_sym, _, _, _, _, first) in enumerate(jobs):
                arr[i].dz, arr[i].x = base[dz_sym[0]][dz_sym[1]], base[x_sym[0]][x_sym[1]]
                arr[i].dw, arr[i].db = dWs[j].data_ptr(), (dbs[j].data_ptr() if first else None)
            ws_bytes = lib.sr_mlp_weight_grad_workspace(cnt, len(jobs), arr)
            if ws_bytes == 0:
                raise ValueError("unsupported weight-gradient job list: " + lib.sr_last_error().decode("utf-8", "replace"))
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.sr_mlp_weight_grad(cnt, len(jobs), arr, C.c_void_p(ws.data_ptr()), ws_bytes,
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        if total_W is None:
            total_W, total_b = dWs, dbs
        else:      # fixed chunk order: still deterministic
            for a_, b_ in zip(total_W, dWs):
                a_.add_(b_)
            for a_, b_ in zip(total_b, dbs):
                a_.add_(b_)
    return total_W, total_b


class _FusedMLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h_in, shape: _Shape, slope: float, grad_enabled: bool, *params):
        L = shape.n_layers
        weights = [_f32c(p) for p in params[:L]]
        biases = [_f32c(p) for p in params[L:]]
        x0 = F.pad(h_in.detach().to(torch.float32), (0, shape.mem_pad - shape.d_in)).contiguous()
        # needs_input_grad reflects requires_grad of the inputs, not the grad mode (inside forward() grad mode is always off):
        # under torch.no_grad() nothing will run backward, so the [L-1, N, hidden] activation stack is neither allocated nor written
        need = grad_enabled and any(ctx.needs_input_grad)
        y, acts, signs = _forward(shape, x0, weights, biases, slope, save=need)
        if need:
            ctx.shape, ctx.slope = shape, slope
            ctx.save_for_backward(x0, acts, signs, y, *weights)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dY):
        shape, slope = ctx.shape, ctx.slope
        x0, acts, signs, y, *weights = ctx.saved_tensors
        L, d_in = shape.n_layers, shape.d_in
        need_input = ctx.needs_input_grad[0]
        dx0, G, dz = _backward(shape, x0, acts, y, dY.to(torch.float32).contiguous(), weights, slope, need_input, signs)
        dWs, dbs = [None] * L, [None] * L
        if any(ctx.needs_input_grad[4:]):
            dWs, dbs = _weight_grads(shape, x0, acts, G, dz, weights)
            dWs = [g if ctx.needs_input_grad[4 + j] else None for j, g in enumerate(dWs)]
            dbs = [g if ctx.needs_input_grad[4 + L + j] else None for j, g in enumerate(dbs)]
        return (dx0[:, :d_in] if need_input else None, None, None, None, *dWs, *dbs)


class _FusedMLPPointsFn(torch.autograd.Function):
    """The same network fed from the points themselves: the input matrix [x | sin / cos encodings | features | padding] is
    built by one kernel (sr_mlp_input_forward) straight into the padded layout the chain kernel reads, and its backward
    (sr_mlp_input_backward) turns dL/dx0 into dL/dxyz and dL/dfeatures -- instead of ~60 small PyTorch kernels per network
    and step for the encoding, the concatenations, the padding and their backward."""

    @staticmethod
    def forward(ctx, xyz, feat, time, shape: _Shape, slope: float, multires: int, time_multires: int, grad_enabled: bool, *params):
        lib = _lib.load()
        L = shape.n_layers
        weights = [_f32c(p) for p in params[:L]]
        biases = [_f32c(p) for p in params[L:]]
        dev, n = xyz.device, xyz.shape[0]
        x32 = _f32c(xyz)
        f32 = _f32c(feat) if feat is not None else None
        t32 = time.detach().to(torch.float32).reshape(-1).contiguous() if time is not None else None
        n_feat = 0 if f32 is None else f32.shape[1]
        x0 = torch.empty(n, shape.mem_pad, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_mlp_input_forward(n, multires, n_feat, time_multires, shape.mem_pad, C.c_void_p(x32.data_ptr()),
                                                C.c_void_p(f32.data_ptr()) if f32 is not None else None,
                                                C.c_void_p(t32.data_ptr()) if t32 is not None else None, C.c_void_p(x0.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        need = grad_enabled and any(ctx.needs_input_grad)
        y, acts, signs = _forward(shape, x0, weights, biases, slope, save=need)
        if need:
            ctx.shape, ctx.slope, ctx.multires, ctx.n_feat = shape, slope, multires, n_feat
            ctx.save_for_backward(x32, x0, acts, signs, y, *weights)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dY):
        lib = _lib.load()
        shape, slope = ctx.shape, ctx.slope
        x32, x0, acts, signs, y, *weights = ctx.saved_tensors
        L, dev, n = shape.n_layers, x0.device, x0.shape[0]
        need_xyz, need_feat = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and ctx.n_feat > 0
        dx0, G, dz = _backward(shape, x0, acts, y, dY.to(torch.float32).contiguous(), weights, slope, need_xyz or need_feat, signs)
        d_xyz = torch.empty(n, 3, dtype=torch.float32, device=dev) if need_xyz else None
        d_feat = torch.empty(n, ctx.n_feat, dtype=torch.float32, device=dev) if need_feat else None
        if need_xyz or need_feat:
            with torch.cuda.device(dev):
                _lib.check(lib.sr_mlp_input_backward(n, ctx.multires, ctx.n_feat, shape.mem_pad, C.c_void_p(x32.data_ptr()),
                                                     C.c_void_p(dx0.data_ptr()), C.c_void_p(d_xyz.data_ptr()) if need_xyz else None,
                                                     C.c_void_p(d_feat.data_ptr()) if need_feat else None,
                                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        dWs, dbs = [None] * L, [None] * L
        if any(ctx.needs_input_grad[8:]):
            dWs, dbs = _weight_grads(shape, x0, acts, G, dz, weights)
            dWs = [g if ctx.needs_input_grad[8 + j] else None for j, g in enumerate(dWs)]
            dbs = [g if ctx.needs_input_grad[8 + L + j] else None for j, g in enumerate(dbs)]
        return (d_xyz, d_feat, None, None, None, None, None, None, *dWs, *dbs)


def fused_general_mlp_points(xyz: torch.Tensor, feat: Optional[torch.Tensor], multires: int, weights: Sequence[torch.Tensor],
                             biases: Sequence[torch.Tensor], skips: Sequence[int] = (), negative_slope: float = 0.01,
                             _shape: Optional[_Shape] = None, time: Optional[torch.Tensor] = None, time_multires: int = 0) -> torch.Tensor:
    """`fused_general_mlp(cat([positional_encoding(xyz, multires), feat, positional_encoding(time, time_multires)]), ...)` with the
    input matrix built on the device in one kernel: xyz [N, 3], feat [N, F] or None, time [N] / [N, 1] or None (one time per point,
    no gradient), float32 on a HIP device."""
    _lib.load()
    if not xyz.is_cuda:
        raise RuntimeError("fused_general_mlp_points has no CPU path: tensors must be on a HIP ('cuda') device")
    n_feat = 0 if feat is None else feat.shape[1]
    time_multires = int(max(time_multires, 0))
    n_time = 0 if time is None else 1 + 2 * time_multires
    d_in = 3 * (1 + 2 * max(multires, 0)) + n_feat + n_time
    shape = _shape or _Shape(weights, d_in, skips)
    if xyz.dim() != 2 or xyz.shape[1] != 3 or shape.d_in != d_in or (feat is not None and (feat.dim() != 2 or feat.shape[0] != xyz.shape[0])):
        raise ValueError(f"xyz must be [N, 3] and feat [N, {shape.d_in - 3 * (1 + 2 * max(multires, 0)) - n_time}]")
    if xyz.dtype != torch.float32 or (feat is not None and (feat.dtype != torch.float32 or feat.device != xyz.device)):
        raise ValueError("xyz and feat must be float32 tensors on the same device")
    if time is not None and (time.numel() != xyz.shape[0] or time.dtype != torch.float32 or time.device != xyz.device or time.requires_grad):
        raise ValueError("time must hold one float32 value per point on the same device and take no gradient")
    if not (0.0 <= negative_slope < 1.0) or len(biases) != shape.n_layers or len(weights) != shape.n_layers:
        raise ValueError("negative_slope must be in [0, 1) and there must be one weight and one bias per layer")
    if xyz.shape[0] == 0:
        return xyz.new_zeros(0, shape.out_features) + 0.0 * (xyz.sum() + sum(w.sum() for w in weights) + sum(b.sum() for b in biases))
    return _FusedMLPPointsFn.apply(xyz, feat, time, shape, float(negative_slope), int(max(multires, 0)), time_multires,
                                   torch.is_grad_enabled(), *weights, *biases)


def fused_general_mlp(h_in: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], skips: Sequence[int] = (),
                      negative_slope: float = 0.01, _shape: Optional[_Shape] = None) -> torch.Tensor:
    """h_in [N, d_in] (positional encoding ++ features, as the reference builds it) -> [N, out_features]; differentiable with
    respect to h_in, the weights and the biases.  weights[j]: [out_j, in_j] on the device, biases[j]: [out_j]; in_0 = d_in,
    in_j = hidden (+ d_in when j - 1 is in `skips`); hidden width 64 or 128."""
    _lib.load()
    if not h_in.is_cuda:
        raise RuntimeError("fused_general_mlp has no CPU path: tensors must be on a HIP ('cuda') device")
    shape = _shape or _Shape(weights, h_in.shape[1] if h_in.dim() == 2 else -1, skips)
    if h_in.dim() != 2 or h_in.shape[1] != shape.d_in:
        raise ValueError(f"h_in must be [N, {shape.d_in}]")
    if not (0.0 <= negative_slope < 1.0):
        raise ValueError("negative_slope must be in [0, 1)")
    if len(biases) != shape.n_layers:
        raise ValueError("one bias per layer")
    for j, (W, b) in enumerate(zip(weights, biases)):
        if tuple(b.shape) != (W.shape[0],):
            raise ValueError(f"layer {j}: bias must be [{W.shape[0]}]")
        for t in (W, b):
            if t.device != h_in.device or t.dtype != torch.float32:
                raise ValueError(f"layer {j}: weights and biases must be float32 tensors on {h_in.device}")
    if h_in.dtype != torch.float32:
        raise ValueError("h_in must be float32 (the reference network runs in fp32)")
    if h_in.shape[0] == 0:      # nothing to launch; keep the graph connected so that parameters still receive (zero) gradients
        return h_in.new_zeros(0, shape.out_features) + 0.0 * (h_in.sum() + sum(w.sum() for w in weights) + sum(b.sum() for b in biases))
    return _FusedMLPFn.apply(h_in, shape, float(negative_slope), torch.is_grad_enabled(), *weights, *biases)


class _PointLinearFn(torch.autograd.Function):
    """y = x W^T + b for a tall matrix of per-point rows ([N, in], N ~ 10^5, in and out a few dozen): the products with the
    weight are library GEMMs, but dL/dW = dY^T x and dL/db = column sums of dY contract over the POINTS -- the shape library
    GEMMs and reductions serialise (0.29 ms per 48 x 48 x 100k product and 0.14 ms per bias on MI355X; the tri-plane refine MLP
    and the flow head were 1.4 ms of an 8.5 ms network step) -- and go through the slab kernel of the fused MLPs
    (sr_mlp_weight_grad: deterministic, ~0.03 ms)."""

    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return torch.addmm(b, x, W.t()) if b is not None else x @ W.t()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dY):
        lib = _lib.load()
        x, W = ctx.saved_tensors
        dev, n, m, k = x.device, x.shape[0], W.shape[0], W.shape[1]
        need_x, need_W = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        dY = dY.contiguous()
        dX = dY @ W if need_x else None
        dW = db = None
        if need_W or need_b:
            m_pad = (m + 3) // 4 * 4
            dz = dY if m_pad == m else F.pad(dY, (0, m_pad - m))
            dW = torch.empty_like(W)
            db = torch.empty(m, dtype=torch.float32, device=dev)
            job = (_lib.SrMlpGradJob * 1)()
            job[0].dz_row, job[0].m, job[0].x_row, job[0].k, job[0].dw_row, job[0].dw_col0 = m_pad, m, k, k, k, 0
            n_max = max(64, ((1 << 31) - 1) // (4 * max(m_pad, k)) // 64 * 64)
            for lo in range(0, n, n_max):
                cnt = min(n_max, n - lo)
                pW, pb = (dW, db) if lo == 0 else (torch.empty_like(W), torch.empty_like(db))
                job[0].dz, job[0].x = dz.data_ptr() + 4 * lo * m_pad, x.data_ptr() + 4 * lo * k
                job[0].dw, job[0].db = pW.data_ptr(), pb.data_ptr()
                ws_bytes = lib.sr_mlp_weight_grad_workspace(cnt, 1, job)
                if ws_bytes == 0:
                    raise ValueError("unsupported weight-gradient job: " + lib.sr_last_error().decode("utf-8", "replace"))
                ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    _lib.check(lib.sr_mlp_weight_grad(cnt, 1, job, C.c_void_p(ws.data_ptr()), ws_bytes,
                                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
                if lo:
                    dW.add_(pW)
                    db.add_(pb)
        return dX, (dW if need_W else None), (db if need_b else None)


def point_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`F.linear(x, weight, bias)` for per-point rows x [N, in]; on a HIP device (float32, in a multiple of 4) the weight and
    bias gradients come from the fused MLPs' slab kernel instead of a library GEMM over N (see _PointLinearFn)."""
    ok = x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.shape[1] % 4 == 0 and \
        x.shape[0] > 0 and weight.device == x.device and (bias is None or (bias.dtype == torch.float32 and bias.device == x.device)) and \
        weight.shape[0] <= 64 * 255 and weight.shape[1] <= 64 * 255
    if not ok or not (torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad))):
        return F.linear(x, weight, bias)
    return _PointLinearFn.apply(x.contiguous(), weight.contiguous(), bias)


class PointLinear(torch.nn.Linear):
    """`nn.Linear` (same parameters, initialisation and state-dict keys) applied to per-point rows through `point_linear`."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return point_linear(x, self.weight, self.bias)


class FusedGeneralMLP:
    """Module-style wrapper: holds the weight / bias tensors (leaf parameters or per-step composed ResField weights may be
    swapped in through `weights` / `biases`) and the static shape."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], d_in: int, skips: Sequence[int] = (),
                 negative_slope: float = 0.01):
        self.weights, self.biases = list(weights), list(biases)
        self.slope = float(negative_slope)
        self.shape = _Shape(self.weights, d_in, skips)
        self.d_in, self.hidden, self.out_features = self.shape.d_in, self.shape.hidden, self.shape.out_features

    def __call__(self, h_in: torch.Tensor) -> torch.Tensor:
        return fused_general_mlp(h_in, self.weights, self.biases, negative_slope=self.slope, _shape=self.shape)
