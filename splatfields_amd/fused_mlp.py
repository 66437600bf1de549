"""Fused forward of the SplatFields deform network's MLPs (include/splatraster.h: sr_mlp_forward; csrc/mlp.hip).

`FusedGeneralMLP` evaluates one `GeneralMLP` of reference utils/time_utils.py:123-191 -- `h = act(layer_i(h))` for every
layer, `h = cat([h_in, h])` after the layers listed in `skips` -- for all points in one kernel (activations in registers,
exact fp32 MFMA).  It takes the layers' *effective* weights: for ResField layers (reference utils/resfields.py:378-405) the
caller composes `W + delta(frame)` first.  Forward only (rendering / evaluation of trained 4-D models, reference render.py);
the backward is the next step of this row.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch
import torch.nn.functional as F

from . import _lib


def pack_layer_weight(W: torch.Tensor, n_mem: int, mem_pad: int, reg_width: int, out_tiles: int) -> torch.Tensor:
    """W [M, n_mem + n_reg] (the reference's column order: network input first, hidden state after it) -> the packed K order
    of csrc/mlp.hip: float ((((c MT + mt) 2 + tl) 64 + 16 k + m) 4 + i) = W'[16 mt + m][16 (2 c + tl) + 4 k + i], where W' is W
    with the input block zero-padded to `mem_pad` columns, the hidden block to `reg_width`, and the rows to 16 `out_tiles`."""
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


class FusedGeneralMLP:
    """weights[j]: [out_j, in_j] float32 on the device, biases[j]: [out_j]; in_0 = d_in, in_j = hidden (+ d_in when j - 1 is in
    `skips`), out_last = the network's output width.  Hidden width 64 or 128."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], d_in: int, skips: Sequence[int] = (),
                 negative_slope: float = 0.01):
        self.weights, self.biases = list(weights), list(biases)
        self.d_in, self.slope = int(d_in), float(negative_slope)
        self.hidden = int(self.weights[0].shape[0])
        if self.hidden not in (64, 128):
            raise ValueError("FusedGeneralMLP supports hidden widths 64 and 128")
        self.skips = {int(s) for s in skips if 0 <= int(s) < len(self.weights) - 1}
        self.out_features = int(self.weights[-1].shape[0])
        if self.out_features > self.hidden:
            raise ValueError("the output may not be wider than the hidden layers")
        self.mem_pad = (self.d_in + 31) // 32 * 32             # two 16-channel tiles per weight chunk
        self._packed = None
        self._versions = None
        for j, W in enumerate(self.weights):
            want = self.d_in if j == 0 else self.hidden + (self.d_in if (j - 1) in self.skips else 0)
            if W.shape[1] != want:
                raise ValueError(f"layer {j}: expected {want} input features, got {W.shape[1]}")

    def _pack(self):
        versions = tuple((w._version, b._version) for w, b in zip(self.weights, self.biases))
        if self._packed is not None and versions == self._versions:
            return self._packed
        ht = self.hidden // 16
        packed, descs = [], []
        last = len(self.weights) - 1
        for j, (W, b) in enumerate(zip(self.weights, self.biases)):
            W = W.detach().to(torch.float32)
            n_mem = self.d_in if (j == 0 or (j - 1) in self.skips) else 0
            out_tiles = ht if j < last else (self.out_features + 15) // 16
            wp = pack_layer_weight(W, n_mem, self.mem_pad if n_mem else 0, 0 if j == 0 else self.hidden, out_tiles)
            bp = F.pad(b.detach().to(torch.float32), (0, 16 * out_tiles - b.shape[0])).contiguous()
            packed.append((wp, bp))
            descs.append((out_tiles, (self.mem_pad // 16) if n_mem else 0, 0 if j == 0 else ht))
        arr = (_lib.SrMlpLayer * len(descs))()
        for j, ((wp, bp), (ot, mt, rt)) in enumerate(zip(packed, descs)):
            arr[j] = _lib.SrMlpLayer(wp.data_ptr(), bp.data_ptr(), ot, mt, rt)
        self._packed, self._versions = (packed, arr), versions   # `packed` keeps the device buffers alive
        return self._packed

    @torch.no_grad()
    def __call__(self, h_in: torch.Tensor) -> torch.Tensor:
        """h_in [N, d_in] (positional encoding ++ features, as the reference builds it) -> [N, out_features]."""
        lib = _lib.load()
        if not h_in.is_cuda:
            raise RuntimeError("FusedGeneralMLP has no CPU path: tensors must be on a HIP ('cuda') device")
        if h_in.dim() != 2 or h_in.shape[1] != self.d_in:
            raise ValueError(f"h_in must be [N, {self.d_in}]")
        dev, n = h_in.device, h_in.shape[0]
        x0 = F.pad(h_in.detach().to(torch.float32), (0, self.mem_pad - self.d_in)).contiguous()
        y = torch.empty(n, self.out_features, dtype=torch.float32, device=dev)
        _, arr = self._pack()
        with torch.cuda.device(dev):
            _lib.check(lib.sr_mlp_forward(n, self.hidden // 16, len(self.weights), arr, C.c_void_p(x0.data_ptr()), self.mem_pad,
                                          C.c_void_p(y.data_ptr()), self.out_features, self.slope,
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return y
