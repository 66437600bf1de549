"""Drop-in counterpart of the reference's `GeneralMLP` (utils/time_utils.py:123-191) on the fused kernels.

Same constructor arguments, same `forward(xyz, xyz_feat=None, frame_id=None)`, same parameter names and shapes --
`net.<i>.weight`, `net.<i>.bias`, and for the ResField layers (reference utils/resfields.py:9-80, the configuration
GeneralMLP builds: compression 'vm', mode 'lookup', fuse 'add') `net.<i>.weights_t` [capacity, rank] and
`net.<i>.matrix_t` [rank, out * in] -- so a `deform.pth` written by the reference loads with `load_state_dict`
(reference scene/deform_model.py:36-47).  What differs is how it runs:

* the layer loop is one fused kernel forward, one for the activation gradients and one for all weight gradients backward
  (splatfields_amd/fused_mlp.py);
* a ResField layer composes ONLY the current frame's weight, `W + (weights_t[frame_id] @ matrix_t).view_as(W)`; the
  reference builds the [capacity, out * in] matrix of all frames on every call and indexes it
  (utils/resfields.py:229,294-300) -- same values, 1/capacity of the work.  All ResField layers of the network are composed
  by ONE kernel launch (`compose_resfield_weights`, csrc/mlp.hip: sr_resfield_compose) and their `weights_t` / `matrix_t`
  gradients come from one more pair of launches, instead of ~8 small PyTorch kernels per layer and step.

Supported: the configurations `SplatFields` constructs (utils/time_utils.py:343-447) -- act 'leaky_relu' (or 'relu'), hidden width 64 or
128, any out_activation of the reference's table.  Anything else raises; there is no PyTorch-layer or CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import torch
from torch import nn

from . import _lib
from .fused_mlp import _Shape, fused_general_mlp, fused_general_mlp_points


def positional_encoding(x: torch.Tensor, multires: int) -> torch.Tensor:
    """reference utils/time_utils.py:9-57 (`get_embedder`): [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...]."""
    if multires <= 0:
        return x
    out = [x]
    for j in range(multires):
        f = float(2 ** j)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, dim=-1)


class ResFieldLinear(nn.Module):
    """Parameters of one reference `resfields.Linear` in its GeneralMLP configuration; `effective(frame_id)` is the weight the
    layer applies at that frame."""

    def __init__(self, in_features: int, out_features: int, rank: int = 0, capacity: int = 0):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.rank, self.capacity = int(rank or 0), int(capacity or 0)
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))              # torch.nn.Linear's defaults, as the reference inherits them
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)
        if self.rank > 0 and self.capacity > 0:
            self.matrix_t = nn.Parameter(0.01 * torch.randn(self.rank, out_features * in_features))
            self.weights_t = nn.Parameter(0.01 * torch.randn(self.capacity, self.rank))

    @property
    def has_residual(self) -> bool:
        return self.rank > 0 and self.capacity > 0

    def effective(self, frame_id) -> torch.Tensor:
        if not self.has_residual:
            return self.weight
        if frame_id is None:
            raise ValueError("a ResField layer needs frame_id")
        coeff = self.weights_t[frame_id].reshape(1, self.rank)            # frame_id: int or 0-dim tensor (no host sync)
        return self.weight + (coeff @ self.matrix_t).view_as(self.weight)


class _ResFieldCompose(torch.autograd.Function):
    """W_eff of the current frame for ALL ResField layers of a network: one launch forward, two backward
    (include/splatraster.h: sr_resfield_compose / sr_resfield_backward).  Inputs: frame [] int64 on the device, then
    (weight, weights_t, matrix_t) per layer; outputs: one composed weight per layer."""

    @staticmethod
    def forward(ctx, frame: torch.Tensor, *params):
        lib = _lib.load()
        n = len(params) // 3
        if n < 1 or n > _lib.RESFIELD_MAX_JOBS:
            raise ValueError("between 1 and %d ResField layers per call" % _lib.RESFIELD_MAX_JOBS)
        dev = params[0].device
        keep = lambda t_: t_ if t_.is_contiguous() else t_.detach().contiguous()
        Ws, wts, Ms = [keep(p) for p in params[0::3]], [keep(p) for p in params[1::3]], [keep(p) for p in params[2::3]]
        frame = frame.detach().to(device=dev, dtype=torch.int64).reshape(1).contiguous()
        outs = [torch.empty_like(W) for W in Ws]
        jobs = (_lib.SrResFieldJob * n)()
        for j in range(n):
            jobs[j].w, jobs[j].weights_t, jobs[j].matrix_t, jobs[j].out = Ws[j].data_ptr(), wts[j].data_ptr(), Ms[j].data_ptr(), outs[j].data_ptr()
            jobs[j].count, jobs[j].rank, jobs[j].capacity = Ws[j].numel(), wts[j].shape[1], wts[j].shape[0]
        with torch.cuda.device(dev):
            _lib.check(lib.sr_resfield_compose(n, jobs, C.c_void_p(frame.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        ctx.save_for_backward(frame, *wts, *Ms)
        ctx.n = n
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        lib = _lib.load()
        n = ctx.n
        frame, *rest = ctx.saved_tensors
        wts, Ms = rest[:n], rest[n:]
        dev = frame.device
        gs = [(g if g is not None else torch.zeros_like(M[0]).view(-1)).contiguous() for g, M in zip(gouts, Ms)]
        d_wts = [torch.empty_like(t) if ctx.needs_input_grad[2 + 3 * j] else None for j, t in enumerate(wts)]
        d_Ms = [torch.empty_like(t) if ctx.needs_input_grad[3 + 3 * j] else None for j, t in enumerate(Ms)]
        jobs = (_lib.SrResFieldJob * n)()
        for j in range(n):
            jobs[j].weights_t, jobs[j].matrix_t, jobs[j].d_out = wts[j].data_ptr(), Ms[j].data_ptr(), gs[j].data_ptr()
            jobs[j].d_matrix_t = d_Ms[j].data_ptr() if d_Ms[j] is not None else None
            jobs[j].d_weights_t = d_wts[j].data_ptr() if d_wts[j] is not None else None
            jobs[j].count, jobs[j].rank, jobs[j].capacity = Ms[j].shape[1], wts[j].shape[1], wts[j].shape[0]
        ws_bytes = lib.sr_resfield_backward_workspace(n, jobs)
        if ws_bytes == 0:
            raise ValueError("unsupported ResField job list")
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_resfield_backward(n, jobs, C.c_void_p(frame.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        grads = [None]
        for j in range(n):       # dL/dW is dL/dW_eff itself
            grads += [gouts[j] if ctx.needs_input_grad[1 + 3 * j] else None, d_wts[j], d_Ms[j]]
        return tuple(grads)


def compose_resfield_weights(layers: Sequence["ResFieldLinear"], frame_id) -> list:
    """effective weights of all `layers` at `frame_id` (plain weights for layers without a residual): the residual layers are
    composed together by the fused kernel when they live on a HIP device, by `ResFieldLinear.effective` otherwise."""
    res = [l for l in layers if l.has_residual]
    if not res:
        return [l.weight for l in layers]
    if frame_id is None:
        raise ValueError("a ResField layer needs frame_id")
    fusable = all(l.weight.is_cuda and l.weight.dtype == torch.float32 and l.weight.numel() % 4 == 0 and l.rank <= _lib.RESFIELD_MAX_RANK
                  for l in res) and len(res) <= _lib.RESFIELD_MAX_JOBS
    if not fusable:
        return [l.effective(frame_id) for l in layers]
    if not torch.is_tensor(frame_id):
        # host-side frame index: the reference's `mat[frame_id]` raises IndexError outside [-capacity, capacity) -- so does this.
        # (A device-side index cannot be checked without a synchronisation: the kernel then poisons the composed weights with NaN.)
        f = int(frame_id)
        for l in res:
            cap = int(l.weights_t.shape[0])
            if not -cap <= f < cap:
                raise IndexError(f"frame_id {f} is out of range for a ResField layer with capacity {cap} (n_frames)")
        frame_id = f % int(res[0].weights_t.shape[0]) if f < 0 else f
    frame = frame_id if torch.is_tensor(frame_id) else torch.tensor(int(frame_id), dtype=torch.int64, device=res[0].weight.device)
    flat = []
    for l in res:
        flat += [l.weight, l.weights_t, l.matrix_t]
    composed = iter(_ResFieldCompose.apply(frame, *flat))
    return [next(composed) if l.has_residual else l.weight for l in layers]


class _Normalize(torch.autograd.Function):
    """`F.normalize(x, dim=-1)` (eps 1e-12) with a five-kernel backward: autograd's own chain through norm -> clamp -> expand ->
    div is ~12 small kernels per step for the [N, 4] rotations."""

    @staticmethod
    def forward(ctx, x):
        n = x.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        y = x / n
        ctx.save_for_backward(y, n)
        return y

    @staticmethod
    def backward(ctx, g):
        y, n = ctx.saved_tensors
        dot = (g * y).sum(dim=-1, keepdim=True)
        # below the clamp the norm is a constant: d (x / eps) = g / eps
        return torch.where(n > 1e-12, g - y * dot, g) / n


_OUT_ACTIVATIONS = {
    "none": lambda x: x,
    "sigmoid": torch.sigmoid,
    "tanh": torch.tanh,
    "relu": torch.relu,
    "selu": torch.selu,
    "softplus": nn.functional.softplus,
    "softmax": lambda x: nn.functional.softmax(x, dim=-1),
    "elu": nn.functional.elu,
    "normalize": _Normalize.apply,
    "leaky_relu": nn.functional.leaky_relu,
}
_SLOPES = {"leaky_relu": 0.01, "relu": 0.0}


class GeneralMLP(nn.Module):
    def __init__(self, in_features: int = 3, out_features: int = 3, hidden_features: int = 128, num_hidden_layers: int = 8,
                 skips: Sequence[int] = (4,), multires: int = 6, out_activation: str = "none", act: str = "relu",
                 composition_rank: int = 0, n_frames: int = 100):
        super().__init__()
        if act not in _SLOPES:
            raise NotImplementedError(f"act={act!r}: the fused kernels implement leaky_relu and relu")
        if out_activation not in _OUT_ACTIVATIONS:
            raise KeyError(out_activation)
        self.out_features, self.input_ch, self.multires = out_features, in_features, multires
        self.skips = list(skips)
        self.slope = _SLOPES[act]
        self.out_act = _OUT_ACTIVATIONS[out_activation]
        d_in = in_features - 3 + 3 * (1 + 2 * max(multires, 0))
        self.d_in = d_in
        # net[0], net[1 + i] for i < num_hidden_layers, net[-1]; layer ids 1 .. num_hidden_layers - 1 (= net[2:-1]) are
        # ResField layers, net[1 + i] takes [h_in | h] when i is in skips (reference utils/time_utils.py:137-160)
        layers = [ResFieldLinear(d_in, hidden_features)]
        for i in range(num_hidden_layers):
            residual = 1 <= i <= num_hidden_layers
            layers.append(ResFieldLinear(hidden_features + (d_in if i in self.skips else 0), hidden_features,
                                         rank=composition_rank if residual else 0,
                                         capacity=n_frames if residual and composition_rank > 0 else 0))
        layers.append(ResFieldLinear(hidden_features, out_features))
        self.net = nn.ModuleList(layers)
        self._shape: Optional[_Shape] = None

    def _static_shape(self) -> _Shape:
        if self._shape is None:
            # the fused op numbers skips as the reference's forward does: the concatenation follows net[i] for i in skips
            self._shape = _Shape([l.weight for l in self.net], self.d_in, self.skips)
        return self._shape

    def forward(self, xyz: torch.Tensor, xyz_feat: Optional[torch.Tensor] = None, frame_id=None, time: Optional[torch.Tensor] = None,
                time_multires: int = 0) -> torch.Tensor:
        """`time` (optional, one value per point, no gradient): its positional encoding is appended behind `xyz_feat` -- the tail
        of the feature vector the reference passes in (utils/time_utils.py:455-456) -- inside the input kernel."""
        weights = compose_resfield_weights(list(self.net), frame_id)
        biases = [layer.bias for layer in self.net]
        if xyz.is_cuda and xyz.dim() == 2 and xyz.shape[1] == 3 and xyz.dtype == torch.float32 and \
                (xyz_feat is None or (xyz_feat.dtype == torch.float32 and xyz_feat.dim() == 2)):
            # the usual case: positions + features (+ time) -> the padded input matrix in one kernel
            h = fused_general_mlp_points(xyz, xyz_feat, self.multires, weights, biases, skips=self.skips, negative_slope=self.slope,
                                         _shape=self._static_shape(), time=time, time_multires=time_multires)
            return self.out_act(h)
        h_in = positional_encoding(xyz, self.multires)
        parts = [h_in] + ([xyz_feat] if xyz_feat is not None else []) + \
            ([positional_encoding(time.reshape(-1, 1), time_multires)] if time is not None else [])
        h_in = torch.cat(parts, dim=-1) if len(parts) > 1 else h_in
        h = fused_general_mlp(h_in, weights, biases, skips=self.skips, negative_slope=self.slope, _shape=self._static_shape())
        return self.out_act(h)
