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
  (utils/resfields.py:229,294-300) -- same values, 1/capacity of the work; autograd carries dL/dW_effective on to `weight`,
  `weights_t[frame_id]` and `matrix_t`.

Supported: the configurations `SplatFields` constructs (utils/time_utils.py:343-447) -- act 'leaky_relu' (or 'relu'), hidden width 64 or
128, any out_activation of the reference's table.  Anything else raises; there is no PyTorch-layer or CPU path.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
from torch import nn

from .fused_mlp import _Shape, fused_general_mlp


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


_OUT_ACTIVATIONS = {
    "none": lambda x: x,
    "sigmoid": torch.sigmoid,
    "tanh": torch.tanh,
    "relu": torch.relu,
    "selu": torch.selu,
    "softplus": nn.functional.softplus,
    "softmax": lambda x: nn.functional.softmax(x, dim=-1),
    "elu": nn.functional.elu,
    "normalize": nn.functional.normalize,
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

    def forward(self, xyz: torch.Tensor, xyz_feat: Optional[torch.Tensor] = None, frame_id=None) -> torch.Tensor:
        h_in = positional_encoding(xyz, self.multires)
        if xyz_feat is not None:
            h_in = torch.cat([h_in, xyz_feat], dim=-1)
        weights = [layer.effective(frame_id) for layer in self.net]
        biases = [layer.bias for layer in self.net]
        h = fused_general_mlp(h_in, weights, biases, skips=self.skips, negative_slope=self.slope, _shape=self._static_shape())
        return self.out_act(h)
