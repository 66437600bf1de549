"""Counterpart of the reference's `SplatFields` network (utils/time_utils.py:305-508) with its six `GeneralMLP`s on the fused
kernels (splatfields_amd/general_mlp.py).

Same constructor keywords, same `forward(xyz_in, t)` -> dict(scales, opacity, rotations, rgb | rgb_fnc, flow, means3D), same
parameter names (`mlp_deform.net.<i>...`, `mlp_refine_feat.<i>...`, `mlp_flow_head.branch_w...`), so the MLP / flow-head part of a
`deform.pth` checkpoint (reference scene/deform_model.py:36-47) loads with `load_state_dict`.  NOT drop-in for the reference's
DEFAULT encoder: its checkpoints carry the plane generator's weights (`encoder.subs.*`), which only load when the caller supplies
that generator (below); `load_state_dict` says so instead of failing on missing / unexpected keys.

The tri-plane feature encoder (reference scene/tripFields.py:383-436) is `splatfields_amd.triplane.TriPlaneSampler`: the
per-point lookup (three `grid_sample`s + cat) runs on HIP kernels, forward and backward.  With the reference's default
`encoder_type='VarTriPlaneEncoder'` and no `encoder=` argument the sampler owns its planes [3, out_ch, 16 noise_res,
16 noise_res] as a learnable parameter (a decoder-free tri-plane).  The reference GENERATES the planes with a diffusers / mmgen
VAE decoder (`Tensorial2D`, :176-204; those packages exist neither in its checkout nor in this image): pass such a generator
as `TriPlaneSampler(plane_source=...)`, or any module with the encoder's interface (`encoder(x[None]) -> [1, N, out_dim]`,
attribute `out_dim`, e.g. the reference's own `VarTriPlaneEncoder` instance) as `encoder=`.  An `encoder_type` outside the
reference's list runs without plane features (`feat_dim = 0`, utils/time_utils.py:333-334).  State-dict keys of the
decoder-free sampler are `encoder.planes`; a reference checkpoint's `encoder.subs.*` keys need the reference's generator.

`FlowHead` (utils/time_utils.py:194-303): 'offset', 'se3' (default) and 'dct'; the SE(3) exponential is written out per point
(no 4x4 batched matmuls), following the reference's formulas including its `w / theta + 1e-5` convention.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from .fused_mlp import PointLinear
from .general_mlp import GeneralMLP, positional_encoding


def _se3_parts(w: torch.Tensor, v: torch.Tensor, theta: torch.Tensor):
    """rotation R [N, 3, 3] and translation p [N, 3] of exp of the screw (w, v) * theta (see se3_transform), in ~20 batched
    kernels: the cross products are `torch.linalg.cross` calls, [w] is minus the cross product of w with the three basis
    vectors (row j of cross(w, e_j) is column j of [w], and [w] is antisymmetric)."""
    n = w.shape[0]
    s, c = torch.sin(theta), torch.cos(theta)                    # [N, 1]
    eye = torch.eye(3, device=w.device, dtype=w.dtype)
    skew = -torch.linalg.cross(w[:, None, :].expand(n, 3, 3), eye.expand(n, 3, 3), dim=-1)
    ww = (w * w).sum(-1, keepdim=True)
    skew2 = w[:, :, None] * w[:, None, :] - ww[:, :, None] * eye
    omc = 1.0 - c
    R = eye + s[:, :, None] * skew + omc[:, :, None] * skew2
    wv = torch.linalg.cross(w, v, dim=-1)
    wwv = torch.linalg.cross(w, wv, dim=-1)
    p = theta * v + omc * wv + (theta - s) * wwv
    return R, p


def se3_transform(w: torch.Tensor, v: torch.Tensor, theta: torch.Tensor):
    """exp of the screw (w, v) * theta as the reference builds it (utils/rigid_utils.py:40-84; Modern Robotics 3.51 / 3.88),
    without assuming |w| = 1 (the reference adds 1e-5 to the normalised axis):  R = I + sin(theta) [w] + (1 - cos(theta)) [w]^2,
    p = (theta I + (1 - cos(theta)) [w] + (theta - sin(theta)) [w]^2) v, with [w]^2 = w w^T - |w|^2 I.  -> [N, 4, 4]."""
    R, p = _se3_parts(w, v, theta)
    bottom = torch.zeros(1, 1, 4, device=w.device, dtype=w.dtype)   # (device-side fill: no host -> device copy, safe under graph capture)
    bottom[..., 3] = 1.0
    return torch.cat([torch.cat([R, p[:, :, None]], dim=-1), bottom.expand(w.shape[0], 1, 4)], dim=1)


def dct_basis(num_basis: int, num_frames: int) -> torch.Tensor:
    """reference utils/time_utils.py:60-70: sqrt(2 / T) cos(pi / (2 T) (2 t + 1) k), k = 1..K"""
    t = torch.arange(num_frames, dtype=torch.float64)[:, None]
    k = torch.arange(1, num_basis + 1, dtype=torch.float64)[None, :]
    return (math.sqrt(2.0 / num_frames) * torch.cos(math.pi / (2.0 * num_frames) * (2 * t + 1) * k)).to(torch.float32)


class FlowHead(nn.Module):
    def __init__(self, W: int = 256, flow_model: str = "offset", num_basis: int = 4, n_frames: int = 100):
        super().__init__()
        self.W, self.flow_model, self.n_frames, self.num_basis = W, flow_model, n_frames, num_basis
        if flow_model == "offset":
            self.gaussian_warp = PointLinear(W, 3)
        elif flow_model == "se3":
            self.branch_w = PointLinear(W, 3)
            self.branch_v = PointLinear(W, 3)
        elif flow_model == "dct":
            self.branch_coeff = PointLinear(W, 3 * num_basis)
            nn.init.zeros_(self.branch_coeff.weight)
            nn.init.zeros_(self.branch_coeff.bias)
            self.trajectory_basis = nn.Parameter(dct_basis(num_basis, n_frames * 2))
        else:
            raise NotImplementedError(f"flow_model={flow_model!r}: offset, se3 and dct are implemented")

    def forward(self, hidden, pts, time_step=None, frame_id=None):
        if self.flow_model == "offset":
            flow = self.gaussian_warp(hidden)
            return flow, pts + flow
        if self.flow_model == "se3":
            w, v = self.branch_w(hidden), self.branch_v(hidden)
            theta = torch.norm(w, dim=-1, keepdim=True)
            w = w / theta + 1e-5
            v = v / theta + 1e-5
            flow = se3_transform(w, v, theta)
            means3D = (flow[:, :3, :3] * pts[:, None, :]).sum(-1) + flow[:, :3, 3]
            return flow, means3D
        coeff = self.branch_coeff(hidden).view(-1, 3, self.num_basis)
        bases = self.trajectory_basis[frame_id.long() if torch.is_tensor(frame_id) else int(frame_id)]
        flow = (coeff * bases.view(1, 1, -1)).sum(-1)
        return flow, pts + flow


class SplatFields(nn.Module):
    def __init__(self, radius=None, n_frames: int = 0, encoder: Optional[nn.Module] = None, **kwargs):
        super().__init__()
        rank = kwargs.get("composition_rank", 0)
        self.n_frames = n_frames
        self.encoder_type = kwargs.get("encoder_type", "VarTriPlaneEncoder")
        if encoder is None and self.encoder_type in ["VarTriPlaneEncoder"]:
            from .triplane import TriPlaneSampler
            ea = kwargs.get("encoder_args", {}) or {}
            ignored = sorted(k for k in ea if k not in ("out_ch", "noise_res", "fuse_mode", "plane_source"))
            if ignored:
                import warnings
                warnings.warn("SplatFields: encoder_args %s configure the reference's plane GENERATOR (scene/tripFields.py:176-204), which "
                              "is not part of this package -- they are ignored by the decoder-free TriPlaneSampler; pass the generator as "
                              "encoder_args['plane_source'] or a whole encoder as encoder=" % ignored, stacklevel=2)
            encoder = TriPlaneSampler(out_ch=ea.get("out_ch", 16), resolution=16 * ea.get("noise_res", 20),
                                      fuse_mode=ea.get("fuse_mode", "cat"), plane_source=ea.get("plane_source"))
        if encoder is not None:
            self.encoder = encoder
            self.feat_dim = int(encoder.out_dim)
            self.mlp_refine_feat = nn.Sequential(PointLinear(self.feat_dim, self.feat_dim), nn.ReLU(), PointLinear(self.feat_dim, self.feat_dim))
        else:
            self.feat_dim = 0
        if n_frames > 0:
            self.time_multires = kwargs.get("time_multires", 3)
            time_ch = 1 + 2 * self.time_multires
        else:
            self.time_multires, time_ch = 0, 0
        self.deform_weight = kwargs.get("deform_weight", 1.0)
        in_ch = 3 + self.feat_dim + time_ch

        def mlp(prefix, out, w, d, skips, multires, out_act, in_features=in_ch):
            return GeneralMLP(in_features=in_features, out_features=out, hidden_features=kwargs.get(prefix + "_w", w),
                              num_hidden_layers=kwargs.get(prefix + "_d", d), skips=kwargs.get(prefix + "_skips", skips),
                              multires=multires, out_activation=out_act, act="leaky_relu", composition_rank=rank, n_frames=n_frames)

        self.mlp_deform = mlp("deform", 3, 128, 6, [3], kwargs.get("deform_multires", 6), "none")
        self.use_view_dep_rgb = kwargs.get("use_view_dep_rgb", False)
        self.mlp_rgb = mlp("rgb", kwargs.get("rgb_w", 128) if self.use_view_dep_rgb else 3, 128, 6, [3], kwargs.get("rgb_multires", 6),
                           "none" if self.use_view_dep_rgb else "sigmoid")
        if self.use_view_dep_rgb:
            self.mlp_rgb_viewdep = nn.Sequential(nn.Linear(3 + self.mlp_rgb.out_features, 3), nn.Sigmoid())
        self.geo_model_disable_pts = bool(kwargs.get("geo_model_disable_pts", False))
        geo_in = in_ch - (3 if self.geo_model_disable_pts else 0)
        off = self.geo_model_disable_pts
        self.mlp_scale = mlp("scale", 3, 64, 4, [2], 0 if off else kwargs.get("scale_multires", 4), "none", geo_in)
        self.mlp_opacity = mlp("opacity", 1, 64, 4, [2], 0 if off else kwargs.get("opacity_multires", 3), "sigmoid", geo_in)
        self.mlp_rotation = mlp("rotation", 4, 64, 3, [20], 0 if off else kwargs.get("rotation_multires", 3), "normalize", geo_in)
        if n_frames > 0:
            self.mlp_flow = mlp("flow", kwargs.get("flow_w", 128), 128, 6, [3], kwargs.get("flow_multires", 6), "none")
            self.mlp_flow_head = FlowHead(W=self.mlp_flow.out_features, flow_model=kwargs.get("flow_model", "se3"),
                                          num_basis=kwargs.get("dct_basis", 4), n_frames=n_frames)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """`nn.Module.load_state_dict`, plus one loud check: a reference default-config checkpoint holds the weights of its
        time-conditioned plane generators (`encoder.subs.*`, scene/tripFields.py:383-428).  The decoder-free sampler has no place
        for them, and silently training its own free planes instead would be a different model."""
        gen_keys = [k for k in state_dict if k.startswith("encoder.subs.")]
        own = getattr(self, "encoder", None)
        if gen_keys and own is not None and not any(k.startswith("encoder.subs.") for k in self.state_dict()):
            raise RuntimeError(
                "this checkpoint carries the reference's tri-plane GENERATOR (%d `encoder.subs.*` tensors, e.g. %r), but this SplatFields "
                "was built with the decoder-free TriPlaneSampler (planes are a free parameter, they ignore frame_id).  Build it with the "
                "reference's encoder -- SplatFields(encoder=<VarTriPlaneEncoder instance>) or encoder_args['plane_source'] -- to load it."
                % (len(gen_keys), gen_keys[0]))
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    def _time2frame_id(self, t):
        return torch.round(t * (self.n_frames - 1))

    def extract_features(self, x, t):
        t_feat = positional_encoding(t, self.time_multires) if self.n_frames > 0 else None
        x_feat = self.mlp_refine_feat(self.encoder(x[None]).squeeze(0)) if self.feat_dim > 0 else None
        parts = [f for f in (x_feat, t_feat) if f is not None]
        return torch.cat(parts, dim=-1) if parts else None

    def forward(self, xyz_in, t):
        out = {}
        time_step, frame_id = None, None
        if self.n_frames > 0:
            time_step = t.view(-1)[0]
            frame_id = self._time2frame_id(time_step).long()
        # features of every network: [refined tri-plane features | time embedding] (extract_features).  With one float32 time per
        # point that takes no gradient, the time embedding is written by the networks' input kernel instead of ~15 small kernels
        # and a concatenation per step.
        fuse_time = self.n_frames > 0 and torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and not t.requires_grad and \
            t.numel() == xyz_in.shape[0] and not self.geo_model_disable_pts
        if fuse_time:
            feat = self.mlp_refine_feat(self.encoder(xyz_in[None]).squeeze(0)) if self.feat_dim > 0 else None
            tk = dict(time=t.reshape(-1), time_multires=self.time_multires)
        else:
            feat, tk = self.extract_features(xyz_in, t), {}
        if self.deform_weight > 0:
            xyz_can = torch.add(xyz_in, self.mlp_deform(xyz_in, feat, frame_id=frame_id, **tk), alpha=self.deform_weight)   # one kernel
        else:
            xyz_can = xyz_in
        geo_xyz, geo_feat = (feat, None) if self.geo_model_disable_pts else (xyz_can, feat)
        out["scales"] = self.mlp_scale(geo_xyz, geo_feat, frame_id=frame_id, **tk)
        out["opacity"] = self.mlp_opacity(geo_xyz, geo_feat, frame_id=frame_id, **tk)
        out["rotations"] = self.mlp_rotation(geo_xyz, geo_feat, frame_id=frame_id, **tk)
        rgb = self.mlp_rgb(xyz_can, feat, frame_id=frame_id, **tk)
        if self.use_view_dep_rgb:
            out["rgb_fnc"] = lambda viewdir: self.mlp_rgb_viewdep(torch.cat([rgb, viewdir], dim=-1))
        else:
            out["rgb"] = rgb
        if self.n_frames > 0:
            flow_feat = self.mlp_flow(xyz_can, feat, frame_id=frame_id, **tk)
            flow, means3D = self.mlp_flow_head(hidden=flow_feat, pts=xyz_can, time_step=time_step, frame_id=frame_id)
        else:
            flow, means3D = None, xyz_can
        out.update({"flow": flow, "means3D": means3D})
        return out


def expon_lr(step: int, lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0, max_steps: int = 1000000) -> float:
    """the reference's schedule (utils/general_utils.py:86-119): log-linear from lr_init to lr_final over max_steps, optionally
    eased in by lr_delay_mult + (1 - lr_delay_mult) sin(pi/2 clip(step / lr_delay_steps)); 0 for step < 0 or when both rates are 0."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    delay = 1.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1.0 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay * math.exp(math.log(lr_init) * (1.0 - t) + math.log(lr_final) * t)


class SplatFieldsModel:
    """Counterpart of reference scene/deform_model.py:9-54 around `SplatFields`: `step(xyz, time_emb)`, Adam over all network
    parameters at 5 x the position learning rate with the exponential schedule, `deform/iteration_<n>/deform.pth` checkpoints."""

    def __init__(self, hyper_args, radius=None, encoder: Optional[nn.Module] = None, device="cuda"):
        kwargs = dict(hyper_args.__dict__) if hasattr(hyper_args, "__dict__") else dict(hyper_args)
        self.deform = SplatFields(radius=radius, encoder=encoder, **kwargs).to(device)
        self.optimizer = None
        self.spatial_lr_scale = 5

    def step(self, xyz, time_emb):
        return self.deform(xyz, time_emb)

    def train_setting(self, training_args):
        lr0 = training_args.position_lr_init * self.spatial_lr_scale
        self.optimizer = torch.optim.Adam([{"params": list(self.deform.parameters()), "lr": lr0, "name": "deform"}], lr=0.0, eps=1e-15)
        self._schedule = dict(lr_init=lr0, lr_final=training_args.position_lr_final, lr_delay_mult=training_args.position_lr_delay_mult,
                              max_steps=training_args.deform_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "deform":
                group["lr"] = expon_lr(iteration, **self._schedule)
                return group["lr"]

    def save_weights(self, model_path, iteration):
        import os
        out = os.path.join(model_path, "deform/iteration_{}".format(iteration))
        os.makedirs(out, exist_ok=True)
        torch.save(self.deform.state_dict(), os.path.join(out, "deform.pth"))

    def load_weights(self, model_path, iteration=-1):
        import os
        if iteration == -1:
            iteration = max(int(name.split("_")[-1]) for name in os.listdir(os.path.join(model_path, "deform")))
        self.deform.load_state_dict(torch.load(os.path.join(model_path, "deform/iteration_{}/deform.pth".format(iteration))))
