#!/usr/bin/env python3
"""SURVEY.md section 8f row 4, "first measure": how much of a 4-D (Owlii-style) training step is the SplatFields deform
network, and how much the rasterizer?

The network below is a SHAPE-FAITHFUL STAND-IN written for this measurement (the reference's code cannot travel to the GPU
box): the structure of reference utils/time_utils.py:305-508 (`SplatFields`: tri-plane encoder -> refine MLP; time
embedding; six `GeneralMLP`s: deform 128x6, rgb 128x6, flow 128x6 + head, scale 64x4, opacity 64x4, rotation 64x3, skip
connections, NeRF positional encodings), of scene/tripFields.py:383-436 (`VarTriPlaneEncoder`: three 16-channel planes
produced every step by a small CNN decoder from an 8-channel 20x20 noise tensor, sampled with `grid_sample`) and of
utils/resfields.py:378-405 (ResField linear layers: weight = base + per-frame low-rank delta).  Random weights, synthetic
points -- only the cost matters.  Runs on PyTorch-ROCm (rocBLAS / hipBLASLt / MIOpen), fp32 like the reference.

Prints one JSON object: ms per forward+backward of the network for N splats, of the rasterizer (precomputed-colour path,
reference train.py:80-81) for the same N, the network's share of the step, and the same step with the six GeneralMLPs
running through splatfields_amd.fused_mlp (forward and backward)."""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def posenc(x, multires):
    if multires <= 0:
        return x
    freqs = 2.0 ** torch.arange(multires, device=x.device, dtype=x.dtype)
    xb = x[..., None, :] * freqs[:, None]
    return torch.cat([x, torch.sin(xb).flatten(-2), torch.cos(xb).flatten(-2)], dim=-1)


class ResLinear(nn.Module):
    """weight(frame) = W + (coef[frame] @ basis).view(out, in)   (reference utils/resfields.py, 'vm'-style low-rank residual)"""

    def __init__(self, fin, fout, rank, n_frames):
        super().__init__()
        self.lin = nn.Linear(fin, fout)
        self.rank = rank if n_frames > 0 else 0
        if self.rank:
            self.coef = nn.Parameter(0.01 * torch.randn(n_frames, rank))
            self.basis = nn.Parameter(0.01 * torch.randn(rank, fout * fin))

    def forward(self, x, frame_id):
        w = self.lin.weight
        if self.rank:
            w = w + (self.coef[frame_id] @ self.basis).view_as(w)
        return F.linear(x, w, self.lin.bias)


class GeneralMLP(nn.Module):
    def __init__(self, in_features, out_features, hidden, depth, skips, multires, rank, n_frames):
        super().__init__()
        self.multires, self.skips = multires, set(skips)
        d_in = in_features + 3 * 2 * multires           # the xyz part is positionally encoded
        self.d_in = d_in
        dims = [d_in] + [hidden] * depth
        self.layers = nn.ModuleList()
        for i in range(depth):
            fin = dims[i] + (d_in if i in self.skips else 0)
            self.layers.append(ResLinear(fin, hidden, rank, n_frames))
        self.out = ResLinear(hidden, out_features, rank, n_frames)

    def effective_weights(self, frame_id):
        """per-step composed weights W + delta(frame) as autograd expressions (the ResField composition stays in PyTorch)"""
        ws, bs = [], []
        for layer in list(self.layers) + [self.out]:
            w = layer.lin.weight
            if layer.rank:
                w = w + (layer.coef[frame_id] @ layer.basis).view_as(w)
            ws.append(w); bs.append(layer.lin.bias)
        return ws, bs

    def forward_fused(self, xyz, feat, frame_id):
        """the same network through splatfields_amd.fused_mlp (all layers in one kernel forward, one kernel for the activation
        gradients + one for all weight gradients backward).  Note the layer order of this stand-in: the skip input is
        concatenated BEHIND the hidden state here, in front of it in the reference -- the fused kernel implements the
        reference's order, so the stand-in's skip weights are reordered on the way in."""
        from splatfields_amd.fused_mlp import fused_general_mlp
        ws, bs = self.effective_weights(frame_id)
        hidden = ws[0].shape[0]
        skips = []
        for i in sorted(self.skips):
            if 0 < i < len(ws):   # layer i consumes cat([h, h0]): move the h0 block in front
                ws[i] = torch.cat([ws[i][:, hidden:], ws[i][:, :hidden]], dim=1)
                skips.append(i - 1)
        h0 = torch.cat([posenc(xyz, self.multires), feat], dim=-1)
        return fused_general_mlp(h0, ws, bs, skips=skips, negative_slope=0.01)

    def forward(self, xyz, feat, frame_id):
        h0 = torch.cat([posenc(xyz, self.multires), feat], dim=-1)
        h = h0
        for i, layer in enumerate(self.layers):
            if i in self.skips:
                h = torch.cat([h, h0], dim=-1)
            h = F.leaky_relu(layer(h, frame_id))
        return self.out(h, frame_id)


class PlaneDecoder(nn.Module):
    """8 x 20 x 20 noise -> 16 x 160 x 160 plane (three nearest-upsampling conv blocks), cf. Tensorial2D / Decoder"""

    def __init__(self, in_ch=8, out_ch=16):
        super().__init__()
        chans = [in_ch, 128, 128, 64, 32]
        self.convs = nn.ModuleList([nn.Conv2d(chans[i], chans[i + 1], 3, padding=1) for i in range(4)])
        self.last = nn.Conv2d(32, out_ch, 3, padding=1)
        self.register_buffer("noise", torch.randn(1, in_ch, 20, 20))

    def forward(self):
        h = self.noise
        for i, c in enumerate(self.convs):
            h = F.leaky_relu(c(h))
            if i >= 1:
                h = F.interpolate(h, scale_factor=2, mode="nearest")
        return self.last(h)


class SplatFieldsStandIn(nn.Module):
    def __init__(self, n_frames=50, rank=10, time_multires=3):
        super().__init__()
        self.n_frames = n_frames
        self.planes = nn.ModuleList([PlaneDecoder() for _ in range(3)])
        fd = 48
        self.refine = nn.Sequential(nn.Linear(fd, fd), nn.ReLU(), nn.Linear(fd, fd))
        self.time_multires = time_multires
        tch = 1 + 2 * time_multires
        fin = 3 + fd + tch
        mk = lambda out, w, d, skips, mr: GeneralMLP(fin, out, w, d, skips, mr, rank, n_frames)
        self.mlp_deform, self.mlp_rgb, self.mlp_flow = mk(3, 128, 6, [3], 6), mk(3, 128, 6, [3], 6), mk(128, 128, 6, [3], 6)
        self.mlp_scale, self.mlp_opacity, self.mlp_rotation = mk(3, 64, 4, [2], 4), mk(1, 64, 4, [2], 3), mk(4, 64, 3, [20], 3)
        self.flow_head = nn.Linear(128, 6)   # se3 flow head: axis-angle + translation

    def forward(self, xyz, t, fused=False):
        frame_id = int(round(float(t) * (self.n_frames - 1)))
        run = (lambda m, x, h, f: m.forward_fused(x, h, f)) if fused else (lambda m, x, h, f: F.leaky_relu(m(x, h, f)))
        planes = torch.cat([p() for p in self.planes], dim=0)                                  # [3, 16, 160, 160]
        coord = torch.stack([xyz[None, :, [0, 1]], xyz[None, :, [1, 2]], xyz[None, :, [2, 0]]])  # [3, 1, N, 2]
        feat = F.grid_sample(planes, coord, align_corners=False)                                # [3, 16, 1, N]
        feat = self.refine(feat.permute(2, 3, 0, 1).reshape(xyz.shape[0], -1))
        tt = posenc(torch.full((xyz.shape[0], 1), float(t), device=xyz.device), self.time_multires)
        h = torch.cat([feat, tt], dim=-1)
        # as in the reference, the activation follows every layer of a GeneralMLP, the last one included (time_utils.py:185-186)
        xyz_can = xyz + run(self.mlp_deform, xyz, h, frame_id)
        flow = self.flow_head(run(self.mlp_flow, xyz_can, h, frame_id))
        return {"means3D": xyz_can + flow[:, 3:], "scales": run(self.mlp_scale, xyz_can, h, frame_id),
                "opacity": torch.sigmoid(run(self.mlp_opacity, xyz_can, h, frame_id)),
                "rotations": F.normalize(run(self.mlp_rotation, xyz_can, h, frame_id), dim=-1),
                "rgb": torch.sigmoid(run(self.mlp_rgb, xyz_can, h, frame_id))}


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=100_000)   # run_owlii.sh:7 --num_pts 100000
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--miopen-find", action="store_true",
                    help="torch.backends.cudnn.benchmark = True: MIOpen searches a solver per convolution shape instead of its "
                         "immediate-mode fallback (the naive direct kernels of the tri-plane decoder)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = bool(a.miopen_find)
    from splatfields_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from splatfields_amd.synthetic import make_camera, make_upstream_grads
    n, H, W = a.splats, a.size, a.size
    net = SplatFieldsStandIn().to(dev)
    n_params = sum(p.numel() for p in net.parameters())
    xyz = (torch.rand(n, 3, device=dev) * 2 - 1).requires_grad_(True)
    base_scale = torch.full((n, 3), 0.35 * n ** (-1 / 3), device=dev)
    cam = make_camera(1, W, H, device=dev)
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3, device=dev), 1.0,
                                       cam.world_view_transform, cam.full_proj_transform, 0, cam.camera_center, False, False)

    probes = {}
    net_params = list(net.parameters())

    def clear_grads(params):
        # what optimizer.zero_grad(set_to_none=True) does over its flat parameter list; nn.Module.zero_grad walks the module tree in
        # Python (0.6 ms per call for these ~100 tensors), which a step of a few ms should not be charged with
        for p_ in params:
            p_.grad = None

    def net_step(fused=False, generic_loss=False):
        clear_grads(net_params)
        xyz.grad = None
        out = net(xyz, 0.37, fused=fused)
        if generic_loss:   # fixed random cotangents: sum(v^2) is constant for the normalised rotations, whose gradients would be rounding noise
            for k, v in out.items():
                if k not in probes:
                    probes[k] = torch.randn(v.shape, device=v.device, generator=torch.Generator(device=v.device).manual_seed(len(probes)))
            loss = sum((v * probes[k]).mean() for k, v in out.items())
        else:
            loss = sum((v * v).mean() for v in out.values())
        loss.backward()

    def raster_step(out=None):
        if out is None:
            with torch.no_grad():
                out = {k: v.detach() for k, v in net(xyz, 0.37).items()}
            out = {k: v.requires_grad_(True) for k, v in out.items()}
        color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
            means3D=out["means3D"], means2D=torch.zeros_like(out["means3D"], requires_grad=True), opacities=out["opacity"],
            colors_precomp=out["rgb"], scales=base_scale + 0.01 * out["scales"], rotations=out["rotations"])
        torch.autograd.backward((color, depth, alpha), (gi, gd, ga))

    with torch.no_grad():
        fixed = {k: v.detach() for k, v in net(xyz, 0.37).items()}

    def raster_only():
        raster_step({k: v.clone().requires_grad_(True) for k, v in fixed.items()})

    def full_step(fused=False):
        clear_grads(net_params)
        xyz.grad = None
        raster_step(net(xyz, 0.37, fused=fused))

    res = {"splats": n, "image": [H, W], "net_parameters": n_params, "dtype": "fp32", "miopen_find": bool(a.miopen_find),
           "net_fwd_bwd_ms": timed(net_step, a.steps, 5), "rasterizer_fwd_bwd_ms": timed(raster_only, a.steps, 5),
           "full_step_ms": timed(full_step, a.steps, 5)}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        res["net_fwd_bwd_ms_bf16_autocast"] = timed(net_step, a.steps, 5)
    # training with the six GeneralMLPs through the fused kernels (fp32, same arithmetic): gradients of every parameter and of
    # the positions against the PyTorch path, then the same timings
    net_step(False, generic_loss=True)
    ref_grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    ref_grads["xyz"] = xyz.grad.clone()
    net_step(True, generic_loss=True)
    got = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    got["xyz"] = xyz.grad
    assert set(got) == set(ref_grads), sorted(set(got) ^ set(ref_grads))
    diffs = {k: ((got[k] - g).abs().max() / g.abs().max().clamp_min(1e-20)).item() for k, g in ref_grads.items()}
    worst = max(diffs, key=diffs.get)
    res["fused_training_max_rel_grad_diff"] = diffs[worst]
    res["fused_training_worst_tensor"] = worst
    # per-point gradients (xyz) see leaky-ReLU units whose pre-activation sits within rounding of 0 take the other branch;
    # parameter gradients sum over all points
    res["fused_training_max_rel_grad_diff_parameters"] = max(v for k, v in diffs.items() if k != "xyz")
    res["fused_training_worst_parameters"] = {k: diffs[k] for k in sorted((k for k in diffs if k != "xyz"), key=diffs.get, reverse=True)[:5]}
    net_step(False, generic_loss=True)      # PyTorch against itself: the run-to-run noise of its own atomics (grid_sample backward, index_put)
    again = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    res["pytorch_vs_pytorch_max_rel_grad_diff_parameters"] = max(((again[k] - g).abs().max() / g.abs().max().clamp_min(1e-20)).item()
                                                                 for k, g in ref_grads.items() if k != "xyz")
    dx = (got["xyz"] - ref_grads["xyz"]).abs() / ref_grads["xyz"].abs().max()
    res["fused_training_xyz_grad_frac_above_1e-4"] = (dx > 1e-4).float().mean().item()
    res["fused_training_xyz_grad_median_rel_diff"] = dx.median().item()
    res["net_fwd_bwd_fused_mlps_ms"] = timed(lambda: net_step(True), a.steps, 5)
    res["full_step_fused_mlps_ms"] = timed(lambda: full_step(True), a.steps, 5)
    # inference (rendering a trained 4-D model, reference render.py): forward only, PyTorch layers vs the fused MLP kernel
    with torch.no_grad():
        ref_out = net(xyz, 0.37)
        fused_out = net(xyz, 0.37, fused=True)
        res["fused_forward_max_rel_diff"] = max(((ref_out[k] - fused_out[k]).abs().max() / ref_out[k].abs().max().clamp_min(1e-12)).item()
                                                for k in ref_out)
        res["net_forward_ms"] = timed(lambda: net(xyz, 0.37), a.steps, 5)
        res["net_forward_fused_mlps_ms"] = timed(lambda: net(xyz, 0.37, fused=True), a.steps, 5)
    # the PRODUCT network (splatfields_amd.deform_field.SplatFields: tri-plane lookup, ResField composition and the six MLPs on HIP
    # kernels) with the same stand-in plane decoder as generator, and decoder-free (the sampler owns 16 x 320 x 320 planes)
    from splatfields_amd.deform_field import SplatFields
    from splatfields_amd.triplane import TriPlaneSampler

    class PlaneStack(nn.Module):
        def __init__(self):
            super().__init__()
            self.subs = nn.ModuleList([PlaneDecoder() for _ in range(3)])

        def get_planes(self, frame_id=None):
            return torch.cat([p() for p in self.subs], dim=0)

    t_emb = torch.full((n, 1), 0.37, device=dev)
    for key, enc in (("product", TriPlaneSampler(out_ch=16, plane_source=PlaneStack())), ("product_decoderfree", None)):
        pnet = SplatFields(n_frames=50, composition_rank=10, encoder=enc, flow_model="offset").to(dev)   # run_owlii.sh:7 --flow_model offset
        pnet_params = list(pnet.parameters())

        def p_net_step():
            clear_grads(pnet_params)
            xyz.grad = None
            out = pnet(xyz, t_emb)
            sum((v * v).mean() for k, v in out.items() if torch.is_tensor(v) and k != "flow").backward()

        def p_full_step():
            clear_grads(pnet_params)
            xyz.grad = None
            raster_step(pnet(xyz, t_emb))

        res[key + "_net_fwd_bwd_ms"] = timed(p_net_step, a.steps, 5)
        res[key + "_full_step_ms"] = timed(p_full_step, a.steps, 5)
        # kernel launches of one full step (network forward + rasterizer + one backward), counted by the profiler
        try:
            from torch.profiler import ProfilerActivity, profile
            torch.cuda.synchronize()
            # (with the device activity alone the trace loses records from run to run -- 307 / 67 / 26 for the same step; with
            # the host activity enabled as well the count is stable and agrees with rocprofv3's kernel trace)
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                for _ in range(3):
                    p_full_step()
                torch.cuda.synchronize()
            dev_events = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
            res[key + "_launches_per_step"] = len(dev_events) / 3.0   # kernels + memsets / copies on the device
            res[key + "_library_launches_per_step"] = sum(1 for e in dev_events if e.name.startswith(("sr::", "void sr::"))) / 3.0
        except Exception as e:  # noqa: BLE001 -- a profiler that is not available must not cost the timings
            res[key + "_launches_per_step"] = None
            res[key + "_launches_error"] = repr(e)[:200]
        with torch.no_grad():
            res[key + "_net_forward_ms"] = timed(lambda: pnet(xyz, t_emb), a.steps, 5)
        res[key + "_net_parameters"] = sum(p.numel() for p in pnet.parameters())
        # the network's forward and backward as two HIP graphs (torch.cuda.make_graphed_callables: static shapes between two
        # densification steps, no host synchronisation inside): the ~850 launches of a step are replayed without host work
        try:
            keys = ["means3D", "scales", "rotations", "opacity", "rgb"]
            graphed = torch.cuda.make_graphed_callables(lambda x, tt: tuple(pnet(x, tt)[k] for k in keys), (xyz, t_emb))

            def g_full_step():
                clear_grads(pnet_params)
                xyz.grad = None
                raster_step(dict(zip(keys, graphed(xyz, t_emb))))

            def g_net_step():
                clear_grads(pnet_params)
                xyz.grad = None
                sum((v * v).mean() for v in graphed(xyz, t_emb)).backward()

            res[key + "_graphed_net_fwd_bwd_ms"] = timed(g_net_step, a.steps, 5)
            res[key + "_graphed_full_step_ms"] = timed(g_full_step, a.steps, 5)
        except Exception as e:  # noqa: BLE001
            res[key + "_graphed_error"] = repr(e)[:300]
        del pnet
    res["net_share_of_step"] = res["net_fwd_bwd_ms"] / (res["net_fwd_bwd_ms"] + res["rasterizer_fwd_bwd_ms"])
    macs = sum(l.lin.weight.numel() for m in net.modules() if isinstance(m, GeneralMLP) for l in list(m.layers) + [m.out])
    res["mlp_macs_per_splat"] = macs + 2 * 48 * 48
    res["mlp_tflops_fwd_bwd_achieved"] = 3 * 2 * res["mlp_macs_per_splat"] * n / (res["net_fwd_bwd_ms"] * 1e-3) / 1e12
    # the product step against the fp32 matrix roofline (forward + activation-gradient chain + weight gradients = 3 x the MACs;
    # 157.3 TFLOP/s: on gfx950 the fp32 MFMA rate IS the fp32 vector rate, tools/mfma_valu_coissue.hip)
    for key in ("product", "product_decoderfree"):
        if res.get(key + "_full_step_ms"):
            res[key + "_mfma_frac"] = 3 * 2 * res["mlp_macs_per_splat"] * n / (res[key + "_full_step_ms"] * 1e-3) / 157.3e12
    print(json.dumps(res))


if __name__ == "__main__":
    main()
