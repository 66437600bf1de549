#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run ONLY in the build container, where the
reference checkout exists at /root/reference; the fixtures (pure numbers) are what travels.

  ref_pieces.npz   -- outputs of the reference's own in-tree restatements of rasterizer math, computed by
                      importing them: utils/sh_utils.py:57 eval_sh, utils/graphics_utils.py:42,56
                      getWorld2View2 / getProjectionMatrix, scene/cameras.py:68-74 matrix assembly (redone with
                      the imported helpers), utils/general_utils.py:138-171 rotation / covariance builders
                      (device="cuda" literals patched to CPU), extract_geo.py:40-44 colour rule.
  render_contract.json -- the call pattern of gaussian_renderer/__init__.py:30-124 captured with a recording
                      stand-in for diff_gaussian_rasterization (argument names, shapes, dtypes, bg of the
                      second pass, output dict keys / shapes / dtypes).
  tiny_*.npz       -- small scenes: inputs + fp64 torch-oracle outputs and gradients (regression pins for both
                      oracles and for the HIP path).
  general_mlp_*.npz -- the reference's GeneralMLP (utils/time_utils.py:123-191, with the ResField layers of
                      utils/resfields.py) imported and run on CPU in float32: its state dict, inputs, frame id, output
                      and the gradients of sum(output * probe) w.r.t. the inputs and every parameter, for the shapes
                      SplatFields builds (scaled down).  scene.tripFields (diffusers / mmgen, absent here) is not needed by
                      GeneralMLP and is kept out of the import.
  densify_*.npz    -- the reference's own GaussianModel.densify_and_prune (scene/gaussian_model.py:411-425, with the clone /
                      split / prune and optimizer surgery of :272-409) run on CPU on small clouds after one Adam step:
                      the state before (parameters, Adam moments, statistics), the thresholds, the unit normal samples its
                      torch.normal call drew (recorded, with the split mask its prune_points call received), and the state after.
  triplane_*.npz   -- the reference's own VarTriPlaneEncoder.forward (scene/tripFields.py:430-436: grid_sample of the three planes +
                      cat) with stand-in plane generators (its TimeVAEDecoder needs diffusers / mmgen): planes, points (some
                      outside [-1, 1]), features, and the gradients of sum(features * probe) w.r.t. planes and points.
  splatfields_*.npz -- the reference's whole SplatFields network (utils/time_utils.py:305-508: six GeneralMLPs, time embedding,
                      FlowHead) in configurations without plane features, same contents as above for every output of
                      forward(xyz, t).
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def import_reference():
    sys.path.insert(0, REF)
    for name in ["trimesh", "cv2", "plyfile", "imageio", "simple_knn", "simple_knn._C", "lpips", "torchvision",
                 "mmgen", "mmgen.models", "mmgen.models.builder", "mmgen.models.architectures",
                 "mmgen.models.architectures.common", "mmcv", "mmcv.cnn", "mmcv.runner", "diffusers",
                 "diffusers.models", "diffusers.models.resnet", "diffusers.models.attention", "sklearn.neighbors"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                m.__dict__.setdefault("__path__", [])
                sys.modules[name] = m
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: None


def ref_pieces():
    import_reference()
    from utils import sh_utils, graphics_utils
    g = torch.Generator().manual_seed(7)
    out = {}
    # --- SH ---
    n = 64
    dirs = torch.randn(n, 3, generator=g, dtype=torch.float64)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    sh = torch.randn(n, 16, 3, generator=g, dtype=torch.float64)
    out["sh_dirs"], out["sh_coeffs"] = dirs.numpy(), sh.numpy()
    for deg in range(4):
        # reference layout: [..., C, (deg+1)^2] (extract_geo.py:40 transposes features [N,16,3] -> [N,3,16])
        out[f"sh_eval_deg{deg}"] = sh_utils.eval_sh(deg, sh.transpose(1, 2), dirs).numpy()
    out["rgb2sh"] = sh_utils.RGB2SH(torch.linspace(0, 1, 5, dtype=torch.float64)).numpy()
    # --- camera matrices ---
    Rs, Ts, W2V, PRJ, FULL, CEN = [], [], [], [], [], []
    for k in range(4):
        q = torch.randn(4, generator=g, dtype=torch.float64)
        q = q / q.norm()
        r, x, y, z = q.tolist()
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                      [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                      [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
        T = torch.randn(3, generator=g, dtype=torch.float64).numpy()
        fovx, fovy = 0.5 + 0.1 * k, 0.4 + 0.07 * k
        w2v = torch.tensor(graphics_utils.getWorld2View2(R, T)).transpose(0, 1)       # scene/cameras.py:68
        prj = graphics_utils.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)  # :70
        full = (w2v.unsqueeze(0).bmm(prj.unsqueeze(0))).squeeze(0)                    # :72-73
        cen = w2v.inverse()[3, :3]                                                    # :74
        Rs.append(R); Ts.append(T); W2V.append(w2v.numpy()); PRJ.append(prj.numpy()); FULL.append(full.numpy()); CEN.append(cen.numpy())
    out.update(cam_R=np.stack(Rs), cam_T=np.stack(Ts), cam_fov=np.array([[0.5 + 0.1 * k, 0.4 + 0.07 * k] for k in range(4)]),
               cam_world_view=np.stack(W2V), cam_proj=np.stack(PRJ), cam_full=np.stack(FULL), cam_center=np.stack(CEN))
    # --- rotation / covariance builders (device literals patched to CPU) ---
    orig_zeros = torch.zeros
    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return orig_zeros(*a, **k)
    torch.zeros = zeros_cpu
    try:
        from utils import general_utils
        q = torch.randn(32, 4, generator=g)
        s = torch.rand(32, 3, generator=g) + 0.1
        Rm = general_utils.build_rotation(q)  # normalises internally
        L = general_utils.build_scaling_rotation(s, q)
        cov = L @ L.transpose(1, 2)
        out.update(quat=q.numpy(), quat_scales=s.numpy(), quat_rot=Rm.numpy(),
                   cov6=general_utils.strip_symmetric(cov).numpy())
    finally:
        torch.zeros = orig_zeros
    # --- ndc2Pix (scene/dataset_readers.py:515-516) evaluated literally ---
    vv = np.linspace(-1.2, 1.2, 9)
    out["ndc_v"] = vv
    out["ndc_pix_800"] = ((vv + 1.0) * 800 - 1.0) * 0.5
    np.savez_compressed(os.path.join(HERE, "ref_pieces.npz"), **out)
    print("ref_pieces.npz:", sorted(out))


def render_contract():
    import_reference()
    calls = []

    class Settings:  # records keyword construction (gaussian_renderer/__init__.py:59-72)
        def __init__(self, **kw):
            self.kw = kw

    class Rasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, **kw):
            rec = {"settings": {k: (list(v.shape) if torch.is_tensor(v) else v) for k, v in self.rs.kw.items()},
                   "bg": self.rs.kw["bg"].tolist(),
                   "args": {k: (None if v is None else {"shape": list(v.shape), "dtype": str(v.dtype)}) for k, v in kw.items()}}
            calls.append(rec)
            n = kw["means3D"].shape[0]
            H, W = self.rs.kw["image_height"], self.rs.kw["image_width"]
            img = torch.zeros(3, H, W) + kw["means2D"].sum() * 0 + kw["means3D"].sum() * 0
            return img, torch.ones(n, dtype=torch.int32), torch.zeros(1, H, W)

    # gaussian_renderer imports scene.gaussian_model only for a type name (:15); the real package drags in
    # mmgen/diffusers, so give it an empty stand-in for this capture
    scene_stub = types.ModuleType("scene"); scene_stub.__path__ = []
    gm_stub = types.ModuleType("scene.gaussian_model"); gm_stub.GaussianModel = object
    saved_scene = {k: sys.modules.get(k) for k in ("scene", "scene.gaussian_model")}
    sys.modules["scene"], sys.modules["scene.gaussian_model"] = scene_stub, gm_stub
    stub = types.ModuleType("diff_gaussian_rasterization")
    stub.GaussianRasterizationSettings = Settings
    stub.GaussianRasterizer = Rasterizer
    saved = sys.modules.get("diff_gaussian_rasterization")
    sys.modules["diff_gaussian_rasterization"] = stub
    orig_zl = torch.zeros_like
    def zl(t, **k):
        k.pop("device", None)
        return orig_zl(t, **k)
    torch.zeros_like = zl
    try:
        sys.modules.pop("gaussian_renderer", None)
        import gaussian_renderer
        from splatfields_amd.synthetic import make_camera, make_splats
        sp = make_splats(50, seed=3)
        cam = make_camera(1, 64, 48)
        gd = {"means3D": sp["means3D"].requires_grad_(True), "active_sh_degree": 2, "gaussian_opacity": sp["opacities"],
              "gaussian_features": sp["shs"], "gaussian_scales": sp["scales"], "gaussian_rotations": sp["rotations"]}
        pipe = types.SimpleNamespace(debug=False)
        out = gaussian_renderer.render(cam, gd, pipe, torch.tensor([1.0, 1.0, 1.0]))
        out["render"].sum().backward()
        contract = {"calls": calls,
                    "outputs": {k: (None if v is None else {"shape": list(v.shape), "dtype": str(v.dtype)}) for k, v in out.items()},
                    "viewspace_points_is_leaf": bool(out["viewspace_points"].is_leaf),
                    "viewspace_points_grad_populated": out["viewspace_points"].grad is not None}
    finally:
        torch.zeros_like = orig_zl
        sys.modules.pop("gaussian_renderer", None)
        for k, m in saved_scene.items():
            if m is not None:
                sys.modules[k] = m
            else:
                sys.modules.pop(k, None)
        if saved is not None:
            sys.modules["diff_gaussian_rasterization"] = saved
        else:
            sys.modules.pop("diff_gaussian_rasterization", None)
    json.dump(contract, open(os.path.join(HERE, "render_contract.json"), "w"), indent=1, sort_keys=True)
    print("render_contract.json:", len(calls), "rasterizer calls")


def tiny_scenes():
    from oracle import torch_oracle as O
    from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
    specs = {"tiny_sh3": dict(n=300, w=64, h=48, deg=3, use_sh=True, seed=11, bg=(1.0, 1.0, 1.0), mod=1.0, ms=0.06),
             "tiny_rgb": dict(n=200, w=40, h=56, deg=0, use_sh=False, seed=12, bg=(0.0, 0.0, 0.0), mod=1.0, ms=0.08),
             "tiny_sh1_mod": dict(n=150, w=33, h=47, deg=1, use_sh=True, seed=13, bg=(0.2, 0.5, 0.9), mod=0.7, ms=0.1)}
    for name, s in specs.items():
        sp = make_splats(s["n"], seed=s["seed"], mean_scale=s["ms"])
        cam = make_camera(3, s["w"], s["h"])
        st = O.settings_from_camera(cam, torch.tensor(s["bg"]), s["deg"], s["mod"])
        gi, gd, ga = make_upstream_grads(s["h"], s["w"], seed=99)
        out, gr = O.fwd_bwd(sp, st, gi, gd, ga, use_sh=s["use_sh"], dtype=torch.float64)
        arrs = {f"in_{k}": v.numpy() for k, v in sp.items()}
        arrs.update(viewmatrix=st.viewmatrix.contiguous().numpy(), projmatrix=st.projmatrix.contiguous().numpy(),
                    campos=st.campos.contiguous().numpy(), bg=np.array(s["bg"], np.float32),
                    meta=np.array([s["h"], s["w"], s["deg"], int(s["use_sh"])]), tanfov=np.array([st.tanfovx, st.tanfovy]),
                    scale_modifier=np.array(s["mod"]), g_img=gi.numpy(), g_depth=gd.numpy(), g_alpha=ga.numpy(),
                    out_color=out.color.detach().numpy(), out_depth=out.depth.detach().numpy(),
                    out_alpha=out.alpha.detach().numpy(), out_radii=out.radii.numpy(), fragile=out.fragile.numpy(),
                    num_rendered=np.array(out.num_rendered))
        arrs.update({f"grad_{k}": v.numpy() for k, v in gr.items()})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
        print(name, "R =", out.num_rendered, "fragile px =", int(out.fragile.sum()))


GENERAL_MLP_CASES = {
    # name: (constructor kwargs, feature channels, frame id or None)
    "scale": (dict(in_features=3 + 8 + 7, out_features=3, hidden_features=64, num_hidden_layers=4, skips=[2], multires=4,
                   out_activation="none", act="leaky_relu", composition_rank=2, n_frames=10), 15, 3),
    "opacity": (dict(in_features=3 + 8 + 7, out_features=1, hidden_features=64, num_hidden_layers=4, skips=[2], multires=3,
                     out_activation="sigmoid", act="leaky_relu", composition_rank=1, n_frames=6), 15, 5),
    "rotation": (dict(in_features=3 + 8 + 7, out_features=4, hidden_features=64, num_hidden_layers=3, skips=[20], multires=3,
                      out_activation="normalize", act="leaky_relu", composition_rank=1, n_frames=4), 15, 0),
    "deform": (dict(in_features=3 + 16 + 7, out_features=3, hidden_features=128, num_hidden_layers=3, skips=[1], multires=6,
                    out_activation="none", act="leaky_relu", composition_rank=1, n_frames=5), 23, 4),
    "static_rgb": (dict(in_features=3 + 8, out_features=3, hidden_features=64, num_hidden_layers=2, skips=[0], multires=2,
                        out_activation="sigmoid", act="leaky_relu", composition_rank=0, n_frames=0), 8, None),
    "no_features": (dict(in_features=3, out_features=5, hidden_features=64, num_hidden_layers=2, skips=[4], multires=0,
                         out_activation="tanh", act="relu", composition_rank=0, n_frames=0), 0, None),
}


def import_time_utils():
    """imports utils/time_utils.py with scene.tripFields masked out (GeneralMLP / FlowHead / SplatFields without plane
    features do not use it; its own imports -- diffusers, mmgen -- do not exist here)"""
    sys.path.insert(0, REF)
    scene_pkg = types.ModuleType("scene")
    scene_pkg.__path__ = []
    masked = types.ModuleType("scene.tripFields")
    for name in ["TriPlaneEncoder", "VarTriPlaneEncoder", "HexPlaneEncoder", "VarHexPlaneEncoder", "GridEncoder", "VarGridEncoder",
                 "LaplaceDensity", "BellDensity"]:
        setattr(masked, name, object)
    saved = {k: sys.modules.get(k) for k in ("scene", "scene.tripFields")}
    sys.modules["scene"], sys.modules["scene.tripFields"] = scene_pkg, masked
    try:
        from utils import time_utils
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return time_utils


def general_mlp_cases():
    GeneralMLP = import_time_utils().GeneralMLP
    for idx, (name, (kwargs, n_feat, frame)) in enumerate(GENERAL_MLP_CASES.items()):
        torch.manual_seed(100 + idx)
        net = GeneralMLP(**kwargs)
        with torch.no_grad():                      # default init leaves the residuals at 1e-2 scale; make them matter
            for k, p in net.named_parameters():
                if k.endswith("matrix_t") or k.endswith("weights_t"):
                    p.mul_(30.0)
        n = 40
        xyz = (torch.rand(n, 3) * 2 - 1).requires_grad_()
        feat = torch.randn(n, n_feat).requires_grad_() if n_feat else None
        frame_id = None if frame is None else torch.tensor(frame)
        out = net(xyz, feat, frame_id=frame_id)
        probe = torch.randn(out.shape)
        (out * probe).sum().backward()
        data = {"xyz": xyz.detach().numpy(), "out": out.detach().numpy(), "probe": probe.numpy(), "grad_xyz": xyz.grad.numpy(),
                "frame_id": np.array(-1 if frame is None else frame)}
        if feat is not None:
            data["feat"], data["grad_feat"] = feat.detach().numpy(), feat.grad.numpy()
        for k, p in net.named_parameters():
            data["param:" + k] = p.detach().numpy()
            data["grad:" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        np.savez_compressed(os.path.join(HERE, f"general_mlp_{name}.npz"), **data)
        print(name, tuple(out.shape), sum(v.size for v in data.values()), "floats")


_SMALL = dict(deform_w=64, deform_d=2, deform_skips=[1], rgb_w=64, rgb_d=2, rgb_skips=[0], flow_w=64, flow_d=2, flow_skips=[1],
              scale_d=2, scale_skips=[0], opacity_d=2, opacity_skips=[5], rotation_d=2, encoder_type="none")
SPLATFIELDS_CASES = {
    # name: (n_frames, constructor kwargs, time value): networks without plane features (the tri-plane encoder cannot be imported here)
    "dynamic_se3": (5, dict(_SMALL, composition_rank=1, flow_model="se3"), 0.75),
    "dynamic_dct": (6, dict(_SMALL, composition_rank=0, flow_model="dct", dct_basis=3, deform_weight=0.5), 0.4),
    "static_viewdep": (0, dict(_SMALL, composition_rank=0, use_view_dep_rgb=True), 0.0),
}


def splatfields_cases():
    """the reference's SplatFields.forward (utils/time_utils.py:467-508) without plane features: state dict, positions, time,
    every output and the gradients of sum over outputs of (output * probe)."""
    SplatFields = import_time_utils().SplatFields
    for idx, (name, (n_frames, kwargs, tval)) in enumerate(SPLATFIELDS_CASES.items()):
        torch.manual_seed(200 + idx)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):            # the constructor prints the module
            net = SplatFields(radius=None, n_frames=n_frames, **kwargs)
        with torch.no_grad():
            for k, p in net.named_parameters():
                if k.endswith("matrix_t") or k.endswith("weights_t"):
                    p.mul_(30.0)
                if "branch_coeff" in k:                               # zero-initialised in the reference; make the dct path matter
                    p.normal_(0.0, 0.1)
        n = 32
        xyz = (torch.rand(n, 3) * 2 - 1).requires_grad_()
        t = torch.full((n, 1), tval)
        out = net(xyz, t)
        if "rgb_fnc" in out:
            viewdir = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
            out = dict(out, rgb=out["rgb_fnc"](viewdir))
            del out["rgb_fnc"]
        else:
            viewdir = None
        data = {"xyz": xyz.detach().numpy(), "t": t.numpy()}
        if viewdir is not None:
            data["viewdir"] = viewdir.numpy()
        loss = 0.0
        for k in sorted(out):
            if out[k] is None:
                continue
            probe = torch.randn(out[k].shape)
            loss = loss + (out[k] * probe).sum()
            data["out:" + k], data["probe:" + k] = out[k].detach().numpy(), probe.numpy()
        loss.backward()
        data["grad_xyz"] = xyz.grad.numpy()
        for k, p in net.named_parameters():
            data["param:" + k] = p.detach().numpy()
            data["grad:" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        for k, b in net.named_buffers():
            data["param:" + k] = b.detach().numpy()
        np.savez_compressed(os.path.join(HERE, f"splatfields_{name}.npz"), **data)
        print(name, sorted(k for k in data if k.startswith("out:")), sum(v.size for v in data.values()), "floats")


def densify_cases():
    """Runs the reference's GaussianModel.densify_and_prune itself.  Only device literals are patched (torch.zeros(...,
    device="cuda") -> CPU); torch.normal and prune_points are WRAPPED to record what the reference drew / decided, nothing of
    its logic is restated here."""
    import_reference()
    orig_zeros, orig_normal, orig_empty_cache = torch.zeros, torch.normal, torch.cuda.empty_cache

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return orig_zeros(*a, **k)

    torch.zeros = zeros_cpu
    torch.cuda.empty_cache = lambda: None
    try:
        # scene/__init__.py pulls in the whole training stack (deform model, decoders); only the module itself is wanted
        saved_scene = sys.modules.get("scene")
        pkg = types.ModuleType("scene")
        pkg.__path__ = [os.path.join(REF, "scene")]
        sys.modules["scene"] = pkg
        try:
            from scene.gaussian_model import GaussianModel
        finally:
            if saved_scene is None:
                sys.modules.pop("scene", None)
            else:
                sys.modules["scene"] = saved_scene
        from torch import nn
        cases = [("aniso_screen", 900, False, 20.0, 11), ("aniso_noscreen", 900, False, None, 12), ("isotropic_screen", 700, True, 20.0, 13),
                 ("tiny", 5, False, None, 14)]
        for name, n, isotropic, screen, seed in cases:
            g = torch.Generator().manual_seed(seed)
            r = lambda *sh: torch.randn(*sh, generator=g)
            m = GaussianModel(1)                       # sh_degree 1: f_rest [N, 3, 3] keeps the fixture small
            m.use_isotropic = isotropic
            m._xyz = nn.Parameter(r(n, 3))
            m._features_dc = nn.Parameter(r(n, 1, 3))
            m._features_rest = nn.Parameter(r(n, 3, 3) * 0.1)
            m._opacity = nn.Parameter(r(n, 1) * 2.5)
            m._scaling = nn.Parameter(math.log(0.03) + 1.2 * r(n, 1 if isotropic else 3))
            m._rotation = nn.Parameter(r(n, 4))
            args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                         position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3)
            m.training_setup(args)                     # :123-143 (Adam with one named group per tensor)
            for grp in m.optimizer.param_groups:       # one Adam step so that the moments exist and are non-trivial
                grp["params"][0].grad = r(*grp["params"][0].shape)
            m.optimizer.step()
            m.xyz_gradient_accum = torch.rand(n, 1, generator=g) * 0.01
            m.denom = torch.randint(0, 6, (n, 1), generator=g).float()
            m.xyz_gradient_accum[m.denom == 0] = 0.0   # 0 / 0 -> NaN -> 0 inside the reference
            m.max_radii2D = torch.rand(n, generator=g) * 40
            names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
                     "rotation": "_rotation"}
            data = {"max_grad": np.float32(0.0035), "min_opacity": np.float32(0.1), "extent": np.float32(4.0),
                    "max_screen_size": np.float32(screen if screen else 0.0), "percent_dense": np.float32(m.percent_dense),
                    "isotropic": np.int32(isotropic)}
            for grp in m.optimizer.param_groups:
                p_ = grp["params"][0]
                st = m.optimizer.state[p_]
                data["in:" + grp["name"]] = p_.detach().numpy().copy()
                data["in_m:" + grp["name"]] = st["exp_avg"].numpy().copy()
                data["in_v:" + grp["name"]] = st["exp_avg_sq"].numpy().copy()
            data["in:accum"], data["in:denom"], data["in:radii"] = m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy(), m.max_radii2D.numpy().copy()
            drawn, masks = [], []

            def normal_rec(mean=None, std=None, **k):
                out = orig_normal(mean=mean, std=std, generator=g)
                drawn.append((out.detach().clone(), std.detach().clone()))
                return out

            orig_prune = GaussianModel.prune_points

            def prune_rec(self, mask):
                masks.append(mask.clone())
                return orig_prune(self, mask)

            torch.normal, GaussianModel.prune_points = normal_rec, prune_rec
            try:
                m.densify_and_prune(0.0035, 0.1, 4.0, screen)     # the reference itself
            finally:
                torch.normal, GaussianModel.prune_points = orig_normal, orig_prune
            assert len(drawn) == 1 and len(masks) == 2
            samples, stds = drawn[0]
            split_mask = masks[0][:n]                   # prune filter of densify_and_split = cat(selected mask, zeros): the selected splats
            assert not masks[0][:n].numel() == 0 and int(masks[0][n:].sum()) == 0 or n == 0
            idx = torch.nonzero(split_mask).reshape(-1)
            unit = torch.zeros(2, n, 3)
            S = idx.numel()
            assert samples.shape[0] == 2 * S
            unit[0][idx] = samples[:S] / stds[:S]       # the unit normals behind torch.normal(0, std): first and second child
            unit[1][idx] = samples[S:] / stds[S:]
            data["unit_normals"] = unit.numpy()
            data["split_mask"] = split_mask.numpy()
            for grp in m.optimizer.param_groups:
                p_ = grp["params"][0]
                st = m.optimizer.state[p_]
                assert p_ is getattr(m, names[grp["name"]])
                data["out:" + grp["name"]] = p_.detach().numpy().copy()
                data["out_m:" + grp["name"]] = st["exp_avg"].numpy().copy()
                data["out_v:" + grp["name"]] = st["exp_avg_sq"].numpy().copy()
                data["step:" + grp["name"]] = np.float32(float(st["step"]))
            data["out:accum"], data["out:denom"], data["out:radii"] = (m.xyz_gradient_accum.detach().numpy(), m.denom.detach().numpy(),
                                                                       m.max_radii2D.detach().numpy())
            np.savez_compressed(os.path.join(HERE, f"densify_{name}.npz"), **data)
            print(f"densify_{name}: {n} -> {data['out:xyz'].shape[0]} rows, {S} split, {sum(v.size for v in data.values())} numbers")
    finally:
        torch.zeros, torch.normal, torch.cuda.empty_cache = orig_zeros, orig_normal, orig_empty_cache


def triplane_cases():
    """The per-point half of the reference's tri-plane encoder, run through the reference class itself.  scene/time_decoders.py
    (diffusers / mmgen) is masked and mmgen's build_module returns a stand-in generator that hands out a fixed learnable plane;
    VarTriPlaneEncoder.get_planes / forward (the code under test) are the reference's."""
    from torch import nn
    sys.path.insert(0, REF)

    class StandInGenerator(nn.Module):
        shape = (16, 24, 24)

        def __init__(self):
            super().__init__()
            self.plane = nn.Parameter(torch.randn(1, *StandInGenerator.shape, generator=StandInGenerator.gen))

        def forward(self, noise, frame_id=None):
            return self.plane

    saved = {k: sys.modules.get(k) for k in ("scene", "scene.time_decoders", "scene.tripFields", "mmgen", "mmgen.models")}
    pkg = types.ModuleType("scene"); pkg.__path__ = [os.path.join(REF, "scene")]
    td = types.ModuleType("scene.time_decoders"); td.TimeVAEDecoder = object
    mm = types.ModuleType("mmgen"); mm.__path__ = []
    mmm = types.ModuleType("mmgen.models"); mmm.build_module = lambda cfg, *a, **k: StandInGenerator()
    sys.modules.update({"scene": pkg, "scene.time_decoders": td, "mmgen": mm, "mmgen.models": mmm})
    sys.modules.pop("scene.tripFields", None)
    try:
        from scene.tripFields import VarTriPlaneEncoder
        for name, shape, n, seed in (("c16_24x24", (16, 24, 24), 600, 21), ("c8_20x28", (8, 20, 28), 300, 22)):
            g = torch.Generator().manual_seed(seed)
            StandInGenerator.shape, StandInGenerator.gen = shape, g
            enc = VarTriPlaneEncoder({"out_ch": shape[0], "noise_res": 4, "layer_kwargs": {}})
            pts = (torch.rand(1, n, 3, generator=g) * 2.3 - 1.15).requires_grad_(True)   # ~13 % of the coordinates outside [-1, 1]
            with torch.no_grad():
                pts[0, 0] = torch.tensor([-1.0, 1.0, 0.0]); pts[0, 1] = torch.tensor([1.0, -1.0, 1.0])   # plane borders exactly
            feat = enc(pts)                                                                # the reference's forward
            probe = torch.randn(feat.shape, generator=g)
            (feat * probe).sum().backward()
            planes = torch.cat([sub.net.plane for sub in enc.subs], dim=0)
            grad_planes = torch.cat([sub.net.plane.grad for sub in enc.subs], dim=0)
            assert enc.out_dim == 3 * shape[0] and tuple(feat.shape) == (1, n, 3 * shape[0])
            np.savez_compressed(os.path.join(HERE, f"triplane_{name}.npz"), planes=planes.detach().numpy(), pts=pts.detach().numpy(),
                                out=feat.detach().numpy(), probe=probe.numpy(), grad_planes=grad_planes.numpy(), grad_pts=pts.grad.numpy(),
                                axis=np.array(enc.axis))
            print(f"triplane_{name}: planes {tuple(planes.shape)}, {n} points -> {tuple(feat.shape)}")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


if __name__ == "__main__":
    only = sys.argv[1:]          # e.g. `make_golden.py general_mlp_cases` regenerates one family
    for fn in (ref_pieces, render_contract, tiny_scenes, general_mlp_cases, splatfields_cases, densify_cases, triplane_cases):
        if not only or fn.__name__ in only:
            fn()
