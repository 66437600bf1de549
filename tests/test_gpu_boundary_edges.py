"""Edges of the drop-in boundary that SURVEY.md section 3.6 recorded at the reference's call sites and that no other `-m gpu` test
feeds through the HIP path:

* `[1,4,4]` view / projection matrices with an OFF-CENTRE principal point, as `CameraPenoptic` builds them
  (reference scene/cameras.py:127-148: `w2c.unsqueeze(0).transpose(1, 2)` and an OpenGL-style matrix from fx, fy, cx, cy) --
  the screen-space mean goes through `projmatrix`, the covariance through `tanfov` (the published rule), in the oracle and here;
* `means2D.grad` after the reference's TWO rasterizer passes over the same `screenspace_points` (colour pass + white-on-black mask
  pass, gaussian_renderer/__init__.py:49-53, 94-115) = the sum of both passes' gradients;
* `prefiltered=True` (the caller asserts it has culled already): same results as `False` for in-frustum and culled splats.
"""
import math
import types

import pytest
import torch

from oracle import torch_oracle as O
from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads
from tests.helpers import grad_error, image_errors, make_scene, radii_mismatch, run_hip

pytestmark = pytest.mark.gpu

IMG_TOL, FRAGILE_TOL, GRAD_TOL64 = 1e-4, 2e-2, 5e-3


def penoptic_camera(width, height, fx, fy, cx, cy, view=3, near=0.01, far=100.0):
    """The three camera tensors as reference scene/cameras.py:127-148 assembles them, from a synthetic world-to-camera matrix."""
    base = make_camera(view, width, height)
    w2c = base.world_view_transform.t().contiguous()                 # make_camera stores W2C^T (row-vector convention)
    cam_center = torch.inverse(w2c)[:3, 3]
    view_t = w2c.unsqueeze(0).transpose(1, 2)                        # [1,4,4], strides of a transposed view
    opengl_proj = torch.tensor([[2 * fx / width, 0.0, -(width - 2 * cx) / width, 0.0],
                                [0.0, 2 * fy / height, -(height - 2 * cy) / height, 0.0],
                                [0.0, 0.0, far / (far - near), -(far * near) / (far - near)],
                                [0.0, 0.0, 1.0, 0.0]]).unsqueeze(0).transpose(1, 2)
    full_proj = view_t.bmm(opengl_proj)                              # [1,4,4]
    fovx, fovy = 2 * math.atan(width / (2 * fx)), 2 * math.atan(height / (2 * fy))   # utils/graphics_utils.py:83-84
    return view_t, full_proj, cam_center, fovx, fovy


def test_batched_matrices_and_off_centre_principal_point(hip_device):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = hip_device
    W, H = 176, 120
    view_t, full_proj, campos, fovx, fovy = penoptic_camera(W, H, fx=210.0, fy=190.0, cx=0.5 * W + 21.5, cy=0.5 * H - 13.25)
    assert view_t.shape == (1, 4, 4) and not view_t.is_contiguous()
    sp = make_splats(5000, seed=77, mean_scale=0.03)
    grads = make_upstream_grads(H, W)
    bg = torch.tensor([0.3, 0.6, 0.1])
    st = O.OracleSettings(H, W, math.tan(fovx * 0.5), math.tan(fovy * 0.5), bg, 1.0, view_t[0], full_proj[0], 3, campos, False, False)
    ref, gr = O.fwd_bwd(sp, st, *grads, use_sh=True, dtype=torch.float64)
    leaf = {k: v.to(dev).clone().requires_grad_(True) for k, v in sp.items()}
    m2d = torch.zeros_like(leaf["means3D"], requires_grad=True)
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=st.tanfovx, tanfovy=st.tanfovy, bg=bg.to(dev),
                                       scale_modifier=1.0, viewmatrix=view_t.to(dev), projmatrix=full_proj.to(dev), sh_degree=3,
                                       campos=campos.to(dev), prefiltered=False, debug=False)
    assert rs.viewmatrix.shape == (1, 4, 4) and rs.projmatrix.shape == (1, 4, 4)
    color, radii, depth, alpha = GaussianRasterizer(rs).forward_ex(
        means3D=leaf["means3D"], means2D=m2d, opacities=leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
        rotations=leaf["rotations"])
    gi, gd, ga = [g.to(dev) for g in grads]
    ((color * gi).sum() + (depth * gd).sum() + (alpha * ga).sum()).backward()
    out = dict(color=color.detach().cpu(), depth=depth.detach().cpu(), alpha=alpha.detach().cpu())
    for k, r in (("color", ref.color), ("depth", ref.depth), ("alpha", ref.alpha)):
        robust, frag = image_errors(out[k], r.detach(), ref.fragile)
        assert robust <= IMG_TOL and frag <= FRAGILE_TOL, (k, robust, frag)
    assert radii_mismatch(radii.cpu(), ref.pre) == 0 and int((radii > 0).sum()) > 1000
    got = {k: leaf[k].grad.cpu() for k in ("means3D", "scales", "rotations", "opacities", "shs")}
    got["means2D"] = m2d.grad.cpu()
    for k in got:
        assert grad_error(got[k], gr[k]) <= GRAD_TOL64, (k, grad_error(got[k], gr[k]))
    # the principal point really is off-centre: a splat on the optical axis lands (cx, cy), not the image centre
    on_axis = campos + 2.0 * view_t[0][:3, 2]                          # camera centre + 2 x viewing direction (column 2 of W2C^T)
    h = torch.cat([on_axis, torch.ones(1)]) @ full_proj[0]
    px = ((h[0] / h[3] + 1.0) * W - 1.0) * 0.5
    py = ((h[1] / h[3] + 1.0) * H - 1.0) * 0.5
    assert abs(px.item() - (0.5 * W + 21.5 - 0.5)) < 1e-2 and abs(py.item() - (0.5 * H - 13.25 - 0.5)) < 1e-2


def test_means2d_gradient_accumulates_over_the_two_reference_passes(hip_device):
    """gaussian_renderer/__init__.py: ONE `screenspace_points` tensor is handed to the colour pass (:94-102) and to the mask pass
    (:106-114); after `loss.backward()` its `.grad` is the sum of both passes' screen-space gradients, which is what
    `add_densification_stats` reads (scene/gaussian_model.py:427-431).  The literal two-pass call pattern and the fused
    single pass must leave the same `.grad` -- and both must equal the oracle's gradient of the combined loss."""
    from splatfields_amd.render import render
    dev = hip_device
    sp, cam, st, grads = make_scene(4000, 136, 104, mean_scale=0.035, view=4)
    gi, gd, ga = [g.to(dev) for g in grads]
    camd = cam.to(dev)
    bg = st.bg.to(dev)
    pipe = types.SimpleNamespace(debug=False)

    def run(two_pass):
        m3 = sp["means3D"].to(dev).requires_grad_(True)
        gdict = {"means3D": m3, "active_sh_degree": 3, "gaussian_opacity": sp["opacities"].to(dev).requires_grad_(True),
                 "gaussian_features": sp["shs"].to(dev), "gaussian_scales": sp["scales"].to(dev), "gaussian_rotations": sp["rotations"].to(dev)}
        pkg = render(camd, gdict, pipe, bg, two_pass=two_pass)
        ((pkg["render"] * gi).sum() + (pkg["opacity"] * ga).sum()).backward()
        return pkg["viewspace_points"].grad.cpu(), m3.grad.cpu(), gdict["gaussian_opacity"].grad.cpu(), pkg

    g2_two, gm_two, go_two, pkg_two = run(True)
    g2_one, gm_one, go_one, _ = run(False)
    # each pass alone: the colour-only and the mask-only gradient; their sum is what the two-pass tensor holds
    def single(loss_of):
        m3 = sp["means3D"].to(dev).requires_grad_(True)
        gdict = {"means3D": m3, "active_sh_degree": 3, "gaussian_opacity": sp["opacities"].to(dev), "gaussian_features": sp["shs"].to(dev),
                 "gaussian_scales": sp["scales"].to(dev), "gaussian_rotations": sp["rotations"].to(dev)}
        pkg = render(camd, gdict, pipe, bg, two_pass=True)
        loss_of(pkg).backward()
        return pkg["viewspace_points"].grad.cpu()
    g_colour = single(lambda p: (p["render"] * gi).sum())
    g_mask = single(lambda p: (p["opacity"] * ga).sum())
    scale = g2_two.abs().max().item()
    assert scale > 0 and g_colour.abs().max() > 0 and g_mask.abs().max() > 0
    assert (g2_two - (g_colour + g_mask)).abs().max().item() <= 1e-5 * scale
    assert (g2_two - g2_one).abs().max().item() <= 2e-5 * scale
    assert (gm_two - gm_one).abs().max().item() <= 2e-5 * gm_one.abs().max().item()
    assert (go_two - go_one).abs().max().item() <= 2e-5 * go_one.abs().max().item()
    ref, gr = O.fwd_bwd(sp, st, grads[0], torch.zeros_like(grads[1]), grads[2], use_sh=True, dtype=torch.float64)
    assert grad_error(g2_two, gr["means2D"]) <= GRAD_TOL64 and grad_error(g2_one, gr["means2D"]) <= GRAD_TOL64
    assert (g2_two[:, 2] == 0).all() and (g2_two[~pkg_two["visibility_filter"].cpu()] == 0).all()


def test_prefiltered_flag_changes_nothing(hip_device):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = hip_device
    sp, cam, st, grads = make_scene(3000, 112, 80, view=6)
    # a third of the cloud behind the camera: what a caller-side frustum filter would have removed
    sp["means3D"][::3] = cam.camera_center[None] * (1.3 + 0.2 * torch.rand(1000, 1, generator=torch.Generator().manual_seed(5)))
    res = {}
    for flag in (False, True):
        leaf = {k: v.to(dev).clone().requires_grad_(True) for k, v in sp.items()}
        rs = GaussianRasterizationSettings(image_height=80, image_width=112, tanfovx=st.tanfovx, tanfovy=st.tanfovy, bg=st.bg.to(dev),
                                           scale_modifier=1.0, viewmatrix=st.viewmatrix.to(dev), projmatrix=st.projmatrix.to(dev),
                                           sh_degree=3, campos=st.campos.to(dev), prefiltered=flag, debug=False)
        color, radii, depth = GaussianRasterizer(rs)(means3D=leaf["means3D"], means2D=torch.zeros_like(leaf["means3D"]),
                                                     opacities=leaf["opacities"], shs=leaf["shs"], scales=leaf["scales"],
                                                     rotations=leaf["rotations"])
        ((color * grads[0].to(dev)).sum() + (depth * grads[1].to(dev)).sum()).backward()
        res[flag] = (color.detach().cpu(), radii.cpu(), depth.detach().cpu(), leaf["means3D"].grad.cpu(), leaf["shs"].grad.cpu())
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)
    assert (res[True][1][::3] == 0).all() and int((res[True][1] > 0).sum()) > 500


def test_debug_snapshot_holds_the_call_arguments(hip_device, tmp_path, monkeypatch):
    """`debug=True` (reference gaussian_renderer/__init__.py:71 passes pipe.debug): [EXT] synchronises after every kernel and,
    when one fails, dumps the call's arguments to snapshot_fw.dump / snapshot_bw.dump before it re-raises.  The library does
    the same (api.hip: after_launch -> dump_snapshot); the test hook runs that path without breaking the device, and the file
    must hold exactly what was passed."""
    import ctypes as C
    import struct
    import numpy as np
    from splatfields_amd import _lib
    from splatfields_amd.rasterizer import GaussianRasterizationSettings, _ViewPack, _splats_struct
    lib = _lib.load()
    dev = hip_device
    monkeypatch.chdir(tmp_path)
    sp, cam, st, grads = make_scene(777, 48, 32, view=2)
    d = {k: v.to(dev).contiguous() for k, v in sp.items()}
    rs = GaussianRasterizationSettings(image_height=32, image_width=48, tanfovx=st.tanfovx, tanfovy=st.tanfovy, bg=st.bg.to(dev),
                                       scale_modifier=0.75, viewmatrix=st.viewmatrix.to(dev), projmatrix=st.projmatrix.to(dev),
                                       sh_degree=2, campos=st.campos.to(dev), prefiltered=False, debug=True)
    view = _ViewPack.get(rs, dev, 16)
    splats = _splats_struct(777, d["means3D"], d["opacities"].reshape(-1), d["scales"], d["rotations"], None, d["shs"], None)
    gi = grads[0].to(dev).contiguous()
    assert lib.sr_debug_snapshot(C.byref(view.struct), C.byref(splats), C.c_void_p(gi.data_ptr()), None, None, 1) == 0
    assert lib.sr_debug_snapshot(C.byref(view.struct), C.byref(splats), None, None, None, 0) == 0
    for name, want_grad in (("snapshot_bw.dump", True), ("snapshot_fw.dump", False)):
        raw = (tmp_path / name).read_bytes()
        assert raw[:8] == b"SRSNAP1\0" and b"test hook" in raw[8:40]
        recs, off = {}, 40
        while off < len(raw):
            tag = raw[off:off + 16].split(b"\0")[0].decode()
            n, = struct.unpack_from("<Q", raw, off + 16)
            recs[tag] = raw[off + 24:off + 24 + n]
            off += 24 + n
        h, w, deg, k = struct.unpack_from("<4i", recs["view"], 0)
        tfx, tfy, sm = struct.unpack_from("<3f", recs["view"], 16)
        assert (h, w, deg, k) == (32, 48, 2, 16) and abs(sm - 0.75) < 1e-7 and abs(tfx - st.tanfovx) < 1e-6 and abs(tfy - st.tanfovy) < 1e-6
        assert struct.unpack("<i", recs["count"])[0] == 777
        for key, ref in (("means3D", sp["means3D"]), ("opacities", sp["opacities"]), ("scales", sp["scales"]), ("rotations", sp["rotations"]),
                         ("shs", sp["shs"]), ("viewmatrix", st.viewmatrix), ("projmatrix", st.projmatrix), ("campos", st.campos)):
            got = np.frombuffer(recs[key], dtype=np.float32)
            assert np.array_equal(got, ref.contiguous().numpy().reshape(-1)), key
        assert len(recs["colors_precomp"]) == 0 and len(recs["cov3D_precomp"]) == 0 and len(recs["dL_ddepth"]) == 0
        assert (len(recs["dL_dcolor"]) == 3 * 32 * 48 * 4) == want_grad
        if want_grad:
            assert np.array_equal(np.frombuffer(recs["dL_dcolor"], dtype=np.float32), grads[0].numpy().reshape(-1))
    # and a debug render goes through (every stage synchronised and checked, nothing dumped, nothing raised)
    from diff_gaussian_rasterization import GaussianRasterizer
    (tmp_path / "snapshot_fw.dump").unlink()
    color, radii, depth = GaussianRasterizer(rs)(means3D=d["means3D"], means2D=torch.zeros_like(d["means3D"]), opacities=d["opacities"],
                                                 shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
    assert torch.isfinite(color).all() and not (tmp_path / "snapshot_fw.dump").exists()
