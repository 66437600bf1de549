"""Counterpart of the reference's render boundary, gaussian_renderer/__init__.py:30-124.

Same signature, same ``gaussian_dict`` keys in, same result dict out
(``render, viewspace_points, visibility_filter, radii, opacity, depth``).  Two differences, both
inside the boundary: (1) the alpha image comes out of the same rasterization (fused output) instead
of a second full pass with white colours on a black background (:104-115) -- identical values, half
the work; ``two_pass=True`` reproduces the reference's call pattern literally; (2) the device of the
screen-space dummy follows ``means3D`` instead of the literal "cuda" (:49)."""
from __future__ import annotations

import math

import torch

from . import rasterizer as _rz
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _redeemed(render_fn):
    """`render_fn()`'s outputs with every asynchronously launched forward of this host thread checked (rasterizer.py:
    sr_forward_async; only when the caller has switched that on): a forward whose capacity promise did not hold has no result
    and is rendered again -- the estimates are corrected by then."""
    out = render_fn()
    try:
        _rz.resolve_pending()
    except _rz.RasterizerOverflow:
        out = render_fn()
        _rz.resolve_pending()
    return out


def render(viewpoint_camera, gaussian_dict: dict, pipe, bg_color: torch.Tensor, scaling_modifier=1.0,
           return_opacity=True, two_pass=False):
    means3D = gaussian_dict['means3D']
    active_sh_degree = gaussian_dict['active_sh_degree']
    gaussian_opacity = gaussian_dict['gaussian_opacity']
    gaussian_scales = gaussian_dict['gaussian_scales']
    gaussian_rotations = gaussian_dict['gaussian_rotations']
    gaussian_features = gaussian_dict.get('gaussian_features', None)
    gaussian_rgb = gaussian_dict.get('gaussian_rgb', None)
    if gaussian_rgb is None and 'gaussian_rgb_fnc' in gaussian_dict:
        ray_d = means3D - viewpoint_camera.camera_center[None]
        ray_d = ray_d / torch.norm(ray_d, dim=-1, keepdim=True)
        gaussian_rgb = gaussian_dict['gaussian_rgb_fnc'](ray_d)

    # zero tensor whose gradient is dL/d(screen-space mean); non-leaf + retain_grad as at :49-53
    screenspace_points = torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    def settings(bg):
        return GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=bg, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
            projmatrix=viewpoint_camera.full_proj_transform, sh_degree=active_sh_degree,
            campos=viewpoint_camera.camera_center, prefiltered=False, debug=bool(getattr(pipe, "debug", False)))

    rasterizer = GaussianRasterizer(raster_settings=settings(bg_color))
    kw = dict(means3D=means3D, means2D=screenspace_points, shs=gaussian_features, colors_precomp=gaussian_rgb,
              opacities=gaussian_opacity, scales=gaussian_scales, rotations=gaussian_rotations, cov3D_precomp=None)
    if two_pass:   # the reference's literal calls: the three-output `forward` (:94-102), then the mask rasterizer (:104-115)
        fwd = lambda: rasterizer(**kw) + (None,)
    else:
        fwd = lambda: rasterizer.forward_ex(**kw)
    # pipe.debug: nothing leaves this function unchecked (with SPLATRASTER_ASYNC=1 a forward may have been launched on a promise)
    rendered_image, radii, depth, alpha = _redeemed(fwd) if getattr(pipe, "debug", False) else fwd()
    opacity_image = None
    if return_opacity:
        if two_pass:
            rasterizer_mask = GaussianRasterizer(raster_settings=settings(bg_color * 0.0))
            opacity_image = rasterizer_mask(
                means3D=means3D, means2D=screenspace_points, shs=None,
                colors_precomp=torch.ones(gaussian_opacity.shape[0], 3, device=gaussian_opacity.device),
                opacities=gaussian_opacity, scales=gaussian_scales, rotations=gaussian_rotations,
                cov3D_precomp=None)[0][:1]
        else:
            opacity_image = alpha
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "opacity": opacity_image, "depth": depth}


def render_model(viewpoint_camera, gaussians, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, return_opacity=True):
    """``render`` for the static / warm-up branch of reference train.py:41-50, taking the ``GaussianModel`` itself instead
    of the dict of its accessors: the raw parameters (``_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation``,
    scene/gaussian_model.py:38-46) go to the rasterizer as they are stored, and ``exp`` / ``sigmoid`` / ``normalize`` /
    ``cat`` (:64-86) happen inside its preprocess kernels (``GaussianRasterizer.forward_raw``).  Same result dict as
    ``render``; gradients arrive on the raw parameters."""
    means3D = gaussians._xyz
    log_scales = gaussians._scaling
    if log_scales.shape[-1] == 1:  # use_isotropic (gaussian_model.py:64-68): the accessor repeats the single scale
        log_scales = log_scales.expand(-1, 3)
    screenspace_points = torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=int(gaussians.active_sh_degree),
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=bool(getattr(pipe, "debug", False)))
    dc, rest = gaussians._features_dc, gaussians._features_rest
    if rest.shape[1] == 15:
        kw = dict(shs=dc, shs_rest=rest)
    else:  # other SH widths: the concatenated tensor, as the accessor builds it
        kw = dict(shs=torch.cat((dc, rest), dim=1))
    fwd = lambda: GaussianRasterizer(raster_settings=rs).forward_raw(
        means3D=means3D, means2D=screenspace_points, opacity_logits=gaussians._opacity, log_scales=log_scales,
        quaternions=gaussians._rotation, **kw)
    rendered_image, radii, depth, alpha = _redeemed(fwd) if getattr(pipe, "debug", False) else fwd()
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "opacity": alpha if return_opacity else None, "depth": depth}
