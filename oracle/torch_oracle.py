"""CPU oracle #1: differentiable PyTorch restatement of the tile rasterizer.

TEST INFRASTRUCTURE ONLY.  Nothing under ``splatfields_amd/`` or
``diff_gaussian_rasterization/`` may import this file; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do.

PARITY UNPINNED: the algorithm lives in a third-party dependency that is absent from
/root/reference -- ``ingra14m/depth-diff-gaussian-rasterization@f2d8fa9`` (reference
README.md:28, imported at gaussian_renderer/__init__.py:14).  The reference holds no tests,
fixtures or golden images for this path, and the CUDA extension cannot be
built here.  This file therefore restates the *published* 3DGS tile-rasterization algorithm
 and is cross-checked against every piece of it the reference does
restate in-tree (tests/test_oracle_vs_reference_pieces.py, fixtures made by
tests/golden/make_golden.py):

  * SH basis / polynomial / coefficient order ........ utils/sh_utils.py:26-112
  * colour = clamp_min(sh2rgb + 0.5, 0), dir = normalize(mean - campos) ... extract_geo.py:40-44
  * quaternion (r,x,y,z) -> R, Sigma = (R S)(R S)^T, 6-vector order ........ utils/general_utils.py:120-171
  * row-vector camera matrices, znear/zfar ............ utils/graphics_utils.py:42-76, scene/cameras.py:62-74
  * ndc2Pix(v,S) = ((v+1) S - 1)/2 .................... scene/dataset_readers.py:515-516
  * the call contract of the boundary ................. gaussian_renderer/__init__.py:30-124

Gradients come from torch autograd over this forward restatement, with three deliberate
"as upstream" deviations from the exact derivative (they are what the published CUDA backward
computes, so they define the reference's gradients):
  (1) alpha = min(0.99, o*G) passes its gradient straight through the clamp;
  (2) the +-1.3*tanfov clamp of t.x/t.z, t.y/t.z zeroes d/dt.x (d/dt.y) and is treated as a
      constant w.r.t. t.z;
  (3) nothing else -- the 1e-7 regulariser upstream adds to det^2 in the conic backward changes
      gradients by < 1.3e-5 relative (det >= 0.09) and is *not* reproduced here (it is in the
      C oracle and the HIP kernels); the tolerance in the tests covers it.

``means2D`` is the screen-space dummy of gaussian_renderer/__init__.py:49-53: it is added to the
NDC coordinates so that its autograd gradient is dL/d(NDC mean), which is exactly the upstream
``0.5*W / 0.5*H``-scaled convention consumed by scene/gaussian_model.py:427-438.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

TILE = 16  # binning tile edge (pixels), as upstream BLOCK_X = BLOCK_Y = 16
NEAR_CULL_Z = 0.2  # splats with view-space z <= 0.2 are culled
COV2D_DILATION = 0.3  # added to both diagonal entries of the 2D covariance
CLAMP_FOV = 1.3  # t.x/t.z, t.y/t.z clamped to +-1.3 tan(fov/2)
ALPHA_MAX = 0.99
ALPHA_MIN = 1.0 / 255.0
T_STOP = 1e-4
W_EPS = 1e-7  # p_hom.w + 1e-7

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


class OracleSettings(NamedTuple):
    """Same 12 fields as the settings tuple built at gaussian_renderer/__init__.py:59-72."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool = False
    debug: bool = False


def sh_basis(deg: int, d: torch.Tensor) -> torch.Tensor:
    """[N, (deg+1)^2] real SH basis at unit directions d[N,3]; order/signs of utils/sh_utils.py:74-100."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    cols = [torch.full_like(x, SH_C0)]
    if deg > 0:
        cols += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy), SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
    if deg > 2:
        cols += [SH_C3[0] * y * (3.0 * xx - yy), SH_C3[1] * xy * z, SH_C3[2] * y * (4.0 * zz - xx - yy),
                 SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy), SH_C3[4] * x * (4.0 * zz - xx - yy),
                 SH_C3[5] * z * (xx - yy), SH_C3[6] * x * (xx - 3.0 * yy)]
    return torch.stack(cols, dim=1)


def sh_to_rgb(deg: int, shs: torch.Tensor, means: torch.Tensor, campos: torch.Tensor) -> torch.Tensor:
    """shs[N,K,3] (coefficient-major, scene/gaussian_model.py:79-82) -> rgb[N,3] = max(SH + 0.5, 0)."""
    d = means - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    basis = sh_basis(deg, d)  # [N, M]
    rgb = (basis[:, :, None] * shs[:, : basis.shape[1], :]).sum(dim=1) + 0.5
    return torch.clamp_min(rgb, 0.0)


def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    """(r,x,y,z) used AS GIVEN (no normalisation), layout of utils/general_utils.py:149-157."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).reshape(-1, 3, 3)


def covariance3d(scales: torch.Tensor, scale_modifier: float, rotations: torch.Tensor) -> torch.Tensor:
    """Sigma = (R S)(R S)^T as a full [N,3,3] matrix (utils/general_utils.py:162-171)."""
    L = quat_to_rot(rotations) * (scale_modifier * scales)[:, None, :]
    return L @ L.transpose(1, 2)


def sym6_to_mat(c6: torch.Tensor) -> torch.Tensor:
    """(xx,xy,xz,yy,yz,zz) -> symmetric [N,3,3] (order of utils/general_utils.py:120-130)."""
    xx, xy, xz, yy, yz, zz = c6.unbind(dim=1)
    return torch.stack([xx, xy, xz, xy, yy, yz, xz, yz, zz], dim=1).reshape(-1, 3, 3)


def _straight_through(value: torch.Tensor, grad_path: torch.Tensor) -> torch.Tensor:
    """Returns `value` numerically, with the derivative of `grad_path`."""
    return grad_path + (value - grad_path).detach()


class Preprocessed(NamedTuple):
    visible: torch.Tensor  # [N] bool: radius > 0 upstream
    radii: torch.Tensor  # [N] int32
    pix: torch.Tensor  # [N,2] pixel-space centre
    depth: torch.Tensor  # [N] view-space z
    conic: torch.Tensor  # [N,3] (A,B,C)
    cov2d: torch.Tensor  # [N,3] (a,b,c) with dilation
    rgb: torch.Tensor  # [N,3]
    rect: torch.Tensor  # [N,4] int64 tile rect (xmin, ymin, xmax, ymax), max exclusive
    radius_raw: torch.Tensor  # [N] 3*sqrt(lambda_max) before ceil (fragility analysis)


def preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
               s: OracleSettings) -> Preprocessed:
    dt = means3D.dtype
    H, W = int(s.image_height), int(s.image_width)
    V = s.viewmatrix.reshape(4, 4).to(dt)
    P = s.projmatrix.reshape(4, 4).to(dt)
    campos = s.campos.reshape(3).to(dt)
    N = means3D.shape[0]

    p_view = means3D @ V[:3, :3] + V[3, :3]
    hom = means3D @ P[:3, :] + P[3, :]
    inv_w = 1.0 / (hom[:, 3] + W_EPS)
    ndc = hom[:, :2] * inv_w[:, None]
    if means2D is not None:
        ndc = ndc + means2D[:, :2]
    tz = p_view[:, 2]
    visible = tz > NEAR_CULL_Z

    if cov3D_precomp is not None and cov3D_precomp.numel() > 0:
        Sigma = sym6_to_mat(cov3D_precomp)
    else:
        Sigma = covariance3d(scales, s.scale_modifier, rotations)

    # EWA projection; guard the division for culled splats (their values are never used)
    tz_safe = torch.where(visible, tz, torch.ones_like(tz))
    limx, limy = CLAMP_FOV * s.tanfovx, CLAMP_FOV * s.tanfovy
    txtz, tytz = p_view[:, 0] / tz_safe, p_view[:, 1] / tz_safe
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz_safe).detach(), p_view[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz_safe).detach(), p_view[:, 1])
    fx, fy = W / (2.0 * s.tanfovx), H / (2.0 * s.tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -fx * tx / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -fy * ty / (tz_safe * tz_safe)], dim=1).reshape(N, 2, 3)
    Rv = V[:3, :3].transpose(0, 1)  # Rv[r][c] = viewmatrix_flat[4c + r]
    M = J @ Rv  # [N,2,3]
    cov = M @ Sigma @ M.transpose(1, 2)
    a = cov[:, 0, 0] + COV2D_DILATION
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + COV2D_DILATION
    det = a * c - b * b
    visible = visible & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius_raw = (3.0 * torch.sqrt(lam)).detach()
    radius = torch.ceil(radius_raw)
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)

    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pd = pix.detach()
    # (int) casts truncate toward zero; after the clamp to [0, grid] that equals floor
    xmin = torch.trunc((pd[:, 0] - radius) / TILE).clamp(0, gx)
    xmax = torch.trunc((pd[:, 0] + radius + TILE - 1) / TILE).clamp(0, gx)
    ymin = torch.trunc((pd[:, 1] - radius) / TILE).clamp(0, gy)
    ymax = torch.trunc((pd[:, 1] + radius + TILE - 1) / TILE).clamp(0, gy)
    rect = torch.stack([xmin, ymin, xmax, ymax], dim=1)
    rect = torch.nan_to_num(rect, nan=0.0).to(torch.int64)
    visible = visible & (((rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])) > 0)

    if shs is None or shs.numel() == 0:
        rgb = colors_precomp
    else:
        rgb = sh_to_rgb(int(s.sh_degree), shs, means3D, campos)

    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)
    return Preprocessed(visible, radii, pix, tz, conic, torch.stack([a, b, c], dim=1), rgb, rect, radius_raw)


class RasterOut(NamedTuple):
    color: torch.Tensor  # [3,H,W]
    radii: torch.Tensor  # [N] int32
    depth: torch.Tensor  # [1,H,W]
    alpha: torch.Tensor  # [1,H,W] = 1 - T_final
    n_contrib: torch.Tensor  # [H,W] int32, 1-based position of the last blended list entry
    fragile: torch.Tensor  # [H,W] bool: a threshold decision of this pixel sits within the fp32 margin
    num_rendered: int  # tile-splat instances (upstream's num_rendered)
    pre: Preprocessed


def rasterize(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None, settings: OracleSettings = None, *, margin: float = 2e-4) -> RasterOut:
    """Full forward restatement.  Differentiable w.r.t. every float input."""
    s = settings
    dt = means3D.dtype
    dev = means3D.device
    H, W = int(s.image_height), int(s.image_width)
    pre = preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, s)
    bg = s.bg.reshape(3).to(dt)
    op = opacities.reshape(-1)

    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    vis_idx = torch.nonzero(pre.visible).reshape(-1)
    rect = pre.rect[vis_idx]
    depth_key = pre.depth.detach()[vis_idx]
    num_rendered = int(((rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])).sum())

    color_rows, depth_rows, alpha_rows = [], [], []
    ncontrib = torch.zeros(H, W, dtype=torch.int32)
    fragile = torch.zeros(H, W, dtype=torch.bool)
    out_c = [[None] * gx for _ in range(gy)]
    out_d = [[None] * gx for _ in range(gy)]
    out_a = [[None] * gx for _ in range(gy)]

    for ty_ in range(gy):
        row_sel = (rect[:, 1] <= ty_) & (rect[:, 3] > ty_)
        for tx_ in range(gx):
            sel = row_sel & (rect[:, 0] <= tx_) & (rect[:, 2] > tx_)
            x0, y0 = tx_ * TILE, ty_ * TILE
            x1, y1 = min(x0 + TILE, W), min(y0 + TILE, H)
            th, tw = y1 - y0, x1 - x0
            ids = vis_idx[sel]
            if ids.numel() == 0:
                out_c[ty_][tx_] = bg[:, None, None].expand(3, th, tw)
                out_d[ty_][tx_] = torch.zeros(1, th, tw, dtype=dt, device=dev)
                out_a[ty_][tx_] = torch.zeros(1, th, tw, dtype=dt, device=dev)
                continue
            # stable front-to-back order: depth (in working precision), ties by splat index
            order = torch.sort(depth_key[sel], stable=True).indices
            ids = ids[order]
            ys, xs = torch.meshgrid(torch.arange(y0, y1, dtype=dt), torch.arange(x0, x1, dtype=dt), indexing="ij")
            px = xs.reshape(-1, 1)
            py = ys.reshape(-1, 1)
            dx = pre.pix[ids, 0][None, :] - px  # [P,L]
            dy = pre.pix[ids, 1][None, :] - py
            A, B, C = pre.conic[ids, 0][None, :], pre.conic[ids, 1][None, :], pre.conic[ids, 2][None, :]
            power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy
            G = torch.exp(torch.clamp_max(power, 0.0))
            raw = op[ids][None, :] * G
            alpha = _straight_through(torch.clamp_max(raw, ALPHA_MAX), raw)
            valid = (power <= 0) & (alpha >= ALPHA_MIN)
            a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
            T_incl = torch.cumprod(1.0 - a_eff, dim=1)
            blended = valid & (T_incl >= T_STOP)
            a_bl = torch.where(blended, alpha, torch.zeros_like(alpha))
            T_all = torch.cumprod(1.0 - a_bl, dim=1)
            T_excl = torch.cat([torch.ones_like(T_all[:, :1]), T_all[:, :-1]], dim=1)
            w = a_bl * T_excl  # [P,L]
            T_fin = T_all[:, -1]
            col = w @ pre.rgb[ids] + T_fin[:, None] * bg[None, :]
            dep = w @ pre.depth[ids]
            out_c[ty_][tx_] = col.t().reshape(3, th, tw)
            out_d[ty_][tx_] = dep.reshape(1, th, tw)
            out_a[ty_][tx_] = (1.0 - T_fin).reshape(1, th, tw)
            with torch.no_grad():
                pos = torch.arange(1, ids.numel() + 1, dtype=torch.int32)[None, :].expand_as(blended)
                ncontrib[y0:y1, x0:x1] = torch.where(blended, pos, torch.zeros_like(pos)).max(dim=1).values.reshape(th, tw)
                # decisions within `margin` (relative) of their threshold can flip under fp32 rounding
                reached = torch.cat([torch.ones_like(T_incl[:, :1], dtype=torch.bool), T_incl[:, :-1] >= T_STOP * (1 - 50 * margin)], dim=1)
                near_alpha = ((alpha - ALPHA_MIN).abs() < margin * ALPHA_MIN) & reached
                near_t = valid & ((T_incl - T_STOP).abs() < 50 * margin * T_STOP) & reached
                near_p = (power.abs() < 1e-6) & (raw >= ALPHA_MIN) & reached
                fragile[y0:y1, x0:x1] = (near_alpha | near_t | near_p).any(dim=1).reshape(th, tw)

    color = torch.cat([torch.cat(r, dim=2) for r in out_c], dim=1)
    depth = torch.cat([torch.cat(r, dim=2) for r in out_d], dim=1)
    alpha_img = torch.cat([torch.cat(r, dim=2) for r in out_a], dim=1)
    return RasterOut(color, pre.radii, depth, alpha_img, ncontrib, fragile, num_rendered, pre)


def fwd_bwd(inputs: dict, settings: OracleSettings, g_img, g_depth=None, g_alpha=None, *, use_sh=True,
            dtype=torch.float64):
    """Convenience for tests: forward + autograd backward of
    loss = sum(color*g_img) + sum(depth*g_depth) + sum(alpha*g_alpha).  Returns (RasterOut, grads dict)."""
    leaf = {}
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "colors_precomp"):
        if inputs.get(k) is not None:
            leaf[k] = inputs[k].detach().to(dtype).clone().requires_grad_(True)
    means2D = torch.zeros_like(leaf["means3D"], requires_grad=True)
    out = rasterize(leaf["means3D"], means2D, leaf["opacities"],
                    shs=leaf["shs"] if use_sh else None,
                    colors_precomp=None if use_sh else leaf["colors_precomp"],
                    scales=leaf["scales"], rotations=leaf["rotations"], settings=settings)
    loss = (out.color * g_img.to(dtype)).sum()
    if g_depth is not None:
        loss = loss + (out.depth * g_depth.to(dtype)).sum()
    if g_alpha is not None:
        loss = loss + (out.alpha * g_alpha.to(dtype)).sum()
    names = ["means3D", "scales", "rotations", "opacities", "shs" if use_sh else "colors_precomp"]
    gr = torch.autograd.grad(loss, [leaf[n] for n in names] + [means2D], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaf[n])) for n, g in zip(names, gr[:-1])}
    grads["means2D"] = gr[-1] if gr[-1] is not None else torch.zeros_like(means2D)
    return out, grads


def settings_from_camera(cam, bg, sh_degree: int, scale_modifier: float = 1.0) -> OracleSettings:
    """Build the settings tuple the way gaussian_renderer/__init__.py:56-72 does."""
    return OracleSettings(int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx * 0.5),
                          math.tan(cam.FoVy * 0.5), bg, scale_modifier, cam.world_view_transform,
                          cam.full_proj_transform, sh_degree, cam.camera_center, False, False)
