"""Seeded synthetic splat clouds and Blender-lego-like cameras.

These stand in for the datasets the reference loads (scene/dataset_readers.py), which are
not available offline.  The camera object exposes exactly the attributes that the
reference's render boundary reads (gaussian_renderer/__init__.py:56-70):
``FoVx, FoVy, image_height, image_width, world_view_transform, full_proj_transform,
camera_center``.  Matrix conventions follow scene/cameras.py:62-74 and
utils/graphics_utils.py:42-76: ``world_view_transform`` is the *transposed* world-to-camera
matrix (points are row vectors, ``p_row @ M``), the camera looks along +z with x right and
y down, znear = 0.01, zfar = 100.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

ZNEAR = 0.01
ZFAR = 100.0
LEGO_CAMERA_ANGLE_X = 0.6911112070083618  # transforms_train.json of NeRF-synthetic lego
LEGO_DISTANCE = 4.031128874 * (2.0 / 2.4)  # world-scale rule, scene/dataset_readers.py:460-474


def world_to_view(cam_pos, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """4x4 world->camera matrix (column-vector convention), camera: x right, y down, z forward."""
    c = np.asarray(cam_pos, dtype=np.float64)
    fwd = np.asarray(target, dtype=np.float64) - c
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, dtype=np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    rot = np.stack([right, down, fwd], axis=0)  # rows = camera axes in world coordinates
    w2c = np.eye(4)
    w2c[:3, :3] = rot
    w2c[:3, 3] = -rot @ c
    return w2c


def projection(znear: float, zfar: float, fovx: float, fovy: float) -> np.ndarray:
    """Perspective matrix with the reference's z convention (utils/graphics_utils.py:56-76):
    w_clip = z_view, depth mapped to [0, 1]."""
    tx, ty = math.tan(fovx * 0.5), math.tan(fovy * 0.5)
    p = np.zeros((4, 4))
    p[0, 0] = 1.0 / tx
    p[1, 1] = 1.0 / ty
    p[3, 2] = 1.0
    p[2, 2] = zfar / (zfar - znear)
    p[2, 3] = -(zfar * znear) / (zfar - znear)
    return p


@dataclass
class SyntheticCamera:
    """Attribute-compatible stand-in for scene/cameras.py:18 ``Camera`` at the render boundary."""

    FoVx: float
    FoVy: float
    image_height: int
    image_width: int
    world_view_transform: torch.Tensor  # [4,4], transposed W2C; deliberately strided like the reference's
    full_proj_transform: torch.Tensor  # [4,4]
    camera_center: torch.Tensor  # [3]

    def to(self, device) -> "SyntheticCamera":
        return SyntheticCamera(self.FoVx, self.FoVy, self.image_height, self.image_width,
                               self.world_view_transform.to(device), self.full_proj_transform.to(device),
                               self.camera_center.to(device))


def make_camera(view_index: int, width: int, height: int, *, fovx: float = LEGO_CAMERA_ANGLE_X,
                distance: float = LEGO_DISTANCE, elevation_deg: float = 30.0,
                azimuth_step_deg: float = 45.0, device="cpu") -> SyntheticCamera:
    az = math.radians(azimuth_step_deg * view_index)
    el = math.radians(elevation_deg)
    pos = distance * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    focal = width / (2.0 * math.tan(fovx * 0.5))
    fovy = 2.0 * math.atan(height / (2.0 * focal))
    w2c = np.float32(world_to_view(pos))
    # same storage pattern as scene/cameras.py:68: a transposed *view* of the row-major W2C (strides (1,4))
    wvt = torch.tensor(w2c).transpose(0, 1)
    proj_t = torch.tensor(np.float32(projection(ZNEAR, ZFAR, fovx, fovy))).transpose(0, 1)
    full = wvt.unsqueeze(0).bmm(proj_t.unsqueeze(0)).squeeze(0)
    center = wvt.inverse()[3, :3]
    return SyntheticCamera(fovx, fovy, int(height), int(width), wvt.to(device), full.to(device), center.to(device))


def make_splats(n: int, *, seed: int = 1234, sh_coeffs: int = 16, mean_scale: float | None = None,
                device="cpu", dtype=torch.float32) -> dict:
    """Synthetic cloud of SURVEY.md §8d.  Returns activated attributes, i.e. what
    scene/gaussian_model.py:64-86 hands to the rasterizer."""
    g = torch.Generator().manual_seed(seed)
    means = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    sbar = mean_scale if mean_scale is not None else 0.35 * n ** (-1.0 / 3.0)
    lo, hi = math.log(0.5 * sbar), math.log(2.0 * sbar)
    scales = torch.exp(torch.rand(n, 3, generator=g) * (hi - lo) + lo)
    rots = torch.randn(n, 4, generator=g)
    rots = rots / rots.norm(dim=1, keepdim=True)
    opac = torch.rand(n, 1, generator=g) * 0.9 + 0.05
    shs = torch.randn(n, sh_coeffs, 3, generator=g)
    shs[:, 1:, :] *= 0.1
    rgb = torch.rand(n, 3, generator=g)
    out = dict(means3D=means, scales=scales, rotations=rots, opacities=opac, shs=shs, colors_precomp=rgb)
    return {k: v.to(device=device, dtype=dtype) for k, v in out.items()}


def make_upstream_grads(height: int, width: int, *, seed: int = 99, device="cpu", dtype=torch.float32):
    """Fixed dL/dimage, dL/ddepth, dL/dalpha used by parity tests and the benchmark (§8d)."""
    g = torch.Generator().manual_seed(seed)
    g_img = torch.randn(3, height, width, generator=g) / (3 * height * width)
    g_depth = torch.randn(1, height, width, generator=g) / (height * width)
    g_alpha = torch.randn(1, height, width, generator=g) / (height * width)
    return g_img.to(device=device, dtype=dtype), g_depth.to(device=device, dtype=dtype), g_alpha.to(device=device, dtype=dtype)
