"""Splat PLY files in the on-disk layout of the reference (scene/gaussian_model.py:153-205 ``save_ply`` /
:212-250 ``load_ply``), without the ``plyfile`` dependency: one binary little-endian ``vertex`` element of float32
properties ``x y z nx ny nz f_dc_0..2 f_rest_0..(3(K-1)-1) opacity scale_0.. rot_0..3``.

Stored values are the reference's *pre-activation* parameters: opacity logits, log-scales, raw quaternions, and SH
coefficients channel-major (``features.transpose(1, 2).flatten(1)``, so ``f_rest_{c*(K-1)+k}`` is channel c of
coefficient k+1).  ``load_rasterizer_inputs`` applies the activations of scene/gaussian_model.py:64-86, giving exactly
what ``get_gaussian_dict`` (train.py:41-50) hands to the rasterizer.  SURVEY.md §8f row 4 (host-side I/O)."""
from __future__ import annotations

import numpy as np
import torch


def attribute_names(n_rest: int, n_scale: int = 3, n_rot: int = 4) -> list:
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(n_rest)]
    return names + ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]


def save_ply(path: str, xyz, features_dc, features_rest, opacity_logit, log_scales, rotations) -> None:
    """features_dc [N,1,3], features_rest [N,K-1,3] (coefficient-major, as the model stores them)."""
    t = lambda a: np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float32)
    xyz, op, sc, rot = t(xyz), t(opacity_logit).reshape(len(xyz), -1), t(log_scales), t(rotations)
    f_dc = np.transpose(t(features_dc), (0, 2, 1)).reshape(len(xyz), -1)
    f_rest = np.transpose(t(features_rest), (0, 2, 1)).reshape(len(xyz), -1)
    data = np.concatenate([xyz, np.zeros_like(xyz), f_dc, f_rest, op, sc, rot], axis=1).astype("<f4")
    names = attribute_names(f_rest.shape[1], sc.shape[1], rot.shape[1])
    assert data.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % len(xyz)
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(data).tobytes())


def load_ply(path: str) -> dict:
    """-> dict of float32 numpy arrays keyed by the reference's parameter names (pre-activation)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, props = None, None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if count is not None:
                    raise ValueError("only a single vertex element is supported")
                if tok[1] != "vertex":
                    raise ValueError("first element must be 'vertex'")
                count = int(tok[2])
            elif tok[0] == "property":
                if tok[1] not in ("float", "float32"):
                    raise ValueError(f"unsupported property type {tok[1]}")
                props.append(tok[2])
            elif tok[0] == "end_header":
                break
        if fmt != "binary_little_endian":
            raise ValueError("only binary_little_endian PLY files are supported")
        raw = np.frombuffer(f.read(count * len(props) * 4), dtype="<f4").reshape(count, len(props))
    col = {n: i for i, n in enumerate(props)}
    pick = lambda prefix: sorted((n for n in props if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    rest_names, scale_names, rot_names = pick("f_rest_"), pick("scale_"), pick("rot_")
    n_rest = len(rest_names) // 3
    f_dc = np.stack([raw[:, col[f"f_dc_{c}"]] for c in range(3)], axis=1)[:, None, :]  # [N,1,3]
    f_rest = raw[:, [col[n] for n in rest_names]].reshape(count, 3, n_rest).transpose(0, 2, 1)  # [N,K-1,3]
    return {"xyz": raw[:, [col["x"], col["y"], col["z"]]].copy(), "features_dc": f_dc.copy(), "features_rest": f_rest.copy(),
            "opacity": raw[:, [col["opacity"]]].copy(), "scaling": raw[:, [col[n] for n in scale_names]].copy(),
            "rotation": raw[:, [col[n] for n in rot_names]].copy()}


def load_rasterizer_inputs(path: str, device="cpu") -> dict:
    """Activated tensors in the rasterizer's input layout (scene/gaussian_model.py:64-86)."""
    p = {k: torch.from_numpy(v).to(device) for k, v in load_ply(path).items()}
    scales = torch.exp(p["scaling"])
    if scales.shape[1] == 1:  # isotropic models store one scale (gaussian_model.py:64-68)
        scales = scales.repeat(1, 3)
    return {"means3D": p["xyz"], "opacities": torch.sigmoid(p["opacity"]), "scales": scales,
            "rotations": torch.nn.functional.normalize(p["rotation"]),
            "shs": torch.cat([p["features_dc"], p["features_rest"]], dim=1).contiguous()}
