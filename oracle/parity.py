"""Parity figures of one forward+backward against an oracle result (bench.py's `parity` block, the full-size tests).
TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by the product path."""
from __future__ import annotations

import torch

IMAGE_REL_FLOOR = 1e-3     # relative error is taken against max(|ref|, 1e-3) (SURVEY.md Appendix A "Tolerance basis")
IMAGE_TOL = 1e-4           # the north star's image tolerance
GRAD_BULK_TOL = 1e-3       # per element, relative to the tensor's largest |ref| entry


def _pct(x: torch.Tensor, q: float) -> float:
    """q-quantile of a large tensor (torch.quantile is limited to 16 M elements: k-th value instead)"""
    flat = x.reshape(-1)
    k = min(flat.numel(), max(1, int(round(q * flat.numel()))))
    return float(flat.kthvalue(k).values.item())


def image_figures(hip: torch.Tensor, ref: torch.Tensor) -> dict:
    a, b = hip.double(), ref.double()
    err = (a - b).abs()
    rel = err / b.abs().clamp_min(IMAGE_REL_FLOOR)
    return {"median_rel": _pct(rel, 0.5), "p999_rel": _pct(rel, 0.999), "share_above_1e-4": float((rel > IMAGE_TOL).double().mean().item()),
            "max_rel": float(rel.max().item()), "max_abs": float(err.max().item())}


def grad_figures(hip: torch.Tensor, ref: torch.Tensor) -> dict:
    b = ref.double()
    scale = b.abs().max().clamp_min(1e-30)
    err = (hip.double() - b).abs() / scale
    return {"max_rel_to_tensor_max": float(err.max().item()), "share_above_1e-3": float((err > GRAD_BULK_TOL).double().mean().item()),
            "median_rel_to_tensor_max": _pct(err, 0.5)}


def compare(out: dict, grads: dict, ref_out: dict, ref_grads: dict) -> dict:
    """out / ref_out: color, depth, alpha, radii; grads / ref_grads: per input name (CPU tensors)."""
    res = {"radii_equal": bool(torch.equal(out["radii"].to(torch.int64), ref_out["radii"].to(torch.int64))),
           "images": {k: image_figures(out[k], ref_out[k]) for k in ("color", "depth", "alpha")},
           "gradients": {k: grad_figures(grads[k], ref_grads[k]) for k in ref_grads if k in grads},
           "definitions": "images: rel = |hip - ref| / max(|ref|, 1e-3) per pixel and channel, both sides fp32 (a pixel whose alpha >= "
                          "1/255 or T >= 1e-4 decision flips between the two fp32 evaluations differs by one blended pair: those are "
                          "the share above 1e-4); gradients: |hip - ref| / max|ref| of the tensor"}
    res["image_max_rel_on_99.9pct_of_pixels"] = max(v["p999_rel"] for v in res["images"].values())
    res["image_share_above_1e-4"] = max(v["share_above_1e-4"] for v in res["images"].values())
    res["gradient_max_rel_to_tensor_max"] = max(v["max_rel_to_tensor_max"] for v in res["gradients"].values())
    return res
