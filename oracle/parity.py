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


def image_figures_robust(hip: torch.Tensor, ref: torch.Tensor, fragile: torch.Tensor, bound: torch.Tensor = None,
                         tol: float = None) -> dict:
    """The north star's bound, as the small-scene tests state it: <= 1e-4 relative (to max(|ref|, 1e-3)) on every pixel whose
    threshold decisions are not within the fp32 margin of their threshold (`fragile`, [H,W] bool from the oracle); the fragile
    pixels may flip one decision and are bounded absolutely.
    `bound` (same shape as ref; raster_ref.c `cond_bound`): what a couple of float ulps in the splats' stored screen-space
    centres do to the pixel, to first order.  A robust pixel beyond 1e-4 must be within 1e-4 + that bound
    (`conditioning_limited`: the rim of one or two faint splats, where any fp32 pipeline carries ~1e-4 per ulp of the centre);
    what is beyond both is `unexplained`.  `tol`: IMAGE_TOL against the oracle in double; 2 x IMAGE_TOL against the oracle in
    float, which is itself an fp32 evaluation within IMAGE_TOL of the exact value (triangle inequality)."""
    tol = IMAGE_TOL if tol is None else tol
    a, b = hip.double(), ref.double()
    err = (a - b).abs()
    floor = b.abs().clamp_min(IMAGE_REL_FLOOR)
    rel = err / floor
    fr = fragile[None].expand_as(rel)
    rob = rel[~fr]
    above = (rel > IMAGE_TOL) & ~fr
    out = {"robust_max_rel": float(rob.max().item()) if rob.numel() else 0.0,
           "robust_above_1e-4": int(above.sum().item()),
           "fragile_share": float(fragile.double().mean().item()),
           "fragile_max_abs": float(err[fr].max().item()) if fr.any() else 0.0,
           "median_rel": _pct(rel, 0.5)}
    if bound is not None:
        beyond = above & (err > tol * floor + bound.double())
        out["conditioning_limited"] = out["robust_above_1e-4"] - int(beyond.sum().item())
        out["unexplained"] = int(beyond.sum().item())
        # how much of the allowance the worst such pixel uses (1.0 = at the bound)
        out["max_err_over_allowance"] = float((err / (tol * floor + bound.double()))[~fr].max().item()) if rob.numel() else 0.0
    else:
        out["unexplained"] = out["robust_above_1e-4"]
    return out


def grad_figures_flagged(hip: torch.Tensor, ref: torch.Tensor, splat_flag: torch.Tensor) -> dict:
    """Every gradient element beyond GRAD_BULK_TOL of its tensor's maximum must belong to a splat the oracle flagged
    (blended into a threshold-fragile pixel: one flipped decision moves its sums by a discrete amount).  `unexplained`
    counts the elements beyond the bound on splats that are NOT flagged."""
    b = ref.double()
    scale = b.abs().max().clamp_min(1e-30)
    err = ((hip.double() - b).abs() / scale).reshape(b.shape[0], -1)
    per_splat = err.max(dim=1).values
    unflagged = per_splat[~splat_flag]
    return {"max_rel_to_tensor_max": float(err.max().item()),
            "max_unflagged": float(unflagged.max().item()) if unflagged.numel() else 0.0,
            "unexplained": int((unflagged > GRAD_BULK_TOL).sum().item()),
            "splats_above_1e-3": int((per_splat > GRAD_BULK_TOL).sum().item()),
            "median_rel_to_tensor_max": _pct(err, 0.5)}


def radii_figures(hip_radii: torch.Tensor, ref_out: dict) -> dict:
    """radii are integers: equal, except where 3 sqrt(lambda) sits within fp32 rounding of an integer (the ceil flips)."""
    diff = hip_radii.to(torch.int64) != ref_out["radii"].to(torch.int64)
    n = int(diff.sum().item())
    res = {"mismatches": n, "unexplained": n}
    if n and "radius_raw" in ref_out:
        raw = ref_out["radius_raw"].double()
        near_int = (raw - raw.round()).abs() < 2e-4 * raw.clamp_min(1.0)
        res["unexplained"] = int((diff & ~near_int).sum().item())
    return res


def compare_flagged(out: dict, grads: dict, ref_out: dict, ref_grads: dict, oracle_precision: str = "fp64") -> dict:
    """compare() with the oracle's fragile mask, splat flags and conditioning bound (c_oracle.rasterize(..., fragile=True)).
    oracle_precision "fp32": the reference is itself an fp32 evaluation -- the image tolerance is two-sided (2 x IMAGE_TOL)."""
    tol = IMAGE_TOL * (2.0 if oracle_precision == "fp32" else 1.0)
    fr, fl = ref_out["fragile"], ref_out["splat_flag"]
    cb = ref_out.get("cond_bound")
    bounds = {"color": None, "depth": None, "alpha": None} if cb is None else {"color": cb[0:3], "depth": cb[3:4], "alpha": cb[4:5]}
    rad = radii_figures(out["radii"], ref_out)
    res = {"radii_equal": rad["mismatches"] == 0, "radii": rad,
           "images": {k: image_figures_robust(out[k], ref_out[k], fr, bounds[k], tol) for k in ("color", "depth", "alpha")},
           "image_tolerance": tol,
           "gradients": {k: grad_figures_flagged(grads[k], ref_grads[k], fl) for k in ref_grads if k in grads},
           "flagged_splat_share": float(fl.double().mean().item())}
    res["image_robust_max_rel"] = max(v["robust_max_rel"] for v in res["images"].values())
    res["image_robust_above_1e-4"] = sum(v["robust_above_1e-4"] for v in res["images"].values())
    res["image_conditioning_limited"] = sum(v.get("conditioning_limited", 0) for v in res["images"].values())
    res["gradient_max_unflagged"] = max(v["max_unflagged"] for v in res["gradients"].values())
    res["unexplained"] = (rad["unexplained"] + sum(v["unexplained"] for v in res["images"].values())
                          + sum(v["unexplained"] for v in res["gradients"].values()))
    return res
