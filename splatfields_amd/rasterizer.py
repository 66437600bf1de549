"""Python facade of the MI355X splat rasterizer.

Mirrors the interface of the pip dependency SplatFields imports at reference
gaussian_renderer/__init__.py:14 (``diff_gaussian_rasterization``, pinned in reference
README.md:28): ``GaussianRasterizationSettings`` (the 12 keyword fields built at
gaussian_renderer/__init__.py:59-72) and ``GaussianRasterizer`` (an ``nn.Module`` whose
``forward`` is called at gaussian_renderer/__init__.py:94-102 / :106-114 and returns
``(color[3,H,W], radii[N] int32, depth[1,H,W])``), with the same argument names, the same
"exactly one of" validation messages and gradients for
``means3D, means2D, shs | colors_precomp, opacities, scales, rotations | cov3D_precomp``.

Everything below the facade goes through the C ABI of ``libsplatraster.so``
(include/splatraster.h) with raw device pointers; PyTorch only owns memory and streams.
There is no CPU path: tensors must live on a HIP device and the library must be built.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
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
    prefiltered: bool
    debug: bool


LAST_INSTANCES = 0  # tile-splat instances of the most recent forward (diagnostics / bench)

# Depth gradient switch.  The depth image is differentiable here: dL/ddepth reaches opacity, conic,
# 2D mean and (through view-space z) means3D, which is what the depth losses of reference train.py:217-229 need and what the
# pinned fork's name ("depth-diff-gaussian-rasterization") promises.  Its CUDA source is not available to confirm that its
# backward consumes grad_depth, so the behaviour of a rasterizer with a non-differentiable depth output can be selected:
# set_depth_gradient(False), or SPLATRASTER_DEPTH_GRAD=0 in the environment, drops dL/ddepth in the backward.
_DEPTH_GRADIENT = os.environ.get("SPLATRASTER_DEPTH_GRAD", "1") not in ("0", "false", "False", "off")


def set_depth_gradient(enabled: bool) -> bool:
    """Propagate dL/ddepth through the rasterizer (default) or drop it.  Returns the previous setting."""
    global _DEPTH_GRADIENT
    prev, _DEPTH_GRADIENT = _DEPTH_GRADIENT, bool(enabled)
    return prev


def depth_gradient_enabled() -> bool:
    return _DEPTH_GRADIENT

# ---- host-synchronisation-free forward (include/splatraster.h: sr_forward_async) ------------------------------------------
# sr_forward waits, in the middle of every forward, for the instance count of the view ([EXT] does the same: its num_rendered
# read-back).  What the host needs the figure for -- the size of the binning buffer and which sort
# classes to launch -- is known from the LAST render of the same camera with the same number of splats, so that forward can be
# launched without waiting (both figures with 25 % headroom; the kernels exit on the device when a guess was too small) and
# the wait moves to where the figures are needed: `resolve_pending()`, or the backward at the latest.
# A wrong guess (the same view grew by more than 25 % since its last render) leaves that forward WITHOUT a result: the forward
# blend fills colour, depth and alpha with NaN on the device (csrc/render.hip, its overflow exit) -- nothing computed from them
# can pass for a value -- and redeeming the ticket raises RasterizerOverflow; the capacities have been corrected by then, so
# rendering again succeeds.
#
# WHO MAY USE IT.  Only a caller that redeems the tickets before it consumes the outputs can use the asynchronous launch
# safely, so it is OFF for the plain drop-in facade (`GaussianRasterizer.forward / forward_ex` called from an unmodified
# training loop: the reference also renders with gradients enabled and never back-propagates, train.py:379-387, where nothing
# would ever redeem a ticket) and ON inside the step functions that do redeem and re-render transparently
# (splatfields_amd/view_parallel.py `_checked`, bench.py's step: `with async_forward(): ...; resolve_pending()`).
#   SPLATRASTER_ASYNC=1 / set_async_forward(True): on for every forward of cameras seen before -- the caller takes over the duty
#       to call `resolve_pending()` (and to re-render on RasterizerOverflow) before using the outputs;
#   SPLATRASTER_ASYNC=0 / set_async_forward(False): off everywhere, also inside the step functions;
#   unset / set_async_forward(None): the default above.
# First renders of a camera (or of a new splat count: after densification) and renders under no_grad always wait.
_ASYNC = {"1": True, "true": True, "True": True, "on": True, "0": False, "false": False, "False": False, "off": False}.get(
    os.environ.get("SPLATRASTER_ASYNC", ""), None)
_LIST_HEADROOM = 1.25


class RasterizerOverflow(RuntimeError):
    """An asynchronously launched forward (sr_forward_async) met more tile-splat instances, or a longer tile list, than the
    last render of that view had promised: it has no result (its outputs are NaN).  The estimates have been corrected; render
    again."""


def set_async_forward(enabled) -> "Optional[bool]":
    """True: launch forwards of cameras seen before without waiting for their instance count, everywhere (the caller redeems the
    tickets: `resolve_pending()`); False: wait in every forward, also inside the step functions; None: the default (off for the
    plain facade, on inside the step functions that redeem their tickets).  Returns the previous setting."""
    global _ASYNC
    prev, _ASYNC = _ASYNC, (None if enabled is None else bool(enabled))
    return prev


class async_forward:
    """`with async_forward():` -- the forwards launched by this host thread inside the block may skip the host wait.  For code
    that redeems the tickets itself (`resolve_pending()` before the outputs are used, re-render on RasterizerOverflow).  An
    explicit set_async_forward(False) / SPLATRASTER_ASYNC=0 still wins."""

    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = getattr(_TLS, "async_scope", None)
        _TLS.async_scope = self.enabled
        return self

    def __exit__(self, *exc):
        _TLS.async_scope = self.prev
        return False


def async_forward_enabled() -> bool:
    """Would a forward launched by this host thread right now take the asynchronous path (for a camera it has seen)?"""
    if _ASYNC is not None:
        return _ASYNC
    return bool(getattr(_TLS, "async_scope", None))


def host_sync_counters(reset: bool = False) -> dict:
    """Process-wide counters of the library (sr_debug_counters): how many forwards waited on the host, how many did not."""
    out = (C.c_longlong * 4)()
    _lib.check(_lib.load().sr_debug_counters(out, 1 if reset else 0))
    return {"forward_host_waits": int(out[0]), "async_forwards": int(out[1]), "tickets_waited_for": int(out[2]),
            "tickets_created": int(out[3])}


_BWD_KERNELS = {None: 0, "auto": 0, "wave": 1, "quads": 2, "mfma": 2}   # "mfma": the name rounds 2-3 gave the entry-per-lane kernel


def set_backward_kernel(which) -> str:
    """Pin the backward blend kernel (A/B measurements, the test that compares the two): None / "auto" = the product choice
    (entry-per-lane over quad buckets at every footprint since round 6), "wave" = pixel-per-lane (round 1's kernel, kept as a
    second implementation of the same gradient slots), "quads" = entry-per-lane.  Returns the previous setting."""
    prev = _lib.load().sr_set_backward_kernel(_BWD_KERNELS[which])
    return {0: "auto", 1: "wave", 2: "quads"}[prev]


# Per-device estimate of the instance count used to size the binning buffer BEFORE the count is known, so that
# the forward never drains the GPU pipeline (include/splatraster.h: sr_forward).  Grows on demand.
_CAPACITY = {}
_INSTANCES_PER_SPLAT = {}
_CAPACITY_HEADROOM = 1.25


def _round_capacity(instances: int) -> int:
    """Capacity for `instances` tile-splat instances: 25 % headroom, rounded UP to eight steps per power of two.  The sizes of
    the binning buffer and of the backward scratch follow the capacity, and a training loop renders a different view -- a
    slightly different instance count -- every iteration: with the capacity tracking the running maximum exactly, every new
    maximum changed both allocation sizes, and the caching allocator answered with two fresh hipMallocs of ~100 MB (tens of
    milliseconds of host time each; seen as one 80 ms step in a 20-step run).  Quantised, the sizes change only when the
    count grows by more than a step (<= 12.5 %)."""
    c = max(int(instances * _CAPACITY_HEADROOM) + 1024, 1 << 16)
    step = 1 << (c.bit_length() - 4)
    return (c + step - 1) // step * step
# The library takes concurrent calls from host threads that render on their own streams (include/splatraster.h); these two
# module-level estimates are the only state the facade shares between them: updated under a lock (read-modify-write of a max).
import threading
import weakref
_CAPACITY_LOCK = threading.Lock()
_TLS = threading.local()   # .pending: this host thread's forwards whose ticket has not been redeemed yet
_PENDING_LOCK = threading.Lock()   # a thread's pending list is appended to by that thread and pruned by whoever redeems a ticket
                                   # (usually autograd's device thread): every mutation happens under this lock


def _record_view(view, n: int, instances: int, longest: int) -> None:
    """what the next render of this camera with this splat count may assume (kept on the cached view pack)"""
    if longest >= 0:
        if len(view.seen) > 64:   # densification changes the count thousands of times in a long run
            view.seen.clear()
        view.seen[n] = (int(instances), int(longest))


class _Pending:
    """One forward launched by sr_forward_async: its ticket and what it was promised."""

    def __init__(self, lib, ticket, capacity: int, covered: int, view, n: int, key, rkey):
        self.lib, self.ticket, self.capacity, self.covered = lib, ticket, int(capacity), int(covered)
        self.view, self.n, self.key, self.rkey = view, n, key, rkey
        self.instances = None
        self.error = None
        # weak: a forward whose outputs are dropped without a backward (evaluation code that forgot no_grad) must not pin its
        # ticket -- the autograd ctx owns this object, and __del__ hands the ticket back
        lst = getattr(_TLS, "pending", None)
        if lst is None:
            lst = _TLS.pending = []
        with _PENDING_LOCK:
            if len(lst) > 64:
                lst[:] = [r for r in lst if r() is not None]
            lst.append(weakref.ref(self))
        self._list = lst   # the launching thread's list: the ticket is usually redeemed on autograd's device thread

    def resolve(self) -> int:
        """Redeems the ticket (waits for stage 1 of that forward if it is still running); returns the instance count or raises
        RasterizerOverflow.  Idempotent."""
        if self.ticket is not None:
            ticket, self.ticket = self.ticket, None
            inst, longest = C.c_longlong(0), C.c_longlong(0)
            rc = self.lib.sr_ticket_wait(ticket, C.byref(inst), C.byref(longest))
            lst = self._list
            with _PENDING_LOCK:
                lst[:] = [r for r in lst if r() is not None and r() is not self]
            _lib.check(rc)
            self.instances = int(inst.value)
            instances, longest = int(inst.value), int(longest.value)
            with _CAPACITY_LOCK:
                _CAPACITY[self.key] = max(_CAPACITY.get(self.key, 0), _round_capacity(instances))
                _INSTANCES_PER_SPLAT[self.rkey] = max(_INSTANCES_PER_SPLAT.get(self.rkey, 0.0), instances / max(self.n, 1))
            _record_view(self.view, self.n, instances, longest)
            global LAST_INSTANCES
            LAST_INSTANCES = instances
            if instances > self.capacity or longest > max(self.covered, 2048):
                self.error = RasterizerOverflow(
                    f"the forward of this view was launched without waiting for its instance count (sr_forward_async) for at most "
                    f"{self.capacity} tile-splat instances and tile lists of up to {max(self.covered, 2048)} entries, but the view has "
                    f"{instances} instances and a list of {longest}: it has no result (its outputs are NaN).  Nothing has been applied "
                    f"and the estimates are corrected: re-run the step (or splatfields_amd.rasterizer.set_async_forward(False))")
        if self.error is not None:
            raise self.error
        return self.instances

    def __del__(self):   # a forward whose outputs were dropped without a backward: hand the ticket back
        t, self.ticket = getattr(self, "ticket", None), None
        if t is not None:
            try:
                self.lib.sr_ticket_release(t)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass


def resolve_pending() -> None:
    """Redeems the tickets of every forward this host thread launched asynchronously and has not checked yet (each waits for
    stage 1 of its forward only).  Raises RasterizerOverflow if one of them overflowed -- after all have been redeemed, so the
    caller can simply re-render."""
    err = None
    with _PENDING_LOCK:
        refs = list(getattr(_TLS, "pending", None) or [])
    for ref in refs:
        p = ref()
        if p is None:
            continue
        try:
            p.resolve()
        except RasterizerOverflow as e:
            err = err or e
    if err is not None:
        raise err


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    """fp32, on `device`, contiguous -- the camera tensors arrive strided (scene/cameras.py:68,74)."""
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _as_input(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    """The tensor itself when the kernels can read it in place (fp32, contiguous, on `device`: the normal case, no
    Python or GPU work), otherwise a converted copy."""
    if t is None:
        return None
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous():
        return t
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _opt(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if (t is None or t.numel() == 0) else t


class _ViewPack:
    """Keeps the contiguous camera tensors alive next to the C struct that points at them."""

    _cache: "dict" = {}
    _CACHE_MAX = 256

    @classmethod
    def get(cls, rs: GaussianRasterizationSettings, device, sh_coeffs: int) -> "_ViewPack":
        """Training renders the same cameras over and over (reference train.py:158-169), each time through a fresh
        settings tuple that refers to the same camera tensors: the packed copy (three small strided-copy kernels and the
        C struct) is kept per set of CAMERA tensors.  An entry holds its source tensors (so their ids cannot be
        recycled) and is dropped when one of them was modified in place (`_version`).
        The background is not part of the key: callers build it per call (the reference's mask pass does,
        gaussian_renderer/__init__.py:81: `bg_color*0.0`), and a camera stays the same camera under another background -- what its
        last render promised (`seen`, sr_forward_async) carries over.  A call whose background tensor differs from the cached
        pack's gets a light copy of the pack that shares the camera tensors and `seen`."""
        src = (rs.viewmatrix, rs.projmatrix, rs.campos)
        key = (id(src[0]), id(src[1]), id(src[2]), int(rs.image_height), int(rs.image_width), float(rs.tanfovx),
               float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree), int(sh_coeffs), bool(rs.prefiltered),
               bool(rs.debug), device)
        versions = (src[0]._version, src[1]._version, src[2]._version)
        hit = cls._cache.get(key)
        if hit is not None and hit[1] == versions and all(a is b for a, b in zip(hit[2], src)):
            pack = hit[0]
            if pack.bg_src is rs.bg and pack.bg_version == rs.bg._version:
                return pack
            return pack.with_bg(rs.bg, device)
        pack = cls(rs, device, sh_coeffs)
        while len(cls._cache) >= cls._CACHE_MAX:
            cls._cache.pop(next(iter(cls._cache)))   # oldest entry first (dicts keep insertion order)
        cls._cache[key] = (pack, versions, src)
        return pack

    def __init__(self, rs: GaussianRasterizationSettings, device, sh_coeffs: int):
        self.viewmatrix = _f32c(rs.viewmatrix, device).reshape(-1)
        self.projmatrix = _f32c(rs.projmatrix, device).reshape(-1)
        self.campos = _f32c(rs.campos, device).reshape(-1)
        if self.viewmatrix.numel() != 16 or self.projmatrix.numel() != 16:
            raise RuntimeError("viewmatrix and projmatrix must hold 16 elements ([4,4] or [1,4,4])")
        if self.campos.numel() != 3:
            raise RuntimeError("campos and bg must hold 3 elements")
        self.seen = {}   # splat count -> (instances, longest tile list) of the last render of this camera (sr_forward_async)
        self._fields = (int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                        float(rs.scale_modifier), int(rs.sh_degree), int(sh_coeffs), int(bool(rs.prefiltered)),
                        int(bool(rs.debug)))
        self._set_bg(rs.bg, device)

    def _set_bg(self, bg: torch.Tensor, device) -> None:
        self.bg_src, self.bg_version = bg, bg._version
        self.bg = _f32c(bg, device).reshape(-1)   # the tensor itself when it is fp32, contiguous and on the device
        if self.bg.numel() != 3:
            raise RuntimeError("campos and bg must hold 3 elements")
        self.struct = _lib.SrView(*self._fields, self.viewmatrix.data_ptr(), self.projmatrix.data_ptr(),
                                  self.campos.data_ptr(), self.bg.data_ptr())

    def with_bg(self, bg: torch.Tensor, device) -> "_ViewPack":
        other = object.__new__(_ViewPack)
        other.viewmatrix, other.projmatrix, other.campos = self.viewmatrix, self.projmatrix, self.campos
        other.seen, other._fields = self.seen, self._fields
        other._set_bg(bg, device)
        return other


def _splats_struct(n, means3D, opacities, scales, rotations, cov3D, shs, colors, raw_params: int = 0, shs_rest=None) -> _lib.SrSplats:
    g = lambda t: None if t is None else t.data_ptr()
    return _lib.SrSplats(int(n), g(means3D), g(opacities), g(scales), g(rotations), g(cov3D), g(shs), g(colors), int(raw_params),
                         g(shs_rest))


def _aligned16(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return t if (t is None or t.data_ptr() % 16 == 0) else t.clone()


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd bridge; the counterpart of [EXT] ``_RasterizeGaussians``."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings: GaussianRasterizationSettings, color_grad_sink=None, raw_params: int = 0, sh_rest=None,
                slice_hook=None, grad_mode: bool = True):
        lib = _lib.load()
        if not means3D.is_cuda:
            raise RuntimeError("splatfields_amd rasterizer has no CPU path: tensors must be on a HIP ('cuda') device")
        dev = means3D.device
        if means3D.dim() != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        n = means3D.shape[0]
        H, W = int(raster_settings.image_height), int(raster_settings.image_width)
        f = lambda t: _as_input(t, dev)
        means3D_c, opac_c = f(means3D), f(opacities).reshape(-1)
        sh_c, col_c, sc_c, rot_c, cov_c = f(_opt(sh)), f(_opt(colors_precomp)), f(_opt(scales)), f(_opt(rotations)), f(_opt(cov3Ds_precomp))
        if opac_c.numel() != n:
            raise RuntimeError("opacities must have dimensions (num_points, 1)")
        for name, t, last in (("shs", sh_c, None), ("colors_precomp", col_c, 3), ("scales", sc_c, 3),
                              ("rotations", rot_c, 4), ("cov3D_precomp", cov_c, 6)):
            if t is not None and (t.shape[0] != n or (last is not None and t.shape[-1] != last)):
                raise RuntimeError(f"{name} has an unexpected shape {tuple(t.shape)} for {n} points")
        if sh_c is not None and (sh_c.dim() != 3 or sh_c.shape[2] != 3):
            raise RuntimeError("shs must have dimensions (num_points, K, 3)")
        sh_coeffs = 0 if sh_c is None else int(sh_c.shape[1])
        rest_c = None
        if _opt(sh_rest) is not None:
            # the reference's two SH parameters, `_features_dc` [N,1,3] + `_features_rest` [N,15,3], without concatenating them
            rest_c = _aligned16(f(sh_rest))
            sh_c = _aligned16(sh_c)
            if sh_c is None or sh_c.shape[1] != 1 or rest_c.dim() != 3 or tuple(rest_c.shape) != (n, 15, 3):
                raise RuntimeError("with shs_rest, shs must be [N,1,3] (dc) and shs_rest [N,15,3]")
            sh_coeffs = 16

        if slice_hook is not None:
            # checked here, not in the backward: an exception raised inside the autograd thread of ONE rank would leave the
            # other ranks waiting in their collectives
            if cov_c is not None:
                raise ValueError("slice_hook requires scales / rotations inputs (cov3D_precomp is not supported in the sliced backward)")
            if rest_c is not None:
                raise ValueError("slice_hook requires the concatenated shs tensor (shs_rest is not supported in the sliced backward)")
            if sh_c is not None and color_grad_sink is None:
                raise ValueError("slice_hook requires color_grad_sink on the SH path (the sliced backward hands over colour gradients)")
        view = _ViewPack.get(raster_settings, dev, sh_coeffs)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(n, dtype=torch.int32, device=dev)  # k_preprocess writes every element
        ctx.raster_settings = raster_settings
        ctx.color_grad_sink = color_grad_sink
        ctx.slice_hook = slice_hook
        # nothing to differentiate (rendering / evaluation under no_grad): the forward skips what only the backward reads
        # (`needs_input_grad` reflects the inputs' requires_grad flags even under torch.no_grad(): the caller's grad mode --
        # autograd switches it off inside this function -- arrives as an argument)
        if not grad_mode or (not any(ctx.needs_input_grad[:8]) and not (len(ctx.needs_input_grad) > 11 and ctx.needs_input_grad[11])):
            raw_params = int(raw_params) | _lib.SR_FORWARD_ONLY
        ctx.raw_params = int(raw_params)
        ctx.sh_coeffs = sh_coeffs
        ctx.n = n
        ctx.opac_shape = tuple(opacities.shape)
        ctx.in_dtypes = [None if t is None else t.dtype for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)]
        ctx.rest_dtype = None if rest_c is None else sh_rest.dtype
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii)
        if n == 0:
            color[:] = view.bg.reshape(3, 1, 1)
            depth.zero_()
            alpha.zero_()
            ctx.instances = 0
            ctx.pending = None
            ctx.save_for_backward()
            return color, radii, depth, alpha

        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            splats = _splats_struct(n, means3D_c, opac_c, sc_c, rot_c, cov_c, sh_c, col_c, raw_params, rest_c)
            geom = torch.empty(lib.sr_geom_bytes(n, H, W), dtype=torch.uint8, device=dev)
            image = torch.empty(lib.sr_image_bytes(H, W), dtype=torch.uint8, device=dev)
            inst = C.c_longlong(0)
            key = (dev.index, n, H, W)
            # known size: what it needed before; new size (the cloud was densified / pruned): the instances-per-splat ratio
            # this image size has shown so far, so that the first forward after a densification step does not fall into the
            # re-run path
            with _CAPACITY_LOCK:
                ratio = _INSTANCES_PER_SPLAT.get((dev.index, H, W))
                capacity = _CAPACITY.get(key) or (max(4 * n, 1 << 16) if ratio is None else _round_capacity(int(ratio * n)))
            rkey = (dev.index, H, W)
            before = view.seen.get(n) if (async_forward_enabled() and not (int(raw_params) & _lib.SR_FORWARD_ONLY)) else None
            if before is not None and _round_capacity(before[0]) <= capacity:
                # this camera was rendered with this splat count before: launch without waiting (see _ASYNC above)
                binning = torch.empty(lib.sr_binning_bytes(capacity, H, W), dtype=torch.uint8, device=dev)
                hint = int(before[1] * _LIST_HEADROOM) + 1
                covered = 2048 if hint <= 2048 else 4096 if hint <= 4096 else 8192 if hint <= 8192 else 1 << 62
                ticket = C.c_void_p()
                _lib.check(lib.sr_forward_async(C.byref(view.struct), C.byref(splats), _ptr(geom), _ptr(radii), _ptr(binning),
                                                capacity, hint, int(before[1]), _ptr(image), _ptr(color), _ptr(depth), _ptr(alpha),
                                                C.byref(ticket), stream))
                ctx.pending = _Pending(lib, ticket, capacity, covered, view, n, key, rkey)
                ctx.instances = None
                ctx.capacity = capacity
                ctx.view_pack = view
                ctx.save_for_backward(means3D_c, opac_c, sc_c, rot_c, cov_c, sh_c, col_c, radii, geom, binning, image, rest_c)
                return color, radii, depth, alpha
            binning = torch.empty(lib.sr_binning_bytes(capacity, H, W), dtype=torch.uint8, device=dev)
            status = lib.sr_forward(C.byref(view.struct), C.byref(splats), _ptr(geom), _ptr(radii), _ptr(binning),
                                    capacity, _ptr(image), _ptr(color), _ptr(depth), _ptr(alpha), C.byref(inst), stream)
            instances = int(inst.value)
            _record_view(view, n, instances, int(lib.sr_last_longest_list()))
            if status == _lib.SR_NEED_CAPACITY:
                # first call for this size, or the cloud grew: re-run stage 2 with a buffer that fits
                capacity = _round_capacity(instances)
                binning = torch.empty(lib.sr_binning_bytes(capacity, H, W), dtype=torch.uint8, device=dev)
                _lib.check(lib.sr_forward_render(C.byref(view.struct), C.byref(splats), _ptr(geom), _ptr(binning),
                                                 capacity, _ptr(image), _ptr(color), _ptr(depth), _ptr(alpha), stream))
            else:
                _lib.check(status)
            with _CAPACITY_LOCK:
                _CAPACITY[key] = max(_CAPACITY.get(key, 0), _round_capacity(instances))
                if len(_CAPACITY) > 4096:  # a long training run changes the splat count thousands of times
                    _CAPACITY.clear()
                _INSTANCES_PER_SPLAT[rkey] = max(_INSTANCES_PER_SPLAT.get(rkey, 0.0), instances / max(n, 1))
        global LAST_INSTANCES
        LAST_INSTANCES = instances
        ctx.pending = None
        ctx.instances = instances
        ctx.capacity = capacity
        ctx.view_pack = view
        ctx.save_for_backward(means3D_c, opac_c, sc_c, rot_c, cov_c, sh_c, col_c, radii, geom, binning, image, rest_c)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        n = ctx.n
        if n == 0:
            return (None,) * 14
        lib = _lib.load()
        if ctx.pending is not None:
            # the forward did not wait for its instance count: redeem its ticket now (stage 1 of that forward is long over;
            # raises RasterizerOverflow if the capacity it was launched with did not hold -- before any gradient is produced)
            ctx.instances = ctx.pending.resolve()
        means3D, opac, sc, rot, cov, sh, col, radii, geom, binning, image, sh_rest = ctx.saved_tensors
        dev = means3D.device
        H, W = int(rs.image_height), int(rs.image_width)
        g = lambda t: _as_input(t, dev)
        grad_color = g(grad_color)
        if grad_color is None:
            grad_color = torch.zeros(3, H, W, dtype=torch.float32, device=dev)
        grad_depth, grad_alpha = g(grad_depth if _DEPTH_GRADIENT else None), g(grad_alpha)
        view = ctx.view_pack  # camera tensors were made contiguous in forward
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        hook = ctx.slice_hook
        sink = ctx.color_grad_sink if sh is not None else None
        if hook is not None:
            # sliced mode (view-parallel exchange, splatfields_amd/view_parallel.py): the gradients are written into the HOOK's
            # buffers, slice by slice, and the hook is called behind every slice's launch -- it starts that slice's collectives
            # while the next slice is computed.  Autograd gets no gradient for these inputs (the hook's owner sets `.grad`).
            assert cov is None and sh_rest is None and (sh is None or sink is not None)   # validated in forward
            hb = hook.buffers(n)
            d_means3D, d_sc, d_rot, d_opac, d_col = hb["means3D"], hb["scales"], hb["rotations"], hb["opacities"], hb["colors"]
            d_means2D, d_cov, d_sh, d_rest = new(n, 3), None, None, None
        else:
            d_means3D, d_means2D, d_opac = new(n, 3), new(n, 3), new(n, 1)
            d_sc = new(n, 3) if sc is not None else None
            d_rot = new(n, 4) if rot is not None else None
            d_cov = new(n, 6) if cov is not None else None
            d_sh = new(*sh.shape) if (sh is not None and sink is None) else None
            d_col = new(n, 3) if (col is not None or sink is not None) else None
            d_rest = new(*sh_rest.shape) if (sh_rest is not None and d_sh is not None) else None
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            splats = _splats_struct(n, means3D, opac, sc, rot, cov, sh, col, ctx.raw_params, sh_rest)
            scratch = torch.empty(lib.sr_backward_scratch_bytes(ctx.capacity), dtype=torch.uint8, device=dev)
            p = lambda t: None if t is None else t.data_ptr()
            grads = _lib.SrGrads(p(d_means3D), p(d_means2D), p(d_opac), p(d_sc), p(d_rot), p(d_cov), p(d_sh), p(d_col), p(d_rest))
            if hook is None:
                _lib.check(lib.sr_backward(C.byref(view.struct), C.byref(splats), _ptr(geom), _ptr(binning), ctx.capacity, ctx.instances,
                                           _ptr(image), _ptr(radii), _ptr(grad_color), _ptr(grad_depth), _ptr(grad_alpha),
                                           _ptr(scratch), C.byref(grads), stream))
            else:
                _lib.check(lib.sr_backward_blend(C.byref(view.struct), C.byref(splats), _ptr(geom), _ptr(binning), ctx.capacity,
                                                 ctx.instances, _ptr(image), _ptr(grad_color), _ptr(grad_depth), _ptr(grad_alpha),
                                                 _ptr(scratch), stream))
                for j, (lo, hi) in enumerate(slice_ranges(n, hook.slices)):
                    _lib.check(lib.sr_backward_splats(C.byref(view.struct), C.byref(splats), _ptr(geom), _ptr(binning), ctx.capacity,
                                                      ctx.instances, _ptr(image), _ptr(radii), _ptr(scratch), C.byref(grads), lo, hi - lo, stream))
                    hook.on_slice(j, lo, hi)
        if hook is not None:
            return (None, d_means2D) + (None,) * 12
        # order of the forward inputs: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D, settings
        if sink is not None:
            sink.append(d_col)  # clamp-masked dL/dcolour of this view; dL/dsh is rebuilt from all views by the caller
            d_col = None
        grads = [d_means3D, d_means2D, d_sh, d_col, d_opac.reshape(ctx.opac_shape), d_sc, d_rot, d_cov]
        # the kernels compute in fp32; hand each gradient back in its input's dtype (fp64 / fp16 callers)
        grads = [g_ if (g_ is None or dt is None or g_.dtype == dt) else g_.to(dt) for g_, dt in zip(grads, ctx.in_dtypes)]
        if d_rest is not None and ctx.rest_dtype is not None and d_rest.dtype != ctx.rest_dtype:
            d_rest = d_rest.to(ctx.rest_dtype)
        return (*grads, None, None, None, d_rest, None, None)


def slice_ranges(n: int, slices: int) -> list:
    """[lo, hi) ranges of the sliced backward: `slices` pieces of equal size rounded up to a multiple of 256 splats (the
    granule of the per-splat kernels); fewer pieces when the cloud is small."""
    per = -(-n // max(int(slices), 1))          # ceil(n / slices)
    per = max(256, -(-per // 256) * 256)        # ... rounded up to the granule
    return [(lo, min(lo + per, n)) for lo in range(0, n, per)]


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, color_grad_sink=None, raw_params: int = 0, sh_rest=None, slice_hook=None):
    """Returns (color, radii, depth, alpha).  ``alpha`` (= 1 - final transmittance) is the fused equivalent of
    the reference's second rasterization with white colours on a black background
    (gaussian_renderer/__init__.py:104-115)."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, color_grad_sink, raw_params, sh_rest, slice_hook,
                                     torch.is_grad_enabled())


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Near-plane visibility mask ([EXT] ``GaussianRasterizer.markVisible``)."""
        lib = _lib.load()
        if not positions.is_cuda:
            raise RuntimeError("splatfields_amd rasterizer has no CPU path: tensors must be on a HIP ('cuda') device")
        with torch.no_grad():
            rs = self.raster_settings
            dev = positions.device
            pos = positions.detach().to(torch.float32).contiguous()
            vm, pj = _f32c(rs.viewmatrix, dev), _f32c(rs.projmatrix, dev)
            present = torch.zeros(pos.shape[0], dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.sr_mark_visible(pos.shape[0], _ptr(pos), _ptr(vm), _ptr(pj), _ptr(present), _stream_ptr(dev)))
        return present.bool()

    def _check(self, shs, colors_precomp, scales, rotations, cov3D_precomp):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

    def forward_ex(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                   cov3D_precomp=None, color_grad_sink=None, shs_rest=None, slice_hook=None):
        """Same as ``forward`` plus the fused alpha image: (color, radii, depth, alpha).

        ``color_grad_sink`` (a list, SH path only): the backward appends the clamp-masked dL/dcolour [N,3] of this view to
        it and returns no gradient for ``shs`` -- used by the view-parallel step, which exchanges colour gradients and
        rebuilds the SH gradient of all views locally (splatfields_amd/view_parallel.py).

        ``slice_hook`` (view-parallel step only): an object with ``slices`` (int), ``buffers(n) -> dict`` of caller-owned gradient
        tensors (``means3D, scales, rotations, opacities, colors``) and ``on_slice(j, lo, hi)``: the backward writes the
        per-splat gradients of rows [lo, hi) into those buffers slice by slice and calls the hook behind every slice's launch,
        so that the exchange of finished slices overlaps the rest of the backward; autograd receives only ``means2D``'s gradient.

        ``shs_rest``: pass the reference's two SH parameters as they are stored, ``shs=_features_dc`` [N,1,3] and
        ``shs_rest=_features_rest`` [N,15,3] (scene/gaussian_model.py:40-41), instead of ``get_features`` -- the per-iteration
        ``torch.cat`` (:79-82, 192 B/splat copied forward and split again in backward) disappears; each gets its gradient."""
        self._check(shs, colors_precomp, scales, rotations, cov3D_precomp)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings, color_grad_sink, 0, shs_rest, slice_hook)

    def forward_raw(self, means3D, means2D, opacity_logits, shs=None, colors_precomp=None, log_scales=None, quaternions=None,
                    color_grad_sink=None, shs_rest=None):
        """``forward_ex`` on the optimiser's RAW parameters: ``_opacity`` (logits), ``_scaling`` (log-scales, [N,3]) and
        ``_rotation`` (unnormalised quaternions) as ``GaussianModel`` stores them (reference scene/gaussian_model.py:64-86).
        ``sigmoid`` / ``exp`` / ``normalize`` run inside the preprocess kernels and their derivatives inside the backward,
        so the six element-wise kernels (and their six backward kernels) the accessors launch per iteration disappear; the
        gradients returned are w.r.t. the raw parameters.  Returns (color, radii, depth, alpha)."""
        self._check(shs, colors_precomp, log_scales, quaternions, None)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacity_logits, log_scales, quaternions, None,
                                   self.raster_settings, color_grad_sink,
                                   _lib.SR_RAW_SCALES | _lib.SR_RAW_OPACITY | _lib.SR_RAW_ROTATIONS, shs_rest)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        served = _serve_mask_call(self.raster_settings, means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                                  cov3D_precomp)
        if served is not None:
            return served
        color, radii, depth, alpha = self.forward_ex(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                                                     cov3D_precomp)
        _remember_call(self.raster_settings, means3D, means2D, opacities, scales, rotations, cov3D_precomp, radii, depth, alpha)
        return color, radii, depth


# ---- the reference's mask pass served from the first pass (for the UNMODIFIED render()) ---------
# gaussian_renderer/__init__.py:104-115 rasterizes every view a second time -- same splats, same camera, white colours on a
# black background -- and keeps channel 0: that image is 1 - T_final, the alpha the first pass already produced.  A caller that
# edits render() takes `forward_ex`'s fourth output (splatfields_amd/render.py); for the zero-change drop-in the facade
# recognises the second call instead:
#   * host-side, by identity: the same means3D / means2D / opacities / scales / rotations tensor objects (and `_version`s) and the
#     same camera tensors and scalar settings as the forward this host thread ran immediately before, shs=None, and a
#     colors_precomp tensor [N,3] that nothing differentiates;
#   * by value: colours all 1 and background all 0.  The first such call of the process is checked on the host (one
#     synchronisation, once): if it is something else -- a second pass with per-splat feature colours, say -- the shortcut switches
#     itself off and the call is rasterized in full, as every later one.  Once the pattern is confirmed the check stays on the
#     device (a few elementwise kernels whose verdict never travels to the host): the served image is where(verdict, alpha, NaN),
#     so a later call that breaks the pattern gets NaN, not a wrong mask, and the call after it finds the (by then finished)
#     verdict and switches the shortcut off for the process.
# The served tensor is a view of the first pass's alpha output: the mask loss back-propagates into that node's grad_alpha, and
# the means2D gradient (densification statistics, train.py:280-286) receives both contributions in one backward -- the sum the
# reference accumulates over its two backward passes.  SPLATRASTER_MASK_SHORTCUT=0 switches it off (two full passes).
_MASK_SHORTCUT = os.environ.get("SPLATRASTER_MASK_SHORTCUT", "1") not in ("0", "false", "False", "off")
MASK_CALLS_SERVED = 0   # diagnostics / tests: mask passes answered without a rasterization
_MASK_CONFIRMED = False  # the pattern has been seen once with white colours on a black background (checked on the host that once)


def set_mask_shortcut(enabled: bool) -> bool:
    global _MASK_SHORTCUT
    prev, _MASK_SHORTCUT = _MASK_SHORTCUT, bool(enabled)
    return prev


def _camera_signature(rs: GaussianRasterizationSettings):
    return (int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier),
            bool(rs.prefiltered), bool(rs.debug))


def _remember_call(rs, means3D, means2D, opacities, scales, rotations, cov3D_precomp, radii, depth, alpha) -> None:
    if not _MASK_SHORTCUT or cov3D_precomp is not None or scales is None or rotations is None:
        _TLS.last_call = None
        return
    tensors = (means3D, means2D, opacities, scales, rotations, rs.viewmatrix, rs.projmatrix, rs.campos)
    _TLS.last_call = (tensors, tuple(t._version for t in tensors), _camera_signature(rs), (radii, depth, alpha))


def _serve_mask_call(rs, means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp):
    last = getattr(_TLS, "last_call", None)
    _TLS.last_call = None   # one shot: the pattern is "the call right after"
    if last is None or not _MASK_SHORTCUT:
        return None
    if shs is not None or cov3D_precomp is not None or not isinstance(colors_precomp, torch.Tensor):
        return None
    tensors, versions, sig, (radii, depth, alpha) = last
    now = (means3D, means2D, opacities, scales, rotations, rs.viewmatrix, rs.projmatrix, rs.campos)
    if any(a is not b for a, b in zip(tensors, now)) or versions != tuple(t._version for t in now) or sig != _camera_signature(rs):
        return None
    n = means3D.shape[0]
    if (tuple(colors_precomp.shape) != (n, 3) or colors_precomp.requires_grad or colors_precomp.device != means3D.device
            or not isinstance(rs.bg, torch.Tensor) or rs.bg.numel() != 3 or rs.bg.requires_grad):
        return None
    _check_mask_verdicts()
    if not _MASK_SHORTCUT:
        return None
    global _MASK_CONFIRMED
    with torch.no_grad():
        ok = (colors_precomp == 1).all() & (rs.bg.to(colors_precomp.device) == 0).all()
        if not _MASK_CONFIRMED:
            # The FIRST call of this shape in the process is checked on the host (one synchronisation, once): a caller whose second
            # call renders something else -- per-splat feature colours behind the RGB pass, say -- never meets the NaN guard; it
            # simply gets two full passes from here on.
            if not bool(ok):
                set_mask_shortcut(False)
                return None
            _MASK_CONFIRMED = True
        _note_mask_verdict(ok)
    served = torch.where(ok, alpha, torch.full((), float("nan"), dtype=alpha.dtype, device=alpha.device))
    global MASK_CALLS_SERVED
    MASK_CALLS_SERVED += 1
    return served.expand(3, -1, -1), radii, depth


def _note_mask_verdict(ok: torch.Tensor) -> None:
    """the verdict travels to pinned host memory behind the stream, without a wait; looked at by a LATER call"""
    lst = getattr(_TLS, "mask_verdicts", None)
    if lst is None:
        lst = _TLS.mask_verdicts = []
    if len(lst) >= 8:
        return   # enough in flight; every call is still guarded on the device by the where()
    host = torch.empty(1, dtype=torch.bool, pin_memory=True)
    host.copy_(ok.reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(ok.device))
    lst.append((ev, host))


def _check_mask_verdicts() -> None:
    lst = getattr(_TLS, "mask_verdicts", None)
    while lst and lst[0][0].query():
        _, host = lst.pop(0)
        if not bool(host[0]):
            import warnings
            set_mask_shortcut(False)
            warnings.warn("splatfields_amd: a rasterizer call that looked like the reference's mask pass (same splats and camera as "
                          "the call before, shs=None) did not carry white colours on a black background; it was answered with NaN. "
                          "The shortcut is now off for this process: such calls are rasterized in full.", RuntimeWarning)
