"""View-parallel training step across the GPUs of one node (new design; the reference is single-GPU).

The reference renders the views of one iteration in a Python loop over the *same* splat tensors and
averages their losses before one backward (train.py:158-169, :242, :252).  Views are independent, so
they shard: one process per GPU, splat tensors replicated, rank r renders views r, r+G, r+2G, ...;
each rank back-propagates its local mean loss, then a sum all-reduce over RCCL/xGMI
(`torch.distributed`, backend "nccl") followed by a 1/G scale reproduces the gradient of the mean
over all views.  The SH gradient (192 B/splat, the bulk of the 236 B/splat payload; 56 B/splat
with precomputed colours -- SURVEY.md §8e) is reduced in place without a packing copy; the small
geometric gradients share one packed buffer and one collective.  `sh_gather_step` avoids moving the
SH gradient at all.

Densification statistics (`viewspace_points.grad`, `radii`) are per view; this harness keeps the
reference's "last view wins" rule (train.py:178, :282, :307) on every rank for its own last view.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Sequence

import torch
import torch.distributed as dist


def shard_views(views: Sequence, rank: int, world: int) -> list:
    """Round-robin deal of this step's view list (reference train.py:158-163 builds the list)."""
    return [v for i, v in enumerate(views) if i % world == rank]


# Gradient tensors below this size are packed into one flat buffer and reduced by ONE collective: at 1 M splats the four
# geometric gradients are 4-16 MB each, where a collective's fixed cost (launch + 8-rank handshake) is a visible fraction of
# its transfer time; the SH gradient (192 MB) is reduced on its own, without a packing copy.
PACK_BELOW_BYTES = 64 << 20

# Tests only: issue the collectives even in a 1-rank group, so that a single MI355X exercises the RCCL calls
# (ReduceOp.AVG, all_gather_into_tensor, device all_to_all_single) the multi-GPU steps rely on.
FORCE_COLLECTIVES = False


def _exchange(world: int) -> bool:
    return world > 1 or (FORCE_COLLECTIVES and dist.is_initialized())


class ExchangeStats:
    """Optional instrumentation of the exchange part of a step (bench.py's N > 1 line): per step the time between issuing the
    first collective of an exchange phase and the completion of its last one (events on the current stream; ``work.wait()``
    orders it behind the collective), summed over the step's phases; the local compute that ran inside those windows (what
    the exchange is overlapped with); and the bytes every GPU puts on the wire under the ring model RCCL uses on a
    point-to-point xGMI node: all-reduce of S bytes 2 (G-1)/G S, all-gather / all-to-all of S bytes in total (G-1)/G S.
    The caller brackets a step with ``end_step()``.  Off by default (no events are created)."""
    enabled = False
    _win: list = []        # (start, end) events of the current step's exchange windows
    _inner: list = []      # (start, end) events of compute inside them
    _bytes = 0.0
    steps: list = []       # per finished step: (windows, inner, wire bytes)

    @classmethod
    def reset(cls, on: bool):
        cls.enabled, cls._win, cls._inner, cls._bytes, cls.steps = bool(on), [], [], 0.0, []

    @classmethod
    def note(cls, kind: str, nbytes: float, world: int):
        if cls.enabled and world > 0:
            f = (world - 1) / world
            cls._bytes += nbytes * (2.0 * f if kind == "all_reduce" else f)

    @classmethod
    def end_step(cls):
        if cls.enabled:
            cls.steps.append((cls._win, cls._inner, cls._bytes))
            cls._win, cls._inner, cls._bytes = [], [], 0.0

    @classmethod
    def summary(cls) -> dict:
        """Median per-step figures (ms, bytes); synchronises the recorded events."""
        def med(xs):
            xs = sorted(xs)
            return xs[len(xs) // 2] if xs else 0.0
        win = [sum(a.elapsed_time(b) for a, b in w) for w, _, _ in cls.steps]
        inner = [sum(a.elapsed_time(b) for a, b in i) for _, i, _ in cls.steps]
        return {"exchange_ms": med(win), "overlap_ms": med(inner), "wire_bytes_per_gpu": med([x for _, _, x in cls.steps]),
                "steps": len(cls.steps)}


class _Window:
    """`with _Window(dev):` = the exchange window of one step; `with _Window(dev, inner=True):` = compute inside it."""

    def __init__(self, dev, inner: bool = False):
        self.on = ExchangeStats.enabled and torch.device(dev).type == "cuda"
        self.inner = inner
        self.dev = dev

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.dev))
        return self

    def __exit__(self, *exc):
        if self.on:
            self.b.record(torch.cuda.current_stream(self.dev))
            (ExchangeStats._inner if self.inner else ExchangeStats._win).append((self.a, self.b))
        return False


def _checked(render_fn: Callable):
    """``render_fn()`` launches a forward -- without waiting for its instance count when the camera was rendered before
    (rasterizer.py: sr_forward_async) -- and returns its outputs.  The ticket is redeemed HERE, before the caller
    back-propagates: a forward whose capacity guess did not hold is re-rendered (the estimates are corrected by then), so the
    step functions never see RasterizerOverflow in the middle of an autograd pass -- and a rank never re-issues collectives."""
    from . import rasterizer as rz
    with rz.async_forward():   # the tickets are redeemed right here: this is the caller the asynchronous launch is for
        out = render_fn()
        try:
            rz.resolve_pending()
        except rz.RasterizerOverflow:
            out = render_fn()
            rz.resolve_pending()
    return out


def _nbytes(t: torch.Tensor) -> float:
    return float(t.numel() * t.element_size())


# xGMI on an 8 x MI355X node: every GPU has 7 point-to-point links of ~153 GB/s (both directions together; ~76.8 GB/s each
# way), one to each peer (/opt/skills/guides/MI355X_MICROARCH.md; the task statement's figure).  A collective that spreads
# its traffic over all peers (RCCL's direct / multi-ring algorithms on a fully connected node) can therefore move at most
# 7 x 153 = 1071 GB/s per GPU counting sent + received bytes, 537 GB/s counting the bytes a GPU SENDS (the ring model's
# `wire_bytes_per_gpu`).  Measured RCCL bus bandwidths on this class of node are lower (~300 GB/s for 10-100 MB messages).
XGMI_LINKS, XGMI_LINK_GBPS = 7, 153.0
RCCL_TYPICAL_BUS_GBPS = 300.0


def wire_bytes_per_gpu(mode: str, n_splats: int, world: int, sh_coeffs: int = 16, views_per_rank: int = 1) -> float:
    """Bytes every GPU sends per step under the ring model ExchangeStats uses (all-reduce of S bytes: 2 (G-1)/G S; all-gather /
    all-to-all of S bytes in total: (G-1)/G S), for the three exchange schemes of a step with `views_per_rank` views per rank."""
    f = (world - 1) / world if world > 0 else 0.0
    geo = 44.0 * n_splats                                    # means3D 12 + scales 12 + rotations 16 + opacity 4
    if mode == "allreduce":
        return 2.0 * f * (geo + 12.0 * sh_coeffs * n_splats)
    if mode == "gather":
        return f * (12.0 * n_splats * world * views_per_rank) + 2.0 * f * geo
    if mode == "shard":
        shard = -(-n_splats // world)
        return 2.0 * f * (12.0 * shard * world * views_per_rank) + 2.0 * f * geo
    raise ValueError(mode)


def predict_scaling(mode: str, n_splats: int, world: int, compute_ms: float, *, tail_ms: float = 0.0, slices: int = 1,
                    rebuild_ms: float = 0.0, sh_coeffs: int = 16) -> dict:
    """What the first run on a multi-GPU node should show (bench.py's `predicted` block): the exchange time of one step from the
    wire bytes and the link arithmetic above, how much of it the step hides, and the weak-scaling speed-up that follows (one view
    per GPU) -- at the link peak (the yardstick: wire bytes / (7 links x 153 GB/s)), at the one-way peak, and at a bus bandwidth
    RCCL typically reaches on such a node.

    `compute_ms`: the single-GPU step.  `tail_ms`: the last part of it after which the data of the exchange is complete -- the
    per-splat backward, which the gather scheme runs in `slices` ranges, starting every range's collectives behind it.
    `rebuild_ms`: compute the scheme adds (the local SH-gradient rebuild; a range's rebuild needs its all-gather).
    Timeline of the sliced gather step, with the wire kept busy from the first finished range on:
        end = compute + max(rebuild, exchange - tail (1 - 1/slices) + rebuild / slices)
    plain all-reduce (nothing to overlap with): end = compute + exchange;  sharded: end = compute + exchange + rebuild."""
    wire = wire_bytes_per_gpu(mode, n_splats, world, sh_coeffs)
    k = max(int(slices), 1)
    out = {"dp_mode": mode, "world": world, "wire_bytes_per_gpu": wire, "wire_bytes_per_splat_per_gpu": wire / max(n_splats, 1),
           "links": XGMI_LINKS, "link_GBps": XGMI_LINK_GBPS, "compute_ms": compute_ms, "tail_ms": tail_ms, "slices": k,
           "rebuild_ms": rebuild_ms,
           "model": "exchange_ms = wire bytes / bandwidth; gather: exposed = max(rebuild, exchange - tail (1 - 1/slices) + rebuild / slices); "
                    "allreduce: exposed = exchange; shard: exposed = exchange + rebuild; step = compute + exposed; "
                    "speed-up = world x compute / step (weak scaling, one view per GPU)"}
    for name, gbps in (("link_peak", XGMI_LINKS * XGMI_LINK_GBPS), ("one_way_peak", 0.5 * XGMI_LINKS * XGMI_LINK_GBPS),
                       ("rccl_typical", RCCL_TYPICAL_BUS_GBPS)):
        ex = wire / (gbps * 1e9) * 1e3 if world > 1 else 0.0
        if world <= 1:
            exposed = 0.0
        elif mode == "gather":
            exposed = max(rebuild_ms, ex - tail_ms * (1.0 - 1.0 / k) + rebuild_ms / k)
        elif mode == "shard":
            exposed = ex + rebuild_ms
        else:
            exposed = ex
        step = compute_ms + exposed
        out[name] = {"bandwidth_GBps": gbps, "exchange_ms": ex, "exposed_exchange_ms": exposed, "step_ms": step,
                     "speedup": world * compute_ms / step if step > 0 else None}
    out["expected_exposed_exchange_ms"] = out["link_peak"]["exposed_exchange_ms"]
    out["expected_speedup"] = out["link_peak"]["speedup"]
    return out


def pack_gradients(grads: Sequence[torch.Tensor]):
    """One flat buffer holding ``grads`` back to back (one `cat` kernel) and, per tensor, a view of its segment."""
    flat = torch.cat([g.reshape(-1) for g in grads])
    views, off = [], 0
    for g in grads:
        views.append(flat[off:off + g.numel()].view(g.shape))
        off += g.numel()
    return flat, views


def allreduce_gradients(params: Iterable[torch.Tensor], world: int, group=None, *, sh_param: torch.Tensor = None,
                        sh_active_coeffs: int = None, restore_none: bool = False) -> None:
    """Mean over ranks of every ``.grad``: large tensors are all-reduced in place (largest first, so its ring starts while
    the rest is queued), the small ones travel packed in one buffer and ``p.grad`` is re-pointed at its segment.
    All collectives are issued asynchronously back to back and waited together.

    Every rank must pass the same parameters: a parameter without a gradient on this rank (it rendered no view, or the
    parameter did not reach its loss) takes part with zeros, so that all ranks issue the same collectives.

    ``sh_param`` / ``sh_active_coeffs``: while ``active_sh_degree`` is below the stored degree (reference
    scene/gaussian_model.py:118-120 raises it every 1000 iterations) the gradient of the inactive bands is exactly zero on
    every rank; only the leading ``sh_active_coeffs`` coefficients of ``sh_param.grad`` [N, K, 3] are exchanged then.

    A parameter that has no gradient on ANY rank (frozen, or unused by every view of the step) ends up with a zero gradient
    here, whereas the single-process loop leaves ``grad = None`` and Adam skips it (its moments do not decay).  Pass only
    parameters that take part in the loss, or ``restore_none=True``: a has-gradient flag per parameter rides in the packed
    buffer and ``grad`` is set back to ``None`` where no rank had one (costs one small device-to-host read after the
    collectives)."""
    if not _exchange(world):
        return
    params = list(params)
    had_grad = [p.grad is not None for p in params]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    all_params = params
    sh_slice = None
    if sh_param is not None and sh_active_coeffs is not None and sh_param.grad is not None and sh_param.grad.dim() == 3 \
            and 0 < sh_active_coeffs < sh_param.grad.shape[1]:
        params = [p for p in params if p is not sh_param]
        sh_slice = sh_param.grad[:, :sh_active_coeffs].contiguous()
    big = sorted((p for p in params if p.grad.numel() * p.grad.element_size() >= PACK_BELOW_BYTES),
                 key=lambda p: -p.grad.numel())
    small = [p for p in params if p.grad.numel() * p.grad.element_size() < PACK_BELOW_BYTES]
    # RCCL averages in the collective itself (no extra pass over 236 B/splat); gloo (CPU tests) has no AVG
    avg = dist.get_backend(group) == "nccl"
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    dev = all_params[0].grad.device if all_params else "cpu"
    flags = None
    with _Window(dev):
        works = [dist.all_reduce(p.grad, op=op, group=group, async_op=True) for p in big]
        for p in big:
            ExchangeStats.note("all_reduce", _nbytes(p.grad), world)
        if sh_slice is not None:
            works.append(dist.all_reduce(sh_slice, op=op, group=group, async_op=True))
            ExchangeStats.note("all_reduce", _nbytes(sh_slice), world)
        by_dtype = {}
        for p in small:
            by_dtype.setdefault(p.grad.dtype, []).append(p)
        packed = []
        for ps in by_dtype.values():
            flat, views = pack_gradients([p.grad for p in ps])
            works.append(dist.all_reduce(flat, op=op, group=group, async_op=True))
            ExchangeStats.note("all_reduce", _nbytes(flat), world)
            packed.append((ps, flat, views))
        if restore_none:
            flags = torch.tensor([1.0 if h else 0.0 for h in had_grad], dtype=torch.float32, device=dev)
            works.append(dist.all_reduce(flags, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()
    if not avg:
        scale = 1.0 / world
        for p in big:
            p.grad.mul_(scale)
        for _, flat, _ in packed:
            flat.mul_(scale)
        if sh_slice is not None:
            sh_slice.mul_(scale)
    for ps, _, views in packed:
        for p, v in zip(ps, views):
            p.grad = v
    if sh_slice is not None:
        sh_param.grad[:, :sh_slice.shape[1]].copy_(sh_slice)
    if flags is not None:
        for p, f in zip(all_params, flags.tolist()):
            if f == 0.0:
                p.grad = None   # no rank had a gradient for it: as in the single-process loop, the optimizer skips it


def _all_reduce_many(tensors: Sequence[torch.Tensor], group=None):
    """SUM all-reduce of several contiguous tensors in place, asynchronously, as ONE launch where the backend can group them
    (RCCL: ncclGroupStart / End through torch's coalescing manager -- the tensors stay where they are, no packing copy);
    returns an object with .wait()."""
    tensors = [t for t in tensors if t.numel() > 0]

    class _Works:
        def __init__(self, ws):
            self.ws = ws

        def wait(self):
            for w in self.ws:
                w.wait()

    if dist.get_backend(group) == "nccl" and hasattr(dist, "_coalescing_manager") and tensors:
        try:
            with dist._coalescing_manager(group=group, device=tensors[0].device, async_ops=True) as cm:
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return _Works([cm])
        except Exception:  # noqa: BLE001 -- an older torch without async coalescing: one collective per tensor below
            pass
    return _Works([dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in tensors])


class _SlicedGatherHook:
    """`slice_hook` of the rasterizer's backward for `sh_gather_step` (one view per rank): owns the gradient buffers, and behind
    every slice of the per-splat backward starts that slice's two collectives -- the all-gather of its colour gradients and ONE
    grouped sum all-reduce of its rows of the four geometric gradients, in place.  They run on RCCL's stream while the compute
    stream goes on with the next slice (0.09 ms of per-splat backward at 1 M splats) and, afterwards, with the SH-gradient
    rebuild of the slices whose colour gradients have arrived."""

    def __init__(self, slices: int, world: int, group, dev):
        self.slices, self.world, self.group, self.dev = int(slices), world, group, dev
        self.buf = None
        self.pending = []     # (lo, hi, gathered [world, hi - lo, 3], gather work, reduce works)
        self.window = None    # ExchangeStats: opens when the first collective is issued (inside the backward)

    def buffers(self, n: int) -> dict:
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=self.dev)
        self.buf = {"means3D": new(n, 3), "scales": new(n, 3), "rotations": new(n, 4), "opacities": new(n, 1), "colors": new(n, 3)}
        return self.buf

    def on_slice(self, j: int, lo: int, hi: int) -> None:
        b = self.buf
        if self.window is None:
            self.window = _Window(self.dev)
            self.window.__enter__()
        gathered = torch.empty(self.world, hi - lo, 3, dtype=torch.float32, device=self.dev)
        gw = dist.all_gather_into_tensor(gathered.view(-1), b["colors"][lo:hi].reshape(-1), group=self.group, async_op=True)
        ExchangeStats.note("all_gather", _nbytes(gathered), self.world)
        geo = [b[k][lo:hi] for k in ("means3D", "scales", "rotations", "opacities")]
        rw = _all_reduce_many(geo, self.group)
        ExchangeStats.note("all_reduce", sum(_nbytes(t) for t in geo), self.world)
        self.pending.append((lo, hi, gathered, gw, rw))


# slices of the per-splat backward whose exchange is started while the rest is still being computed (sh_gather_step, one
# view per rank); 1 = the unsliced exchange behind the whole backward
GATHER_SLICES = 4


def sh_gather_step(params: dict, cams: Sequence, bg, sh_degree: int, backward_fn: Callable, *, scaling_modifier: float = 1.0,
                   rank: int = None, world: int = None, group=None, slices: int = None) -> None:
    """View-parallel step for the SH colour path with the low-rank gradient exchange.

    ``params``: dict of leaf tensors ``means3D, scales, rotations, opacities, shs`` (replicated on every rank).
    ``cams``: the V cameras of this step (same list on every rank; rank r renders cams[r::world]).
    ``backward_fn(view_index, color, depth, alpha)`` must back-propagate the loss of that view *already divided by V*
    (e.g. ``(loss / V).backward()`` or ``torch.autograd.backward(outputs, upstream_grads / V)``).

    Afterwards every rank holds in ``p.grad`` the gradient of the mean loss over all V views -- the same result as
    all-reducing all five gradient tensors, but the 192 B/splat SH gradient never crosses xGMI: for one view it is
    basis(view direction) (x) dL/dcolour, so ranks all-gather the 12 B/splat colour gradients of all views and rebuild
    the sum locally (sr_sh_backward).  Wire traffic per rank drops from ~413 to ~161 bytes per splat.

    ``slices`` (default `GATHER_SLICES`; used when every rank renders exactly one view): the per-splat part of the rasterizer's
    backward runs in that many splat ranges, and the exchange of a finished range -- all-gather of its colour gradients, one
    grouped in-place all-reduce of its geometric gradients (no packing copy) -- is issued right behind it, so it overlaps the
    remaining ranges and the SH-gradient rebuild of earlier ones.  Results are identical to the unsliced step (same kernels
    per splat, same collectives per row).  Loss terms of ``backward_fn`` that reach the parameters OUTSIDE the rasterizer
    (regularisers on scales / opacities / SH, ...) are supported in both forms: their gradients are summed over the ranks by one
    extra packed all-reduce and added (every rank must run the same ``backward_fn``, as in any data-parallel step)."""
    import math
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from . import sh as shmod
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    V = len(cams)
    if V % world != 0:
        raise ValueError("the number of views must be a multiple of the number of ranks")
    names = ["means3D", "scales", "rotations", "opacities"]
    for p in params.values():
        p.grad = None
    means3D, shs = params["means3D"], params["shs"]
    dev = means3D.device
    n = means3D.shape[0]
    mine = list(range(rank, V, world))
    slices = GATHER_SLICES if slices is None else int(slices)
    if _exchange(world) and len(mine) == 1 and slices > 1 and n >= 512:
        return _sh_gather_step_sliced(params, cams, bg, sh_degree, backward_fn, scaling_modifier, rank, world, group, slices, mine[0])
    dcol_views = []
    for slot, vi in enumerate(mine):
        cam = cams[vi]
        rs = GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
        sink = []
        color, radii, depth, alpha = _checked(lambda: GaussianRasterizer(rs).forward_ex(
            means3D=means3D, means2D=torch.zeros_like(means3D, requires_grad=True), opacities=params["opacities"],
            shs=shs, scales=params["scales"], rotations=params["rotations"], color_grad_sink=sink))
        # the rasterizer's backward accumulates the 4 small gradients (incl. the view-direction term in means3D) and hands
        # over the clamp-masked colour gradient instead of writing 192 B/splat of SH gradient
        backward_fn(vi, color, depth, alpha)
        dcol_views.append(sink.pop())
    for k in names:
        if params[k].grad is None:
            params[k].grad = torch.zeros_like(params[k])
    # an SH gradient that did not come through the rasterizer (it hands over colour gradients here): an SH regulariser of
    # backward_fn; summed over the ranks and added to the rebuilt gradient below
    sh_extra = params["shs"].grad
    dcol_local = dcol_views[0][None] if len(dcol_views) == 1 else torch.stack(dcol_views)
    campos_all = _campos_of(cams, dev)
    if _exchange(world):
        gathered = torch.empty(world, len(mine), n, 3, dtype=torch.float32, device=dev)
        # gathered[r, slot] is view r + slot*world
        order = [r + sl * world for r in range(world) for sl in range(len(mine))]
        dcol_all = gathered.reshape(world * len(mine), n, 3)
        campos_used = campos_all if order == list(range(V)) else campos_all[torch.tensor(order, device=dev)]
        with _Window(dev):
            # the collectives run in issue order on RCCL's stream: the all-gather first (the SH rebuild needs it), then ONE
            # all-reduce of the four geometric gradients packed back to back (44 B/splat), which overlaps the SH rebuild
            gather_work = dist.all_gather_into_tensor(gathered.view(-1), dcol_local.view(-1), group=group, async_op=True)
            ExchangeStats.note("all_gather", _nbytes(gathered), world)
            flat, views = pack_gradients([params[k].grad for k in names])
            reduce_work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
            ExchangeStats.note("all_reduce", _nbytes(flat), world)
            extra_work = None
            if sh_extra is not None:
                sh_extra = sh_extra.detach().to(torch.float32).contiguous()
                extra_work = dist.all_reduce(sh_extra, op=dist.ReduceOp.SUM, group=group, async_op=True)
                ExchangeStats.note("all_reduce", _nbytes(sh_extra), world)
            gather_work.wait()
            with _Window(dev, inner=True):
                d_shs = shmod.sh_backward(means3D, shs, campos_used, dcol_all, sh_degree, want_shs=True)
            reduce_work.wait()
            if extra_work is not None:
                extra_work.wait()
        for k, v in zip(names, views):
            params[k].grad = v
    else:
        d_shs = shmod.sh_backward(means3D, shs, campos_all, dcol_local, sh_degree, want_shs=True)
    if sh_extra is not None:
        d_shs = d_shs + sh_extra.to(d_shs.dtype)
    params["shs"].grad = d_shs if d_shs.dtype == params["shs"].dtype else d_shs.to(params["shs"].dtype)


def _sh_gather_step_sliced(params, cams, bg, sh_degree, backward_fn, scaling_modifier, rank, world, group, slices, vi) -> None:
    """`sh_gather_step` with one view per rank and the exchange started slice by slice from inside the backward."""
    import math
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from . import sh as shmod
    names = ["means3D", "scales", "rotations", "opacities"]
    means3D, shs = params["means3D"], params["shs"]
    dev = means3D.device
    V = len(cams)
    cam = cams[vi]
    rs = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
    hook = _SlicedGatherHook(slices, world, group, dev)
    sink = []
    color, radii, depth, alpha = _checked(lambda: GaussianRasterizer(rs).forward_ex(
        means3D=means3D, means2D=torch.zeros_like(means3D, requires_grad=True), opacities=params["opacities"],
        shs=shs, scales=params["scales"], rotations=params["rotations"], color_grad_sink=sink, slice_hook=hook))
    campos_all = _campos_of(cams, dev)          # gathered[r] is view r (one view per rank: V == world)
    d_shs = torch.empty_like(shs, dtype=torch.float32)
    backward_fn(vi, color, depth, alpha)    # blend, then per slice: per-splat backward + hook.on_slice (collectives issued)
    if hook.buf is None:                    # nothing reached the rasterizer's backward (zero upstream gradients)
        hook.buffers(means3D.shape[0])
        for t in hook.buf.values():
            t.zero_()
        from .rasterizer import slice_ranges
        for j, (lo, hi) in enumerate(slice_ranges(means3D.shape[0], slices)):
            hook.on_slice(j, lo, hi)
    # Gradients that reached the leaves OUTSIDE the rasterizer (scale / opacity regularisers, any extra loss term of
    # backward_fn): the rasterizer's backward handed autograd nothing for these inputs (its part sits in the hook's buffers and is
    # already on the wire), so whatever `.grad` holds now is that extra part.  It is summed over the ranks by one packed
    # all-reduce of its own and added below -- the unsliced path gets the same result through autograd's accumulation.  Every
    # rank runs the same backward_fn, so every rank finds the same set of extras (the usual data-parallel assumption).
    extra_names = [k for k in names + ["shs"] if params[k].grad is not None]
    extra_work, extra_views = None, []
    if extra_names:
        flat, extra_views = pack_gradients([params[k].grad.detach().to(torch.float32) for k in extra_names])
        extra_work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        ExchangeStats.note("all_reduce", _nbytes(flat), world)
    m_det, s_det = means3D.detach(), shs.detach()
    for lo, hi, gathered, gw, rw in hook.pending:
        gw.wait()
        with _Window(dev, inner=True):
            shmod.sh_backward(m_det[lo:hi], s_det[lo:hi], campos_all, gathered, sh_degree, want_shs=True, out=d_shs[lo:hi])
    for _, _, _, _, rw in hook.pending:
        rw.wait()
    if extra_work is not None:
        extra_work.wait()
    if hook.window is not None:
        hook.window.__exit__(None, None, None)
    out = {k: hook.buf[k].view(params[k].shape) for k in names}
    out["shs"] = d_shs
    for k, v in zip(extra_names, extra_views):
        out[k] += v.view(out[k].shape)
    for k, g in out.items():
        # the kernels compute in fp32; a parameter of another dtype gets its gradient in its own (as the unsliced path does)
        params[k].grad = g if g.dtype == params[k].dtype else g.to(params[k].dtype)


_CAMPOS_CACHE: dict = {}


def _campos_of(cams: Sequence, dev) -> torch.Tensor:
    """[V,3] camera centres of a view list; the same cameras come back every step, so the stacked tensor is kept (the entry
    holds the source tensors, so their ids cannot be recycled, and is rebuilt if one was modified in place)."""
    src = tuple(c.camera_center for c in cams)
    key = (tuple(id(t) for t in src), str(dev))
    versions = tuple(t._version for t in src)
    hit = _CAMPOS_CACHE.get(key)
    if hit is not None and hit[1] == versions and all(a is b for a, b in zip(hit[2], src)):
        return hit[0]
    out = torch.stack([t.to(device=dev, dtype=torch.float32).reshape(3) for t in src])
    if len(_CAMPOS_CACHE) > 256:
        _CAMPOS_CACHE.clear()
    _CAMPOS_CACHE[key] = (out, versions, src)
    return out


def _all_to_all(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    """Equal-split all-to-all of contiguous buffers [world, ...].  RCCL moves device buffers directly; gloo (the CPU-backend
    tests) only implements it for host tensors, so it is staged through the host there."""
    if dist.get_backend(group) == "gloo" and inp.is_cuda:
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, group=group)


def sh_sharded_step(params: dict, cams: Sequence, bg, sh_degree: int, backward_fn: Callable, *, scaling_modifier: float = 1.0,
                    rank: int = None, world: int = None, group=None):
    """View-parallel step with the SH coefficients SHARDED by splat range (ZeRO-style for the 192 B/splat tensor that is 80 %
    of the parameters): rank r owns ``shs[lo:hi]``, r's slice of ``ceil(N / world)`` consecutive splats.

    1. every rank evaluates the colours of ITS shard for ALL V views of the step (one pass over its coefficients,
       ``sr_sh_forward_views``) and an all-to-all hands every rank the colours of all splats for the views it renders
       (12 B/splat/view received instead of reading 192 B/splat of SH locally);
    2. each rank rasterizes its views forward + backward on the precomputed-colour path (no SH traffic in the kernels);
    3. a second all-to-all returns the colour gradients to the shard owners, which rebuild the SH gradient of their shard
       from all views (``sr_sh_backward``) and add the view-direction term to their rows of ``means3D.grad``;
    4. ONE packed all-reduce sums the geometric gradients (44 B/splat).

    Wire traffic per rank: 2 x 10.5 + 77 = ~98 B/splat (plain all-reduce: ~413, `sh_gather_step`: ~161), and the per-rank
    step is the cheaper precomputed-colour step.  Afterwards ``means3D / scales / rotations / opacities`` hold the full
    gradient of the mean loss over all views on every rank (as after `sh_gather_step`); the SH gradient exists only for the
    owned shard and is returned as ``(lo, hi, d_shs[hi - lo, K, 3])`` -- what a sharded optimizer consumes; ``shs.grad`` is
    left ``None``.  ``params["shs"]`` may be the full tensor (only rows lo:hi are read) -- ``backward_fn`` as in
    `sh_gather_step`."""
    import math
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from . import sh as shmod
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    V = len(cams)
    if V % world != 0:
        raise ValueError("the number of views must be a multiple of the number of ranks")
    k = V // world
    names = ["means3D", "scales", "rotations", "opacities"]
    for p in params.values():
        p.grad = None
    means3D, shs = params["means3D"], params["shs"]
    dev = means3D.device
    n = means3D.shape[0]
    shard = (n + world - 1) // world
    lo, hi = min(n, rank * shard), min(n, (rank + 1) * shard)
    n_own = hi - lo
    # views in (destination rank d, slot) order: rank d renders views d, d + world, ...
    order = [d + sl * world for d in range(world) for sl in range(k)]
    campos = _campos_of([cams[vi] for vi in order], dev)
    m_own, sh_own = means3D.detach()[lo:hi], shs.detach()[lo:hi]

    # 1. colours of my shard for every view -> the ranks that render them
    send = torch.zeros(world, k, shard, 3, dtype=torch.float32, device=dev) if n_own < shard else \
        torch.empty(world, k, shard, 3, dtype=torch.float32, device=dev)
    keep = None
    if n_own == shard:   # the usual case: the kernel writes straight into the communication buffer
        _, keep = shmod.sh_forward_views(m_own, sh_own, campos, sh_degree, out=send.view(V, shard, 3))
    elif n_own > 0:      # short last shard: pad rows stay zero
        col_own, keep = shmod.sh_forward_views(m_own, sh_own, campos, sh_degree)     # [V, n_own, 3] each
        send.view(V, shard, 3)[:, :n_own].copy_(col_own)
    if _exchange(world):
        recv = torch.empty_like(send)
        with _Window(dev):
            _all_to_all(recv, send, group)
            ExchangeStats.note("all_to_all", _nbytes(send), world)
    else:
        recv = send
    # recv[s, slot] = colours of shard s for my view `slot`

    # 2. my views on the precomputed-colour path
    dcol_send = torch.zeros(world, k, shard, 3, dtype=torch.float32, device=dev) if world * shard != n else \
        torch.empty(world, k, shard, 3, dtype=torch.float32, device=dev)
    for slot in range(k):
        vi = rank + slot * world
        cam = cams[vi]
        cols = recv[:, slot].reshape(world * shard, 3)[:n].contiguous().requires_grad_(True)
        rs = GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
        color, radii, depth, alpha = _checked(lambda: GaussianRasterizer(rs).forward_ex(
            means3D=means3D, means2D=torch.zeros_like(means3D, requires_grad=True), opacities=params["opacities"],
            colors_precomp=cols, scales=params["scales"], rotations=params["rotations"]))
        backward_fn(vi, color, depth, alpha)
        g = cols.grad if cols.grad is not None else torch.zeros(n, 3, dtype=torch.float32, device=dev)
        dcol_send[:, slot].reshape(world * shard, 3)[:n].copy_(g) if k == 1 else \
            dcol_send[:, slot].copy_(torch.nn.functional.pad(g, (0, 0, 0, world * shard - n)).view(world, shard, 3))
    for name in names:
        if params[name].grad is None:
            params[name].grad = torch.zeros_like(params[name])

    # 3. colour gradients back to the shard owners; SH gradient (and the view-direction term) of my shard from all views
    if _exchange(world):
        dcol_recv = torch.empty_like(dcol_send)
        with _Window(dev):
            _all_to_all(dcol_recv, dcol_send, group)
            ExchangeStats.note("all_to_all", _nbytes(dcol_send), world)
    else:
        dcol_recv = dcol_send
    d_shs = None
    if n_own > 0:
        dcol = dcol_recv.view(V, shard, 3)[:, :n_own] * keep                          # clamp mask of each view
        d_means = torch.empty(n_own, 3, dtype=torch.float32, device=dev)
        d_shs = shmod.sh_backward(m_own, sh_own, campos, dcol, sh_degree, want_shs=True, means_grad=d_means,
                                  accumulate_means=False)
        params["means3D"].grad[lo:hi] += d_means

    # 4. the geometric gradients of all views
    if _exchange(world):
        with _Window(dev):
            flat, views = pack_gradients([params[name].grad for name in names])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            ExchangeStats.note("all_reduce", _nbytes(flat), world)
        for name, v_ in zip(names, views):
            params[name].grad = v_
    return lo, hi, d_shs


def view_parallel_step(params: List[torch.Tensor], views: Sequence, render_loss: Callable, *, rank: int = None,
                       world: int = None, group=None) -> torch.Tensor:
    """One data-parallel step.  ``render_loss(view) -> scalar loss`` renders one view with the shared
    parameters.  After the call every rank holds, in ``p.grad``, the gradient of
    ``mean_{v in views} render_loss(v)`` -- exactly what the single-process loop of the reference
    computes -- and the returned tensor is that mean loss (all-reduced)."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    for p in params:
        p.grad = None
    mine = shard_views(views, rank, world)

    def losses():
        tot = None
        for v in mine:   # the forwards of the rank's views are enqueued back to back (none waits once its camera is known)
            l = render_loss(v)
            tot = l if tot is None else tot + l
        return tot

    total = _checked(losses)
    n_views = len(views)
    if total is not None:
        # local contribution to the global mean; summed (not averaged) across ranks below
        (total / n_views).backward()
        local = (total / n_views).detach()
    else:
        local = torch.zeros((), device=params[0].device, dtype=params[0].dtype)
    if _exchange(world):
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        big = sorted((p for p in params if p.grad.numel() * p.grad.element_size() >= PACK_BELOW_BYTES or p.grad.dtype != local.dtype),
                     key=lambda q: -q.numel())
        small = [p for p in params if not any(p is b for b in big)]
        works = [dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True) for p in big]
        # the small gradients and the scalar loss share one buffer and one collective
        flat, views = pack_gradients([p.grad for p in small] + [local.reshape(1)])
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for w in works:
            w.wait()
        for p, v in zip(small, views):
            p.grad = v
        local = views[-1].reshape(())
    return local


def field_view_parallel_step(compute_splats: Callable[[], dict], views: Sequence, render_loss: Callable, *, rank: int = None,
                             world: int = None, group=None) -> torch.Tensor:
    """Data-parallel step of the NEURAL path (reference train.py:62-101: the splat attributes of an iteration are the output of
    the deform network -- `means3D`, `scales`, `rotations`, `opacity`, `rgb` -- and every view of the iteration renders them).

    The network is replicated and evaluated once per rank (same points, same time -> the same attributes everywhere); rank r
    renders views r, r+G, ... of them; what crosses the wire is the gradient with respect to the ATTRIBUTES, 56 B/splat with
    precomputed colours, in one packed sum all-reduce -- not the network's parameters (5 M floats for
    the SplatFields shapes) and not one collective per parameter tensor.  Each rank then back-propagates the reduced attribute
    gradient through its replica, which leaves identical parameter / position gradients on every rank, equal to those of the
    single-process loop over all views.

    ``compute_splats() -> dict`` runs the network (tensors that require grad take part; other entries pass through);
    ``render_loss(splats, view) -> scalar`` renders one view from the given attribute tensors.  Gradients accumulate into the
    ``.grad`` of whatever leaves ``compute_splats`` used (the caller zeroes them, as with a plain backward).  Returns the mean
    loss over all views."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    outputs = compute_splats()
    keys = [k for k, v in outputs.items() if torch.is_tensor(v) and v.requires_grad]
    # detached copies are the "parameters" of the rendering part of the step
    splats = dict(outputs)
    for k in keys:
        splats[k] = outputs[k].detach().requires_grad_(True)
    mine = shard_views(views, rank, world)

    def losses():
        tot = None
        for v in mine:
            l = render_loss(splats, v)
            tot = l if tot is None else tot + l
        return tot

    total = _checked(losses)
    n_views = len(views)
    ref = outputs[keys[0]]
    if total is not None:
        (total / n_views).backward()
        local = (total / n_views).detach().to(ref.dtype)
    else:
        local = torch.zeros((), device=ref.device, dtype=ref.dtype)
    grads = [splats[k].grad if splats[k].grad is not None else torch.zeros_like(splats[k]) for k in keys]
    if _exchange(world):
        flat, views_ = pack_gradients([g.to(ref.dtype) for g in grads] + [local.reshape(1)])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        grads, local = views_[:-1], views_[-1].reshape(())
    torch.autograd.backward([outputs[k] for k in keys], [g.to(outputs[k].dtype) for k, g in zip(keys, grads)])
    return local
