"""The forward that does not block the host (include/splatraster.h: sr_forward_async; SURVEY.md section 8b "... or is avoided with
a capacity-bounded workspace"; [EXT] reads num_rendered back in the middle of every forward).

A camera that was rendered before (same splat count) is launched without waiting for its instance count; the ticket is redeemed
at the backward.  Checked here: the library's own counters prove that no forward waited, V forwards go out back to back, results
are bit-identical to the waiting path, a capacity guess that does not hold is detected before any gradient leaves and the retry
succeeds, and dropped forwards hand their tickets back."""
import math

import pytest
import torch

from splatfields_amd.synthetic import make_camera, make_splats, make_upstream_grads

pytestmark = pytest.mark.gpu
N, W, H = 6000, 176, 128
NAMES = ["means3D", "scales", "rotations", "opacities", "shs"]


def _settings(cam, bg, deg=3):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                         bg, 1.0, cam.world_view_transform, cam.full_proj_transform, deg, cam.camera_center, False, False)


def _leaves(sp):
    return {k: sp[k].clone().requires_grad_(True) for k in NAMES}


def _forward(p, cam, bg):
    from diff_gaussian_rasterization import GaussianRasterizer
    m2 = torch.zeros_like(p["means3D"], requires_grad=True)
    out = GaussianRasterizer(_settings(cam, bg)).forward_ex(means3D=p["means3D"], means2D=m2, opacities=p["opacities"], shs=p["shs"],
                                                            scales=p["scales"], rotations=p["rotations"])
    return out, m2


def _step(sp, cams, bg, grads):
    """the reference's iteration (train.py:158-252): every view of the step rendered, losses added, ONE backward"""
    p = _leaves(sp)
    gi, gd, ga = grads
    loss, keep = 0.0, []
    for cam in cams:
        (c, r, d, a), m2 = _forward(p, cam, bg)
        loss = loss + (c * gi).sum() + (d * gd).sum() + (a * ga).sum()
        keep.append((c.detach(), d.detach(), a.detach(), r, m2))
    loss.backward()
    torch.cuda.synchronize()
    return {k: v.grad.clone() for k, v in p.items()}, keep


@pytest.fixture()
def scene(hip_device):
    from splatfields_amd import rasterizer as rz
    prev = rz.set_async_forward(True)
    sp = make_splats(N, seed=21, device=hip_device, mean_scale=0.02)
    cams = [make_camera(v, W, H, device=hip_device) for v in range(4)]   # persistent camera tensors, as a training loop has them
    bg = torch.ones(3, device=hip_device)
    grads = make_upstream_grads(H, W, device=hip_device)
    yield sp, cams, bg, grads
    rz.set_async_forward(prev)


def test_known_cameras_are_launched_without_a_host_wait_and_give_identical_results(scene):
    from splatfields_amd import rasterizer as rz
    sp, cams, bg, grads = scene
    rz.set_async_forward(False)
    ref_g, ref_out = _step(sp, cams, bg, grads)                 # every forward waits (rounds 1-4); the cameras are known now
    rz.set_async_forward(True)
    rz.host_sync_counters(reset=True)
    g, out = _step(sp, cams, bg, grads)
    c = rz.host_sync_counters()
    assert c["forward_host_waits"] == 0 and c["async_forwards"] == len(cams), c   # V forwards enqueued back to back, none waited
    for k in NAMES:
        assert torch.equal(g[k], ref_g[k]), k
    for a, b in zip(out, ref_out):
        for x, y in zip(a[:4], b[:4]):
            assert torch.equal(x, y)
        assert torch.equal(a[4].grad, b[4].grad)                  # means2D.grad of every view
    assert rz.LAST_INSTANCES > 0


def test_first_render_of_a_camera_waits_and_the_second_does_not(scene):
    from splatfields_amd import rasterizer as rz
    sp, cams, bg, grads = scene
    cam = make_camera(6, W, H, device=sp["means3D"].device)      # a camera nobody has rendered yet
    rz.host_sync_counters(reset=True)
    _step(sp, [cam], bg, grads)
    assert rz.host_sync_counters()["forward_host_waits"] == 1 and rz.host_sync_counters()["async_forwards"] == 0
    _step(sp, [cam], bg, grads)
    c = rz.host_sync_counters()
    assert c["forward_host_waits"] == 1 and c["async_forwards"] == 1, c
    # a new splat count (densification) is a new situation: waits again
    sp2 = {k: v[:N - 512].contiguous() for k, v in sp.items()}
    _step(sp2, [cam], bg, grads)
    assert rz.host_sync_counters()["forward_host_waits"] == 2
    # rendering under no_grad never goes asynchronous (nothing would redeem the ticket before the image is used)
    with torch.no_grad():
        _forward({k: v for k, v in sp.items()}, cam, bg)
    assert rz.host_sync_counters()["forward_host_waits"] == 3


def test_a_capacity_guess_that_does_not_hold_raises_before_any_gradient_and_the_retry_succeeds(scene):
    """same camera, same splat count, but the splats have grown 12 x since its last render: far more instances than promised
    (the smallest capacity the facade ever uses is 65536 instances: the grown cloud must exceed it)"""
    from splatfields_amd import rasterizer as rz
    sp, cams, bg, grads = scene
    cam = cams[1]
    _step(sp, [cam], bg, grads)                                   # the camera's record: ~ N * 3 instances
    big = dict(sp, scales=sp["scales"] * 12.0)
    rz.set_async_forward(False)
    ref_g, ref_out = _step(big, [cam], bg, grads)
    assert rz.LAST_INSTANCES > 2 * 65536
    rz.set_async_forward(True)
    # (the waiting render above refreshed the record: make it stale again)
    pack = rz._ViewPack.get(_settings(cam, bg), sp["means3D"].device, 16)
    _step(sp, [cam], bg, grads)
    inst_small = pack.seen[N][0]
    key = (sp["means3D"].device.index, N, H, W)
    rz._CAPACITY[key] = rz._round_capacity(inst_small)            # as if only the small cloud had ever been seen
    p = _leaves(big)
    (c, r, d, a), m2 = _forward(p, cam, bg)
    loss = (c * grads[0]).sum() + (d * grads[1]).sum() + (a * grads[2]).sum()
    with pytest.raises(rz.RasterizerOverflow, match="re-run the step"):
        loss.backward()
    assert all(v.grad is None for v in p.values())                # nothing was applied
    assert rz._CAPACITY[key] >= pack.seen[N][0] > inst_small      # the estimates are corrected ...
    g, out = _step(big, [cam], bg, grads)                         # ... so the retry goes through (asynchronously again)
    for k in NAMES:
        assert torch.equal(g[k], ref_g[k]), k
    assert torch.equal(out[0][0], ref_out[0][0])


def test_a_list_longer_than_the_launched_sort_classes_is_detected(hip_device):
    """capacity fine, but one tile list is longer than the classes the hint selected: the blend must not touch the unsorted
    list (its ids were never written) and the ticket reports it"""
    from splatfields_amd import rasterizer as rz
    prev = rz.set_async_forward(True)
    try:
        sp = make_splats(40000, seed=5, device=hip_device, mean_scale=0.2)       # 64 x 64 image: lists of thousands of entries
        cam = make_camera(2, 64, 64, device=hip_device)
        bg = torch.ones(3, device=hip_device)
        grads = make_upstream_grads(64, 64, device=hip_device)
        ref_g, ref_out = _step(sp, [cam], bg, grads)
        pack = rz._ViewPack.get(_settings(cam, bg), hip_device, 16)
        inst, longest = pack.seen[40000]
        assert longest > 2048
        pack.seen[40000] = (inst, 100)                                           # a stale promise: "lists of 100 entries"
        p = _leaves(sp)
        (c, r, d, a), m2 = _forward(p, cam, bg)
        with pytest.raises(rz.RasterizerOverflow, match=f"a list of {longest}"):
            ((c * grads[0]).sum()).backward()
        assert pack.seen[40000] == (inst, longest)
        g, out = _step(sp, [cam], bg, grads)
        for k in NAMES:
            assert torch.equal(g[k], ref_g[k]), k
    finally:
        rz.set_async_forward(prev)


def test_resolve_pending_and_the_step_functions_re_render_transparently(scene):
    from splatfields_amd import rasterizer as rz
    from splatfields_amd.view_parallel import sh_gather_step
    sp, cams, bg, grads = scene
    sp = dict(sp, scales=sp["scales"] * 12.0)                     # ~50 tiles per splat
    gi, gd, ga = grads
    V = len(cams)

    def run():
        p = _leaves(sp)

        def bwd(vi, c, d, a):
            torch.autograd.backward((c, d, a), (gi / V, gd / V, ga / V))
        sh_gather_step(p, cams, bg, 3, bwd, rank=0, world=1)
        torch.cuda.synchronize()
        return {k: v.grad.clone() for k, v in p.items()}

    ref = run()
    assert rz.LAST_INSTANCES > 2 * 65536                          # (65536 = the smallest capacity the facade uses)
    for cam in cams:                                              # stale promises for every camera of the step
        pack = rz._ViewPack.get(_settings(cam, bg), sp["means3D"].device, 16)
        pack.seen[N] = (pack.seen[N][0] // 8, pack.seen[N][1])
    key = (sp["means3D"].device.index, N, H, W)
    rz._CAPACITY[key] = rz._round_capacity(rz.LAST_INSTANCES // 8)
    rz.host_sync_counters(reset=True)
    got = run()                                                   # no exception: redeemed before the backward, re-rendered
    assert rz.host_sync_counters()["async_forwards"] > len(cams)  # at least one forward was launched twice
    for k in NAMES:
        assert torch.equal(got[k], ref[k]), k
    # explicit form
    p = _leaves(sp)
    _forward(p, cams[0], bg)
    rz.resolve_pending()                                          # nothing pending afterwards
    assert not [r for r in (getattr(rz._TLS, "pending", None) or []) if r() is not None and r().ticket is not None]


def test_dropped_forwards_hand_their_tickets_back(scene):
    from splatfields_amd import rasterizer as rz
    sp, cams, bg, grads = scene
    _step(sp, cams[:1], bg, grads)
    before = rz.host_sync_counters()["tickets_created"]
    p = _leaves(sp)
    for _ in range(40):                                           # evaluation code that forgot no_grad: outputs dropped, no backward
        out = _forward(p, cams[0], bg)
        del out
    torch.cuda.synchronize()
    assert rz.host_sync_counters()["tickets_created"] - before <= 4


def test_plain_facade_waits_by_default_and_step_functions_do_not(scene):
    """an unmodified training loop never redeems a ticket before it reads the image (the reference even
    renders with gradients enabled and never back-propagates, train.py:379-387), so the plain facade waits in every forward
    unless the caller opts in; the step functions, which redeem and re-render, launch asynchronously on their own."""
    from splatfields_amd import rasterizer as rz
    from splatfields_amd.view_parallel import sh_gather_step
    sp, cams, bg, grads = scene
    prev = rz.set_async_forward(None)                             # the default policy
    try:
        assert not rz.async_forward_enabled()
        _step(sp, cams, bg, grads)                                # cameras known
        rz.host_sync_counters(reset=True)
        _step(sp, cams, bg, grads)
        c = rz.host_sync_counters()
        assert c["async_forwards"] == 0 and c["forward_host_waits"] == len(cams), c
        with rz.async_forward():                                  # a caller that redeems the tickets itself
            assert rz.async_forward_enabled()
            rz.host_sync_counters(reset=True)
            p = _leaves(sp)
            (col, r, d, a), m2 = _forward(p, cams[0], bg)
            rz.resolve_pending()
            assert rz.host_sync_counters()["async_forwards"] == 1
        assert not rz.async_forward_enabled()
        gi, gd, ga = grads
        p = _leaves(sp)
        rz.host_sync_counters(reset=True)
        sh_gather_step(p, cams, bg, 3, lambda vi, c_, d_, a_: torch.autograd.backward((c_, d_, a_), (gi, gd, ga)), rank=0, world=1)
        c = rz.host_sync_counters()
        assert c["async_forwards"] == len(cams) and c["forward_host_waits"] == 0, c
        rz.set_async_forward(False)                               # an explicit "off" also binds the step functions
        rz.host_sync_counters(reset=True)
        sh_gather_step(_leaves(sp), cams, bg, 3, lambda vi, c_, d_, a_: torch.autograd.backward((c_, d_, a_), (gi, gd, ga)), rank=0, world=1)
        assert rz.host_sync_counters()["async_forwards"] == 0
    finally:
        rz.set_async_forward(prev)


def test_a_stale_promise_cannot_be_consumed_silently(scene):
    """a loop that reads loss.item() (or saves the image) between a forward launched on a stale
    promise and its backward must not see a plausible number: the forward blend's overflow exit fills colour, depth and alpha
    with NaN on the device, and the ticket raises RasterizerOverflow."""
    from splatfields_amd import rasterizer as rz
    sp, cams, bg, grads = scene
    cam = cams[2]
    _step(sp, [cam], bg, grads)
    big = dict(sp, scales=sp["scales"] * 12.0)
    pack = rz._ViewPack.get(_settings(cam, bg), sp["means3D"].device, 16)
    inst_small = pack.seen[N][0]
    key = (sp["means3D"].device.index, N, H, W)
    rz._CAPACITY[key] = rz._round_capacity(inst_small)            # only the small cloud has ever been seen
    p = _leaves(big)
    (c, r, d, a), m2 = _forward(p, cam, bg)
    loss = (c * grads[0]).sum() + (d * grads[1]).sum() + (a * grads[2]).sum()
    assert math.isnan(loss.item())                                # what a logging line between forward and backward prints
    assert torch.isnan(c).all() and torch.isnan(d).all() and torch.isnan(a).all()
    with pytest.raises(rz.RasterizerOverflow, match="no result"):
        rz.resolve_pending()
    with pytest.raises(rz.RasterizerOverflow):                    # and the backward still refuses
        loss.backward()
    assert all(v.grad is None for v in p.values())
    # render(): with pipe.debug nothing unchecked leaves the function -- the stale forward is rendered again inside
    from types import SimpleNamespace
    from splatfields_amd.render import render
    rz._CAPACITY[key] = rz._round_capacity(inst_small)
    pack.seen[N] = (inst_small, pack.seen[N][1])
    gd_ = {"means3D": big["means3D"], "active_sh_degree": 3, "gaussian_opacity": big["opacities"], "gaussian_scales": big["scales"],
           "gaussian_rotations": big["rotations"], "gaussian_features": big["shs"]}
    res = render(cam, gd_, SimpleNamespace(debug=True), bg)
    assert torch.isfinite(res["render"]).all() and torch.isfinite(res["opacity"]).all()


def test_a_fresh_background_tensor_per_call_keeps_the_camera_known(scene):
    """the reference builds `bg_color*0.0` per call (gaussian_renderer/__init__.py:81): the camera is the same camera"""
    from splatfields_amd import rasterizer as rz
    sp, cams, bg, grads = scene
    _step(sp, cams[:1], bg, grads)
    rz.host_sync_counters(reset=True)
    p = _leaves(sp)
    (c, r, d, a), m2 = _forward(p, cams[0], bg * 0.0)             # a new tensor object
    rz.resolve_pending()
    assert rz.host_sync_counters()["async_forwards"] == 1
    assert torch.allclose(a, _forward(_leaves(sp), cams[0], bg)[0][3])   # alpha does not depend on the background


def test_the_reference_mask_pass_is_served_from_the_first_pass(hip_device):
    """the UNMODIFIED reference render() (gaussian_renderer/__init__.py:94-115) calls the rasterizer
    a second time with white colours on a black background.  The facade answers that call with the first pass's alpha output
    -- same values, same gradients as two full passes, no second rasterization and no host wait."""
    from types import SimpleNamespace
    from splatfields_amd import rasterizer as rz
    from splatfields_amd.render import render
    dev = hip_device
    sp = make_splats(N, seed=3, device=dev, mean_scale=0.03)
    cam = make_camera(1, W, H, device=dev)
    bg = torch.tensor([1.0, 0.5, 0.25], device=dev)
    gi, gd, ga = make_upstream_grads(H, W, device=dev)
    pipe = SimpleNamespace(debug=False)

    def run(shortcut: bool):
        prev = rz.set_mask_shortcut(shortcut)
        try:
            p = _leaves(sp)
            gdict = {"means3D": p["means3D"], "active_sh_degree": 3, "gaussian_opacity": p["opacities"], "gaussian_scales": p["scales"],
                     "gaussian_rotations": p["rotations"], "gaussian_features": p["shs"]}
            res = render(cam, gdict, pipe, bg, two_pass=True)      # the literal call pattern of the reference
            loss = (res["render"] * gi).sum() + (res["depth"] * gd).sum() + (res["opacity"] * ga).sum()
            loss.backward()
            torch.cuda.synchronize()
            return res, {k: v.grad.clone() for k, v in p.items()}, res["viewspace_points"].grad.clone()
        finally:
            rz.set_mask_shortcut(prev)

    ref, gref, m2ref = run(False)                                  # two full rasterizations
    before = rz.MASK_CALLS_SERVED
    rz.host_sync_counters(reset=True)
    got, g, m2 = run(True)
    c = rz.host_sync_counters()
    assert rz.MASK_CALLS_SERVED == before + 1
    assert c["forward_host_waits"] + c["async_forwards"] == 1, c   # ONE rasterization, and the mask pass waited for nothing
    assert got["opacity"].shape == ref["opacity"].shape == (1, H, W)
    assert torch.allclose(got["opacity"], ref["opacity"], rtol=0, atol=1e-6)
    assert torch.equal(got["render"], ref["render"]) and torch.equal(got["depth"], ref["depth"])
    for k in NAMES:
        scale = gref[k].abs().max().item()
        assert torch.allclose(g[k], gref[k], rtol=0, atol=2e-5 * scale), (k, (g[k] - gref[k]).abs().max().item() / scale)
    assert torch.allclose(m2, m2ref, rtol=0, atol=2e-5 * m2ref.abs().max().item())
    # a process whose FIRST pattern-shaped call carries other colours (feature rendering behind the RGB pass) never meets the
    # NaN guard: that call is checked on the host, switches the shortcut off and is rasterized in full
    from diff_gaussian_rasterization import GaussianRasterizer
    prev, prev_conf = rz.set_mask_shortcut(True), rz._MASK_CONFIRMED
    try:
        rz._MASK_CONFIRMED = False
        p = _leaves(sp)
        m2_ = torch.zeros_like(p["means3D"], requires_grad=True)
        kw = dict(means3D=p["means3D"], means2D=m2_, opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])
        GaussianRasterizer(_settings(cam, bg))(shs=p["shs"], **kw)
        feat = GaussianRasterizer(_settings(cam, bg * 0.0))(colors_precomp=torch.full((N, 3), 0.5, device=dev), **kw)[0]
        assert torch.isfinite(feat).all() and not rz._MASK_SHORTCUT and not rz._MASK_CONFIRMED
    finally:
        rz.set_mask_shortcut(prev)
        rz._MASK_CONFIRMED = prev_conf
    # once the pattern is confirmed, a call that only LOOKS like the mask pass (other colours) is answered with NaN, never with
    # a wrong mask, and switches the shortcut off for the process once its verdict has arrived
    import warnings
    prev = rz.set_mask_shortcut(True)
    rz._MASK_CONFIRMED = True
    try:
        p = _leaves(sp)
        m2_ = torch.zeros_like(p["means3D"], requires_grad=True)
        kw = dict(means3D=p["means3D"], means2D=m2_, opacities=p["opacities"], scales=p["scales"], rotations=p["rotations"])
        GaussianRasterizer(_settings(cam, bg))(shs=p["shs"], **kw)
        fake = GaussianRasterizer(_settings(cam, bg * 0.0))(colors_precomp=torch.full((N, 3), 0.5, device=dev), **kw)[0]
        assert torch.isnan(fake).all()
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            GaussianRasterizer(_settings(cam, bg))(shs=p["shs"], **kw)
            real = GaussianRasterizer(_settings(cam, bg * 0.0))(colors_precomp=torch.full((N, 3), 0.5, device=dev), **kw)[0]
        assert any("mask pass" in str(x.message) for x in w)
        assert not rz._MASK_SHORTCUT and torch.isfinite(real).all()   # rasterized in full from now on
    finally:
        rz.set_mask_shortcut(prev)
