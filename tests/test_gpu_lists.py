"""Integer work of the binning stage, read back from the opaque buffers (sr_debug_layout) and compared with the oracle:
per-tile lists (membership and order) and the instance accounting.

The kernels emit a (splat, tile) instance only where the splat's alpha >= 1/255 support can reach the tile, inside
upstream's 3-sigma tile rectangle (NOTEBOOK.md section 4 item 2: the rectangle is clipped to the support's bounding box, and
for rectangles of >= 4 tiles each tile is tested against the ellipse itself).  So, per tile:
  * the HIP list is a subset of the oracle's 3-sigma list, in the same (depth, splat index) order;
  * every oracle entry the HIP list leaves out has alpha < 1/255 on every pixel of the tile (nothing blended is lost);
  * sum of list lengths <= instance count == what the facade reports."""
import ctypes as C

import pytest
import torch

from oracle import torch_oracle as O
from tests.helpers import make_scene

pytestmark = pytest.mark.gpu


def hip_tile_lists(sp, st, dev):
    from splatfields_amd import _lib, rasterizer as rz
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    lib = _lib.load()
    leaf = {k: v.detach().to(dev).clone().requires_grad_(True) for k, v in sp.items()}
    rs = GaussianRasterizationSettings(
        image_height=st.image_height, image_width=st.image_width, tanfovx=st.tanfovx, tanfovy=st.tanfovy, bg=st.bg.to(dev),
        scale_modifier=st.scale_modifier, viewmatrix=st.viewmatrix.to(dev), projmatrix=st.projmatrix.to(dev),
        sh_degree=st.sh_degree, campos=st.campos.to(dev), prefiltered=False, debug=False)
    color, radii, depth, alpha = rz.rasterize_gaussians(leaf["means3D"], torch.zeros_like(leaf["means3D"]), leaf["shs"], None,
                                                        leaf["opacities"], leaf["scales"], leaf["rotations"], None, rs)
    torch.cuda.synchronize()
    fn = color.grad_fn
    geom, binning = fn.saved_tensors[8], fn.saved_tensors[9]
    n, H, W = leaf["means3D"].shape[0], st.image_height, st.image_width
    off = (C.c_size_t * 4)()
    assert lib.sr_debug_layout(n, H, W, int(fn.capacity), off) == 0
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    i32 = lambda buf, o, cnt: buf[o:o + 4 * cnt].view(torch.int32).cpu().to(torch.int64)
    tile_start = i32(geom, off[0], tiles + 1)
    total = i32(geom, off[2], 4)
    sorted_id = i32(binning, off[1], int(tile_start[-1]))      # list entries <= instances (tile_reached, csrc/common.h)
    return tile_start, sorted_id, int(total[0]), int(rz.LAST_INSTANCES), radii.cpu()


@pytest.mark.parametrize("n,w,h,scale", [(20000, 320, 240, None), (3000, 200, 152, 0.06), (60000, 400, 304, None)])
def test_tile_lists_match_the_oracle(hip_device, n, w, h, scale):
    sp, cam, st, grads = make_scene(n, w, h, mean_scale=scale, view=4)
    check_tile_lists(sp, st, n, w, h, hip_device)


def test_a_few_lists_between_2048_and_4096_entries(hip_device):
    """A sparse view with ONE list in the 2049..4096 class (the headline has such a tile in about half of its views): the long
    classes walk only the head of the launch order (Geom::total[4..6], csrc/binning.hip k_scan_small) -- a tight cluster of
    2600 small splats in front of the camera's target puts that many entries into the centre tile."""
    n, w, h = 12000, 336, 272   # the image centre (the camera's target) is the centre of a tile
    sp, cam, st, grads = make_scene(n, w, h, view=4)
    g = torch.Generator().manual_seed(7)
    idx = torch.randperm(n, generator=g)[:2600]
    sp["means3D"][idx] = 0.004 * torch.randn(2600, 3, generator=g)
    sp["scales"][idx] = sp["scales"][idx].clamp(max=0.01)
    lengths = check_tile_lists(sp, st, n, w, h, hip_device)
    long = (lengths > 2048) & (lengths <= 4096)
    assert 1 <= int(long.sum()) <= 8 and int(lengths.max()) <= 4096, lengths.max()


def check_tile_lists(sp, st, n, w, h, hip_device):
    tile_start, sorted_id, total, reported, radii = hip_tile_lists(sp, st, hip_device)
    # instances = tiles of every splat's rectangle (what the facade reports and sizes the per-instance buffers by); the lists hold
    # the instances whose tile the splat can reach
    assert total == reported >= int(tile_start[-1]) == sorted_id.numel()
    assert (tile_start[1:] >= tile_start[:-1]).all()

    d64 = {k: v.double() for k, v in sp.items()}
    pre = O.preprocess(d64["means3D"], None, d64["opacities"], d64["shs"], None, d64["scales"], d64["rotations"], None, st)
    d32 = {k: v.float() for k, v in sp.items()}
    pre32 = O.preprocess(d32["means3D"], None, d32["opacities"], d32["shs"], None, d32["scales"], d32["rotations"], None, st)
    # splats whose integer decisions (radius, tile rectangle, visibility) do not depend on the working precision
    stable = (pre.rect == pre32.rect).all(dim=1) & (pre.visible == pre32.visible) & (pre.radii == radii)
    assert stable.float().mean().item() > 0.99
    gx, gy = (w + 15) // 16, (h + 15) // 16
    op = d64["opacities"].reshape(-1)
    A, B, Cc = pre.conic[:, 0], pre.conic[:, 1], pre.conic[:, 2]
    checked_missing = 0
    for t in range(gx * gy):
        tx, ty = t % gx, t // gx
        hip = sorted_id[int(tile_start[t]):int(tile_start[t + 1])]
        # a splat appears at most once per tile
        assert hip.unique().numel() == hip.numel()
        in_rect = pre.visible & (pre.rect[:, 0] <= tx) & (pre.rect[:, 2] > tx) & (pre.rect[:, 1] <= ty) & (pre.rect[:, 3] > ty)
        ref = torch.nonzero(in_rect).reshape(-1)
        # --- membership: subset of the 3-sigma list (precision-stable splats) ---
        hip_stable = hip[stable[hip]]
        assert in_rect[hip_stable].all(), f"tile {t}: an instance outside upstream's tile rectangle"
        # --- nothing blended is lost: the left-out entries never reach alpha 1/255 on this tile ---
        member = torch.zeros(n, dtype=torch.bool)
        member[hip] = True
        missing = ref[~member[ref] & stable[ref]]
        if missing.numel():
            ys, xs = torch.meshgrid(torch.arange(ty * 16, min(ty * 16 + 16, h), dtype=torch.float64),
                                    torch.arange(tx * 16, min(tx * 16 + 16, w), dtype=torch.float64), indexing="ij")
            dx = pre.pix[missing, 0][:, None] - xs.reshape(1, -1)
            dy = pre.pix[missing, 1][:, None] - ys.reshape(1, -1)
            power = -0.5 * (A[missing, None] * dx * dx + Cc[missing, None] * dy * dy) - B[missing, None] * dx * dy
            amax = (op[missing, None] * torch.exp(power)).max(dim=1).values
            assert (amax < (1.0 / 255.0) * (1 + 1e-4)).all(), f"tile {t}: a contributing instance is missing"
            checked_missing += int(missing.numel())
        # --- order: front to back by view depth, ties by splat index ---
        if hip.numel() > 1:
            dz = pre.depth[hip]
            assert (dz[1:] >= dz[:-1] - 1e-6 * dz[:-1].abs()).all(), f"tile {t}: list not sorted by depth"
            f32 = dz.float()
            tie = f32[1:] == f32[:-1]
            assert (hip[1:][tie] > hip[:-1][tie]).all() or not stable[hip].all(), f"tile {t}: equal depths not in splat order"
    # the exact-support culling does leave entries out, and all of them were checked
    assert checked_missing > 0
    return tile_start[1:] - tile_start[:-1]
