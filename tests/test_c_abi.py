"""The C-ABI shared library loads and exports every symbol include/splatraster.h declares; the host-only
entry points work without a GPU; the product path refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from splatfields_amd import build, _lib
    build.build_library()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "splatraster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from splatfields_amd import _lib
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names  # the ctypes table binds exactly the declared ABI


def test_host_only_entry_points(lib):
    from splatfields_amd import _lib
    assert lib.sr_version() == _lib.SR_VERSION == 4
    g1, g2 = lib.sr_geom_bytes(1000, 64, 64), lib.sr_geom_bytes(2000, 64, 64)
    assert 0 < g1 < g2 and g1 % 256 == 0
    assert lib.sr_binning_bytes(1000, 64, 64) >= 1000 * 28  # ent 8 + merge ping-pong 2 x 8 + sorted ids 4 bytes per instance
    assert lib.sr_image_bytes(800, 800) >= 800 * 800 * 8
    assert lib.sr_backward_scratch_bytes(1000) >= 1000 * 48
    assert lib.sr_profile_stage_name(5) == b"render_backward"


def test_binding_refuses_a_library_of_another_abi_version(lib, monkeypatch):
    """struct layouts / workspace contracts belong to an ABI version; the binding checks it at load."""
    from splatfields_amd import _lib
    monkeypatch.setattr(_lib, "SR_VERSION", _lib.SR_VERSION + 1)
    with pytest.raises(RuntimeError, match="version 4 of the splatraster ABI"):
        _lib.bind(_lib.LIB_PATH)


def test_host_sync_counters_start_at_zero_and_reset(lib):
    out = (C.c_longlong * 4)()
    assert lib.sr_debug_counters(out, 1) == 0
    assert lib.sr_debug_counters(out, 0) == 0 and list(out) == [0, 0, 0, 0]
    assert lib.sr_debug_counters(None, 0) != 0
    assert lib.sr_ticket_wait(None, None, None) != 0 and b"null ticket" in lib.sr_last_error()


def test_struct_layouts_match_header():
    from splatfields_amd import _lib
    # 9 x 4-byte scalars, padded to 8, then 4 pointers
    assert C.sizeof(_lib.SrView) == 40 + 4 * 8
    assert C.sizeof(_lib.SrSplats) == 8 + 7 * 8 + 8 + 8   # count (padded), 7 pointers, raw_params (padded), shs_rest
    assert C.sizeof(_lib.SrGrads) == 9 * 8


def test_argument_validation_errors_without_gpu(lib):
    from splatfields_amd import _lib
    view = _lib.SrView(64, 64, 0.5, 0.5, 1.0, 0, 0, 0, 0, None, None, None, None)
    splats = _lib.SrSplats(0, None, None, None, None, None, None, None, 0, None)
    inst = C.c_longlong(0)
    rc = lib.sr_forward_prepare(C.byref(view), C.byref(splats), None, None, C.byref(inst), None)
    assert rc != 0 and b"device pointers" in lib.sr_last_error()
    # tile coordinates are stored in 12 bits (the top nibbles of Geom::rect carry the reach mask): 65520 pixels a side at most
    fake = C.c_void_p(4096)   # never dereferenced: the size check comes first
    for h, w, ok_size in ((64, 65536, False), (65537, 64, False), (65520, 65520, True)):
        view = _lib.SrView(h, w, 0.5, 0.5, 1.0, 0, 0, 0, 0, fake, fake, fake, fake)
        rc = lib.sr_forward_prepare(C.byref(view), C.byref(splats), None, None, C.byref(inst), None)
        assert rc != 0 and (b"image too large" in lib.sr_last_error()) == (not ok_size), (h, w, lib.sr_last_error())


def test_facade_validation_and_no_cpu_fallback():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5, bg=torch.zeros(3),
                                       scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
                                       campos=torch.zeros(3), prefiltered=False, debug=False)
    r = GaussianRasterizer(raster_settings=rs)
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), scales=x, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x, scales=x, rotations=torch.zeros(4, 4))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from splatfields_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "libsplatraster.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    for pkg in ("splatfields_amd", "diff_gaussian_rasterization"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in src and "from oracle" not in src and "raster_ref" not in src, f


def test_header_constants_match_the_binding():
    """#define values of include/splatraster.h == the Python binding's constants."""
    from splatfields_amd import _lib
    text = open(os.path.join(ROOT, "include", "splatraster.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define\s+(SR_[A-Z_]+)\s+(\d+)\b", text, re.M)}
    assert defs["SR_NEED_CAPACITY"] == _lib.SR_NEED_CAPACITY
    assert defs["SR_VERSION"] == _lib.SR_VERSION
    assert (defs["SR_RAW_SCALES"], defs["SR_RAW_OPACITY"], defs["SR_RAW_ROTATIONS"], defs["SR_FORWARD_ONLY"]) == \
        (_lib.SR_RAW_SCALES, _lib.SR_RAW_OPACITY, _lib.SR_RAW_ROTATIONS, _lib.SR_FORWARD_ONLY)
    assert defs["SR_PROFILE_STAGES"] == _lib.PROFILE_STAGES
    assert defs["SR_MLP_MAX_PACK_JOBS"] == _lib.MLP_MAX_PACK_JOBS
    assert (defs["SR_MLP_MAX_GRAD_JOBS"], defs["SR_MLP_MAX_GRAD_TASKS"]) == (_lib.MLP_MAX_GRAD_JOBS, _lib.MLP_MAX_GRAD_TASKS)
    assert (defs["SR_MLP_MAX_OPS"], defs["SR_MLP_NONE"], defs["SR_MLP_LEAKY"], defs["SR_MLP_MASK"]) == \
        (_lib.MLP_MAX_OPS, _lib.MLP_NONE, _lib.MLP_LEAKY, _lib.MLP_MASK)


def test_mlp_entry_points_refuse_bad_descriptions_without_touching_a_device():
    """argument validation of sr_mlp_chain / sr_mlp_pack / sr_mlp_weight_grad happens on the host before any launch."""
    from splatfields_amd import _lib
    lib = _lib.load()
    buf = (C.c_float * 4096)()
    base = C.addressof(buf)
    base += (-base) % 16
    op = _lib.SrMlpOp(w_packed=base, bias=None, src=base, mask=None, store=None, out_tiles=8, mem_tiles=2, reg_tiles=0, src_row=32,
                      epilogue=_lib.MLP_LEAKY, mask_row=0, store_row=0, store_channels=0, store_accumulate=0, keep_state=0)
    ops = (_lib.SrMlpOp * 1)(op)
    assert lib.sr_mlp_chain(-1, 8, 1, ops, 0.01, None) != 0
    assert lib.sr_mlp_chain(0, 8, 1, ops, 1.5, None) != 0 and b"negative_slope" in lib.sr_last_error()
    assert lib.sr_mlp_chain(0, 8, 0, ops, 0.01, None) != 0
    assert lib.sr_mlp_chain(0, 8, _lib.MLP_MAX_OPS + 1, ops, 0.01, None) != 0
    assert lib.sr_mlp_chain(0, 6, 1, ops, 0.01, None) != 0 and b"unsupported op list" in lib.sr_last_error()   # out_tiles 8 > hidden_tiles 6
    for field, bad in [("mem_tiles", 3), ("src_row", 16), ("epilogue", 7), ("w_packed", base + 4), ("out_tiles", 0)]:
        broken = (_lib.SrMlpOp * 1)(op)
        setattr(broken[0], field, bad)
        assert lib.sr_mlp_chain(0, 8, 1, broken, 0.01, None) != 0 and b"unsupported op list" in lib.sr_last_error(), field
    masked = (_lib.SrMlpOp * 1)(op)
    masked[0].epilogue = _lib.MLP_MASK                                           # mask epilogue without a mask pointer
    assert lib.sr_mlp_chain(0, 8, 1, masked, 0.01, None) != 0 and b"unsupported op list" in lib.sr_last_error()
    assert lib.sr_mlp_pack(_lib.MLP_MAX_PACK_JOBS + 1, (_lib.SrMlpPackJob * 1)(), None) != 0 and b"sr_mlp_pack" in lib.sr_last_error()
    J = _lib.SrMlpGradJob
    good = (J * 2)(J(dz=base, x=base, dw=base, db=None, dz_row=128, m=128, x_row=96, k=94, dw_row=222, dw_col0=0),
                   J(dz=base, x=base, dw=base, db=base, dz_row=32, m=3, x_row=128, k=128, dw_row=128, dw_col0=0))
    # 2 x 2 + 1 x 2 blocks of 64 x 64, 256 slabs of 392 points, (4096 + 64) floats per block and slab
    assert lib.sr_mlp_weight_grad_workspace(100_000, 2, good) == 6 * 256 * (4096 + 64) * 4
    assert lib.sr_mlp_weight_grad_workspace(50, 2, good) == 6 * 1 * (4096 + 64) * 4
    bad = (J * 1)(J(dz=base, x=base, dw=base, db=None, dz_row=126, m=100, x_row=96, k=94, dw_row=94, dw_col0=0))   # row stride not a multiple of 4
    assert lib.sr_mlp_weight_grad_workspace(1000, 1, bad) == 0
    assert lib.sr_mlp_weight_grad(1000, 1, bad, base, 1 << 30, None) != 0 and b"unsupported job list" in lib.sr_last_error()
    assert lib.sr_mlp_weight_grad(0, 2, good, base, 1 << 30, None) != 0


def test_capacity_steps_are_coarse_and_sufficient():
    """The facade sizes the binning buffer and the backward scratch by a capacity that follows the largest instance count seen:
    quantised (eight steps per power of two) so that the per-view fluctuation of a training loop does not change the two
    allocation sizes every time a new maximum appears (splatfields_amd/rasterizer.py: _round_capacity)."""
    from splatfields_amd.rasterizer import _round_capacity
    prev = 0
    for inst in [0, 1, 999, 65_535, 65_536, 100_000, 2_291_306, 2_300_000, 2_350_000, 6_660_000, 25_400_000, 3_000_000_000]:
        c = _round_capacity(inst)
        assert c >= int(inst * 1.25) + 1024 and c >= 1 << 16          # never below what the count needs with its headroom
        assert c <= (int(inst * 1.25) + 1024) * 1.125 + 1 or c == 1 << 16   # at most one step above it
        assert c >= prev                                               # monotone
        prev = c
    # the headline's views (2.27 .. 2.31 M instances: 9 distinct counts, 9 sizes if the capacity followed them exactly) see at most
    # one change of size
    assert len({_round_capacity(i) for i in range(2_270_000, 2_310_001, 5_000)}) <= 2
