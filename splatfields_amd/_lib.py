"""ctypes binding of the C ABI declared in include/splatraster.h.

The product path has no CPU fallback: if libsplatraster.so is missing this module raises."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

# SPLATRASTER_LIB selects an alternative build of the same ABI (A/B experiments); default: the in-tree library.
LIB_PATH = Path(os.environ.get("SPLATRASTER_LIB") or (Path(__file__).resolve().parent / "libsplatraster.so"))

c_float_p = C.c_void_p  # raw device pointers travel as integers


class SrView(C.Structure):
    _fields_ = [("image_height", C.c_int), ("image_width", C.c_int), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float), ("sh_degree", C.c_int), ("sh_coeffs", C.c_int), ("prefiltered", C.c_int),
                ("debug", C.c_int), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
                ("bg", C.c_void_p)]


class SrSplats(C.Structure):
    _fields_ = [("count", C.c_int), ("means3D", C.c_void_p), ("opacities", C.c_void_p), ("scales", C.c_void_p),
                ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p), ("shs", C.c_void_p),
                ("colors_precomp", C.c_void_p), ("raw_params", C.c_int), ("shs_rest", C.c_void_p)]


class SrGrads(C.Structure):
    _fields_ = [("dL_dmeans3D", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dopacity", C.c_void_p),
                ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p), ("dL_dcov3D", C.c_void_p),
                ("dL_dshs", C.c_void_p), ("dL_dcolors", C.c_void_p), ("dL_dshs_rest", C.c_void_p)]


class SrMlpOp(C.Structure):
    _fields_ = [("w_packed", C.c_void_p), ("bias", C.c_void_p), ("src", C.c_void_p), ("mask", C.c_void_p), ("store", C.c_void_p),
                ("sign_store", C.c_void_p), ("mask_bits", C.c_void_p),
                ("out_tiles", C.c_int), ("mem_tiles", C.c_int), ("reg_tiles", C.c_int), ("src_row", C.c_int), ("epilogue", C.c_int),
                ("mask_row", C.c_int), ("store_row", C.c_int), ("store_channels", C.c_int), ("store_accumulate", C.c_int),
                ("keep_state", C.c_int)]


class SrMlpPackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias_src", C.c_void_p), ("dst", C.c_void_p), ("bias_dst", C.c_void_p), ("ld", C.c_int),
                ("transposed", C.c_int), ("row0", C.c_int), ("n_rows", C.c_int), ("n_mem", C.c_int), ("mem_pad", C.c_int),
                ("mem_col0", C.c_int), ("n_reg", C.c_int), ("reg_width", C.c_int), ("reg_col0", C.c_int), ("out_tiles", C.c_int),
                ("n_bias", C.c_int)]


class SrMlpGradJob(C.Structure):
    _fields_ = [("dz", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("db", C.c_void_p), ("dz_row", C.c_int), ("m", C.c_int),
                ("x_row", C.c_int), ("k", C.c_int), ("dw_row", C.c_int), ("dw_col0", C.c_int)]


class SrResFieldJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("weights_t", C.c_void_p), ("matrix_t", C.c_void_p), ("out", C.c_void_p), ("d_out", C.c_void_p),
                ("d_matrix_t", C.c_void_p), ("d_weights_t", C.c_void_p), ("count", C.c_int), ("rank", C.c_int), ("capacity", C.c_int)]


RESFIELD_MAX_JOBS, RESFIELD_MAX_RANK = 16, 64
MLP_MAX_GRAD_JOBS, MLP_MAX_GRAD_TASKS = 16, 128
MLP_MAX_PACK_JOBS = 32
MLP_MAX_OPS, MLP_NONE, MLP_LEAKY, MLP_MASK = 24, 0, 1, 2      # include/splatraster.h: SR_MLP_*


# every symbol include/splatraster.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sr_version": (C.c_int, []),
    "sr_last_error": (C.c_char_p, []),
    "sr_geom_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "sr_binning_bytes": (C.c_size_t, [C.c_longlong, C.c_int, C.c_int]),
    "sr_image_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "sr_backward_scratch_bytes": (C.c_size_t, [C.c_longlong]),
    "sr_forward_prepare": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_longlong), C.c_void_p]),
    "sr_forward_render": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p, C.c_longlong,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_forward": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p]),
    "sr_forward_async": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                   C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p),
                                   C.c_void_p]),
    "sr_ticket_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "sr_ticket_release": (C.c_int, [C.c_void_p]),
    "sr_last_longest_list": (C.c_longlong, []),
    "sr_debug_counters": (C.c_int, [C.POINTER(C.c_longlong), C.c_int]),
    "sr_backward": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SrGrads),
                              C.c_void_p]),
    "sr_backward_blend": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_backward_splats": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(SrGrads), C.c_int, C.c_int, C.c_void_p]),
    "sr_debug_snapshot": (C.c_int, [C.POINTER(SrView), C.POINTER(SrSplats), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "sr_set_backward_kernel": (C.c_int, [C.c_int]),
    "sr_sh_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_sh_forward_views": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_sh_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "sr_knn_workspace_bytes": (C.c_size_t, [C.c_int]),
    "sr_knn3_mean_dist2": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_densify_workspace_bytes": (C.c_size_t, [C.c_int]),
    "sr_densify_plan": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                  C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_longlong), C.c_void_p]),
    "sr_densify_gather": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "sr_mlp_pack": (C.c_int, [C.c_int, C.POINTER(SrMlpPackJob), C.c_void_p]),
    "sr_mlp_weight_grad_workspace": (C.c_size_t, [C.c_int, C.c_int, C.POINTER(SrMlpGradJob)]),
    "sr_mlp_weight_grad": (C.c_int, [C.c_int, C.c_int, C.POINTER(SrMlpGradJob), C.c_void_p, C.c_size_t, C.c_void_p]),
    "sr_mlp_chain": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(SrMlpOp), C.c_float, C.c_void_p]),
    "sr_mlp_input_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_mlp_top_gradient": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "sr_mlp_input_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_resfield_compose": (C.c_int, [C.c_int, C.POINTER(SrResFieldJob), C.c_void_p, C.c_void_p]),
    "sr_resfield_backward_workspace": (C.c_size_t, [C.c_int, C.POINTER(SrResFieldJob)]),
    "sr_resfield_backward": (C.c_int, [C.c_int, C.POINTER(SrResFieldJob), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sr_triplane_backward_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "sr_triplane_forward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_triplane_backward": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "sr_debug_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_longlong, C.POINTER(C.c_size_t)]),
    "sr_debug_backward_stats": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "sr_profile_enable": (C.c_int, [C.c_int]),
    "sr_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "sr_profile_stage_name": (C.c_char_p, [C.c_int]),
    "sr_mark_visible": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sr_densification_stats": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

PROFILE_STAGES = 7
SR_VERSION = 4          # include/splatraster.h: the ABI this binding was written against
SR_NEED_CAPACITY = 2
SR_RAW_SCALES, SR_RAW_OPACITY, SR_RAW_ROTATIONS, SR_FORWARD_ONLY = 1, 2, 4, 8
_lib = None


def bind(path) -> C.CDLL:
    """dlopen a build of the library and attach the prototypes of every declared symbol."""
    lib = C.CDLL(str(path))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    got = lib.sr_version()
    if got != SR_VERSION:   # struct layouts, workspace sizes and buffer contracts belong to a version: never mix them
        raise RuntimeError(f"{path} implements version {got} of the splatraster ABI, this binding expects {SR_VERSION}: "
                           "rebuild it with `python -m splatfields_amd.build`")
    return lib


def load() -> C.CDLL:
    """Loads libsplatraster.so once; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m splatfields_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = bind(LIB_PATH)
    return _lib


class use_library:
    """Context manager: route the facade through another build of the same ABI (bench.py's counting build)."""

    def __init__(self, path):
        self.handle = bind(path)

    def __enter__(self):
        global _lib
        self.prev, _lib = _lib, self.handle
        return self.handle

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def check(status: int) -> None:
    if status != 0:
        raise RuntimeError("libsplatraster: " + load().sr_last_error().decode("utf-8", "replace"))
