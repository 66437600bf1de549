"""Builds libsplatraster.so (hand-written HIP kernels + C ABI) for gfx950, in-tree."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libsplatraster.so"
# same ABI with the work counters of the backward blend compiled in (-DSR_BWD_STATS): measurement aid of bench.py
# (pairs_evaluated / pairs_blended), never the timed path
STATS_LIB_PATH = PKG_DIR / "libsplatraster_stats.so"
SOURCES = ["api.hip", "preprocess.hip", "binning.hip", "render.hip", "blend_bwd.hip", "knn.hip", "sh.hip", "densify.hip", "mlp.hip", "triplane.hip"]
HEADERS = ["common.h", "kernels.h", "expand.h", "sh_stage.h", "quadmask.h", "../../include/splatraster.h"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libsplatraster.so cannot be built")
    return exe


def is_stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HEADERS)


def strip_comments(text: str) -> str:
    """C / C++ source without comments and without blank or indentation differences (string and character literals are kept)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":                                   # literal: copy up to the closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    lines = (" ".join(line.split()) for line in "".join(out).splitlines())
    return "\n".join(line for line in lines if line)


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the CODE of the kernel sources and headers the library is built from (comments and
    layout do not count: strip_comments): recorded counter files under profiles/ carry it, and bench.py refuses them when the
    code has changed since they were collected."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        h.update(f.encode())
        h.update(strip_comments((CSRC / f).read_text()).encode())
    return h.hexdigest()[:16]


def build_library(force: bool = False, verbose: bool = False, out: Path = None, defines=()) -> Path:
    """hipcc --offload-arch=gfx950 ... -> splatfields_amd/libsplatraster.so (cross-compiles without a GPU).

    `out` / `defines`: an experimental variant of the same ABI (A/B timing through SPLATRASTER_LIB, tools/ab.sh)."""
    if out is None and not force and not is_stale():
        return LIB_PATH
    target = Path(out) if out is not None else LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           # the SLP vectoriser pairs fp32 adds into v_pk_add_f32, which blocks DPP fusion (v_add_f32_dpp) in the
           # wave reductions and costs extra v_mov shuffles in the VALU-bound blend loops
           "-fno-slp-vectorize"] + [f"-D{d}" for d in defines] + [
           "-o", str(target)] + [str(CSRC / f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return target


def build_stats_library(force: bool = False, verbose: bool = False) -> Path:
    if not force and STATS_LIB_PATH.exists() and not any(
            (CSRC / f).stat().st_mtime > STATS_LIB_PATH.stat().st_mtime for f in SOURCES + HEADERS):
        return STATS_LIB_PATH
    # the counters cost registers: with the product's cap of 128 (four workgroups per CU) the counting build spills 22 vector +
    # 41 scalar registers and its phase shares measure the spills; three wavefronts per SIMD (168 registers) hold them
    return build_library(force=True, verbose=verbose, out=STATS_LIB_PATH, defines=["SR_BWD_STATS", "SR_BWD_WAVES_PER_SIMD=3"])


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:   # python -m splatfields_amd.build OUT.so [DEFINE ...]
        print(build_library(force=True, verbose=True, out=Path(sys.argv[1]).resolve(), defines=sys.argv[2:]))
    else:
        print(build_library(force=True, verbose=True))
