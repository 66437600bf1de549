#!/usr/bin/env python3
"""Build libsplatraster.so from the kernel sources of a git revision (A/B baselines for tools/ab.sh):
    python tools/build_rev.py HEAD gpurun_exp/lib_head.so [DEFINE ...]"""
import os, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
rev, out = sys.argv[1], Path(sys.argv[2]).resolve()
defines = sys.argv[3:]
with tempfile.TemporaryDirectory() as td:
    subprocess.run(f"git -C {ROOT} archive {rev} splatfields_amd/csrc include | tar -x -C {td}", shell=True, check=True)
    csrc = Path(td) / "splatfields_amd" / "csrc"
    srcs = sorted(str(p) for p in csrc.glob("*.hip"))
    out.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-fno-slp-vectorize"] + [f"-D{d}" for d in defines] + ["-o", str(out)] + srcs
    subprocess.run(cmd, check=True, cwd=str(csrc))
print(out)
