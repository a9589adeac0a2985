"""Builds the HIP extension in-tree (muax_amd/lib/libmzsearch.so) for gfx950."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmzsearch.so")
SOURCES = ["mz_api.hip"]
HEADERS = ["mz_spec.cuh", "mz_fused.cuh", "mz_step.cuh", "mz_step_jump.cuh", "mz_train.cuh", "mz_conv.cuh", "mz_dirichlet.cuh",
           os.path.join("..", "..", "include", "mzsearch.h")]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build muax_amd's gfx950 kernels)")


def stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> muax_amd/lib/libmzsearch.so"""
    if not force and not stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans",
           "-mllvm", "-amdgpu-mfma-vgpr-form",  # MFMA accumulators in VGPRs: the recurrent kernel's folds need no accvgpr moves
           "-fPIC", "-shared", "-Wno-unused-value",
           "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
