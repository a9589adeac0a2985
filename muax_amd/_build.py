"""Builds the HIP extension in-tree (muax_amd/lib/libmzsearch.so) for gfx950.

The library is several translation units compiled in parallel (each `hipcc -c`, objects under muax_amd/lib/obj/)
and linked into one shared object: the C-ABI and the step-wise / training / Dirichlet kernels (mz_api.hip), the
fused act() kernel instances in five groups (mz_fused_g*.hip, listed in mz_instances.def) and the ResNet
recurrent kernel (mz_conv.hip).  Only the units whose sources changed are recompiled."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libmzsearch.so")
_ABI = os.path.join("..", "..", "include", "mzsearch.h")
_FUSED = ["mz_fused_group.inc", "mz_fused_launch.h", "mz_fused.cuh", "mz_spec.cuh", "mz_instances.def", "mz_host.h", _ABI]
# translation unit -> the headers it is rebuilt for
UNITS = {
    "mz_api.hip": ["mz_host.h", "mz_fused_launch.h", "mz_fused.cuh", "mz_spec.cuh", "mz_step.cuh", "mz_step_jump.cuh",
                   "mz_mlp_generic.cuh", "mz_train.cuh", "mz_dirichlet.cuh", _ABI],
    "mz_fused_g0.hip": _FUSED,
    "mz_fused_g1.hip": _FUSED,
    "mz_fused_g2.hip": _FUSED,
    "mz_fused_g3.hip": _FUSED,
    "mz_fused_g4.hip": _FUSED,
    "mz_conv.hip": ["mz_host.h", "mz_conv_host.h", "mz_conv.cuh", "mz_spec.cuh", _ABI],
    "mz_search_conv.hip": ["mz_host.h", "mz_conv_host.h", "mz_conv.cuh", "mz_step.cuh", "mz_step_jump.cuh", "mz_spec.cuh", _ABI],
    "mz_norm.hip": ["mz_host.h", "mz_norm.cuh", "mz_repr.cuh", "mz_repr_host.h", "mz_spec.cuh", _ABI],
    "mz_repr.hip": ["mz_host.h", "mz_repr.cuh", "mz_repr_host.h", "mz_norm.cuh", "mz_spec.cuh", _ABI],
    "mz_ez.hip": ["mz_host.h", "mz_ez.cuh", "mz_spec.cuh", _ABI],
}
SOURCES = list(UNITS)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans",
         "-mllvm", "-amdgpu-mfma-vgpr-form",  # MFMA accumulators in VGPRs: the recurrent kernel's folds need no accvgpr moves
         "-fno-slp-vectorize",  # SLP packs the two scalar adds of paired DPP butterflies into mov_dpp x 2 + v_pk_add (-1.6 % without)
         "-fPIC", "-Wno-unused-value"]
# per-unit additions, measured on one box (profiles/r02_fused_variants.txt): the fused instances are issue-bound single
# wavefronts -- the scheduler's max-ILP strategy is worth 5 % on the small-embedding shapes and 1 % on the E = 32 ones
# (before their layers became v_fmac_f32_dpp chain blocks it cost those 2 %), nothing on the other translation units
UNIT_FLAGS = {u: ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
              for u in ("mz_fused_g0.hip", "mz_fused_g1.hip", "mz_fused_g2.hip", "mz_fused_g3.hip", "mz_fused_g4.hip")}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build muax_amd's gfx950 kernels)")


def _obj(unit: str, tag: str = "") -> str:
    return os.path.join(OBJ_DIR, os.path.splitext(unit)[0] + tag + ".o")


def _unit_stale(unit: str, tag: str = "") -> bool:
    o = _obj(unit, tag)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in [unit] + UNITS[unit])


def stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for u in UNITS for f in [u] + UNITS[u])


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str = None) -> str:
    """hipcc --offload-arch=gfx950 -c <unit> (in parallel) ... ; hipcc -shared -> muax_amd/lib/libmzsearch.so.
    `extra_flags` / `out`: variant builds of the same ABI for the tools (-DMZ_PROFILE, A/B defines)."""
    out = out or LIB_PATH
    variant = bool(extra_flags) or out != LIB_PATH
    if not force and not variant and not stale():
        return LIB_PATH
    tag = "" if not variant else "_" + os.path.splitext(os.path.basename(out))[0]
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    import fcntl
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        # ranks that find the library missing at the same time build one after the other, not into each other's files
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not variant and not stale():
            return LIB_PATH  # another process built it while this one waited
        return _build_locked(force, verbose, extra_flags, out, variant, tag)


def _build_locked(force, verbose, extra_flags, out, variant, tag) -> str:
    cc = hipcc()
    jobs = []
    for unit in UNITS:
        if force or variant or _unit_stale(unit, tag):
            cmd = [cc] + FLAGS + UNIT_FLAGS.get(unit, []) + list(extra_flags) + ["-c", os.path.join(CSRC, unit), "-o", _obj(unit, tag)]
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    failed = None
    for cmd, p in jobs:  # every compiler is waited for, also after a failure: no orphans writing into lib/obj
        if p.wait() != 0 and failed is None:
            failed = (p.returncode, cmd)
            for _, q in jobs:
                if q.poll() is None:
                    q.terminate()
    if failed:
        raise subprocess.CalledProcessError(*failed)
    tmp = out + f".tmp{os.getpid()}"
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [_obj(u, tag) for u in UNITS]
    if verbose:
        print(" ".join(link))
    try:
        subprocess.check_call(link)
        os.replace(tmp, out)  # a process that has the old library mapped keeps its inode
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    if variant:
        for u in UNITS:
            os.remove(_obj(u, tag))
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
