"""On-demand instances of the fused act() kernel (muax_amd/csrc/mz_fused_jit.hip).

libmzsearch.so carries the (num_actions, embedding_dim, support slots, tree size) instances listed in
muax_amd/csrc/mz_instances.def; the reference's act() takes ANY num_simulations / network widths (muax/model.py:82-96,
muax/nn.py:59-115).  When mzs_act_mlp has no instance for a shape and a hipcc is present, `ensure_instance` compiles one
translation unit for that shape (~25 s, once: cached as muax_amd/lib/jit/<shape>-<source hash>.so), loads it and registers
it with the library (mzs_register_fused_dispatch) -- the shape then runs as ONE launch per act() like a listed one, same
kernel source, same bits.  Shapes outside the kernel's own limits (more than 16 actions, more than 255 simulations,
embeddings wider than 64) cannot be instantiated: they take the one-launch generic search (mzs_mlp_search) or the
step-wise path.  MUAX_AMD_JIT=0 turns the on-demand build off.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

from . import _build

JIT_DIR = os.path.join(_build.LIB_DIR, "jit")
_SOURCES = ["mz_fused_jit.hip", "mz_fused_group.inc", "mz_fused_launch.h", "mz_fused.cuh", "mz_spec.cuh", "mz_host.h"]
_loaded = {}   # shape -> CDLL (kept alive: the library calls into it)
_failed = set()
_TRAIN_SOURCES = ["mz_train_jit.hip", "mz_train.cuh", "mz_spec.cuh", "mz_host.h"]
# Bump when plan() changes its answer for any shape: the number is part of every cached file's name (and this file is
# hashed into it as well), so a cache directory that survives a planner change cannot serve an instance whose record
# kind / roots per workgroup differ from what the planner now says.
PLAN_VERSION = 2
last_build_log = None  # path of the compiler log of the most recent FAILED on-demand build (surfaced in MuZero's warning)


def _ceil_log2(n: int) -> int:
    return 0 if n <= 1 else 1 + _ceil_log2((n + 1) // 2)


def lds_bytes(A: int, E: int, NMAX: int, WAVES: int, long_paths: bool = False, gumbel: bool = True) -> int:
    """FusedCfg::LDS_BYTES of a plain (non-compact) instance -- mz_fused.cuh -- for the widest record the instance is
    compiled with: `gumbel` = the Gumbel modes too (five words per child: the prior logit; the raw value in the header;
    root noise behind the tree), else the MuZero policy's modes only (four words per child; round 6: an on-demand
    instance is planned per policy).  `long_paths`: FusedCfg::LONG -- root paths (and embeddings) in HBM."""
    selw = ((2 * A + 3) // 4) * 4
    st0 = selw + (4 if (gumbel or not long_paths) else 3)  # HDRW: the raw value leaves a PH_ record unless MODE 3 reads it
    path0 = st0 + (5 if gumbel else 4) * A + (E if (E <= 16 and not long_paths) else 0)  # (LONG: embeddings in HBM too)
    entry = 8 if _ceil_log2(NMAX) + max(1, _ceil_log2(A)) <= 8 else 16
    pathw = (NMAX * entry + 31) // 32
    ns = (path0 + (0 if long_paths else pathw)) | 1
    tree = ((ns * NMAX + 3) // 4) * 4 + (((A + 3) // 4) * 4 if (long_paths and gumbel) else 0)
    root = tree + ((8 - tree % 32 + 32) % 32)
    tbl = 2 * (((NMAX + 2 + 3) // 4) * 4)
    return 4 * (tbl + 4 * WAVES * root)


def plan(A: int, E: int, F: int, S: int, gumbel: bool = True):
    """(FS, NMAX, WAVES, LONG) of an instance that serves the shape, or None when the kernel's own limits rule it out:
    A <= 16 (all scores of a node in one lane's registers; four action bits per JUMP word), S <= 255 (depths are bytes).
    Up to 127 simulations the nodes' root paths may live in the LDS record; beyond -- or when that lets a workgroup
    hold more roots -- in HBM (FusedCfg::LONG).  `gumbel` False: an instance of the MuZero policy's modes only, sized for
    their four-word children (e.g. CartPole's shape at 160 simulations: 16 roots per workgroup instead of 12)."""
    if not (1 <= A <= 16 and 17 <= F <= 63 and 1 <= S <= 255):
        return None
    if E < 1 or E > 64:  # (round 5: any width up to 64; widths above 16 that are no multiple of 8 take the packed-fma
        return None      # first layers instead of the eight-input v_fmac_f32_dpp blocks)
    FS = 2 if F <= 32 else 4
    for NMAX in (51, 64, 101, 128, 161, 201, 256):
        if S + 1 > NMAX:
            continue
        entry = 8 if _ceil_log2(NMAX) + max(1, _ceil_log2(A)) <= 8 else 16
        pathw = (NMAX * entry + 31) // 32
        best = None
        for long_paths in (False, True):  # (paths in LDS are the faster record: taken unless HBM paths hold more roots)
            # (ties go to the LDS record, the faster one; the LONG record is taken where it holds more roots per workgroup)
            if (pathw + 15) // 16 > (8 if long_paths else 4) or (A > pathw and not long_paths):
                continue
            for W in (4, 3, 2, 1):
                if lds_bytes(A, E, NMAX, W, long_paths, gumbel) <= 160 * 1024:
                    if best is None or W > best[2]:
                        best = (FS, NMAX, W, long_paths)
                    break
        if best is not None:
            return best
    return None


def _source_hash() -> str:
    # everything the side library's code depends on: its sources, the ABI header, the compiler flags (they name the
    # offload arch) -- a cache directory shared across flag or header changes must not serve a stale instance
    h = hashlib.sha256()
    for f in _SOURCES + [_build._ABI]:
        with open(os.path.join(_build.CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.abspath(__file__), "rb") as fh:  # the planner and the compile recipe live here
        h.update(fh.read())
    h.update("\0".join(_build.FLAGS + _build.UNIT_FLAGS["mz_fused_g0.hip"]).encode())
    h.update(f"plan{PLAN_VERSION}".encode())
    return h.hexdigest()[:12]


def _run_compiler(cmd, log, verbose) -> bool:
    """hipcc with its output kept: `log` holds the command and everything the compiler said (a failed on-demand build
    used to be silent); the file stays after a failure and is removed after a success."""
    global last_build_log
    with open(log, "w") as lf:
        lf.write(" ".join(cmd) + "\n")
        lf.flush()
        rc = subprocess.call(cmd, stdout=lf, stderr=subprocess.STDOUT)
    if verbose:
        print(open(log).read())
    if rc != 0:
        last_build_log = log
        return False
    os.remove(log)
    return True


def build_log_tail(lines: int = 6) -> str:
    """The last lines of the most recent failed on-demand build's compiler log ('' when none failed)."""
    if not last_build_log or not os.path.exists(last_build_log):
        return ""
    with open(last_build_log, errors="replace") as f:
        return "".join(f.readlines()[-lines:]).strip()


def _compile(cc, so, tag, A, E, FS, NMAX, W, LONG, gumbel, verbose, extra_flags=()) -> bool:
    """One translation unit for one shape -> `so` (under the cache directory's file lock).  OSError propagates."""
    import fcntl
    with open(os.path.join(JIT_DIR, ".jit.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)  # ranks that miss the same shape build it once
        if os.path.exists(so):
            return True
        deff = os.path.join(JIT_DIR, f"inst_{tag}.def")
        with open(deff, "w") as f:
            f.write(f"MZS_INST(100, {A}, {E}, {FS}, {NMAX}, {W}, {'2' if LONG else 'false'})\n")
        tmp = so + f".tmp{os.getpid()}"
        cmd = [cc] + _build.FLAGS + _build.UNIT_FLAGS["mz_fused_g0.hip"] + [
            f'-DMZ_INSTANCES_FILE="{deff}"', "-DMZ_FUSED_GROUP=100", f"-DMZ_FUSED_MUZERO_ONLY={0 if gumbel else 1}"] + \
            list(extra_flags) + ["-shared", os.path.join(_build.CSRC, "mz_fused_jit.hip"), "-o", tmp]
        try:
            if not _run_compiler(cmd, os.path.join(JIT_DIR, f"mzfused_{tag}.log"), verbose):
                return False
            os.replace(tmp, so)
            return True
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)


def instance_tag(A: int, E: int, FS: int, NMAX: int, W: int, LONG: bool, gumbel: bool) -> str:
    """File tag of an on-demand instance: EVERYTHING the planner decided (record kind and policy class included) and the
    hash of everything the code depends on (sources, ABI header, flags, this file, PLAN_VERSION)."""
    return f"a{A}_e{E}_fs{FS}_n{NMAX}_w{W}_l{int(LONG)}_p{int(gumbel)}-{_source_hash()}"


def instance_file(A: int, E: int, F: int, S: int, gumbel: bool = True):
    """Path of the side library ensure_instance() would load for the shape (None when no instance can exist)."""
    if not gumbel and plan(A, E, F, S, False) == plan(A, E, F, S, True):
        gumbel = True
    pl = plan(A, E, F, S, gumbel)
    return None if pl is None else os.path.join(JIT_DIR, f"mzfused_{instance_tag(A, E, *pl, gumbel)}.so")


def ensure_instance(A: int, E: int, F: int, S: int, verbose: bool = False, gumbel: bool = True, extra_flags=()) -> bool:
    """Make sure mzs_act_mlp can serve (A, E, F, S) with one launch: True when an on-demand instance is registered
    (built now or earlier), False when the shape cannot be instantiated, there is no compiler, the build fails (its
    compiler output is then in muax_amd/lib/jit/mzfused_<tag>.log, see build_log_tail()) or MUAX_AMD_JIT=0.
    `gumbel`: the policy class the instance is planned for -- True (default): every mode, sized for the Gumbel modes'
    five-word children; False: the MuZero policy's modes only, taken where that record holds more roots per workgroup
    (where it does not, the all-modes instance is used: one file per shape).  `extra_flags`: more compiler flags (tests)."""
    if os.environ.get("MUAX_AMD_JIT", "1") == "0":
        return False
    if not gumbel and plan(A, E, F, S, False) == plan(A, E, F, S, True):
        gumbel = True  # nothing to gain from a policy-specific instance
    pl = plan(A, E, F, S, gumbel)
    if pl is None:
        return False
    FS, NMAX, W, LONG = pl
    shape = (A, E, FS, NMAX, W, bool(LONG), bool(gumbel)) + tuple(extra_flags)
    if shape in _loaded:
        return True
    if shape in _failed:
        return False
    try:
        cc = _build.hipcc()
    except RuntimeError:
        return False
    tag = instance_tag(A, E, FS, NMAX, W, LONG, gumbel)
    if extra_flags:
        tag += "-x" + hashlib.sha256("\0".join(extra_flags).encode()).hexdigest()[:6]
    so = os.path.join(JIT_DIR, f"mzfused_{tag}.so")
    try:  # a read-only install (no cache directory, no lock file) is "no instance", not an exception out of act()
        os.makedirs(JIT_DIR, exist_ok=True)
        if not os.path.exists(so) and not _compile(cc, so, tag, A, E, FS, NMAX, W, LONG, gumbel, verbose, extra_flags):
            _failed.add(shape)
            return False
    except OSError:
        _failed.add(shape)
        return False
    from . import _lib
    L = _lib.load()
    side = C.CDLL(so)
    side.mzs_jit_dispatch.restype = C.c_void_p
    side.mzs_jit_abi.restype = C.c_int
    register = L.mzs_register_fused_dispatch if gumbel else L.mzs_register_fused_dispatch_muzero
    _lib.check(register(C.c_void_p(side.mzs_jit_dispatch()), side.mzs_jit_abi()))
    _loaded[shape] = side
    return True


def _train_hash() -> str:
    h = hashlib.sha256()
    for f in _TRAIN_SOURCES + [_build._ABI]:
        with open(os.path.join(_build.CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.abspath(__file__), "rb") as fh:
        h.update(fh.read())
    h.update("\0".join(_build.FLAGS).encode())
    return h.hexdigest()[:12]


def ensure_train_instance(A: int, E: int, F: int, verbose: bool = False) -> bool:
    """Make sure mzs_mlp_loss_grad can serve (num_actions, embedding_dim, F = 2 support_size + 1): True when an on-demand
    instance of the fused training-step kernel (muax_amd/csrc/mz_train_jit.hip) is registered -- built now (one
    translation unit, cached as muax_amd/lib/jit/mztrain_<shape>-<source hash>.so) or earlier --, False when the shape is
    outside the kernel's limits (more than 16 actions, embeddings wider than 64, F outside 17..63), there is no compiler,
    the build fails, or MUAX_AMD_JIT=0.  Round 5: a model whose act() runs on an on-demand instance keeps the fused
    update() too (the reference's update() takes any widths, muax/model.py:181-201)."""
    if os.environ.get("MUAX_AMD_JIT", "1") == "0":
        return False
    if not (1 <= A <= 16 and 1 <= E <= 64 and 17 <= F <= 63):
        return False
    shape = ("train", A, E, F)
    if shape in _loaded:
        return True
    if shape in _failed:
        return False
    try:
        cc = _build.hipcc()
    except RuntimeError:
        return False
    so = os.path.join(JIT_DIR, f"mztrain_a{A}_e{E}_f{F}-{_train_hash()}.so")
    try:
        os.makedirs(JIT_DIR, exist_ok=True)
        if not os.path.exists(so):
            import fcntl
            with open(os.path.join(JIT_DIR, ".jit.lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)  # ranks that miss the same shape build it once
                if not os.path.exists(so):
                    tmp = so + f".tmp{os.getpid()}"
                    cmd = [cc] + _build.FLAGS + [f"-DMZ_TRAIN_A={A}", f"-DMZ_TRAIN_E={E}", f"-DMZ_TRAIN_F={F}", "-shared",
                                                 os.path.join(_build.CSRC, "mz_train_jit.hip"), "-o", tmp]
                    try:
                        if not _run_compiler(cmd, so[:-3] + ".log", verbose):
                            _failed.add(shape)
                            return False
                        os.replace(tmp, so)
                    finally:
                        if os.path.exists(tmp):
                            os.remove(tmp)
    except OSError:
        _failed.add(shape)
        return False
    from . import _lib
    L = _lib.load()
    side = C.CDLL(so)
    side.mzs_jit_train_abi.restype = C.c_int
    got = (C.c_int32(), C.c_int32(), C.c_int32())
    side.mzs_jit_train_shape(*(C.byref(x) for x in got))
    if tuple(x.value for x in got) != (A, E, F):  # the file name said (A, E, F); the code inside must agree
        raise RuntimeError(f"{so}: built for shape {tuple(x.value for x in got)}, asked for {(A, E, F)}")
    fn = C.cast(side.mzs_jit_train_launch, C.c_void_p)
    _lib.check(L.mzs_register_train_dispatch(fn, A, E, F, side.mzs_jit_train_abi()))
    _loaded[shape] = side
    return True
