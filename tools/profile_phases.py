"""In-kernel phase timing of the fused search kernel (s_memtime), via a -DMZ_PROFILE build.

    python tools/profile_phases.py build      # here (cross-compiles)  -> tools/bin/libmzsearch_prof.so
    python tools/profile_phases.py run        # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.environ.get("MZ_PROF_LIB", os.path.join(ROOT, "tools", "bin", "libmzsearch_prof.so"))  # variant A/Bs: MZ_PROF_LIB + MZ_PROF_FLAGS
PHASES = ["loop top", "select (jump words)", "network pass", "-", "expand stores", "backward + refresh",
          "bw: path entry", "bw: node loads", "bw: G chain", "bw: value update", "bw: scores",
          "bw: decide + JUMP scan", "prologue (x S)", "epilogue (x S)", "-", "-"]


def build():
    from muax_amd import _build
    print(_build.build(extra_flags=["-DMZ_PROFILE"] + os.environ.get("MZ_PROF_FLAGS", "").split(), out=LIB))


def run():
    import numpy as np
    import torch
    from muax_amd import _build, _lib
    _build.LIB_PATH = LIB
    _lib._lib = None
    import bench
    from muax_amd import MuZeroSearch, SearchConfig
    name = sys.argv[2] if len(sys.argv) > 2 else "cartpole"
    if ":" in name:  # "A:E:S:B" -- a shape of the profiling build's own instance list (MZ_PROF_FLAGS=-DMZ_INSTANCES_FILE=...)
        A, E, S, B = (int(x) for x in name.split(":"))
        obs_dim, support = 4, 10
    else:
        B, obs_dim, E, A, support, S = bench.WORKLOADS[name]
    w = bench.haiku_style_weights(0, obs_dim, E, A, 2 * support + 1)
    g = torch.Generator().manual_seed(1000)
    obs = (torch.rand(B, obs_dim, generator=g) * 2 - 1).cuda()
    noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
    s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
    s.set_mlp_weights(w, obs_dim, support)
    waves = (B + 3) // 4
    prof = torch.zeros(waves, 16, dtype=torch.int64, device="cuda")
    L = _lib.load()
    L.mzs_debug_profile.argtypes = [C.c_void_p, C.c_void_p]
    assert L.mzs_debug_profile(s._h, C.c_void_p(prof.data_ptr())) == 0
    for i in range(3):
        s.act_mlp(obs, (0, i), dirichlet_noise=noise)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    tot = p[:, :14].sum(1)
    print(f"waves {waves}; cycles per wave: mean {tot.mean():.0f} max {tot.max():.0f} min {tot.min():.0f}")
    depth = s.depth_sum.cpu().numpy().reshape(waves, 4)
    print(f"mean selection depth {depth.mean() / S:.2f}; per-wave sum of max-of-4 is not tracked here")
    for k, name in enumerate(PHASES[:14]):
        if name == "-":
            continue
        print(f"  {name:26s} {p[:, k].mean() / S:9.0f} cycles/sim  ({100 * p[:, k].sum() / tot.sum():5.1f}%)   "
              f"slowest wave {p[:, k].max() / S:9.0f}")
    print(f"near-tie evaluations per simulation (lane 0's row) {p[:, 14].mean() / S:.3f}, key-walk levels hashed per simulation "
          f"{p[:, 15].mean() / S:.3f}")
    slow = int(tot.argmax())
    print("slowest wave", slow, "phases/sim:", (p[slow] / S).round(0).tolist(), "depth sums", depth[slow].tolist())


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
