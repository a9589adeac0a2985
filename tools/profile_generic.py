"""Tree-step phase timers (s_memtime, -DMZ_PROFILE build: python tools/profile_search.py build) of the generic one-launch
search of the default trio (mz_mlp_generic.cuh), per simulation.   python tools/profile_generic.py"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ["MUAX_AMD_LIB"] = "/root/repo/tools/bin/libmzsearch_prof.so"
os.environ["MZS_FORCE_GENERIC"] = "1"
from bench import haiku_style_weights
from muax_amd import MuZeroSearch, SearchConfig, _lib
L = _lib.load()
L.mzs_debug_generic_jump_profile.argtypes = [C.c_void_p, C.c_int32]
B, od, E, A, S = 4096, 4, 8, 2, 50
if len(sys.argv) > 1:  # python tools/profile_generic.py A [S] [E] [B]
    A = int(sys.argv[1]); S = int(sys.argv[2]) if len(sys.argv) > 2 else S; E = int(sys.argv[3]) if len(sys.argv) > 3 else E
    B = int(sys.argv[4]) if len(sys.argv) > 4 else B
w = haiku_style_weights(0, od, E, A, 21)
s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
s.set_mlp_weights(w, od, 10, 0.99)
s.allow_generic()
obs = (torch.rand(B, od) * 2 - 1).cuda()
noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
buf = (C.c_uint64 * (1024 * 8))()
for i in range(3): s.act_mlp(obs, (0, i), dirichlet_noise=noise)
torch.cuda.synchronize(); L.mzs_debug_generic_jump_profile(buf, 1024 * 8)
n = 5
import time
t0 = time.perf_counter()
for i in range(n): s.act_mlp(obs, (0, 10 + i), dirichlet_noise=noise)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
L.mzs_debug_generic_jump_profile(buf, 1024 * 8)
jp = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
# blockIdx & 1023 aliases 4 blocks per slot
jp = jp / (n * S * max(1, B // 1024))
names = ["path + expand", "per-level inputs", "chain", "values + write back", "decisions", "JUMP + stores", "next selection"]
print(f"B={B} A={A} E={E} S={S} depth {float(s.depth_sum.float().mean()) / S:.1f}: generic act {dt*1e3:.3f} ms = {dt/S*1e6:.2f} us/sim; tree step phases (us at 2.1 GHz, per sim per root): " + ", ".join(f"{nm} {jp[:, k].mean()/2100:.2f}" for k, nm in enumerate(names)), "sum", jp[:, :7].sum(1).mean()/2100)
