"""Where does the runtime stall the host?  N synced fused acts, per-step wall time; prints the steps above 1 ms.
    python tools/diag_stall.py [N] [events-per-step 0/1] [throwaway events recorded up front]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import haiku_style_weights  # noqa: E402
from muax_amd import MuZeroSearch, SearchConfig  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
EV = int(sys.argv[2]) if len(sys.argv) > 2 else 0
B, obs_dim, E, A, S = 4096, 4, 8, 2, 50
s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
s.set_mlp_weights(haiku_style_weights(0, obs_dim, E, A, 21), obs_dim, 10, 0.99)
obs = (torch.rand(B, obs_dim) * 2 - 1).cuda()
noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
PRE = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # throwaway events recorded up front (kept alive): grows the runtime's signal pool
pre = [torch.cuda.Event(enable_timing=True) for _ in range(PRE)]
for e in pre:
    e.record()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N if EV else 0)]
for a, b in evs:
    a.record(); b.record()
torch.cuda.synchronize()
ts = []
for i in range(N):
    t0 = time.perf_counter()
    if EV:
        evs[i][0].record()
    s.act_mlp(obs, (0, i), dirichlet_noise=noise)
    if EV:
        evs[i][1].record()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
slow = [(i, round(t * 1e3, 2)) for i, t in enumerate(ts) if t > 1e-3]
print(f"N={N} events={EV} pre-recorded={PRE}: median {sorted(ts)[N // 2] * 1e6:.1f} us; steps above 1 ms: {slow}")
