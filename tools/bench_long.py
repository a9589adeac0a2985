"""Long searches / wide action sets of the default trio through mzs_act_mlp: per-act time at several batch sizes
(median of 20 synchronised acts).   python tools/bench_long.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import haiku_style_weights  # noqa: E402
from muax_amd import MuZeroSearch, SearchConfig, _jit  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()
for (A, E, od, S) in ((2, 8, 4, 63), (2, 8, 4, 100), (2, 8, 4, 127), (2, 8, 4, 160), (2, 8, 4, 255), (4, 32, 8, 100), (4, 32, 8, 200),
                      (6, 8, 4, 100), (9, 8, 4, 50), (12, 8, 4, 50), (16, 8, 4, 50)):
    row = []
    for B in (64, 256, 1024, 4096):
        _jit.ensure_instance(A, E, 21, S)
        s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
        s.set_mlp_weights(haiku_style_weights(0, od, E, A, 21), od, 10, 0.99)
        obs = (torch.rand(B, od) * 2 - 1).cuda()
        noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
        for i in range(4):
            s.act_mlp(obs, (0, i), dirichlet_noise=noise)
        torch.cuda.synchronize()
        ts = []
        for i in range(20):
            t0 = time.perf_counter()
            s.act_mlp(obs, (0, 10 + i), dirichlet_noise=noise)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        row.append(sorted(ts)[10] * 1e3)
        s.close()
    print(f"A={A:2d} E={E:2d} S={S:3d} plan {_jit.plan(A, E, 21, S)}: " + "  ".join(f"B={B}: {t:7.3f} ms" for B, t in zip((64, 256, 1024, 4096), row))
          + f"   | 4096 roots: {row[-1] / S * 1e3:6.2f} us/sim", flush=True)
