"""EfficientZero-style plugin nets (muax/nn.py:180-309) through MuZero.act(): 128 roots x 50 simulations, A = 18, 84x84x4
frames, the search loop as one hipGraph -- recurrent_fn as torch modules between the tree kernels against the
one-launch HIP kernel (mzs_ez_recurrent).  python tools/bench_ez.py [roots] [S] [channels]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
C = int(sys.argv[3]) if len(sys.argv) > 3 else 32
g = torch.Generator().manual_seed(0)
mods = (mx.nn.EZRepresentation(C, generator=g), mx.nn.EZPrediction(18, 21, 1.0, generator=g),
        mx.nn.EZDynamic(C, 18, 21, 1.0, generator=g))
obs = torch.randint(0, 256, (B, 84, 84, 4), generator=g).float().cuda()
for hip in (False, True):
    mods[2].use_hip_recurrent = hip
    m = mx.MuZero(*mods, capture_graph=True)
    m.init(0, np.zeros((1, 84, 84, 4), np.float32))
    kw = dict(obs_from_batch=True, num_simulations=S, device_outputs=True)
    for i in range(2):
        m.act(i, obs, **kw)
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for i in range(n):
        m.act(10 + i, obs, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"EZ nets C={C} roots={B} S={S}: recurrent_fn {'one HIP launch' if hip else 'torch modules '}: {dt * 1e3:8.2f} ms/act "
          f"({dt / S * 1e6:7.1f} us per simulation)  {B / dt:9.1f} env-steps/s", flush=True)
if mods[2].use_hip_recurrent:
    s = torch.rand(B, 6, 6, C, generator=g).cuda()
    a = torch.randint(0, 18, (B,), generator=g).cuda()
    for _ in range(5):
        mods[2].hip_recurrent(mods[1], s, a, 10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        mods[2].hip_recurrent(mods[1], s, a, 10)
    e1.record()
    torch.cuda.synchronize()
    print(f"mzs_ez_recurrent alone: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per launch")
