"""BASELINE config 1 shape: one root (B=1), CartPole MLP, S=10 and S=50 -- the latency of a single act() as
muax.fit drives it (host overhead + one kernel + the .item() sync of the reference's contract)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)

g = torch.Generator().manual_seed(0)
net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                      mx.nn.Dynamic(8, 2, 21, generator=g))
m = mx.MuZero(net)
m.init(0, np.zeros((1, 4)))
obs = np.zeros(4, np.float32)
key = mx.prng.PRNGKey(0)
for S in (10, 50):
    for i in range(20):
        m.act(key, obs, with_pi=True, with_value=True, num_simulations=S)
    n = 300
    t0 = time.perf_counter()
    for i in range(n):
        key, sub = mx.prng.split(key)
        m.act(sub, obs, with_pi=True, with_value=True, num_simulations=S)
    dt = (time.perf_counter() - t0) / n
    print(f"B=1 S={S}: {dt * 1e6:8.1f} us per act() incl. key split, host sync")
