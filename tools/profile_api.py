"""cProfile of MuZero.act() on the host: B=4096 NumPy in / NumPy out (the bench line's api.numpy) and B=1 S=10."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402

g = torch.Generator().manual_seed(0)
net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                      mx.nn.Dynamic(8, 2, 21, generator=g))
m = mx.MuZero(net)
m.init(0, np.zeros((1, 4)))
for B, S in ((4096, 50), (1, 10)):
    obs = np.random.default_rng(0).uniform(-1, 1, (B, 4)).astype(np.float32)
    for i in range(30):
        m.act(i, obs, obs_from_batch=True, num_simulations=S)

    def loop():
        for i in range(300):
            m.act(100 + i, obs, obs_from_batch=True, num_simulations=S)

    pr = cProfile.Profile()
    pr.enable()
    loop()
    pr.disable()
    print(f"===== B={B} S={S}")
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
