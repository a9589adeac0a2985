"""cProfile of the B=1 act() loop (host side of BASELINE config 1)."""
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
obs = np.zeros(4, np.float32)
key = mx.prng.PRNGKey(0)
for i in range(50):
    m.act(key, obs, with_pi=True, with_value=True, num_simulations=50)


def loop():
    k = key
    for i in range(500):
        k, sub = mx.prng.split(k)
        m.act(sub, obs, with_pi=True, with_value=True, num_simulations=50)


pr = cProfile.Profile()
pr.enable()
loop()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
