"""cProfile of MuZero.update() on the host (fused HIP loss + gradient kernels, torch optimiser): B=4096, k=10."""
import cProfile
import os
import pstats
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402

B, L = 4096, 10
g = torch.Generator().manual_seed(0)
rng = np.random.default_rng(0)
batch = mx.Transition(obs=torch.rand(B, L, 4).cuda(), a=torch.randint(0, 2, (B, L)).cuda(),
                      r=torch.rand(B, L).cuda(), Rn=(torch.rand(B, L) * 20).cuda(),
                      pi=torch.as_tensor(rng.dirichlet([1, 1], (B, L)).astype(np.float32)).cuda())
net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                      mx.nn.Dynamic(8, 2, 21, generator=g))
m = mx.MuZero(net)
m.init(0, np.zeros((1, 4)))
for _ in range(20):
    m.update(batch)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    m.update(batch)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
