"""Step-wise (plugin nets) path: eager launches vs the hipGraph-captured simulation loop.

    python tools/bench_stepwise.py [B] [S]      # on the GPU box
The nets are the default MLP trio wrapped so the fused kernel does not claim them: per simulation the
path runs two HIP tree kernels plus the plugin's own torch kernels, so it is launch-bound."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)


class Wrap(torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward(self, *a):
        return self.inner(*a)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    g = torch.Generator().manual_seed(0)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                          mx.nn.Dynamic(8, 2, 21, generator=g))
    mods = [Wrap(x) for x in net]
    obs = torch.rand(B, 4, generator=g).cuda() * 2 - 1
    noise = torch.distributions.Dirichlet(torch.full((2,), 0.3)).sample((B,)).cuda()
    for name, cap in (("eager", False), ("hipGraph", True)):
        m = mx.MuZero(*mods, capture_graph=cap)
        m.init(0, np.zeros((1, 4)))
        kw = dict(obs_from_batch=True, num_simulations=S, dirichlet_noise=noise, device_outputs=True)
        for i in range(3):
            m.act(i, obs, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for i in range(n):
            m.act(10 + i, obs, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:9s} B={B} S={S}: {dt * 1e3:8.3f} ms/act  {B / dt / 1e6:7.3f} M env-steps/s")


if __name__ == "__main__":
    main()
