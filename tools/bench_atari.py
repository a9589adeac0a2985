"""BASELINE config 4 shape on ONE GPU's shard: Atari 84x84x4 frames, ResNet plugin nets, A=18, S=200, 128 roots
(1024 roots / 8 GPUs).  The nets are torch modules (interim, see muax_amd/nn.py); the tree kernels are ours.

    python tools/bench_atari.py [roots] [S]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    g = torch.Generator().manual_seed(0)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    obs = torch.randint(0, 256, (B, 84, 84, 4), generator=g).float().cuda()
    for name, cap in (("eager", False), ("hipGraph", True)):
        m = mx.MuZero(*mods, capture_graph=cap)
        m.init(0, np.zeros((1, 84, 84, 4), np.float32))
        kw = dict(obs_from_batch=True, num_simulations=S, device_outputs=True)
        for i in range(2):
            m.act(i, obs, **kw)
        torch.cuda.synchronize()
        n = 3
        t0 = time.perf_counter()
        for i in range(n):
            m.act(10 + i, obs, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        h = list(m._policy._handles.values())[0]
        depth = float(h.depth_sum.float().mean()) / S
        print(f"{name:9s} roots={B} S={S} A=18 E=2304: {dt * 1e3:9.2f} ms/act  {B / dt:10.1f} env-steps/s  "
              f"(mean selection depth {depth:.1f})")


if __name__ == "__main__":
    main()
