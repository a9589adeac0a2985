"""Per-layer time of the ResNet representation net (config 4's root inference, torch/MIOpen): which convolutions
fall back to MIOpen's naive NHWC kernel.  python tools/prof_repr.py [roots]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g = torch.Generator().manual_seed(0)
rep = mx.nn.ResNetRepresentation(32, generator=g)
obs = torch.randint(0, 256, (B, 84, 84, 4), generator=g).float()
with torch.no_grad():
    rep(obs[:1])
rep.cuda()
obs = obs.cuda()
rows = []
orig = mx.nn.HkConv2D.forward


def timed(self, x):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = orig(self, x)
    torch.cuda.synchronize()
    rows.append((tuple(x.shape), self.k, self.stride, self.out_channels, (time.perf_counter() - t0) * 1e3))
    return y


with torch.no_grad():
    for _ in range(3):
        rep(obs)
    mx.nn.HkConv2D.forward = timed
    rep(obs)
    mx.nn.HkConv2D.forward = orig
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        rep(obs)
    torch.cuda.synchronize()
    print(f"whole representation net, {B} roots: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
for r in rows:
    print(f"  in {r[0]} k={r[1]} stride={r[2]} -> {r[3]} ch: {r[4]:.3f} ms")
