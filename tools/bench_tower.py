"""Launch time of the fused recurrent_fn of the ResNet nets (mzs_resnet_tower with heads), one workgroup per
root against pair mode.  python tools/bench_tower.py [B ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)

g = torch.Generator().manual_seed(0)
mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
        mx.nn.ResNetDynamic(18, 21, generator=g))
m = mx.MuZero(*mods)
m.init(0, np.zeros((1, 84, 84, 4), np.float32))
d, pred = mods[2], mods[1]
for B in [int(x) for x in sys.argv[1:]] or [16, 64, 128]:
    s = torch.rand(B, 6, 6, 64, generator=g).cuda()
    a = torch.randint(0, 18, (B,), generator=g).cuda()
    row = []
    for pair in (False, True):
        d.use_pair_tower = pair
        for _ in range(5):
            out = d.hip_recurrent(pred, s, a, 10)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            out = d.hip_recurrent(pred, s, a, 10)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 50 * 1e3)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            out = d._tower_hip(s, a)  # tower only: no reward / prediction heads
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"B={B:4d}: one workgroup per root {row[0]:7.1f} us (tower only {row[1]:6.1f})   pair mode {row[2]:7.1f} us "
          f"(tower only {row[3]:6.1f})   status {d.pair_status()}", flush=True)
