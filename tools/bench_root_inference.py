"""Root inference of the convolutional plugin nets (representation + prediction + value decode, muax/model.py:251-263)
on config 4's shard: torch expressions for the LayerNorm chains against the fused HIP calls (mzs_layernorm_act), eager
and as one hipGraph.  python tools/bench_root_inference.py [roots]"""
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
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""  # "resnet_fused" / "ez_fused": one variant only (for a kernel trace)
for name, make in (("ResNet", lambda g: (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
                                         mx.nn.ResNetDynamic(18, 21, generator=g))),
                   ("EZ", lambda g: (mx.nn.EZRepresentation(32, generator=g), mx.nn.EZPrediction(18, 21, 1.0, generator=g),
                                     mx.nn.EZDynamic(32, 18, 21, 1.0, generator=g)))):
    g = torch.Generator().manual_seed(0)
    mods = make(g)
    obs = torch.randint(0, 256, (B, 84, 84, 4), generator=g).float().cuda()
    if ONLY and name != {"resnet_fused": "ResNet", "ez_fused": "EZ"}.get(ONLY, "ResNet"):
        continue
    for fused in ((True,) if ONLY else (False, True)):
        mx.nn.HkLayerNorm.use_hip = fused
        for cap in ((False,) if ONLY else (False, True)):
            m = mx.MuZero(*mods, capture_graph=cap)
            m.init(0, np.zeros((1, 84, 84, 4), np.float32))
            for _ in range(3):
                m._root_inference(m.params, None, obs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                m._root_inference(m.params, None, obs)
            torch.cuda.synchronize()
            print(f"{name:6s} roots={B} LayerNorm chains {'fused HIP' if fused else 'torch    '} "
                  f"{'hipGraph' if cap else 'eager   '}: {(time.perf_counter() - t0) / n * 1e3:7.3f} ms", flush=True)
mx.nn.HkLayerNorm.use_hip = True
