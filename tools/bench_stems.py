"""The stride-2 stems of the representation nets on 128 frames: mzs_conv3x3_stride2_nhwc against the library path.
    python tools/bench_stems.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)

B = 128
for cin, cout, H, div in ((4, 32, 84, 255.0), (4, 32, 84, None), (32, 64, 42, None)):
    g = torch.Generator().manual_seed(0)
    conv = mx.nn.HkConv2D(cout, 3, 2, in_channels=cin, generator=g).cuda()
    x = torch.rand(B, H, H, cin, generator=g).cuda()
    out = {}
    with torch.no_grad():
        for hip in (False, True):
            conv.use_hip = hip
            for _ in range(5):
                conv.scaled(x, div, relu=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                conv.scaled(x, div, relu=True)
            e1.record()
            torch.cuda.synchronize()
            out[hip] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"stem {B} x {H}x{H}x{cin} -> {cout}, stride 2, / {div}, relu: library path {out[False]:7.1f} us   mzs_conv3x3_stride2_nhwc {out[True]:7.1f} us")
