"""mzs_conv3x3_nhwc against the library's convolution on the shapes of config 4's root inference (128 roots).
    python tools/bench_repr_conv.py [roots]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for C, H in ((64, 21), (64, 11), (64, 6), (32, 21), (32, 11), (32, 42)):
    g = torch.Generator().manual_seed(0)
    conv = mx.nn.HkConv2D(C, 3, 1, in_channels=C, generator=g).cuda()
    x = torch.rand(B, H, H, C, generator=g).cuda()
    out = {}
    for hip in (False, True):
        conv.use_hip = hip
        with torch.no_grad():
            if hip and not conv._hip_ok(x):
                out[hip] = float("nan")
                continue
            for _ in range(5):
                conv(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                conv(x)
            e1.record()
            torch.cuda.synchronize()
        out[hip] = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2 * B * H * H * 9 * C * C
    print(f"{B} x {H}x{H}x{C}: library {out[False]:7.1f} us ({fl / out[False] / 1e6:5.1f} TFLOP/s)   mzs_conv3x3_nhwc {out[True]:7.1f} us "
          f"({fl / out[True] / 1e6:5.1f} TFLOP/s)")

# whole residual blocks: mzs_resblock_v1 (three launches) against the module path (three convolutions + two LayerNorm
# chains = seven launches)
for C, H, proj in ((32, 42, True), (64, 21, True), (64, 11, True), (64, 21, False), (64, 11, False), (64, 24, False), (64, 12, False)):
    g = torch.Generator().manual_seed(0)
    blk = mx.nn.ResidualConvBlockV1(C, 1, proj, generator=g)
    x = torch.rand(B, H, H, C, generator=g)
    with torch.no_grad():
        blk(x[:1])
        blk.cuda()
        x = x.cuda()
        out = {}
        for hip in (False, True):
            blk.use_hip = hip
            for _ in range(5):
                blk(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                blk(x)
            e1.record()
            torch.cuda.synchronize()
            out[hip] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"block {B} x {H}x{H}x{C} {'projected' if proj else 'identity '} shortcut: module path {out[False]:7.1f} us   mzs_resblock_v1 {out[True]:7.1f} us")
