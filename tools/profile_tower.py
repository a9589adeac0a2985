"""In-kernel phase timing of the fused recurrent kernel of the ResNet nets (s_memtime), via a -DMZ_PROFILE build.

    python tools/profile_tower.py build      # here (cross-compiles)  -> tools/bin/libmzsearch_prof.so
    python tools/profile_tower.py run [B]    # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libmzsearch_prof.so")
PHASES = ["LDS init + state load", "reward head (one workgroup per root: up front)", "stem", "conv pass A (projection + conv_0)",
          "moments (both passes)", "message stores + post (both)", "wait for the partner (both)",
          "normalise + boundary + store A", "conv pass B (conv_1)", "normalise + boundary + residual + store B",
          "min-max, message C, y", "heads after the tower", "-", "-", "-", "whole workgroup"]


def build():
    from muax_amd import _build
    print(_build.build(extra_flags=["-DMZ_PROFILE"], out=LIB))


def run():
    import numpy as np
    import torch
    from muax_amd import _build, _lib
    _build.LIB_PATH = LIB
    _lib._lib = None
    import muax_amd as mx
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    g = torch.Generator().manual_seed(0)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    m = mx.MuZero(*mods)
    m.init(0, np.zeros((1, 84, 84, 4), np.float32))
    d, pred = mods[2], mods[1]
    s = torch.rand(B, 6, 6, 64, generator=g).cuda()
    a = torch.randint(0, 18, (B,), generator=g).cuda()
    L = _lib.load()
    L.mzs_debug_tower_profile.argtypes = [C.c_void_p, C.c_int32]
    buf = (C.c_uint64 * (1024 * 16))()
    n = 20
    for pair in (False, True):
        d.use_pair_tower = pair
        for _ in range(3):
            d.hip_recurrent(pred, s, a, 10)
        torch.cuda.synchronize()
        L.mzs_debug_tower_profile(buf, 1024 * 16)  # clear
        for _ in range(n):
            d.hip_recurrent(pred, s, a, 10)
        torch.cuda.synchronize()
        assert L.mzs_debug_tower_profile(buf, 1024 * 16) == 0
        p = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64) / n
        nwg = 2 * 8 * ((B + 7) // 8) if pair else B
        p = p[:nwg]
        p = p[p[:, 15] > 0]
        print(f"## {'pair mode' if pair else 'one workgroup per root'}, {B} roots, {len(p)} workgroups; microseconds at 2.4 GHz (s_memtime counts shader clocks)")
        groups = [("all", p)] if not pair else [("half 0 (pixels 0..15, prediction heads)", p[(np.arange(len(p)) // 8) % 2 == 0]),
                                                ("half 1 (pixels 16..35, reward head)", p[(np.arange(len(p)) // 8) % 2 == 1])]
        for name, q in groups:
            print(f"# {name}: whole workgroup {q[:, 15].mean() / 2400:.1f} us")
            for k, ph in enumerate(PHASES[:12]):
                if q[:, k].mean() > 0:
                    print(f"   {ph:48s} {q[:, k].mean() / 2400:7.2f} us")


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
