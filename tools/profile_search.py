"""In-kernel phase timing (s_memtime) of the one-launch ResNet search (mz_search_conv.hip) via a -DMZ_PROFILE build, and
end-to-end act() times of the routes it replaces, on config 4's shard (128 roots x 200 simulations).

    python tools/profile_search.py build      # here (cross-compiles)  -> tools/bin/libmzsearch_prof.so
    python tools/profile_search.py run [B] [S]    # on the GPU box
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# MZ_PROF_HEADS=1 (build AND run): slots 3..9 time the pieces of "heads after the tower" instead of the passes
HEADS = bool(os.environ.get("MZ_PROF_HEADS"))
LIB = os.path.join(ROOT, "tools", "bin", "libmzsearch_prof_heads.so" if HEADS else "libmzsearch_prof.so")
PHASES = ["LDS init + state load", "reward head (single: up front; pair half 0: front convs + one pixel per pass)", "stem", "conv pass A (projection + conv_0)",
          "moments (both passes)", "message stores + post (both)", "wait for the partner (both)",
          "normalise + boundary + store A", "conv pass B (conv_1)", "normalise + boundary + residual + store B",
          "min-max, message C, y", "heads after the tower",
          "half 0: wait for the reward | half 1: post the reward | single: -", "half 0 / single: TREE STEP | half 1: wait for the next selection",
          "half 0 / single: post the next selection", "recurrent_fn passes (sum)"]


if HEADS:
    PHASES[3:10] = ["heads: reward head's tail (partial sums -> vector -> logits -> decode)",
                    "heads: the partner's 20 pixels received and normalised", "-",
                    "heads: first 1x1 convolutions (64 -> 16, both heads)", "heads: value head's second 1x1 convolution",
                    "heads: flatten -> Linear(576 -> 16), both heads", "heads: last layers + value decode"]


def build():
    from muax_amd import _build
    print(_build.build(extra_flags=["-DMZ_PROFILE"] + (["-DMZ_PROF_HEADS"] if HEADS else []), out=LIB))


def act_times(mx, np, torch, B, S, deep):
    g = torch.Generator().manual_seed(0)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    go = g if deep else torch.Generator().manual_seed(100)
    obs = torch.randint(0, 256, (B, 84, 84, 4), generator=go).float().cuda()
    out = {}
    for name, env in (("one launch, pair", {}), ("one launch, 1 wg/root", {"MZS_TOWER_PAIR": "0"}),
                      ("per-simulation launches (hipGraph), pair", {"MZS_RESNET_SEARCH": "0"})):
        os.environ.update(env)
        m = mx.MuZero(*mods, capture_graph=True)
        m.init(0, np.zeros((1, 84, 84, 4), np.float32))
        kw = dict(obs_from_batch=True, num_simulations=S, device_outputs=True)
        for i in range(2):
            m.act(i, obs, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            m.act(10 + i, obs, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        h = list(m._policy._handles.values())[0]
        out[name] = (dt * 1e3, float(h.depth_sum.float().mean()) / S)
        for k in env:
            del os.environ[k]
    return out


def run():
    import numpy as np
    import torch
    from muax_amd import _build, _lib
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    import muax_amd as mx
    for deep in (True, False):
        for name, (ms, depth) in act_times(mx, np, torch, B, S, deep).items():
            print(f"act() {B} roots x {S} sims, mean selection depth {depth:5.1f}: {ms:8.2f} ms  [{name}]")
    if not os.path.exists(LIB):
        return
    _build.LIB_PATH = LIB
    _lib._lib = None
    L = _lib.load()
    L.mzs_debug_search_profile.argtypes = [C.c_void_p, C.c_int32]
    buf = (C.c_uint64 * (1024 * 16))()
    L.mzs_debug_jump_profile.argtypes = [C.c_void_p, C.c_int32]
    jbuf = (C.c_uint64 * (1024 * 8))()
    g = torch.Generator().manual_seed(0)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    obs = torch.randint(0, 256, (B, 84, 84, 4), generator=g).float().cuda()
    for pair in (True, False):
        mx.nn.ResNetDynamic.use_pair_tower = pair
        m = mx.MuZero(*mods)
        m.init(0, np.zeros((1, 84, 84, 4), np.float32))
        kw = dict(obs_from_batch=True, num_simulations=S, device_outputs=True)
        m.act(0, obs, **kw)
        torch.cuda.synchronize()
        L.mzs_debug_search_profile(buf, 1024 * 16)  # clear
        L.mzs_debug_jump_profile(jbuf, 1024 * 8)
        n = 2
        for i in range(n):
            m.act(1 + i, obs, **kw)
        torch.cuda.synchronize()
        assert L.mzs_debug_search_profile(buf, 1024 * 16) == 0
        p = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.float64) / (n * S)
        nwg = 2 * 8 * ((B + 7) // 8) if pair else B
        p = p[:nwg]
        p = p[p[:, 15] > 0]
        print(f"## one-launch search, {'pair mode' if pair else 'one workgroup per root'}, {B} roots x {S} simulations, {len(p)} "
              f"workgroups; microseconds PER SIMULATION at 2.4 GHz")
        groups = [("all", p)] if not pair else [("half 0 (pixels 0..15, prediction heads, tree)", p[(np.arange(len(p)) // 8) % 2 == 0]),
                                                ("half 1 (pixels 16..35, reward head)", p[(np.arange(len(p)) // 8) % 2 == 1])]
        for name, q in groups:
            print(f"# {name}: sum {q[:, [12, 13, 14, 15]].sum(1).mean() / 2400:.1f} us")
            for k, ph in enumerate(PHASES):
                if q[:, k].mean() > 0:
                    print(f"   {ph:80s} {q[:, k].mean() / 2400:7.2f} us   (max over workgroups {q[:, k].max() / 2400:7.2f})")
        assert L.mzs_debug_jump_profile(jbuf, 1024 * 8) == 0
        jp = np.frombuffer(jbuf, dtype=np.uint64).reshape(1024, 8).astype(np.float64) / (n * S)
        jp = jp[jp.sum(1) > 0]
        # (pair mode: path and inputs are prefetched in idle convolution passes and the chain runs beside the expansion on
        # the second wavefront -- slots 1 and 2 stay empty there)
        names = ["path + expand (pair mode: || chain)", "per-level inputs", "discounted-return chain", "new values + write back",
                 "decisions of the path",
                 "JUMP records (pointer jumping) + stores", "next selection"]
        print(f"# tree step by phase ({len(jp)} workgroups): " + ", ".join(f"{nm} {jp[:, k].mean() / 2400:.2f}" for k, nm in enumerate(names))
              + (f" [slot 7: {jp[:, 7].mean() / 2400:.2f}]" if jp[:, 7].any() else ""))


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
