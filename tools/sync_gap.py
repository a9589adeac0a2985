"""Where the time of ONE synchronised fused act() goes on the metric's workload (4096 CartPole roots x 50 simulations):
the host's part of the call (key derivation, argument block, launch), the kernel, and the completion's way back to the
host.  `value` of bench.py is roots / (all three); `roofline.kernel_ms` is the second alone.

    python tools/sync_gap.py [acts]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from muax_amd import MuZeroSearch, SearchConfig  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    warm_runtime()
    B, obs_dim, E, A, support, S = bench.WORKLOADS["cartpole"]
    w = bench.haiku_style_weights(0, obs_dim, E, A, 2 * support + 1)
    g = torch.Generator().manual_seed(1000)
    obs = (torch.rand(B, obs_dim, generator=g) * 2 - 1).cuda()
    noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
    s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
    s.set_mlp_weights(w, obs_dim, support, 0.99)
    for i in range(300):  # clocks settled
        s.act_mlp(obs, (0, i), dirichlet_noise=noise)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:  # (an event's first record allocates its signal)
        a.record()
        b.record()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    for name, sync in (("torch.cuda.synchronize()", torch.cuda.synchronize), ("stream.synchronize()", stream.synchronize)):
        call, total = [], []
        for i in range(n):
            t0 = time.perf_counter()
            s.act_mlp(obs, (1, i), dirichlet_noise=noise)
            t1 = time.perf_counter()
            sync()
            t2 = time.perf_counter()
            call.append(t1 - t0)
            total.append(t2 - t0)
        print(f"{name}: act + sync {sorted(total)[n // 2] * 1e6:.1f} us, of which the call {sorted(call)[n // 2] * 1e6:.1f} us")
    kern = []
    for i, (a, b) in enumerate(ev):  # the kernel alone: events around launches of a busy stream
        a.record()
        s.act_mlp(obs, (2, i), dirichlet_noise=noise)
        b.record()
    torch.cuda.synchronize()
    kern = sorted(a.elapsed_time(b) for a, b in ev)
    med = lambda x: sorted(x)[len(x) // 2] * 1e6  # noqa: E731
    k = kern[len(kern) // 2] * 1e3
    print(f"synchronised act(): {med(total):.1f} us = call returns after {med(call):.1f} us (host: key walk, argument block, "
          f"launch) + kernel {k:.1f} us (events, busy stream) + {med(total) - med(call) - k:.1f} us (launch-to-start and "
          f"completion-to-host, minus what of the kernel overlaps the call)")
    print(f"  -> {B / med(total):.2f} M env-steps/s synchronised; the kernel alone would be {B / k:.2f} M")


if __name__ == "__main__":
    main()
