"""Training step (BASELINE config 5 shape: 4096 trajectories, k_steps=10, default MLP trio): the fused HIP
forward+backward (mzs_mlp_loss_grad) vs the torch autograd route, loss+gradients only and whole update().

    python tools/bench_train.py [B] [L]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    g = torch.Generator().manual_seed(0)
    rng = np.random.default_rng(0)
    batch = mx.Transition(obs=torch.rand(B, L, 4).cuda(), a=torch.randint(0, 2, (B, L)).cuda(),
                          r=torch.rand(B, L).cuda(), Rn=(torch.rand(B, L) * 20).cuda(),
                          pi=torch.as_tensor(rng.dirichlet([1, 1], (B, L)).astype(np.float32)).cuda())
    res = {}
    for backend in ("hip", "torch"):
        net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                              mx.nn.Dynamic(8, 2, 21, generator=g))
        m = mx.MuZero(net)
        m.init(0, np.zeros((1, 4)))
        res[backend] = timeit(lambda: m.update(batch, backend=backend))
        print(f"update() backend={backend:5s} B={B} L={L}: {res[backend] * 1e3:8.3f} ms/step "
              f"{B * L / res[backend] / 1e6:8.2f} M transitions/s")
        if backend == "hip":
            f = m._fused_train
            t = timeit(lambda: f(batch), n=50)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(50):
                f(batch)
            ev1.record()
            torch.cuda.synchronize()
            print(f"  loss+grad kernels only: host-timed {t * 1e6:8.1f} us/call, device {ev0.elapsed_time(ev1) / 50 * 1e3:8.1f} us/call")


if __name__ == "__main__":
    main()
