"""BASELINE config 5 shape on one GPU: Gumbel MuZero act() on 4096 roots (S=50) + one k_steps=10 training
step on 4096 trajectories, default MLP trio.  Both halves are single HIP launches (fused search, fused
forward+backward); with torch.distributed initialised the gradient mean is one flat all-reduce."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muax_amd as mx  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)

B, L, S = 4096, 10, 50
g = torch.Generator().manual_seed(0)
net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                      mx.nn.Dynamic(8, 2, 21, generator=g))
m = mx.MuZero(net, policy="gumbel")
m.init(0, np.zeros((1, 4)))
obs = (torch.rand(B, 4, generator=g) * 2 - 1).cuda()
rng = np.random.default_rng(0)
batch = mx.Transition(obs=torch.rand(B, L, 4).cuda(), a=torch.randint(0, 2, (B, L)).cuda(), r=torch.rand(B, L).cuda(),
                      Rn=(torch.rand(B, L) * 20).cuda(),
                      pi=torch.as_tensor(rng.dirichlet([1, 1], (B, L)).astype(np.float32)).cuda())


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t_act = timeit(lambda: m.act(1, obs, obs_from_batch=True, num_simulations=S, device_outputs=True))
t_upd = timeit(lambda: m.update(batch))
print(f"gumbel act  B={B} S={S}: {t_act * 1e3:7.3f} ms  {B / t_act / 1e6:6.2f} M env-steps/s")
print(f"update      B={B} L={L}: {t_upd * 1e3:7.3f} ms  {B * L / t_upd / 1e6:6.2f} M transitions/s")
print(f"act + update           : {(t_act + t_upd) * 1e3:7.3f} ms per iteration")
