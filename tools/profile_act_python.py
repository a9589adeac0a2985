import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import muax_amd as mx
from muax_amd import search
g = torch.Generator().manual_seed(0)
net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g), mx.nn.Dynamic(8, 2, 21, generator=g))
m = mx.MuZero(net); m.init(0, np.zeros((1, 4), np.float32))
obs = np.random.default_rng(0).uniform(-1, 1, (4096, 4)).astype(np.float32)
for i in range(30): m.act(i, obs, obs_from_batch=True, num_simulations=50)
acc = [0.0]
orig = search.MuZeroSearch.act_mlp_host
L = None
def timed(self, *a, **k):
    global L
    if L is None:
        L = self._L.mzs_act_mlp_host
        def wrapped(*x):
            t0 = time.perf_counter(); r = L(*x); acc[0] += time.perf_counter() - t0; return r
        self._L.mzs_act_mlp_host = wrapped
    return orig(self, *a, **k)
search.MuZeroSearch.act_mlp_host = timed
m.act(0, obs, obs_from_batch=True, num_simulations=50)
acc[0] = 0.0
n = 500
t0 = time.perf_counter()
for i in range(n): m.act(100 + i, obs, obs_from_batch=True, num_simulations=50)
tot = time.perf_counter() - t0
print(f"act() total {tot/n*1e6:.1f} us, inside the C call {acc[0]/n*1e6:.1f} us, python around it {(tot-acc[0])/n*1e6:.1f} us")
