import sys, time, os
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import muax_amd as mx
for od in (4, 16, 64, 256):
    g = torch.Generator().manual_seed(0)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g), mx.nn.Dynamic(8, 2, 21, generator=g))
    m = mx.MuZero(net)
    m.init(0, np.zeros((1, od), np.float32))
    obs = np.random.default_rng(0).uniform(-1, 1, (4096, od)).astype(np.float32)
    obs_d = torch.from_numpy(obs).cuda()
    for label, x, kw in (("numpy", obs, {}), ("device", obs_d, {"device_outputs": True})):
        for i in range(20): m.act(i, x, obs_from_batch=True, num_simulations=50, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(200): m.act(100 + i, x, obs_from_batch=True, num_simulations=50, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"obs_dim {od:4d} {label:6s}: {dt*1e6:7.1f} us per act")
