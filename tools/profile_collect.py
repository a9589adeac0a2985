"""Where the acting half of fit_vector spends its time (host side): cProfile of VectorCollector.collect
on the GPU box.  python tools/profile_collect.py [envs] [steps]"""
import cProfile
import os
import pstats
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import muax_amd as muax  # noqa: E402
from muax_amd import nn  # noqa: E402
from cartpole_env import VectorCartPole  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
net = muax.create_muzero_network(nn.Representation, nn.Prediction, nn.Dynamic, embedding_dim=8, num_actions=2,
                                 full_support_size=21)
model = muax.MuZero(net, discount=0.99, support_size=10)
venv = VectorCartPole(N, seed=0)
model.init(muax.prng.PRNGKey(0), venv.reset()[:1].astype(float))
col = muax.VectorCollector(venv, 10, 0.99, 0.5)
key = muax.prng.PRNGKey(1)
for _ in range(3):
    t = time.perf_counter()
    trajs, key, n = col.collect(model, key, T, 50, 1.0)
    dt = time.perf_counter() - t
    print(f"collect: {n} env steps, {len(trajs)} episodes, {dt * 1e3:.1f} ms -> {n / dt / 1e6:.2f} M env-steps/s")
obs = venv.reset()
for _ in range(5):
    model.act(key, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=50)
t = time.perf_counter()
for _ in range(50):
    model.act(key, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=50)
print(f"act() alone at B={N}: {(time.perf_counter() - t) / 50 * 1e3:.3f} ms per call")
pr = cProfile.Profile()
pr.enable()
col.collect(model, key, T, 50, 1.0)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
