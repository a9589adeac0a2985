"""Regenerates tests/golden/rollout_cartpole_s10.npz: BASELINE config 1 (CartPole-v1, MLP embed 8,
num_simulations=10, batch 1) as the inner loop of muax.fit runs it (muax/train.py:153-170): per environment step
`key, subkey = split(key)`, `act(subkey, obs, with_pi, with_value, obs_from_batch=False, num_simulations=10,
temperature=1.)`, `env.step(a)` -- recorded as the trace (subkey, obs) -> (a, pi, v) that SURVEY.md 8 row a10
asks for.  The trace comes from the CPU ORACLE driven with the same sub-keys (root noise = its restatement of
jax.random.dirichlet from split(subkey, 3)[1], tie-break noise and the final Gumbel draw from the key): NOT a
reference output (jax / mctx cannot be imported here), it freezes the oracle on the fit-loop path so that the
GPU test compares muax_amd.rollout() with data that does not depend on rebuilding the oracle.

    python tests/golden/make_rollout_trace.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from cartpole_env import CartPole  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

SEED, ENV_SEED, KEY, S, STEPS = 0, 0, (0, 42), 10, 20
OBS_DIM, E, A, F = 4, 8, 2, 21


def oracle_rollout(w, key, env, steps=STEPS, num_simulations=S, temperature=1.0):
    mlp = po.Mlp(w, OBS_DIM, E, A, F)
    cfg = po.SearchCfg(num_simulations, tiebreak=1)
    obs, _ = env.reset()
    rows = []
    key = np.asarray(key, np.uint32)
    for t in range(steps):
        key, subkey = po.split(key, 2)
        k_dir = po.split(subkey, 3)[1]
        noise = po.dirichlet(k_dir, 0.3, 1, A)
        out = po.act_mlp(mlp, cfg, obs[None].astype(np.float32), subkey, noise, 0.25, None, temperature, None)
        a = int(out["action"][0])
        rows.append(dict(subkey=np.asarray(subkey, np.uint32), obs=obs.astype(np.float32), noise=noise[0],
                         a=a, pi=out["action_weights"].reshape(1, A), v=np.float32(out["root_value"][0])))
        obs, r, done, truncated, _ = env.step(a)
        if done or truncated:
            break
    return rows


def main():
    w = po.random_mlp_weights(SEED, OBS_DIM, E, A, F, bias_scale=0.1)
    rows = oracle_rollout(w, KEY, CartPole(seed=ENV_SEED))
    assert len(rows) == STEPS, f"episode ended after {len(rows)} steps: pick another ENV_SEED"
    data = {"meta": np.array([SEED, ENV_SEED, S, STEPS, OBS_DIM, E, A], np.int64), "key": np.array(KEY, np.uint32)}
    for f in ("subkey", "obs", "noise", "a", "pi", "v"):
        data[f] = np.stack([np.asarray(r[f]) for r in rows])
    data.update({"w_" + k: v for k, v in w.items()})
    np.savez_compressed(os.path.join(HERE, "rollout_cartpole_s10.npz"), **data)
    print("wrote rollout_cartpole_s10.npz:", len(rows), "steps, actions", data["a"].tolist())


if __name__ == "__main__":
    main()
