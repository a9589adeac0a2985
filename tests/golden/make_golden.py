"""Regenerates tests/golden/act_mlp_*.npz from the CPU oracle (oracle/mz_oracle.c).

These are NOT reference outputs: jax/mctx cannot be imported in the build container and the reference
holds no vectors for this path ("parity unpinned").  They freeze the oracle's own
results on small seeded cases so that (a) accidental drift of the oracle is caught on CPU and (b) the
GPU path is checked against data that does not depend on the oracle being rebuilt on the GPU box.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle as po  # noqa: E402

CASES = {
    # name: (seed, B, obs_dim, E, A, S, tiebreak, key)
    "cartpole_s10": (0, 8, 4, 8, 2, 10, 1, (0, 42)),      # BASELINE config 1 shapes (S=10), 8 roots
    "cartpole_s50": (1, 8, 4, 8, 2, 50, 1, (0, 7)),       # BASELINE config 2 shapes, 8 roots
    "lunarlander_s50": (2, 6, 8, 32, 4, 50, 1, (3, 9)),   # BASELINE config 3 shapes, 6 roots
}


def make(name):
    seed, B, obs_dim, E, A, S, tb, key = CASES[name]
    F = 21
    w = po.random_mlp_weights(seed, obs_dim, E, A, F, bias_scale=0.1)
    rng = np.random.default_rng(seed + 100)
    obs = rng.uniform(-1, 1, (B, obs_dim)).astype(np.float32)
    noise = rng.dirichlet([0.3] * A, B).astype(np.float32)
    out = po.act_mlp(po.Mlp(w, obs_dim, E, A, F), po.SearchCfg(S, tiebreak=tb), obs, list(key), noise, 0.25)
    data = {"obs": obs, "dirichlet_noise": noise, "key": np.array(key, np.uint32),
            "meta": np.array([B, obs_dim, E, A, S, tb], np.int64),
            "action": out["action"], "action_weights": out["action_weights"], "root_value": out["root_value"],
            "depth_sum": out["depth_sum"]}
    data.update({"w_" + k: v for k, v in w.items()})
    data.update({"tree_" + k: v for k, v in out["tree"].arrays().items()})
    np.savez_compressed(os.path.join(HERE, f"act_mlp_{name}.npz"), **data)


def gumbel_act(w, obs_dim, E, A, S, obs, key, kind, maxc):
    """mctx.gumbel_muzero_policy composed from the oracle's pieces (root Gumbel noise from split(key)[1])."""
    B = obs.shape[0]
    mlp = po.Mlp(w, obs_dim, E, A, 21)
    pl, v, emb = po.root_inference(mlp, obs)
    g = po.gumbel(po.split(list(key), 2)[1], B * A).reshape(B, A)
    tree = po.Tree(B, S + 1, A, E)
    cfg = po.SearchCfg(S)
    po.tree_init(tree, po.mask_root_logits(pl, None), v, emb, None)
    dsum = np.zeros(B, np.int64)
    for sim in range(S):
        p_, a_, d_ = po.gumbel_step_select(tree, cfg, g, kind, maxc)
        dsum += d_
        po.step_expand_backup(tree, sim, p_, a_, *po.recurrent_inference(mlp, a_, tree.embeddings[np.arange(B), p_]))
    action, weights = po.gumbel_finish(tree, g, kind)
    return {"action": action, "action_weights": weights, "root_value": v, "depth_sum": dsum, "tree": tree}


def make_gumbel():
    """Gumbel MuZero (config 5's policy), CartPole shapes, 8 roots, S=32, mix-value qtransform, 16 considered."""
    seed, B, obs_dim, E, A, S, key = 5, 8, 4, 8, 2, 32, (11, 13)
    w = po.random_mlp_weights(seed, obs_dim, E, A, 21, bias_scale=0.1)
    obs = np.random.default_rng(seed + 100).uniform(-1, 1, (B, obs_dim)).astype(np.float32)
    out = gumbel_act(w, obs_dim, E, A, S, obs, key, 1, 16)
    data = {"obs": obs, "key": np.array(key, np.uint32), "meta": np.array([B, obs_dim, E, A, S, 16], np.int64),
            "action": out["action"], "action_weights": out["action_weights"], "root_value": out["root_value"],
            "depth_sum": out["depth_sum"]}
    data.update({"w_" + k: v for k, v in w.items()})
    data.update({"tree_" + k: v for k, v in out["tree"].arrays().items()})
    np.savez_compressed(os.path.join(HERE, "act_gumbel_cartpole_s32.npz"), **data)


if __name__ == "__main__":
    for n in CASES:
        make(n)
        print("wrote", n)
    make_gumbel()
    print("wrote gumbel_cartpole_s32")
