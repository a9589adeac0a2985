"""mctx.muzero_policy on the CPU backend, driven by THIS build's own jnp restatement of the default MLP trio.

Used by bench.py's cpu_baseline leg only, and only on a host where `import jax, mctx` works (SURVEY.md section 7 step 0;
BASELINE.md section 3, third baseline row: "third-party search, the build's own glue").  No file of the reference is
needed: the nets are muax/nn.py:37-115 restated over the C-ABI's 18 weight arrays (include/mzsearch.h), root_fn /
recurrent_fn are muax/model.py:251-282 restated, the call is the one muax/policy.py:18-30 makes.  Two uses:

* time `jax.jit(mctx.muzero_policy)` on the metric's workload -> {"mctx_cpu": {value, unit, ...}} (a reported
  third-party baseline, never the thing measured);
* dump the same call at 8 roots, seeds {0, 1, 2} x {CartPole shapes at 50 simulations, LunarLander shapes at 50}, in
  tests/golden/mctx_fixture.py's format under gpurun_out/mctx_capture/ (gpurun merges that directory back; committing
  the files under tests/golden/ un-skips tests/test_mctx_pin_cpu.py and tests/test_gpu_mctx_pin.py for rows a7-a9).
  meta["route"] says the nets are this build's restatement, not the reference's haiku modules: such a capture pins the
  SEARCH (mctx) and the PRNG walk, not muax's own net definitions.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _nets(w, A, support, discount):
    import jax
    import jax.numpy as jnp
    import mctx

    W = {k: jnp.asarray(v, jnp.float32) for k, v in w.items()}

    def min_max(s):  # muax/nn.py:37-44
        lo, hi = s.min(-1, keepdims=True), s.max(-1, keepdims=True)
        sc = hi - lo
        sc = jnp.where(sc < 1e-5, sc + 1e-5, sc)
        return (s - lo) / sc

    def mlp(x, net):
        return jax.nn.elu(x @ W[net + "_w1"] + W[net + "_b1"]) @ W[net + "_w2"] + W[net + "_b2"]

    def decode(logits):  # muax/utils.py:70-102: softmax expectation over [-support, support], then the inverse scaling
        p = jax.nn.softmax(logits, -1)
        x = (p * jnp.arange(-support, support + 1, dtype=jnp.float32)).sum(-1)
        eps = 0.001
        return jnp.sign(x) * (((jnp.sqrt(1 + 4 * eps * (jnp.abs(x) + 1 + eps)) - 1) / (2 * eps)) ** 2 - 1)

    def root_fn(obs):  # muax/model.py:251-263
        s = min_max(obs @ W["repr_w"] + W["repr_b"])
        return mctx.RootFnOutput(prior_logits=mlp(s, "pp"), value=decode(mlp(s, "pv")), embedding=s)

    def recurrent_fn(params, rng_key, action, embedding):  # muax/model.py:265-282 (prediction on the child's state)
        sa = jnp.concatenate([embedding, jax.nn.one_hot(action, A, dtype=jnp.float32)], -1)
        ns = min_max(mlp(sa, "dn"))
        out = mctx.RecurrentFnOutput(reward=decode(mlp(sa, "dr")), discount=jnp.full_like(action, discount, jnp.float32),
                                     prior_logits=mlp(ns, "pp"), value=decode(mlp(ns, "pv")))
        return out, ns

    return root_fn, recurrent_fn


def _policy(w, A, support, S, discount=0.99):
    import jax
    import mctx
    root_fn, recurrent_fn = _nets(w, A, support, discount)

    def plan(key, obs):  # muax/policy.py:18-30 with MuZero.act's defaults (muax/model.py:161-171)
        root = root_fn(obs)
        out = mctx.muzero_policy(None, key, root, recurrent_fn, num_simulations=S, dirichlet_fraction=0.25,
                                 dirichlet_alpha=0.3, pb_c_init=1.25, pb_c_base=19652, temperature=1.0,
                                 qtransform=mctx.qtransform_by_parent_and_siblings)
        return out, root.value

    return jax.jit(plan, backend="cpu")


def time_and_capture(w, obs, noise, A, E, F, S, support, budget_s, out_dir):
    import jax
    import jax.numpy as jnp
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import capture_from_mctx as cap
    import mctx_fixture as fx

    res = {}
    plan = _policy(w, A, support, S)
    x = jnp.asarray(obs)
    plan(jax.random.PRNGKey(0), x)[0].action.block_until_ready()  # compile
    reps, t = 0, 0.0
    while t < budget_s and reps < 50:
        t0 = time.perf_counter()
        plan(jax.random.PRNGKey(1 + reps), x)[0].action.block_until_ready()
        t += time.perf_counter() - t0
        reps += 1
    res["mctx_cpu"] = {"value": round(obs.shape[0] * reps / t, 1), "unit": "env-steps/s", "kind": "third-party",
                       "sample": f"{obs.shape[0]} roots x S={S}, {reps} acts in {t:.1f}s, jax.jit(mctx.muzero_policy, "
                                 f"backend='cpu') around the build's own jnp trio (tools/mctx_cpu_glue.py)"}
    os.makedirs(out_dir, exist_ok=True)
    written = []
    shapes = {"cartpole": dict(obs_dim=4, E=8, A=2), "lunarlander": dict(obs_dim=8, E=32, A=4)}
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po  # (weights generator only: inside bench.py's cpu_baseline leg)
    for name, shp in shapes.items():
        for seed in (0, 1, 2):
            ww = po.random_mlp_weights(seed, shp["obs_dim"], shp["E"], shp["A"], 2 * support + 1,
                                       bias_scale=0.0 if seed == 0 else 0.1)
            o = np.random.default_rng(1000 + seed).uniform(-1, 1, (8, shp["obs_dim"])).astype(np.float32)
            key = jax.random.PRNGKey(100 + 7 * seed + 50)
            out, rv = _policy(ww, shp["A"], support, 50)(key, jnp.asarray(o))
            meta = {"policy": "muzero", "num_simulations": 50, "support_size": support, "discount": 0.99, "temperature": 1.0,
                    "dirichlet_fraction": 0.25, "dirichlet_alpha": 0.3, "pb_c_init": 1.25, "pb_c_base": 19652.0,
                    "max_depth": None, "recurrent_pred_on": "child", "qtransform": "qtransform_by_parent_and_siblings",
                    "max_num_considered_actions": 16, "gumbel_scale": 1.0, "seed": seed, "shape": name, **shp,
                    "route": "mctx.muzero_policy around muax_amd's jnp restatement of the trio (tools/mctx_cpu_glue.py); "
                             "pins the search and the PRNG walk, not the reference's haiku nets",
                    "versions": {"jax": jax.__version__, "mctx": getattr(__import__("mctx"), "__version__", "unknown"),
                                 "jax_threefry_partitionable": bool(getattr(jax.config, "jax_threefry_partitionable", False))}}
            inter = cap.rng_intermediates("muzero", key, 8, shp["A"], 50, 0.3)
            path = os.path.join(out_dir, f"mctx_{name}_muzero_s50_seed{seed}.npz")
            fx.save_case(path, meta, ww, o, np.asarray(key, np.uint32),
                         {"action": np.asarray(out.action), "action_weights": np.asarray(out.action_weights),
                          "root_value": np.asarray(rv)}, cap.tree_arrays(out.search_tree), inter)
            written.append(os.path.basename(path))
    res["captures_written"] = written
    return res
