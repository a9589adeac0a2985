#!/usr/bin/env python
"""Capture golden vectors of the reference's act() path from a REAL jax + mctx + dm-haiku install.

This is the one-command route to pinning the oracle (last row).  It cannot run in the build container
or on the GPU box (no jax / mctx / haiku there, no network): run it once on any machine with

    pip install "jax[cpu]" mctx dm-haiku optax scipy           # jax 0.4.x matches the reference's pins (setup.py:40-42)
    python tests/golden/capture_from_mctx.py --muax-path /path/to/bwfbowen-muax-checkout
    git add tests/golden/mctx_*.npz tests/golden/mctx_checkpoint.npy tests/golden/mctx_checkpoint_expected.npz

and from then on `pytest tests/test_mctx_pin_cpu.py` (oracle vs capture, CPU) and `pytest -m gpu
tests/test_gpu_mctx_pin.py` (HIP path vs capture) stop skipping and compare: integer arrays exact, floats to 1e-5.

What runs: the reference's own glue.  With --muax-path the top-level `muax.model.MuZero` of the checkout
(muax/model.py:16-283; policy adapters muax/policy.py:13-47; nets muax/nn.py:59-115, factories :417-433) is imported
around the package `__init__` (which needs dm-acme) and around the muax.model <-> muax.loss import cycle (loss is not
on this path and is replaced by an inert module).  Without it the pip release is used (`import muax`, README API:
`muax.MuZero(repr_fn, pred_fn, dy_fn, policy='muzero')`, i.e. muax/frameworks/coax/model.py:101-110).  Either way
every capture is one `model._plan(params, key, obs, ...)` call -- root inference, `mctx.muzero_policy` /
`mctx.gumbel_muzero_policy`, returned `(PolicyOutput, root.value)` -- and the PRNG intermediates are re-drawn with plain
jax.random calls along mctx's key walk so that sampler and search can be checked separately.

Captured: seeds {0, 1, 2} x {CartPole shapes A=2, E=8, obs 4 at num_simulations 1 / 10 / 50 (+ 160 and a
twelve-action trio at seed 0, round 6: the HIP side's LONG / wide instances);
LunarLander shapes A=4, E=32, obs 8 at 50}, B = 8; Gumbel MuZero on both shapes with both qtransforms (seed 0); one
checkpoint written by the reference's own save (muax/model.py:203-212) with the flattened weights beside it; when
gymnasium is importable, 20 CartPole-v1 steps of the fit loop's act() calls (muax/train.py:153-170); and one case of the
convolutional ResNet nets (root inference + one recurrent_fn call on two Atari-shaped frames, parameters by manifest).
Nothing of the reference's source is written anywhere: the outputs are arrays.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mctx_fixture as fx  # noqa: E402

SHAPES = {"cartpole": dict(obs_dim=4, E=8, A=2), "lunarlander": dict(obs_dim=8, E=32, A=4),
          # round 6: twelve actions -- on the HIP side an on-demand instance that keeps all of a node's scores in one lane
          "wide12": dict(obs_dim=4, E=8, A=12)}
SUPPORT = 10
TIEBREAK_LEVELS = 12


def import_reference(muax_path):
    """-> (module holding MuZero, muax.nn, muax.policy or None, route string)."""
    if muax_path:
        root = os.path.abspath(muax_path)
        if not os.path.isfile(os.path.join(root, "muax", "model.py")):
            raise SystemExit(f"--muax-path {root}: no muax/model.py there")
        pkg = types.ModuleType("muax")
        pkg.__path__ = [os.path.join(root, "muax")]  # skip muax/__init__.py (imports acme._metadata)
        sys.modules["muax"] = pkg
        inert = types.ModuleType("muax.loss")        # muax.model <-> muax.loss import each other; loss is off this path
        inert.default_loss_fn = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("loss is not captured"))
        sys.modules["muax.loss"] = inert
        model = importlib.import_module("muax.model")
        return model, importlib.import_module("muax.nn"), importlib.import_module("muax.policy"), \
            "reference checkout, top-level muax.model.MuZero (muax/model.py)"
    muax = importlib.import_module("muax")
    return muax, importlib.import_module("muax.nn"), None, \
        f"pip release muax {getattr(muax, '__version__', '?')} (README API, frameworks/coax/model.py)"


def build_model(ref, nn, policy_mod, route, shape, policy):
    F = 2 * SUPPORT + 1
    repr_fn = nn._init_representation_func(nn.Representation, shape["E"])
    pred_fn = nn._init_prediction_func(nn.Prediction, shape["A"], F)
    dy_fn = nn._init_dynamic_func(nn.Dynamic, shape["E"], shape["A"], F)
    if policy_mod is not None:  # top-level class: MuZero(network, policy_class=...)
        cls = policy_mod.MuZeroPolicy if policy == "muzero" else policy_mod.GumbelMuZeroPolicy
        return ref.MuZero(nn.MZNetwork(repr_fn, pred_fn, dy_fn), policy_class=cls, discount=0.99, support_size=SUPPORT)
    return ref.MuZero(repr_fn, pred_fn, dy_fn, policy="muzero", discount=0.99, support_size=SUPPORT)


def flatten_params(params, shape):
    """haiku parameter dicts -> the C-ABI's 18 names.  Layers in creation order (linear, linear_1, ...):
    representation: repr; prediction: v_func (2), pi_func (2); dynamic: ns_func (2), r_func (2) (muax/nn.py:59-115)."""
    def linears(tree):
        def order(name):
            tail = name.rsplit("/", 1)[-1]
            return int(tail.rsplit("_", 1)[1]) if "_" in tail and tail.rsplit("_", 1)[1].isdigit() else 0
        return [{k: np.asarray(v, np.float32) for k, v in tree[n].items()} for n in sorted(tree, key=order)]

    (rep,), pred, dyn = linears(params.representation), linears(params.prediction), linears(params.dynamic)
    names = {"repr": rep, "pv_1": pred[0], "pv_2": pred[1], "pp_1": pred[2], "pp_2": pred[3],
             "dn_1": dyn[0], "dn_2": dyn[1], "dr_1": dyn[2], "dr_2": dyn[3]}
    w = {"repr_w": rep["w"], "repr_b": rep["b"]}
    for k, layer in names.items():
        if k != "repr":
            net, idx = k.split("_")
            w[f"{net}_w{idx}"], w[f"{net}_b{idx}"] = layer["w"], layer["b"]
    E, A, F = shape["E"], shape["A"], 2 * SUPPORT + 1
    want = {"repr_w": (shape["obs_dim"], E), "pv_w2": (16, F), "pp_w2": (16, A), "dr_w1": (E + A, 16), "dr_w2": (16, F),
            "dn_w2": (16, E)}
    for k, s in want.items():
        if w[k].shape != s:
            raise SystemExit(f"haiku layer order is not what muax/nn.py suggests: {k} has shape {w[k].shape}, expected {s}")
    return w


def check_flattening(w, model, params, obs, shape):
    """Evaluate the trio from the flattened arrays in NumPy (float64) and compare with the reference's own
    _root_inference / _recurrent_inference logits -- catches a wrong layer mapping before anything is written.
    Returns which embedding the recurrent prediction reads ('child' as muax/model.py:272, 'parent' as the pip release)."""
    import jax
    import jax.numpy as jnp

    def elu(x):
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))

    def minmax(s):
        lo, hi = s.min(1, keepdims=True), s.max(1, keepdims=True)
        sc = hi - lo
        sc = np.where(sc < 1e-5, sc + 1e-5, sc)
        return (s - lo) / sc

    W = {k: v.astype(np.float64) for k, v in w.items()}
    s = minmax(obs.astype(np.float64) @ W["repr_w"] + W["repr_b"])
    pi = elu(s @ W["pp_w1"] + W["pp_b1"]) @ W["pp_w2"] + W["pp_b2"]
    key = jax.random.PRNGKey(0)
    root = model._root_inference(params, key, jnp.asarray(obs))
    if not np.allclose(np.asarray(root.embedding), s, atol=1e-5) or not np.allclose(np.asarray(root.prior_logits), pi, atol=1e-4):
        raise SystemExit("flattened weights do not reproduce the reference's root inference")
    a = np.arange(obs.shape[0]) % shape["A"]
    sa = np.concatenate([s, np.eye(shape["A"])[a]], 1)
    ns = minmax(elu(sa @ W["dn_w1"] + W["dn_b1"]) @ W["dn_w2"] + W["dn_b2"])
    rec, nxt = model._recurrent_inference(params, key, jnp.asarray(a, jnp.int32), root.embedding)
    if not np.allclose(np.asarray(nxt), ns, atol=1e-5):
        raise SystemExit("flattened weights do not reproduce the reference's dynamics")
    pi_child = elu(ns @ W["pp_w1"] + W["pp_b1"]) @ W["pp_w2"] + W["pp_b2"]
    pi_parent = elu(s @ W["pp_w1"] + W["pp_b1"]) @ W["pp_w2"] + W["pp_b2"]
    got = np.asarray(rec.prior_logits)
    if np.allclose(got, pi_child, atol=1e-4):
        return "child"
    if np.allclose(got, pi_parent, atol=1e-4):
        return "parent"
    raise SystemExit("recurrent prior logits match neither the child's nor the parent's embedding")


def rng_intermediates(policy, key, B, A, S, alpha, gumbel_scale=1.0):
    """mctx's key walk with plain jax.random calls (mctx/_src/policies.py muzero_policy / gumbel_muzero_policy,
    search.py search / simulate, action_selection.py muzero_action_selection)."""
    import jax
    import jax.numpy as jnp
    out = {}
    if policy == "muzero":
        k_sample, k_dir, k_search = jax.random.split(key, 3)
        out["dirichlet"] = jax.random.dirichlet(k_dir, jnp.full([A], alpha, jnp.float32), (B,))
        out["final_gumbel"] = jax.random.gumbel(k_sample, (B, A), jnp.float32)
        D = min(S, TIEBREAK_LEVELS)
        tb = np.zeros((S, B, D, A), np.float32)
        rk = k_search
        for s in range(S):
            rk, k_sim, _k_expand = jax.random.split(rk, 3)
            roots = jax.random.split(k_sim, B)
            for b in range(B):
                kb = roots[b]
                for d in range(D):
                    kb, k_sel = jax.random.split(kb)
                    tb[s, b, d] = np.asarray(jax.random.uniform(k_sel, (A,)))
        out["tiebreak"] = tb
    else:
        _k_search, k_gumbel = jax.random.split(key)
        out["root_gumbel"] = gumbel_scale * jax.random.gumbel(k_gumbel, (B, A), jnp.float32)
    return {k: np.asarray(v, np.float32) for k, v in out.items()}


def tree_arrays(tree):
    names = fx.TREE_INT + fx.TREE_FLOAT
    return {n: np.asarray(getattr(tree, n)) for n in names}


def versions():
    import haiku
    import jax
    import mctx
    v = {"jax": jax.__version__, "mctx": getattr(mctx, "__version__", "unknown"), "haiku": haiku.__version__,
         "numpy": np.__version__,
         # (older jax has no such flag: its threefry stream is the non-partitionable one the oracle restates)
         "jax_threefry_partitionable": bool(getattr(jax.config, "jax_threefry_partitionable", False)),
         "jax_enable_x64": bool(getattr(jax.config, "jax_enable_x64", False)), "backend": jax.default_backend()}
    try:
        import jaxlib
        v["jaxlib"] = jaxlib.__version__
    except Exception:
        pass
    return v


def capture(ref, nn, policy_mod, route, name, shape, seed, S, policy="muzero", qtransform=None, maxc=16, B=8, out_dir=HERE):
    import jax
    import jax.numpy as jnp
    import mctx
    model = build_model(ref, nn, policy_mod, route, shape, policy)
    rng = np.random.default_rng(1000 + seed)
    obs = rng.uniform(-1, 1, (B, shape["obs_dim"])).astype(np.float32)
    params = model.init(jax.random.PRNGKey(seed), jnp.asarray(obs))
    if seed > 0:  # haiku initialises biases to zero; seeds 1, 2 exercise them
        leaves, treedef = jax.tree_util.tree_flatten(params)
        leaves = [x + 0.1 * jnp.asarray(rng.standard_normal(x.shape), x.dtype) if x.ndim == 1 else x for x in leaves]
        params = jax.tree_util.tree_unflatten(treedef, leaves)
        model._params = params
    w = flatten_params(params, shape)
    pred_on = check_flattening(w, model, params, obs, shape)
    key = jax.random.PRNGKey(100 + 7 * seed + S)
    meta = {"policy": policy, "num_simulations": S, "support_size": SUPPORT, "discount": 0.99, "temperature": 1.0,
            "dirichlet_fraction": 0.25, "dirichlet_alpha": 0.3, "pb_c_init": 1.25, "pb_c_base": 19652.0, "max_depth": None,
            "recurrent_pred_on": pred_on, "qtransform": qtransform or "qtransform_by_parent_and_siblings",
            "max_num_considered_actions": maxc, "gumbel_scale": 1.0, "route": route, "versions": versions(),
            "seed": seed, "shape": name, **shape}
    qt = getattr(mctx, meta["qtransform"])
    # the keyword set MuZero.act hands to _plan (muax/model.py:161-171)
    kw = dict(num_simulations=S, temperature=1.0, invalid_actions=None, max_depth=None, loop_fn=jax.lax.fori_loop,
              qtransform=None, dirichlet_fraction=0.25, dirichlet_alpha=0.3, pb_c_init=1.25, pb_c_base=19652)
    if policy == "muzero":
        out, root_value = model._plan(params, key, jnp.asarray(obs), **kw)
    elif policy_mod is not None:
        # muax/model.py:230-231 overrides qtransform=None with by_parent_and_siblings for every policy; the
        # mix-value capture passes it explicitly
        out, root_value = model._plan(params, key, jnp.asarray(obs), **dict(kw, qtransform=qt))
    else:  # pip release: the gumbel branch is commented out there -> drive mctx with the model's own callbacks
        root = model._root_inference(params, key, jnp.asarray(obs))
        out = mctx.gumbel_muzero_policy(params, key, root, model._recurrent_inference, num_simulations=S,
                                        qtransform=qt, max_num_considered_actions=maxc, gumbel_scale=1.0)
        root_value = root.value
    outputs = {"action": np.asarray(out.action), "action_weights": np.asarray(out.action_weights),
               "root_value": np.asarray(root_value)}
    inter = rng_intermediates(policy, key, B, shape["A"], S, 0.3)
    tag = f"{name}_{policy}_s{S}_seed{seed}" + ("_mix" if (qtransform or "").endswith("mix_value") else "")
    path = os.path.join(out_dir, f"mctx_{tag}.npz")
    fx.save_case(path, meta, w, obs, np.asarray(key, np.uint32), outputs, tree_arrays(out.search_tree), inter)
    print(f"wrote {os.path.relpath(path)}  ({os.path.getsize(path) / 1024:.0f} KB)  pred_on={pred_on}")
    return model, params, w, pred_on


def capture_checkpoint(model, w, out_dir):
    """The reference's own save: jnp.save of {'params', 'optimizer_state'} (muax/model.py:203-212 / coax :315-318)."""
    path = os.path.join(out_dir, "mctx_checkpoint")
    if hasattr(model, "save_load"):
        model.save_load(path, save=True)
    else:
        model.save(path)
    np.savez_compressed(os.path.join(out_dir, "mctx_checkpoint_expected.npz"), **w)
    print("wrote tests/golden/mctx_checkpoint.npy + mctx_checkpoint_expected.npz")


def capture_rollout(model, params, w, out_dir, pred_on="child", steps=20, S=10):
    """20 CartPole-v1 steps of the fit loop's acting half (muax/train.py:153-170): key, subkey = split(key);
    a, pi, v = act(subkey, obs, with_pi=True, with_value=True, obs_from_batch=False, num_simulations=S, temperature=T)."""
    try:
        import gymnasium as gym
    except Exception as e:  # noqa: BLE001
        print("gymnasium not importable -> no fit-loop trace:", e)
        return
    import jax
    env = gym.make("CartPole-v1")
    obs, _ = env.reset(seed=0)
    key = jax.random.PRNGKey(0)
    rec = {k: [] for k in ("subkey", "obs", "action", "pi", "v")}
    for _ in range(steps):
        key, subkey = jax.random.split(key)
        a, pi, v = model.act(subkey, obs, with_pi=True, with_value=True, obs_from_batch=False, num_simulations=S,
                             temperature=1.0)
        rec["subkey"].append(np.asarray(subkey, np.uint32)); rec["obs"].append(np.asarray(obs, np.float32))
        rec["action"].append(int(a)); rec["pi"].append(np.asarray(pi, np.float32)); rec["v"].append(float(v))
        obs, _, done, trunc, _ = env.step(int(a))
        if done or trunc:
            obs, _ = env.reset()
    data = {k: np.asarray(v) for k, v in rec.items()}
    data.update({"w_" + k: v for k, v in w.items()})
    data["meta"] = np.array(json.dumps({"num_simulations": S, "recurrent_pred_on": pred_on, "versions": versions(),
                                        "format_version": fx.FORMAT_VERSION}))
    np.savez_compressed(os.path.join(out_dir, "mctx_rollout_cartpole_s10.npz"), **data)
    print("wrote tests/golden/mctx_rollout_cartpole_s10.npz")


def _representation_stages(model, params, x, nn):
    """One output per stage of the reference's ResNetRepresentation (muax/nn.py:291-310) -- the two stems (after their
    relu), every residual block, the two average pools -- recorded with haiku's method interceptor while the
    reference's own transformed function runs, in call order = mctx_fixture.STAGE_NAMES.  Returns None (no stage
    digests in the file; everything else is still captured) when this haiku has no interceptor or the call order is
    not the expected 2 stems + 8 blocks + 2 pools."""
    import haiku as hk
    import jax
    calls = []

    def interceptor(next_f, args, kwargs, context):
        out = next_f(*args, **kwargs)
        if context.method_name == "__call__":
            cls = type(context.module).__name__
            inside_block = "residual_conv_block" in context.module.module_name and cls != "ResidualConvBlockV1"
            if cls in ("ResidualConvBlockV1", "AvgPool") or (cls == "Conv2D" and not inside_block):
                calls.append((cls, out))
        return out

    try:
        with hk.intercept_methods(interceptor):
            model.repr_func.apply(params.representation, x)
    except Exception as e:  # noqa: BLE001
        print("no per-stage digests (haiku interceptor unavailable):", e)
        return None
    kinds = [c for c, _ in calls]
    want = ["Conv2D", "ResidualConvBlockV1", "ResidualConvBlockV1", "Conv2D"] + ["ResidualConvBlockV1"] * 3 + ["AvgPool"] + \
           ["ResidualConvBlockV1"] * 3 + ["AvgPool"]
    if kinds != want:
        print("no per-stage digests: the representation net's call order is", kinds)
        return None
    return [(name, np.asarray(jax.nn.relu(out) if kind == "Conv2D" else out))
            for name, (kind, out) in zip(fx.STAGE_NAMES, calls)]


def capture_resnet(ref, nn, policy_mod, out_dir, seed=0, B=2, A=18, F=21):
    """One capture of the convolutional plugin nets (muax/nn.py:118-148,313-395): root inference on B Atari-shaped
    frames and ONE recurrent_fn call, through the reference's own `_root_inference` / `_recurrent_inference`
    (muax/model.py:251-282) plus the heads' raw logits.  The parameters are NOT haiku's draws: every array of the
    parameter tree is overwritten, in the tree's own (call) order, with mctx_fixture.resnet_weights(manifest, seed), so
    that the file holds a manifest + a seed instead of 5 MB of weights and the consumer regenerates the same arrays.
    Pins the torch mirrors (muax_amd/nn.py ResNet*) and the one-launch HIP recurrent kernel (mz_conv.cuh): their
    SAME-padding geometry, LayerNorm axes, average pooling and head order."""
    import jax
    import jax.numpy as jnp
    if policy_mod is None:
        print("pip release route: the ResNet nets are captured from a checkout only (--muax-path)")
        return
    repr_fn = nn._init_resnet_representation_func(nn.ResNetRepresentation, 32)
    pred_fn = nn._init_resnet_prediction_func(nn.ResNetPrediction, A, F, 16)
    dy_fn = nn._init_resnet_dynamic_func(nn.ResNetDynamic, A, F, 64)
    model = ref.MuZero(nn.MZNetwork(repr_fn, pred_fn, dy_fn), policy_class=policy_mod.MuZeroPolicy, discount=0.99,
                       support_size=SUPPORT)
    rng = np.random.default_rng(2000 + seed)
    obs = rng.integers(0, 256, (B, 84, 84, 4)).astype(np.uint8)
    action = (np.arange(B) * 7 % A).astype(np.int32)
    params = model.init(jax.random.PRNGKey(seed), jnp.asarray(obs, jnp.float32))
    manifest = [(net, module, pname, list(np.shape(arr))) for net in ("representation", "prediction", "dynamic")
                for module, sub in getattr(params, net).items() for pname, arr in sub.items()]
    arrays = iter(fx.resnet_weights(manifest, seed))
    new = {net: {module: {pname: jnp.asarray(next(arrays)) for pname in sub} for module, sub in getattr(params, net).items()}
           for net in ("representation", "prediction", "dynamic")}
    params = type(params)(new["representation"], new["prediction"], new["dynamic"])
    key = jax.random.PRNGKey(0)
    x = jnp.asarray(obs, jnp.float32)
    root = model._root_inference(params, key, x)
    stages = _representation_stages(model, params, x, nn)
    v_lg, p_lg = model.pred_func.apply(params.prediction, root.embedding)
    rec, ns = model._recurrent_inference(params, key, jnp.asarray(action), root.embedding)
    r_lg, ns2 = model.dy_func.apply(params.dynamic, root.embedding, jnp.asarray(action))
    v2, p2 = model.pred_func.apply(params.prediction, ns2)
    assert np.allclose(np.asarray(ns), np.asarray(ns2)) and np.allclose(np.asarray(rec.prior_logits), np.asarray(p2))
    meta = {"manifest": manifest, "weights_seed": seed, "A": A, "F": F, "support_size": SUPPORT, "input_channels": 32,
            "dynamic_channels": 64, "versions": versions(), "route": "reference checkout, muax.nn ResNet* through muax.model.MuZero"}
    path = os.path.join(out_dir, os.path.basename(fx.RESNET_PATH))
    fx.save_resnet(path, meta, obs, action,
                   {"embedding": root.embedding, "value_logits": v_lg, "prior_logits": root.prior_logits, "value": root.value},
                   {"reward_logits": r_lg, "next_embedding": ns, "value_logits": v2, "prior_logits": rec.prior_logits,
                    "reward": rec.reward, "value": rec.value}, stages)
    print(f"wrote {os.path.relpath(path)}  ({os.path.getsize(path) / 1024:.0f} KB; {len(manifest)} parameter arrays by manifest)")


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--muax-path", default=os.environ.get("MUAX_PATH"),
                    help="checkout of bwfbowen/muax (uses its top-level muax.model); default: the pip release `muax`")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--quick", action="store_true", help="seed 0 only")
    args = ap.parse_args()
    try:
        import jax
        import mctx  # noqa: F401
        import haiku  # noqa: F401
    except ImportError as e:
        raise SystemExit(f"capture_from_mctx.py needs jax, mctx and dm-haiku ({e}); see the docstring")
    jax.config.update("jax_platform_name", "cpu")
    ref, nn, policy_mod, route = import_reference(args.muax_path)
    first = None
    for seed in ((0,) if args.quick else (0, 1, 2)):
        for S in (1, 10, 50):
            got = capture(ref, nn, policy_mod, route, "cartpole", SHAPES["cartpole"], seed, S, out_dir=args.out)
            if first is None and S == 10:
                first = got
        capture(ref, nn, policy_mod, route, "lunarlander", SHAPES["lunarlander"], seed, 50, out_dir=args.out)
    for name in ("cartpole", "lunarlander"):
        for qt in ("qtransform_by_parent_and_siblings", "qtransform_completed_by_mix_value"):
            capture(ref, nn, policy_mod, route, name, SHAPES[name], 0, 50, policy="gumbel", qtransform=qt, out_dir=args.out)
    # round 6: the record kinds the HIP side added since the kit was written -- 160 simulations (FusedCfg::LONG: root
    # paths in HBM; an instance planned per policy) and twelve actions (in-lane scores), both policies
    capture(ref, nn, policy_mod, route, "cartpole", SHAPES["cartpole"], 0, 160, out_dir=args.out)
    capture(ref, nn, policy_mod, route, "cartpole", SHAPES["cartpole"], 0, 160, policy="gumbel",
            qtransform="qtransform_completed_by_mix_value", out_dir=args.out)
    capture(ref, nn, policy_mod, route, "wide12", SHAPES["wide12"], 0, 50, out_dir=args.out)
    capture(ref, nn, policy_mod, route, "wide12", SHAPES["wide12"], 0, 50, policy="gumbel",
            qtransform="qtransform_completed_by_mix_value", out_dir=args.out)
    model, params, w, pred_on = first
    capture_checkpoint(model, w, args.out)
    capture_rollout(model, params, w, args.out, pred_on)
    capture_resnet(ref, nn, policy_mod, args.out)


if __name__ == "__main__":
    main()
