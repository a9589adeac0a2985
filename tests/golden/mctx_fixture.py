"""Format, reader and comparer of the mctx/JAX golden captures (tests/golden/mctx_*.npz).

A capture is written by tests/golden/capture_from_mctx.py on a machine that HAS jax + mctx + dm-haiku (and the
reference's muax glue); this build container and the GPU box have none of them, so the files are absent until
someone runs that script ("Pinning the oracle").  When present, tests/test_mctx_pin_cpu.py compares
the CPU oracle with them and tests/test_gpu_mctx_pin.py the HIP path: integers exact, floats to 1e-5
(BASELINE.json's bar).  This module is test infrastructure: nothing under muax_amd/ imports it.

One capture = one `MuZero._plan(params, key, obs, ...)` call of the reference (muax/model.py:222-243) on the default
MLP trio, B roots:

  meta                json: policy, shapes, keyword arguments, recurrent_pred_on, library versions, route
  w_<name>            the 18 weight arrays in the C-ABI's naming (include/mzsearch.h; haiku layout w[in][out])
  obs [B, obs_dim]    f32;  key uint32[2] (jax.random.PRNGKey data)
  action [B] i32, action_weights [B, A] f32, root_value [B] f32
  tree_<field>        mctx.Tree arrays: node_visits, parents, action_from_parent [B, N] i32; raw_values, node_values
                      [B, N] f32; children_index, children_visits [B, N, A] i32; children_prior_logits,
                      children_values, children_rewards, children_discounts [B, N, A] f32; embeddings [B, N, E] f32
  rng_dirichlet [B, A]            jax.random.dirichlet(split(key, 3)[1], alpha * ones(A), (B,))        (muzero)
  rng_tiebreak [S, B, D, A]       uniform(k_sel, (A,)) of simulation s, root b, selection level d < D    (muzero)
  rng_final_gumbel [B, A]         gumbel(split(key, 3)[0], (B, A)): the categorical draw of the action   (muzero)
  rng_root_gumbel [B, A]          gumbel(split(key)[1], (B, A))                                          (gumbel)
"""
from __future__ import annotations

import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FORMAT_VERSION = 1
WEIGHT_NAMES = ("repr_w", "repr_b", "pv_w1", "pv_b1", "pv_w2", "pv_b2", "pp_w1", "pp_b1", "pp_w2", "pp_b2",
                "dr_w1", "dr_b1", "dr_w2", "dr_b2", "dn_w1", "dn_b1", "dn_w2", "dn_b2")
TREE_INT = ("node_visits", "parents", "action_from_parent", "children_index", "children_visits")
TREE_FLOAT = ("raw_values", "node_values", "children_prior_logits", "children_values", "children_rewards",
              "children_discounts", "embeddings")
FLOAT_TOL = 1e-5  # BASELINE.json: "within 1e-5 on value/policy logits"


def fixture_paths():
    """Real per-call captures present in tests/golden (mctx_<shape>_<policy>_s<S>_seed<k>[_mix].npz; not the
    checkpoint's expected weights, not the fit-loop trace, never the synthetic self-test files)."""
    return sorted(glob.glob(os.path.join(HERE, "mctx_*_seed*.npz")))


ROLLOUT_PATH = os.path.join(HERE, "mctx_rollout_cartpole_s10.npz")


def load_rollout(path=ROLLOUT_PATH):
    """The fit-loop trace of the capture script (muax/train.py:153-170: key, subkey = split(key); act(subkey, obs,
    with_pi=True, with_value=True, obs_from_batch=False, num_simulations=10, temperature=1)): per step the sub-key, the
    observation and what act() returned, plus the flattened weights."""
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return {"meta": meta, "w": {k: z["w_" + k] for k in WEIGHT_NAMES}, "subkey": z["subkey"], "obs": z["obs"],
            "action": z["action"], "pi": z["pi"], "v": z["v"]}


def save_case(path, meta: dict, weights: dict, obs, key, outputs: dict, tree: dict, rng: dict):
    meta = dict(meta, format_version=FORMAT_VERSION)
    data = {"meta": np.array(json.dumps(meta, sort_keys=True)), "obs": np.asarray(obs, np.float32),
            "key": np.asarray(key, np.uint32).reshape(2)}
    for k in WEIGHT_NAMES:
        data["w_" + k] = np.asarray(weights[k], np.float32)
    data["action"] = np.asarray(outputs["action"], np.int32)
    data["action_weights"] = np.asarray(outputs["action_weights"], np.float32)
    data["root_value"] = np.asarray(outputs["root_value"], np.float32)
    for k in TREE_INT:
        data["tree_" + k] = np.asarray(tree[k], np.int32)
    for k in TREE_FLOAT:
        data["tree_" + k] = np.asarray(tree[k], np.float32)
    for k, v in rng.items():
        data["rng_" + k] = np.asarray(v, np.float32)
    np.savez_compressed(path, **data)


def load_case(path) -> dict:
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    if meta.get("format_version") != FORMAT_VERSION:
        raise ValueError(f"{path}: capture format {meta.get('format_version')}, this reader knows {FORMAT_VERSION}")
    case = {"path": path, "meta": meta, "obs": z["obs"], "key": z["key"],
            "w": {k: z["w_" + k] for k in WEIGHT_NAMES},
            "action": z["action"], "action_weights": z["action_weights"], "root_value": z["root_value"],
            "tree": {k: z["tree_" + k] for k in TREE_INT + TREE_FLOAT},
            "rng": {k[4:]: z[k] for k in z.files if k.startswith("rng_")}}
    return case


# --------------------------------------------------------------------------------------------------------------
# comparing
# --------------------------------------------------------------------------------------------------------------
def _diff(name, want, got, exact, tol=FLOAT_TOL):
    want, got = np.asarray(want), np.asarray(got)
    if want.shape != got.shape:
        return [f"{name}: shape {got.shape}, capture has {want.shape}"]
    if exact:
        bad = np.argwhere(want != got)
        if bad.size:
            i = tuple(bad[0])
            return [f"{name}: {len(bad)} of {want.size} differ, first at {i}: {got[i]} (capture {want[i]})"]
        return []
    err = np.abs(want.astype(np.float64) - got.astype(np.float64))
    lim = tol * np.maximum(1.0, np.abs(want.astype(np.float64)))
    bad = np.argwhere(err > lim)
    if bad.size:
        i = tuple(bad[0])
        return [f"{name}: {len(bad)} of {want.size} beyond {tol:g}, worst {err.max():.3g}, first at {i}: "
                f"{got[i]!r} (capture {want[i]!r})"]
    return []


def compare_outputs(case, got, with_tree=True):
    """`got`: {"action", "action_weights", "root_value"[, "tree": {field: array}]} -> list of mismatch messages.
    Integers (actions, every index / visit array) exact; floats to FLOAT_TOL relative-or-absolute.  Embeddings of
    nodes no simulation created are compared too (both sides leave them zero)."""
    msgs = _diff("action", case["action"], got["action"], True)
    msgs += _diff("action_weights", case["action_weights"], got["action_weights"], False)
    msgs += _diff("root_value", case["root_value"], got["root_value"], False)
    if with_tree and got.get("tree") is not None:
        for k in TREE_INT:
            msgs += _diff("tree." + k, case["tree"][k], got["tree"][k], True)
        for k in TREE_FLOAT:
            msgs += _diff("tree." + k, case["tree"][k], got["tree"][k], False)
    return msgs


# --------------------------------------------------------------------------------------------------------------
# running the oracle on a capture's inputs
# --------------------------------------------------------------------------------------------------------------
def _mlp(po, case):
    m = case["meta"]
    return po.Mlp(case["w"], m["obs_dim"], m["E"], m["A"], 2 * m["support_size"] + 1, discount=m["discount"],
                  support_size=m["support_size"], recurrent_pred_on=1 if m["recurrent_pred_on"] == "parent" else 0)


def oracle_rng(po, case) -> dict:
    """The PRNG intermediates of the capture, recomputed with the oracle's threefry walk (RNG
    stream).  Uniform bits are exact integers scaled by powers of two -> compared exactly; Dirichlet / Gumbel go
    through log / erf_inv -> FLOAT_TOL."""
    m, key = case["meta"], [int(x) for x in case["key"]]
    B, A, S = case["obs"].shape[0], m["A"], m["num_simulations"]
    out = {}
    if m["policy"] == "muzero":
        k_sample, k_dir, sim_keys = po.sim_keys_from_act_key(key, S)
        out["dirichlet"] = po.dirichlet(k_dir, m["dirichlet_alpha"], B, A)
        out["final_gumbel"] = po.gumbel(k_sample, B * A).reshape(B, A)
        if "tiebreak" in case["rng"]:
            D = case["rng"]["tiebreak"].shape[2]
            tb = np.zeros((S, B, D, A), np.float32)
            for s in range(S):
                roots = po.split(sim_keys[s], B)
                for b in range(B):
                    rk = roots[b]
                    for d in range(D):
                        two = po.split(rk, 2)
                        rk = two[0]
                        tb[s, b, d] = po.uniform(two[1], A)
            out["tiebreak"] = tb
    else:
        out["root_gumbel"] = po.gumbel(po.split(key, 2)[1], B * A).reshape(B, A)
    return out


def compare_rng(case, got) -> list:
    msgs = []
    for k, want in case["rng"].items():
        if k in got:
            msgs += _diff("rng." + k, want, got[k], exact=(k == "tiebreak"))
    return msgs


def oracle_run(po, case, dirichlet_from="oracle", tiebreak_from="oracle", gumbel_from="oracle", override=None) -> dict:
    """The oracle's act() on the capture's inputs.  `dirichlet_from`: "oracle" draws the root noise with the oracle's
    restatement of jax.random.dirichlet (everything from the key, as a caller gets it); "capture" injects the
    captured array (isolates the search from the sampler's float bits).  `tiebreak_from` / `gumbel_from` = "capture"
    inject the captured per-level tie-break uniforms and the final / root Gumbel array the same way (step-wise driver:
    PRNG and search pinned independently).  `override` {(simulation, root): (parent, action)} forces decisions (the
    self-tests build a "reference" that decided differently somewhere)."""
    m, key = case["meta"], [int(x) for x in case["key"]]
    B, A, S, E = case["obs"].shape[0], m["A"], m["num_simulations"], m["E"]
    mlp = _mlp(po, case)
    if m["policy"] == "muzero":
        if dirichlet_from == "capture":
            noise = case["rng"]["dirichlet"]
        else:
            noise = po.dirichlet(po.split(key, 3)[1], m["dirichlet_alpha"], B, A)
        cfg = po.SearchCfg(S, max_depth=m.get("max_depth") or 0, pb_c_init=m["pb_c_init"], pb_c_base=m["pb_c_base"],
                           tiebreak=1)
        if tiebreak_from == "oracle" and gumbel_from == "oracle" and not override:
            out = po.act_mlp(mlp, cfg, case["obs"], key, noise, m["dirichlet_fraction"], None, m["temperature"])
            return {"action": out["action"], "action_weights": out["action_weights"], "root_value": out["root_value"],
                    "tree": out["tree"].arrays()}
        k_sample, _, sim_keys = po.sim_keys_from_act_key(key, S)
        pl, v, emb = po.root_inference(mlp, case["obs"])
        tree = po.Tree(B, S + 1, A, E)
        po.tree_init(tree, po.root_prior(pl, noise, m["dirichlet_fraction"]), v, emb, None)
        tb = case["rng"]["tiebreak"] if tiebreak_from == "capture" else None
        for sim in range(S):
            if tb is not None:
                p_, a_, _ = po.step_select_injected(tree, cfg, sim, sim_keys[sim], tb[sim])
            else:
                p_, a_, _ = po.step_select(tree, cfg, sim, sim_keys[sim])
            for (s_, b_), (fp, fa) in (override or {}).items():
                if s_ == sim:
                    p_[b_], a_[b_] = fp, fa
            po.step_expand_backup(tree, sim, p_, a_, *po.recurrent_inference(mlp, a_, tree.embeddings[np.arange(B), p_]))
        g = case["rng"]["final_gumbel"] if gumbel_from == "capture" else po.gumbel(k_sample, B * A).reshape(B, A)
        action, weights = po.summary_sample(tree, m["temperature"], g)
        return {"action": action, "action_weights": weights, "root_value": v, "tree": tree.arrays()}
    kind = 1 if m["qtransform"].endswith("mix_value") else 0
    pl, v, emb = po.root_inference(mlp, case["obs"])
    if gumbel_from == "capture":
        g = case["rng"]["root_gumbel"]
    else:
        g = po.gumbel(po.split(key, 2)[1], B * A).reshape(B, A) * np.float32(m.get("gumbel_scale", 1.0))
    tree = po.Tree(B, S + 1, A, E)
    cfg = po.SearchCfg(S, max_depth=m.get("max_depth") or 0)
    po.tree_init(tree, po.mask_root_logits(pl, None), v, emb, None)
    for sim in range(S):
        p_, a_, _ = po.gumbel_step_select(tree, cfg, g, kind, m["max_num_considered_actions"])
        for (s_, b_), (fp, fa) in (override or {}).items():
            if s_ == sim:
                p_[b_], a_[b_] = fp, fa
        po.step_expand_backup(tree, sim, p_, a_, *po.recurrent_inference(mlp, a_, tree.embeddings[np.arange(B), p_]))
    action, weights = po.gumbel_finish(tree, g, kind)
    return {"action": action, "action_weights": weights, "root_value": v, "tree": tree.arrays()}


def synthetic_case(po, path, policy="muzero", seed=0, B=8, obs_dim=4, E=8, A=2, S=10, key=(0, 42), D=4, weights=None,
                   override=None):
    """A file in the capture format whose 'reference' side is the oracle itself -- NOT a pin: it exists so that the
    reader, the comparers and the HIP-side harness are exercised (and shown to fail on a perturbed file) before any
    real capture exists.  Written to a temporary path by the tests, never into tests/golden."""
    support = 10
    w = weights or po.random_mlp_weights(seed, obs_dim, E, A, 2 * support + 1, bias_scale=0.1)
    obs = np.random.default_rng(seed + 7).uniform(-1, 1, (B, obs_dim)).astype(np.float32)
    meta = {"policy": policy, "A": A, "E": E, "obs_dim": obs_dim, "num_simulations": S, "support_size": support,
            "discount": 0.99, "temperature": 1.0, "dirichlet_fraction": 0.25, "dirichlet_alpha": 0.3,
            "pb_c_init": 1.25, "pb_c_base": 19652.0, "max_depth": None, "recurrent_pred_on": "child",
            "qtransform": "qtransform_by_parent_and_siblings", "max_num_considered_actions": 16, "gumbel_scale": 1.0,
            "route": "synthetic (oracle output in the capture format; not a reference output)", "versions": {}}
    case = {"meta": meta, "w": w, "obs": obs, "key": np.array(key, np.uint32), "rng": {"tiebreak": np.zeros((S, B, D, A))}}
    rng = oracle_rng(po, case)
    out = oracle_run(po, case, override=override)  # `override`: a "reference" that decided differently somewhere
    save_case(path, meta, w, obs, key, out, out["tree"], rng)
    return path


def oracle_rollout_mismatches(po, tr) -> list:
    """The oracle's act() for every recorded (sub-key, observation) of a fit-loop trace against what the trace holds:
    action exact, pi and value to FLOAT_TOL."""
    S = tr["meta"]["num_simulations"]
    mlp = po.Mlp(tr["w"], 4, 8, 2, 21, recurrent_pred_on=1 if tr["meta"].get("recurrent_pred_on") == "parent" else 0)
    msgs = []
    for t in range(len(tr["action"])):
        key = [int(x) for x in tr["subkey"][t]]
        noise = po.dirichlet(po.split(key, 3)[1], 0.3, 1, 2)
        out = po.act_mlp(mlp, po.SearchCfg(S, tiebreak=1), tr["obs"][t][None], key, noise, 0.25, None, 1.0)
        msgs += _diff(f"step {t} action", np.asarray([tr["action"][t]]), out["action"], True)
        msgs += _diff(f"step {t} pi", np.asarray(tr["pi"][t]).reshape(1, 2), out["action_weights"], False)
        msgs += _diff(f"step {t} value", np.asarray([tr["v"][t]], np.float32), out["root_value"], False)
    return msgs


def synthetic_rollout(po, path, steps=6, S=10, seed=0):
    """A fit-loop trace in the capture script's format whose 'reference' side is the oracle (NOT a pin; exercises the
    reader and both consumers before a real trace exists)."""
    w = po.random_mlp_weights(seed, 4, 8, 2, 21, bias_scale=0.1)
    rng = np.random.default_rng(seed)
    mlp = po.Mlp(w, 4, 8, 2, 21)
    rec = {k: [] for k in ("subkey", "obs", "action", "pi", "v")}
    key = [0, 7]
    for _ in range(steps):
        key, sub = [list(map(int, k)) for k in po.split(key, 2)]
        obs = rng.uniform(-1, 1, 4).astype(np.float32)
        noise = po.dirichlet(po.split(sub, 3)[1], 0.3, 1, 2)
        out = po.act_mlp(mlp, po.SearchCfg(S, tiebreak=1), obs[None], sub, noise, 0.25, None, 1.0)
        rec["subkey"].append(np.asarray(sub, np.uint32)); rec["obs"].append(obs)
        rec["action"].append(int(out["action"][0])); rec["pi"].append(out["action_weights"].copy())
        rec["v"].append(float(out["root_value"][0]))
    data = {k: np.asarray(v) for k, v in rec.items()}
    data.update({"w_" + k: v for k, v in w.items()})
    data["meta"] = np.array(json.dumps({"num_simulations": S, "recurrent_pred_on": "child", "versions": {},
                                        "format_version": FORMAT_VERSION}))
    np.savez_compressed(path, **data)
    return path


# --------------------------------------------------------------------------------------------------------------
# the convolutional (ResNet) plugin nets: one capture of root inference + one recurrent_fn call
# --------------------------------------------------------------------------------------------------------------
RESNET_PATH = os.path.join(HERE, "mctx_resnet_nets_seed0.npz")


def resnet_weights(manifest, seed):
    """The parameters of a ResNet-net capture, regenerated from its manifest [(net, haiku module, param, shape), ...]
    (the haiku parameter tree in CALL order): the capture script writes these very arrays INTO the haiku tree before
    it runs the reference, so the file carries a manifest and a seed instead of megabytes of weights.  NumPy's
    default_rng stream is stable across versions."""
    rng = np.random.default_rng(seed)
    out = []
    for _net, _module, param, shape in manifest:
        shape = tuple(int(x) for x in shape)
        x = rng.standard_normal(shape).astype(np.float32)
        if param == "w":
            x = x / np.float32(np.sqrt(float(np.prod(shape[:-1]))))
        elif param == "scale":
            x = np.float32(1) + np.float32(0.1) * x
        else:  # offsets and biases
            x = np.float32(0.1) * x
        out.append(np.ascontiguousarray(x, np.float32))
    return out


def resnet_param_slots(mods):
    """The torch mirror's parameters (muax_amd/nn.py ResNet* modules, materialised by one forward pass) in the CALL
    order of the reference's modules (muax/nn.py:118-148,313-378: projection before conv_0 in a block; v_func before
    pi_func in ResNetPrediction.__call__; r_func before ns_func in ResNetDynamic.__call__) as [(label, param, tensor)]."""
    rep, pred, dyn = mods

    def conv(m, label):
        return [(label, "w", m.w)]

    def ln(m, label):
        return [(label, "scale", m.scale), (label, "offset", m.offset)]

    def block(b, label):
        return (conv(b.proj_conv, label + ".proj_conv") + ln(b.proj_ln, label + ".proj_ln") + conv(b.conv_0, label + ".conv_0")
                + ln(b.ln_0, label + ".ln_0") + conv(b.conv_1, label + ".conv_1") + ln(b.ln_1, label + ".ln_1"))

    def head(seq, label):
        out = []
        for i, m in enumerate(seq):
            if hasattr(m, "w") and getattr(m, "b", None) is not None:
                out += [(f"{label}[{i}]", "w", m.w), (f"{label}[{i}]", "b", m.b)]
            elif hasattr(m, "w"):
                out += conv(m, f"{label}[{i}]")
        return out

    slots = {"representation": conv(rep.stem0, "stem0")}
    for i, b in enumerate(rep.blocks0):
        slots["representation"] += block(b, f"blocks0[{i}]")
    slots["representation"] += conv(rep.stem1, "stem1")
    for name in ("blocks1", "blocks2"):
        for i, b in enumerate(getattr(rep, name)):
            slots["representation"] += block(b, f"{name}[{i}]")
    slots["prediction"] = head(pred.v_func, "v_func") + head(pred.pi_func, "pi_func")
    slots["dynamic"] = head(dyn.r_func, "r_func") + conv(dyn.ns_stem, "ns_stem")
    for i, b in enumerate(dyn.ns_blocks):
        slots["dynamic"] += block(b, f"ns_blocks[{i}]")
    return slots


def resnet_assign(mods, manifest, seed):
    """Write the capture's parameters into the torch mirror, slot by slot in call order; any disagreement in count,
    parameter kind or shape is a loud error that prints both sides (the haiku naming is spec-to-confirm)."""
    import torch
    slots = resnet_param_slots(mods)
    arrays = resnet_weights(manifest, seed)
    by_net = {}
    for (net, module, param, shape), arr in zip(manifest, arrays):
        by_net.setdefault(net, []).append((module, param, tuple(shape), arr))
    for net, want in by_net.items():
        have = slots[net]
        if len(want) != len(have) or any(w[1] != h[1] or w[2] != tuple(h[2].shape) for w, h in zip(want, have)):
            lines = [f"  {w[0]}/{w[1]} {w[2]}   <->   {h[0]}.{h[1]} {tuple(h[2].shape)}" for w, h in zip(want, have)]
            raise AssertionError(f"{net}: the reference's parameters in call order do not line up with the torch mirror's "
                                 f"({len(want)} vs {len(have)}):\n" + "\n".join(lines))
        with torch.no_grad():
            for (_m, _p, _s, arr), (_l, _k, t) in zip(want, have):
                t.copy_(torch.from_numpy(arr).to(t.device))


def load_resnet(path=RESNET_PATH):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    stages = None
    if "stage_names" in z.files:
        stages = {"names": [str(n) for n in z["stage_names"]], "stats": z["stage_stats"], "samples": z["stage_samples"]}
    return {"meta": meta, "manifest": [tuple(x) for x in meta["manifest"]], "seed": meta["weights_seed"],
            "obs": z["obs"], "action": z["action"], "stages": stages,
            "root": {k: z["root_" + k] for k in ("embedding", "value_logits", "prior_logits", "value")},
            "rec": {k: z["rec_" + k] for k in ("reward_logits", "next_embedding", "value_logits", "prior_logits", "reward", "value")}}


# --- one digest per STAGE of the representation net's root inference (round 5): when a capture exists and the
# embedding disagrees, the first stage whose digest differs names the layer group (stem / block / pool) at fault --
# haiku's SAME geometry of a strided stem, a LayerNorm axis, the AvgPool's counting -- instead of "the net".
STAGE_NAMES = ("stem0", "blocks0[0]", "blocks0[1]", "stem1", "blocks1[0]", "blocks1[1]", "blocks1[2]", "pool0",
               "blocks2[0]", "blocks2[1]", "blocks2[2]", "pool1")
STAGE_SAMPLES = 256


def stage_digest(x):
    """(mean, std, max |x|) and a fixed pseudo-random sample of 256 elements of one stage's output [B, H, W, C]."""
    x = np.asarray(x, np.float32).reshape(-1)
    idx = np.random.default_rng(x.size).integers(0, x.size, STAGE_SAMPLES)
    return np.array([x.mean(dtype=np.float64), x.std(dtype=np.float64), np.abs(x).max()], np.float32), x[idx]


def resnet_mirror_stages(mx, rep, obs):
    """The torch mirror's ResNetRepresentation, stage by stage (muax_amd/nn.py; the same calls as its forward()):
    [(name, tensor)] in STAGE_NAMES' order.  A stem's entry is relu(conv(x)), as the reference applies it
    (muax/nn.py:299-303)."""
    import torch
    out = []
    with torch.no_grad():
        x = rep.stem0.scaled(obs.to(torch.float32), 255., relu=True)
        out.append(x)
        for b in rep.blocks0:
            x = b(x)
            out.append(x)
        x = rep.stem1.scaled(x, relu=True)
        out.append(x)
        for b in rep.blocks1:
            x = b(x)
            out.append(x)
        x = mx.nn.avg_pool_same(x)
        out.append(x)
        for b in rep.blocks2:
            x = b(x)
            out.append(x)
        out.append(mx.nn.avg_pool_same(x))
    assert len(out) == len(STAGE_NAMES)
    return list(zip(STAGE_NAMES, out))


def compare_resnet_stages(case, stages, tol=2e-4):
    """`stages`: [(name, array)] of the side under test; the capture's digests in case["stages"] (absent in captures
    written before round 5: nothing to compare).  Messages name the FIRST diverging stage."""
    want = case.get("stages")
    if not want:
        return []
    if list(want["names"]) != [n for n, _ in stages]:
        return [f"stages: the capture recorded {list(want['names'])}, this side has {[n for n, _ in stages]}"]
    for i, (name, arr) in enumerate(stages):
        stats, sample = stage_digest(arr.detach().cpu().numpy() if hasattr(arr, "detach") else arr)
        lim = tol * max(1.0, float(want["stats"][i][2]))
        e_s = float(np.abs(sample.astype(np.float64) - want["samples"][i]).max())
        e_m = float(np.abs(stats.astype(np.float64) - want["stats"][i]).max())
        if e_s > lim or e_m > lim:
            return [f"stage {i} ({name}) is the first to diverge: sample error {e_s:.3g}, (mean, std, max) error {e_m:.3g}, "
                    f"beyond {lim:.3g}; stages before it agree"]
    return []


def save_resnet(path, meta, obs, action, root, rec, stages=None):
    data = {"meta": np.array(json.dumps(dict(meta, format_version=FORMAT_VERSION), sort_keys=True)),
            "obs": np.asarray(obs, np.uint8), "action": np.asarray(action, np.int32)}
    data.update({"root_" + k: np.asarray(v, np.float32) for k, v in root.items()})
    data.update({"rec_" + k: np.asarray(v, np.float32) for k, v in rec.items()})
    if stages:  # [(name, array)]
        digests = [stage_digest(np.asarray(a)) for _, a in stages]
        data["stage_names"] = np.array([n for n, _ in stages])
        data["stage_stats"] = np.stack([d[0] for d in digests])
        data["stage_samples"] = np.stack([d[1] for d in digests])
    np.savez_compressed(path, **data)


def resnet_mirror_outputs(mx, mods, case, support_size=10, device="cpu", hip=False):
    """Root inference and ONE recurrent_fn call of the torch mirror on the capture's inputs (muax/model.py:251-282).
    hip=True: the recurrent call through the one-launch HIP kernel (ResNetDynamic.hip_recurrent)."""
    import torch
    rep, pred, dyn = mods
    obs = torch.as_tensor(case["obs"].astype(np.float32), device=device)
    act = torch.as_tensor(case["action"], device=device)
    dec = lambda lg: mx.utils.support_to_scalar(torch.softmax(lg, -1), support_size).flatten()  # noqa: E731
    with torch.no_grad():
        s = rep(obs)
        v_lg, p_lg = pred(s)
        root = {"embedding": s, "value_logits": v_lg, "prior_logits": p_lg, "value": dec(v_lg)}
        if hip:
            r, v, pl, ns = dyn.hip_recurrent(pred, s, act, support_size)
            rec = {"next_embedding": ns, "prior_logits": pl, "reward": r, "value": v}
        else:
            r_lg, ns = dyn(s, act)
            v2, p2 = pred(ns)
            rec = {"reward_logits": r_lg, "next_embedding": ns, "value_logits": v2, "prior_logits": p2,
                   "reward": dec(r_lg), "value": dec(v2)}
    tonp = lambda d: {k: v.detach().cpu().numpy() for k, v in d.items()}  # noqa: E731
    return tonp(root), tonp(rec)


def compare_resnet(case, root, rec, tol=2e-4):
    """Floats through ~30 fp32 convolutions + LayerNorms: `tol` relative to the array's largest entry (the search's
    1e-5 bar is for the MLP trio's logits; two fp32 convolution orders differ by more than that)."""
    msgs = []
    for name, want, got in [("root." + k, case["root"][k], v) for k, v in root.items()] + \
                           [("rec." + k, case["rec"][k], v) for k, v in rec.items()]:
        want = np.asarray(want).reshape(np.asarray(got).shape)
        err = float(np.abs(want.astype(np.float64) - got.astype(np.float64)).max())
        lim = tol * max(1.0, float(np.abs(want).max()))
        if err > lim:
            msgs.append(f"{name}: max error {err:.3g} beyond {lim:.3g}")
    return msgs


def synthetic_resnet(mx, path, seed=0, B=1, hw=32, A=6, F=21, c=8, dc=16, device="cpu"):
    """A file in the ResNet capture's format whose 'reference' side is the torch mirror itself with haiku-style names
    (NOT a pin: proves manifest -> weights -> assignment -> comparison work, on small frames)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    mods = (mx.nn.ResNetRepresentation(c, generator=g), mx.nn.ResNetPrediction(A, F, generator=g),
            mx.nn.ResNetDynamic(A, F, output_channels=dc, generator=g))
    obs = np.random.default_rng(seed).integers(0, 256, (B, hw, hw, 4)).astype(np.uint8)
    act = (np.arange(B) % A).astype(np.int32)
    with torch.no_grad():
        s = mods[0](torch.as_tensor(obs.astype(np.float32)))
        mods[1](s)
        mods[2](s, torch.as_tensor(act))
    for m in mods:
        m.to(device)
    manifest = [(net, label, param, list(t.shape)) for net, sl in resnet_param_slots(mods).items() for label, param, t in sl]
    resnet_assign(mods, manifest, seed)
    case = {"obs": obs, "action": act}
    root, rec = resnet_mirror_outputs(mx, mods, case, device=device)
    stages = [(n, t.cpu().numpy()) for n, t in resnet_mirror_stages(mx, mods[0], torch.as_tensor(obs.astype(np.float32), device=device))]
    meta = {"manifest": manifest, "weights_seed": seed, "A": A, "F": F, "support_size": 10, "input_channels": c,
            "dynamic_channels": dc, "route": "synthetic (torch mirror output in the capture format; not a reference output)"}
    save_resnet(path, meta, obs, act, root, rec, stages)
    return path
