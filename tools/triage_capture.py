#!/usr/bin/env python
"""Triage of a golden capture of the reference (tests/golden/mctx_*.npz) against the CPU oracle.

When a real capture exists and the oracle's tree differs from it, this tool answers WHY, decision by decision:

    python tools/triage_capture.py tests/golden/mctx_cartpole_muzero_s50_seed0.npz [--own-rng] [--flip-margin 2e-6]

It replays the search simulation by simulation with the oracle's step-wise entry points.  The reference's decision of
simulation s is read off the captured tree (node s + 1 was created by it: `parents[s + 1]`, `action_from_parent[s + 1]`
and the chain of parents up to the root give the whole selection path).  Wherever the oracle's own selection differs,
the FIRST level of the descent at which the two part is reported with the oracle's arithmetic for that decision --
value scores, policy scores, the tie-break noise of that level, visit counts -- and the MARGIN by which the oracle
preferred its action.  Then the reference's decision is forced ("teacher forcing") so that every later simulation is
still compared on the reference's tree, and at the end the forced tree's floats are compared with the capture's.

How to read the report:
  * margin <= --flip-margin (default 2e-6: a few ulps of a score of order 1, above the 1e-7 tie-break noise): a NEAR-TIE
    FLIP -- two correct fp32 evaluations (XLA's exp/log polynomials, Eigen's matmul order vs this restatement's) may
    legitimately order such a pair differently; the trees diverge from there without either side being wrong;
  * a larger margin, or forced-tree floats beyond 1e-5: a SEMANTIC difference -- the restatement of mctx is wrong
    somewhere (the report names the simulation, root, node and level to look at).

By default the capture's own PRNG intermediates are injected (Dirichlet noise, per-level tie-break uniforms, final
Gumbel) so that the SEARCH is judged independently of the sampler restatements; `--own-rng` draws everything from the
key with the oracle's threefry / Dirichlet restatement instead, and the PRNG arrays themselves are always compared
first.  Test infrastructure: imports oracle/ and tests/golden/mctx_fixture.py, nothing under muax_amd/ imports it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import mctx_fixture as fx  # noqa: E402

FLIP_MARGIN = 2e-6


def reference_decisions(case):
    """Per simulation s and root b: the reference's (parent, action) and its selection path [(node, action), ...]
    from the root, read off the captured tree.  -1 / None where node s + 1 was never created (a `max_depth` cut
    re-expanded an existing node: the capture alone does not say which)."""
    t = case["tree"]
    B, N = t["parents"].shape
    S = N - 1
    parent = np.full((S, B), -1, np.int32)
    action = np.full((S, B), -1, np.int32)
    paths = [[None] * B for _ in range(S)]
    for s in range(S):
        for b in range(B):
            p, a = int(t["parents"][b, s + 1]), int(t["action_from_parent"][b, s + 1])
            if p < 0:
                continue
            parent[s, b], action[s, b] = p, a
            path, node = [(p, a)], p
            while node != 0:
                path.append((int(t["parents"][b, node]), int(t["action_from_parent"][b, node])))
                node = path[-1][0]
            paths[s][b] = path[::-1]
    return parent, action, paths


def _level_report(po, tree, cfg, b, path, own_path, uniforms, flip_margin):
    """First level at which the oracle's descent (`own_path`) leaves the reference's (`path`); the oracle's scores there."""
    for d, (node, a_ref) in enumerate(path):
        if d >= len(own_path) or own_path[d][0] != node:
            return {"level": d, "note": "the oracle's descent ended above this level (a child the reference has visited "
                                        "is unvisited here, or max_depth cut differently)"}
        a_own = own_path[d][1]
        if a_own == a_ref:
            continue
        vs, ps = po.action_scores(tree, cfg, b, node)
        u = None if uniforms is None or d >= uniforms.shape[0] else np.asarray(uniforms[d], np.float32)
        total = vs + ps
        if u is not None:
            total = total + np.float32(1e-7) * u
        margin = float(total[a_own]) - float(total[a_ref])
        return {"level": d, "node": int(node), "action_reference": int(a_ref), "action_oracle": int(a_own),
                "margin": margin, "value_score": vs.tolist(), "policy_score": ps.tolist(),
                "tiebreak_uniform": None if u is None else u.tolist(), "score_with_noise": total.tolist(),
                "node_visits": int(tree.node_visits[b, node]), "children_visits": tree.children_visits[b, node].tolist(),
                "kind": "near-tie flip" if abs(margin) <= flip_margin else "semantic"}
    return {"level": len(path), "note": "same path, different leaf bookkeeping"}


def _own_path(po, tree, cfg, b, sim_key, uniforms, D_cap):
    """The oracle's own descent for root b as [(node, action), ...] (python walk over the oracle's select_action via
    its score export: value + policy + noise, first max wins, invalid root actions masked)."""
    A = tree.A
    max_depth = cfg.max_depth or cfg.num_simulations
    key = po.split(sim_key, cfg.global_batch or tree.B)[cfg.root_offset + b] if cfg.tiebreak else None
    node, depth, path = 0, 0, []
    while True:
        noise = None
        if cfg.tiebreak:
            two = po.split(key, 2)
            key = two[0]
            u = uniforms[depth] if (uniforms is not None and depth < D_cap) else po.uniform(two[1], A)
            noise = np.float32(1e-7) * np.asarray(u, np.float32)
        vs, ps = po.action_scores(tree, cfg, b, node)
        score = vs + ps
        if noise is not None:
            score = score + noise
        if depth == 0:
            score = np.where(tree.root_invalid_actions[b] != 0, -np.inf, score)
        a = int(np.argmax(score))  # first max
        path.append((node, a))
        nxt = int(tree.children_index[b, node, a])
        depth += 1
        if nxt == -1 or depth >= max_depth:
            return path
        node = nxt


def triage(po, case, own_rng=False, flip_margin=FLIP_MARGIN, max_reports=50):
    """-> report dict (see the module docstring).  MuZero policy captures; Gumbel captures get the decision
    comparison without per-level scores (their interior selection has no noise and no near ties by construction)."""
    m, key = case["meta"], [int(x) for x in case["key"]]
    B, A, S, E = case["obs"].shape[0], m["A"], m["num_simulations"], m["E"]
    mlp = fx._mlp(po, case)
    rep = {"capture": os.path.basename(case.get("path", "?")), "policy": m["policy"], "roots": B, "num_simulations": S,
           "versions": m.get("versions", {}), "rng_from": "oracle (key)" if own_rng else "capture (injected)",
           "flip_margin": flip_margin}
    rep["rng_mismatches"] = fx.compare_rng(case, fx.oracle_rng(po, case))
    ref_parent, ref_action, ref_paths = reference_decisions(case)
    pl, v, emb = po.root_inference(mlp, case["obs"])
    rep["root_mismatches"] = fx._diff("root_value", case["root_value"], v, False) + \
        fx._diff("root embedding", case["tree"]["embeddings"][:, 0], emb, False)
    tree = po.Tree(B, S + 1, A, E)
    divergences = []
    if m["policy"] == "muzero":
        cfg = po.SearchCfg(S, max_depth=m.get("max_depth") or 0, pb_c_init=m["pb_c_init"], pb_c_base=m["pb_c_base"], tiebreak=1)
        k_sample, k_dir, sim_keys = po.sim_keys_from_act_key(key, S)
        noise = po.dirichlet(k_dir, m["dirichlet_alpha"], B, A) if own_rng or "dirichlet" not in case["rng"] \
            else case["rng"]["dirichlet"]
        tb = None if own_rng else case["rng"].get("tiebreak")
        po.tree_init(tree, po.root_prior(pl, noise, m["dirichlet_fraction"]), v, emb, None)
        rep["root_mismatches"] += fx._diff("root prior logits (after the noise mix)",
                                           case["tree"]["children_prior_logits"][:, 0], tree.children_prior_logits[:, 0], False)
        for s in range(S):
            if tb is not None:
                p_, a_, _ = po.step_select_injected(tree, cfg, s, sim_keys[s], tb[s])
            else:
                p_, a_, _ = po.step_select(tree, cfg, s, sim_keys[s])
            for b in range(B):
                if ref_parent[s, b] < 0 or (p_[b] == ref_parent[s, b] and a_[b] == ref_action[s, b]):
                    continue
                own = _own_path(po, tree, cfg, b, sim_keys[s], None if tb is None else tb[s, b], 0 if tb is None else tb.shape[2])
                assert own[-1] == (int(p_[b]), int(a_[b])), "the python walk disagrees with the C oracle's walk"
                d = {"simulation": s, "root": b, "reference": [int(ref_parent[s, b]), int(ref_action[s, b])],
                     "oracle": [int(p_[b]), int(a_[b])]}
                d.update(_level_report(po, tree, cfg, b, ref_paths[s][b], own, None if tb is None else tb[s, b], flip_margin))
                d.setdefault("kind", "semantic")
                divergences.append(d)
            known = ref_parent[s] >= 0  # teacher forcing: expand what the REFERENCE selected
            fp = np.where(known, ref_parent[s], p_).astype(np.int32)
            fa = np.where(known, ref_action[s], a_).astype(np.int32)
            po.step_expand_backup(tree, s, fp, fa, *po.recurrent_inference(mlp, fa, tree.embeddings[np.arange(B), fp]))
        g = po.gumbel(k_sample, B * A).reshape(B, A) if own_rng or "final_gumbel" not in case["rng"] else case["rng"]["final_gumbel"]
        action, weights = po.summary_sample(tree, m["temperature"], g)
    else:
        kind = 1 if m["qtransform"].endswith("mix_value") else 0
        cfg = po.SearchCfg(S, max_depth=m.get("max_depth") or 0)
        g = po.gumbel(po.split(key, 2)[1], B * A).reshape(B, A) * np.float32(m.get("gumbel_scale", 1.0)) \
            if own_rng or "root_gumbel" not in case["rng"] else case["rng"]["root_gumbel"]
        po.tree_init(tree, po.mask_root_logits(pl, None), v, emb, None)
        for s in range(S):
            p_, a_, _ = po.gumbel_step_select(tree, cfg, g, kind, m["max_num_considered_actions"])
            for b in range(B):
                if ref_parent[s, b] >= 0 and (p_[b] != ref_parent[s, b] or a_[b] != ref_action[s, b]):
                    q = po.qtransform(tree, np.where(ref_parent[s] >= 0, ref_parent[s], 0), kind)[b]
                    divergences.append({"simulation": s, "root": b, "reference": [int(ref_parent[s, b]), int(ref_action[s, b])],
                                        "oracle": [int(p_[b]), int(a_[b])], "kind": "semantic",
                                        "completed_q_at_reference_parent": np.asarray(q).tolist(),
                                        "note": "Gumbel MuZero selection is deterministic in the node statistics"})
            known = ref_parent[s] >= 0
            fp = np.where(known, ref_parent[s], p_).astype(np.int32)
            fa = np.where(known, ref_action[s], a_).astype(np.int32)
            po.step_expand_backup(tree, s, fp, fa, *po.recurrent_inference(mlp, fa, tree.embeddings[np.arange(B), fp]))
        action, weights = po.gumbel_finish(tree, g, kind)
    got = {"action": action, "action_weights": weights, "root_value": v, "tree": tree.arrays()}
    rep["forced_tree_mismatches"] = fx.compare_outputs(case, got)
    rep["n_divergences"] = len(divergences)
    rep["n_near_tie_flips"] = sum(d["kind"] == "near-tie flip" for d in divergences)
    rep["n_semantic"] = sum(d["kind"] == "semantic" for d in divergences)
    rep["divergences"] = divergences[:max_reports]
    if not divergences and not rep["forced_tree_mismatches"] and not rep["root_mismatches"]:
        rep["verdict"] = "PINNED: every decision of every simulation equals the reference's, floats within 1e-5"
    elif rep["n_semantic"] == 0 and not rep["forced_tree_mismatches"] and not rep["root_mismatches"]:
        rep["verdict"] = (f"NEAR-TIE FLIPS ONLY: {rep['n_near_tie_flips']} decisions differ by margins <= {flip_margin:g}; with the "
                          "reference's decisions forced the tree agrees to 1e-5 -- last-bit float differences, not semantics")
    else:
        rep["verdict"] = ("SEMANTIC DIFFERENCE: see `divergences` with kind == 'semantic' (first: simulation, root, node, level) "
                          "and `forced_tree_mismatches` / `root_mismatches`")
    return rep


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("capture", nargs="+")
    ap.add_argument("--own-rng", action="store_true", help="draw Dirichlet / tie-break / Gumbel from the key with the oracle")
    ap.add_argument("--flip-margin", type=float, default=FLIP_MARGIN)
    ap.add_argument("--json", action="store_true", help="one JSON report per capture instead of text")
    args = ap.parse_args()
    from oracle import pyoracle as po
    worst = 0
    for path in args.capture:
        rep = triage(po, fx.load_case(path), args.own_rng, args.flip_margin)
        if args.json:
            print(json.dumps(rep))
        else:
            print(f"== {rep['capture']}  ({rep['policy']}, {rep['roots']} roots x {rep['num_simulations']} simulations, "
                  f"rng from {rep['rng_from']}; {rep['versions']})")
            for k in ("rng_mismatches", "root_mismatches", "forced_tree_mismatches"):
                for msg in rep[k]:
                    print(f"   {k[:-11]}: {msg}")
            for d in rep["divergences"]:
                print(f"   sim {d['simulation']:3d} root {d['root']:3d}: reference (parent, action) = {tuple(d['reference'])}, oracle "
                      f"{tuple(d['oracle'])}  [{d['kind']}]")
                if "margin" in d:
                    print(f"       level {d['level']} node {d['node']} (visits {d['node_visits']}, children {d['children_visits']}): "
                          f"margin {d['margin']:.3g}\n       value_score {d['value_score']}\n       policy_score {d['policy_score']}\n"
                          f"       tie-break uniform {d['tiebreak_uniform']}")
                elif "note" in d:
                    print("       " + d["note"])
            print("   ->", rep["verdict"])
        worst = max(worst, 2 if rep["verdict"].startswith("SEMANTIC") else (1 if rep["verdict"].startswith("NEAR") else 0))
    raise SystemExit(worst)


if __name__ == "__main__":
    main()
