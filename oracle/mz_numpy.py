"""Independent NumPy restatement of the reference path (TEST INFRASTRUCTURE).

Second opinion for ``oracle/mz_oracle.c``: same algorithm, written the way
``mctx`` writes it (simulation-major, whole batch at once, masked while-loops),
with NumPy's own exp/log/sum instead of the pinned MZ-F32 routines.  Floats
therefore agree with the C oracle only to rounding (~1e-6); integer outputs
agree exactly wherever no argmax was decided by less than that rounding, which
``min_margin`` reports per root.

Reference anchors: muax/model.py:222-282 (glue), muax/nn.py:37-44,59-115 (nets),
muax/utils.py:70-102 (codec); mctx 0.0.5 policies/search/action_selection/
qtransforms/tree (third-party, restated from the published algorithm).
PARITY UNPINNED -- see oracle/mz_oracle.h.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
TINY = np.finfo(np.float32).tiny
FMIN = np.finfo(np.float32).min


# ---- muax/utils.py:70-102 ------------------------------------------------

def inv_scaling(x, eps=1e-3):
    x = np.asarray(x, F32)
    return (np.sign(x) * (((np.sqrt(F32(1) + F32(4 * eps) * (np.abs(x) + F32(1) + F32(eps)))
                            - F32(1)) / F32(2 * eps)) ** 2 - F32(1))).astype(F32)


def scaling(x, eps=1e-3):
    x = np.asarray(x, F32)
    return (np.sign(x) * (np.sqrt(np.abs(x) + F32(1)) - F32(1)) + F32(eps) * x).astype(F32)


def softmax(x):
    x = np.asarray(x, F32)
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return (e / e.sum(axis=-1, keepdims=True)).astype(F32)


def support_to_scalar(probs, support_size):
    bins = (np.arange(2 * support_size + 1) - support_size).astype(F32)
    return inv_scaling((bins * probs).sum(axis=-1).astype(F32))


def scalar_to_support(x, support_size):
    x = np.clip(scaling(x), -support_size, support_size)
    low = np.floor(x).astype(np.int32)
    high = np.ceil(x).astype(np.int32)
    p_high = x - low
    p_low = F32(1) - p_high
    out = np.zeros(x.shape + (2 * support_size + 1,), F32)
    idx = np.indices(x.shape)
    np.add.at(out, tuple(idx) + (low + support_size,), p_low)
    np.add.at(out, tuple(idx) + (high + support_size,), p_high)
    return out


# ---- muax/nn.py -----------------------------------------------------------

def min_max_normalize(s):
    mn = s.min(axis=1, keepdims=True)
    mx = s.max(axis=1, keepdims=True)
    scale = mx - mn
    scale = np.where(scale < F32(1e-5), scale + F32(1e-5), scale)
    return ((s - mn) / scale).astype(F32)


def elu(x):
    return np.where(x > 0, x, np.expm1(np.where(x > 0, 0, x))).astype(F32)


def _mlp2(x, w1, b1, w2, b2):
    return (elu(x @ w1 + b1) @ w2 + b2).astype(F32)


def prediction(w, s):
    return _mlp2(s, w["pv_w1"], w["pv_b1"], w["pv_w2"], w["pv_b2"]), \
        _mlp2(s, w["pp_w1"], w["pp_b1"], w["pp_w2"], w["pp_b2"])


def root_inference(w, obs, support_size):
    s = min_max_normalize((obs @ w["repr_w"] + w["repr_b"]).astype(F32))
    v_logits, pi_logits = prediction(w, s)
    return pi_logits, support_to_scalar(softmax(v_logits), support_size), s


def recurrent_inference(w, action, emb, support_size, discount, A, pred_on="child"):
    sa = np.concatenate([emb, np.eye(A, dtype=F32)[action]], axis=1)
    r_logits = _mlp2(sa, w["dr_w1"], w["dr_b1"], w["dr_w2"], w["dr_b2"])
    ns = min_max_normalize(_mlp2(sa, w["dn_w1"], w["dn_b1"], w["dn_w2"], w["dn_b2"]))
    v_logits, pi_logits = prediction(w, emb if pred_on == "parent" else ns)
    r = support_to_scalar(softmax(r_logits), support_size)
    v = support_to_scalar(softmax(v_logits), support_size)
    return r, np.full_like(r, discount), pi_logits, v, ns


# ---- mctx restated --------------------------------------------------------

class Tree:
    def __init__(self, B, N, A, E):
        self.B, self.N, self.A, self.E = B, N, A, E
        self.node_visits = np.zeros((B, N), np.int32)
        self.raw_values = np.zeros((B, N), F32)
        self.node_values = np.zeros((B, N), F32)
        self.parents = np.full((B, N), -1, np.int32)
        self.action_from_parent = np.full((B, N), -1, np.int32)
        self.children_index = np.full((B, N, A), -1, np.int32)
        self.children_prior_logits = np.zeros((B, N, A), F32)
        self.children_values = np.zeros((B, N, A), F32)
        self.children_visits = np.zeros((B, N, A), np.int32)
        self.children_rewards = np.zeros((B, N, A), F32)
        self.children_discounts = np.zeros((B, N, A), F32)
        self.embeddings = np.zeros((B, N, E), F32)
        self.root_invalid_actions = np.zeros((B, A), np.uint8)

    def arrays(self):
        return {k: v for k, v in self.__dict__.items() if isinstance(v, np.ndarray)
                and k != "root_invalid_actions"}


def root_prior(prior_logits, noise, fraction, invalid=None):
    p = softmax(prior_logits)
    if noise is not None:
        p = (F32(1 - fraction) * p + F32(fraction) * noise).astype(F32)
    logits = np.log(np.maximum(p, TINY)).astype(F32)
    if invalid is not None:
        logits = logits - logits.max(axis=-1, keepdims=True)
        logits = np.where(invalid.astype(bool), FMIN, logits).astype(F32)
    return logits


def _select(tree, node, depth, pb_c_init, pb_c_base, noise, rows):
    """muzero_action_selection for roots `rows` at `node[rows]`.  Returns the
    action and the top1-top2 score margin."""
    vc = tree.children_visits[rows, node]
    nv = tree.node_visits[rows, node]
    pb_c = F32(pb_c_init) + np.log((nv + F32(pb_c_base) + F32(1)) / F32(pb_c_base)).astype(F32)
    prior = softmax(tree.children_prior_logits[rows, node])
    policy = (np.sqrt(nv.astype(F32)) * pb_c)[:, None] * prior / (vc + 1).astype(F32)
    q = tree.children_rewards[rows, node] + tree.children_discounts[rows, node] * \
        tree.children_values[rows, node]
    nval = tree.node_values[rows, node][:, None]
    safe = np.where(vc > 0, q, nval)
    lo = np.minimum(nval, safe.min(axis=-1, keepdims=True))
    hi = np.maximum(nval, safe.max(axis=-1, keepdims=True))
    value = (np.where(vc > 0, q, lo) - lo) / np.maximum(hi - lo, F32(1e-8))
    score = (value + policy).astype(F32)
    if noise is not None:
        score = score + noise
    inv = tree.root_invalid_actions[rows].astype(bool) & (depth[rows] == 0)[:, None]
    score = np.where(inv, -np.inf, score)
    srt = np.sort(score, axis=-1)
    margin = srt[:, -1] - srt[:, -2] if score.shape[1] > 1 else np.full(len(rows), np.inf)
    return score.argmax(axis=-1).astype(np.int32), margin


def search(tree, recurrent_fn, S, max_depth=None, pb_c_init=1.25, pb_c_base=19652.0,
           noise_fn=None):
    """mctx.search.  noise_fn(sim, level, rows) -> [len(rows), A] scaled tie-break
    noise or None.  Returns (min_margin[B], depth_sum[B])."""
    B, A = tree.B, tree.A
    max_depth = S if not max_depth or max_depth <= 0 else max_depth
    br = np.arange(B)
    min_margin = np.full(B, np.inf)
    depth_sum = np.zeros(B, np.int64)
    for sim in range(S):
        # simulate: vmapped while_loop == masked loop until every root stops
        node = np.zeros(B, np.int32)
        parent = np.zeros(B, np.int32)
        action = np.zeros(B, np.int32)
        depth = np.zeros(B, np.int32)
        cont = np.ones(B, bool)
        level = 0
        while cont.any():
            rows = br[cont]
            nz = noise_fn(sim, level, rows) if noise_fn else None
            a, m = _select(tree, node[rows], depth, pb_c_init, pb_c_base, nz, rows)
            min_margin[rows] = np.minimum(min_margin[rows], m)
            parent[rows] = node[rows]
            action[rows] = a
            nxt = tree.children_index[rows, node[rows], a]
            depth[rows] += 1
            go = (nxt != -1) & (depth[rows] < max_depth)
            node[rows[go]] = nxt[go]
            cont[rows[~go]] = False
            level += 1
        depth_sum += depth
        nxt = tree.children_index[br, parent, action]
        nxt = np.where(nxt == -1, sim + 1, nxt).astype(np.int32)
        # expand
        r, d, pl, v, ne = recurrent_fn(action, tree.embeddings[br, parent])
        tree.children_prior_logits[br, nxt] = pl
        tree.raw_values[br, nxt] = v
        tree.node_values[br, nxt] = v
        tree.node_visits[br, nxt] += 1
        tree.embeddings[br, nxt] = ne
        tree.children_index[br, parent, action] = nxt
        tree.children_rewards[br, parent, action] = r
        tree.children_discounts[br, parent, action] = d
        tree.parents[br, nxt] = parent
        tree.action_from_parent[br, nxt] = action
        # backward
        leaf_value = tree.node_values[br, nxt].copy()
        idx = nxt.copy()
        while (idx != 0).any():
            rows = br[idx != 0]
            i = idx[rows]
            p = tree.parents[rows, i]
            cnt = tree.node_visits[rows, p]
            a = tree.action_from_parent[rows, i]
            lv = tree.children_rewards[rows, p, a] + tree.children_discounts[rows, p, a] * leaf_value[rows]
            leaf_value[rows] = lv
            tree.node_values[rows, p] = (tree.node_values[rows, p] * cnt + lv) / (cnt + F32(1.0))
            tree.node_visits[rows, p] = cnt + 1
            tree.children_values[rows, p, a] = tree.node_values[rows, i]
            tree.children_visits[rows, p, a] += 1
            idx[rows] = p
    return min_margin, depth_sum


def summary_sample(tree, temperature, gumbel):
    vc = tree.children_visits[:, 0].astype(F32)
    total = vc.sum(axis=-1, keepdims=True)
    probs = vc / np.maximum(total, 1)
    probs = np.where(total > 0, probs, F32(1 / tree.A)).astype(F32)
    logits = np.log(np.maximum(probs, TINY))
    logits = logits - logits.max(axis=-1, keepdims=True)
    with np.errstate(over="ignore", invalid="ignore"):
        logits = logits / np.maximum(TINY, F32(temperature))
    return (logits + gumbel).argmax(axis=-1).astype(np.int32), probs


def act_mlp(w, obs, S, A, E, support_size=10, discount=0.99, dirichlet_noise=None,
            dirichlet_fraction=0.25, invalid_actions=None, temperature=1.0, gumbel=None,
            max_depth=None, pb_c_init=1.25, pb_c_base=19652.0, pred_on="child", noise_fn=None):
    obs = np.asarray(obs, F32)
    B = obs.shape[0]
    pl, v0, emb = root_inference(w, obs, support_size)
    tree = Tree(B, S + 1, A, E)
    tree.children_prior_logits[:, 0] = root_prior(pl, dirichlet_noise, dirichlet_fraction,
                                                  invalid_actions)
    tree.raw_values[:, 0] = v0
    tree.node_values[:, 0] = v0
    tree.node_visits[:, 0] = 1
    tree.embeddings[:, 0] = emb
    if invalid_actions is not None:
        tree.root_invalid_actions[:] = invalid_actions

    def rec(action, e):
        return recurrent_inference(w, action, e, support_size, discount, A, pred_on)

    margin, dsum = search(tree, rec, S, max_depth, pb_c_init, pb_c_base, noise_fn)
    g = np.zeros((B, A), F32) if gumbel is None else gumbel
    action, weights = summary_sample(tree, temperature, g)
    return {"action": action, "action_weights": weights, "root_value": v0, "tree": tree,
            "min_margin": margin, "depth_sum": dsum}


# ---- Gumbel MuZero (mctx gumbel_muzero_policy), independent restatement ---------------------------

def considered_visits(m, num_simulations):
    """seq_halving.get_sequence_of_considered_visits"""
    import math
    if m <= 1:
        return list(range(num_simulations))
    log2max = int(math.ceil(math.log2(m)))
    seq, visits, nc = [], [0] * m, m
    while len(seq) < num_simulations:
        extra = max(1, int(num_simulations / (log2max * nc)))
        for _ in range(extra):
            seq.extend(visits[:nc])
            for i in range(nc):
                visits[i] += 1
        nc = max(2, nc // 2)
    return seq[:num_simulations]


def qtransform_completed_by_mix_value(tree, rows, node, value_scale=0.1, maxvisit_init=50.0, eps=1e-8):
    vc = tree.children_visits[rows, node]
    q = tree.children_rewards[rows, node] + tree.children_discounts[rows, node] * tree.children_values[rows, node]
    raw = tree.raw_values[rows, node]
    prior = np.maximum(TINY, softmax(tree.children_prior_logits[rows, node]))
    sum_visits = vc.sum(-1)
    sum_probs = np.where(vc > 0, prior, 0).sum(-1, keepdims=True)
    weighted_q = np.where(vc > 0, prior * q / np.where(vc > 0, sum_probs, 1.0), 0.0).sum(-1)
    value = (raw + sum_visits * weighted_q) / (sum_visits + 1)
    cq = np.where(vc > 0, q, value[:, None])
    lo, hi = cq.min(-1, keepdims=True), cq.max(-1, keepdims=True)
    cq = (cq - lo) / np.maximum(hi - lo, F32(eps))
    return ((F32(maxvisit_init) + vc.max(-1))[:, None] * F32(value_scale) * cq).astype(F32)


def qtransform_by_parent_and_siblings(tree, rows, node, eps=1e-8):
    vc = tree.children_visits[rows, node]
    q = tree.children_rewards[rows, node] + tree.children_discounts[rows, node] * tree.children_values[rows, node]
    nval = tree.node_values[rows, node][:, None]
    safe = np.where(vc > 0, q, nval)
    lo = np.minimum(nval, safe.min(-1, keepdims=True))
    hi = np.maximum(nval, safe.max(-1, keepdims=True))
    return ((np.where(vc > 0, q, lo) - lo) / np.maximum(hi - lo, F32(eps))).astype(F32)


def _score_considered(considered_visit, gumbel, logits, qv, vc):
    logits = logits - logits.max(-1, keepdims=True)
    penalty = np.where(vc == considered_visit[:, None], 0, -np.inf)
    return np.maximum(-1e9, gumbel + logits + qv) + penalty


def gumbel_search(tree, recurrent_fn, S, root_gumbel, max_considered=16, qtransform="mix", max_depth=None):
    """mctx search with gumbel_muzero_{root,interior}_action_selection. Returns depth_sum."""
    B, A = tree.B, tree.A
    qt = qtransform_completed_by_mix_value if qtransform == "mix" else qtransform_by_parent_and_siblings
    max_depth = S if not max_depth or max_depth <= 0 else max_depth
    br = np.arange(B)
    table = [considered_visits(m, S) for m in range(max_considered + 1)]
    num_valid = (1 - tree.root_invalid_actions.astype(np.int32)).sum(-1)
    num_considered = np.minimum(max_considered, num_valid)
    depth_sum = np.zeros(B, np.int64)
    for sim in range(S):
        node = np.zeros(B, np.int32)
        parent = np.zeros(B, np.int32)
        action = np.zeros(B, np.int32)
        depth = np.zeros(B, np.int32)
        cont = np.ones(B, bool)
        level = 0
        while cont.any():
            rows = br[cont]
            n = node[rows]
            vc = tree.children_visits[rows, n]
            qv = qt(tree, rows, n)
            logits = tree.children_prior_logits[rows, n]
            if level == 0:
                cv = np.array([table[num_considered[b]][vc[i].sum()] for i, b in enumerate(rows)])
                score = _score_considered(cv, root_gumbel[rows], logits, qv, vc)
                score = np.where(tree.root_invalid_actions[rows].astype(bool), -np.inf, score)
            else:
                probs = softmax(logits + qv)
                score = probs - vc / (1 + vc.sum(-1, keepdims=True))
            a = score.argmax(-1).astype(np.int32)
            parent[rows] = n
            action[rows] = a
            nxt = tree.children_index[rows, n, a]
            depth[rows] += 1
            go = (nxt != -1) & (depth[rows] < max_depth)
            node[rows[go]] = nxt[go]
            cont[rows[~go]] = False
            level += 1
        depth_sum += depth
        nxt = tree.children_index[br, parent, action]
        nxt = np.where(nxt == -1, sim + 1, nxt).astype(np.int32)
        r, d, pl, v, ne = recurrent_fn(action, tree.embeddings[br, parent])
        tree.children_prior_logits[br, nxt] = pl
        tree.raw_values[br, nxt] = v
        tree.node_values[br, nxt] = v
        tree.node_visits[br, nxt] += 1
        tree.embeddings[br, nxt] = ne
        tree.children_index[br, parent, action] = nxt
        tree.children_rewards[br, parent, action] = r
        tree.children_discounts[br, parent, action] = d
        tree.parents[br, nxt] = parent
        tree.action_from_parent[br, nxt] = action
        leaf_value = tree.node_values[br, nxt].copy()
        idx = nxt.copy()
        while (idx != 0).any():
            rows = br[idx != 0]
            i = idx[rows]
            p = tree.parents[rows, i]
            cnt = tree.node_visits[rows, p]
            a = tree.action_from_parent[rows, i]
            lv = tree.children_rewards[rows, p, a] + tree.children_discounts[rows, p, a] * leaf_value[rows]
            leaf_value[rows] = lv
            tree.node_values[rows, p] = (tree.node_values[rows, p] * cnt + lv) / (cnt + F32(1.0))
            tree.node_visits[rows, p] = cnt + 1
            tree.children_values[rows, p, a] = tree.node_values[rows, i]
            tree.children_visits[rows, p, a] += 1
            idx[rows] = p
    return depth_sum


def gumbel_finish(tree, root_gumbel, qtransform="mix"):
    qt = qtransform_completed_by_mix_value if qtransform == "mix" else qtransform_by_parent_and_siblings
    B = tree.B
    br = np.arange(B)
    root = np.zeros(B, np.int32)
    vc = tree.children_visits[:, 0]
    qv = qt(tree, br, root)
    logits = tree.children_prior_logits[:, 0]
    score = _score_considered(vc.max(-1), root_gumbel, logits, qv, vc)
    inv = tree.root_invalid_actions.astype(bool)
    action = np.where(inv, -np.inf, score).argmax(-1).astype(np.int32)
    x = logits + qv
    masked = np.where(inv, FMIN, x - x.max(-1, keepdims=True))
    x = np.where(inv.any(-1, keepdims=True), masked, x)
    return action, softmax(x)
