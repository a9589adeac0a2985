"""Known-answer and cross-restatement tests of the Gumbel MuZero oracle (mctx gumbel_muzero_policy,
reached from muax/policy.py:33-47).  Parity with jax+mctx itself is unpinned (see mz_oracle.h)."""
import numpy as np
import pytest

from oracle import mz_numpy as mn

F32 = np.float32


def test_sequence_of_considered_visits(oracle):
    # hand derivation for m=4, n=8: log2max=2; 4 considered once -> [0,0,0,0]; then 2 considered,
    # int(8/(2*2))=2 extra rounds -> [1,1],[2,2]
    assert oracle.considered_visits(4, 8).tolist() == [0, 0, 0, 0, 1, 1, 2, 2]
    assert oracle.considered_visits(1, 5).tolist() == [0, 1, 2, 3, 4]
    assert oracle.considered_visits(0, 3).tolist() == [0, 1, 2]
    assert oracle.considered_visits(2, 6).tolist() == [0, 0, 1, 1, 2, 2]
    for m in range(0, 20):
        for n in (1, 7, 50, 200):
            assert oracle.considered_visits(m, n).tolist() == mn.considered_visits(m, n)
    # budget accounting: with m considered actions, the k-th visit of a slot never precedes its (k-1)-th
    seq = oracle.considered_visits(16, 50)
    assert seq[:16].tolist() == [0] * 16 and (np.diff(seq) >= -max(seq)).all()


def _const_rec(A, E, value, reward, logits):
    def rec(action, emb):
        B = len(action)
        return (np.full(B, reward, F32), np.full(B, 0.99, F32), np.tile(np.asarray(logits, F32), (B, 1)),
                np.full(B, value, F32), np.zeros((B, E), F32))
    return rec


def _run_both(oracle, B, A, E, S, root_logits, root_value, rec, gumbel, kind, max_considered=16, invalid=None,
              max_depth=0):
    pl = oracle.mask_root_logits(np.asarray(root_logits, F32).reshape(-1, A) * np.ones((B, 1), F32), invalid)
    tree = oracle.Tree(B, S + 1, A, E)
    oracle.tree_init(tree, pl, np.full(B, root_value, F32), np.zeros((B, E), F32), invalid)
    cfg = oracle.SearchCfg(S, max_depth=max_depth)
    for s in range(S):
        p, a, d = oracle.gumbel_step_select(tree, cfg, gumbel, kind, max_considered)
        oracle.step_expand_backup(tree, s, p, a, *rec(a, tree.embeddings[np.arange(B), p]))
    action, weights = oracle.gumbel_finish(tree, gumbel, kind)
    nt = mn.Tree(B, S + 1, A, E)
    nt.children_prior_logits[:, 0] = pl
    nt.raw_values[:, 0] = nt.node_values[:, 0] = root_value
    nt.node_visits[:, 0] = 1
    if invalid is not None:
        nt.root_invalid_actions[:] = invalid
    mn.gumbel_search(nt, rec, S, gumbel, max_considered, "mix" if kind == 1 else "pas", max_depth or None)
    an, wn = mn.gumbel_finish(nt, gumbel, "mix" if kind == 1 else "pas")
    return tree, action, weights, nt, an, wn


@pytest.mark.parametrize("kind", [0, 1])
def test_gumbel_c_and_numpy_restatements_agree(oracle, kind):
    B, obs_dim, E, A, S = 48, 4, 8, 4, 24
    w = oracle.random_mlp_weights(13, obs_dim, E, A, 21, bias_scale=0.2)
    obs = np.random.default_rng(3).uniform(-1, 1, (B, obs_dim)).astype(F32)
    mlp = oracle.Mlp(w, obs_dim, E, A, 21)
    pl, v, emb = oracle.root_inference(mlp, obs)
    gumbel = np.random.default_rng(4).gumbel(size=(B, A)).astype(F32)
    invalid = np.zeros((B, A), np.uint8)
    invalid[::5, 1] = 1

    def rec(action, e):
        return oracle.recurrent_inference(mlp, action, e)

    def rec_np(action, e):
        return mn.recurrent_inference(w, action, e, 10, 0.99, A)

    mpl = oracle.mask_root_logits(pl, invalid)
    tree = oracle.Tree(B, S + 1, A, E)
    oracle.tree_init(tree, mpl, v, emb, invalid)
    cfg = oracle.SearchCfg(S)
    for s in range(S):
        p, a, d = oracle.gumbel_step_select(tree, cfg, gumbel, kind, 3)
        oracle.step_expand_backup(tree, s, p, a, *rec(a, tree.embeddings[np.arange(B), p]))
    action, weights = oracle.gumbel_finish(tree, gumbel, kind)
    nt = mn.Tree(B, S + 1, A, E)
    nt.children_prior_logits[:, 0] = mpl
    nt.raw_values[:, 0] = nt.node_values[:, 0] = v
    nt.node_visits[:, 0] = 1
    nt.embeddings[:, 0] = emb
    nt.root_invalid_actions[:] = invalid
    mn.gumbel_search(nt, rec_np, S, gumbel, 3, "mix" if kind == 1 else "pas")
    an, wn = mn.gumbel_finish(nt, gumbel, "mix" if kind == 1 else "pas")
    same = (tree.children_index == nt.children_index).reshape(B, -1).all(1)
    assert same.mean() > 0.8  # float rounding may flip a near-tie; everything else is identical
    assert (tree.children_visits[same] == nt.children_visits[same]).all()
    assert (action[same] == an[same]).all()
    assert np.allclose(weights[same], wn[same], rtol=1e-3, atol=1e-4)
    assert (weights[::5, 1] == 0).all() and (action[::5] != 1).all()
    # every simulation passes through the root: visits add up; at most 3 root actions are considered
    assert (tree.children_visits[:, 0].sum(1) == S).all()
    assert ((tree.children_visits[:, 0] > 0).sum(1) <= 3).all()


def test_gumbel_sequential_halving_visit_pattern(oracle):
    """Equal Q everywhere: the root budget follows the halving table exactly. m=4, S=8: the top-4 by
    gumbel+logits get one visit each, the top-2 get the remaining four."""
    B, A, E, S = 1, 6, 1, 8
    gumbel = np.array([[0.5, 3.0, 2.0, -1.0, 1.0, 0.0]], F32)
    rec = _const_rec(A, E, value=0.0, reward=0.0, logits=[0.0] * A)
    tree, action, weights, nt, an, wn = _run_both(oracle, B, A, E, S, [0.0] * A, 0.0, rec, gumbel, 1, max_considered=4)
    assert tree.children_visits[0, 0].tolist() == [1, 3, 3, 0, 1, 0]
    assert action[0] == 1 and an[0] == 1   # best gumbel + logits + q among the most visited
    assert np.array_equal(tree.children_visits, nt.children_visits)
    assert abs(weights.sum() - 1) < 1e-6 and np.allclose(weights, wn, atol=1e-6)


def test_gumbel_interior_selection_is_deterministic_and_spreads_visits(oracle):
    """Interior nodes: argmax(softmax(logits + q) - visits/(1+sum)); with uniform logits and equal Q the
    children of a node are visited round-robin in action order."""
    B, A, E, S = 1, 3, 1, 10
    gumbel = np.array([[5.0, 0.0, 0.0]], F32)  # the root keeps choosing action 0 while allowed
    rec = _const_rec(A, E, value=0.0, reward=0.0, logits=[0.0] * A)
    tree, *_ = _run_both(oracle, B, A, E, S, [0.0] * A, 0.0, rec, gumbel, 1, max_considered=1)
    assert tree.children_visits[0, 0].tolist() == [S, 0, 0]
    child = tree.children_index[0, 0, 0]
    cv = tree.children_visits[0, child]
    assert cv.sum() == S - 1 and cv.max() - cv.min() <= 1


def test_mix_value_qtransform_formula(oracle):
    """qtransform_completed_by_mix_value on a hand-made node."""
    t = oracle.Tree(1, 3, 3, 1)
    t.children_prior_logits[0, 0] = np.log(np.array([0.5, 0.3, 0.2], F32))
    t.children_visits[0, 0] = [2, 0, 1]
    t.children_rewards[0, 0] = [0.1, 0.0, 0.4]
    t.children_discounts[0, 0] = [0.9, 0.0, 0.9]
    t.children_values[0, 0] = [1.0, 0.0, -1.0]
    t.raw_values[0, 0] = 0.3
    q = np.array([0.1 + 0.9, 0.0, 0.4 - 0.9])
    p = np.array([0.5, 0.3, 0.2])
    wq = (p[0] * q[0] + p[2] * q[2]) / (p[0] + p[2])
    value = (0.3 + 3 * wq) / 4
    cq = np.array([q[0], value, q[2]])
    cq = (cq - cq.min()) / (cq.max() - cq.min())
    expect = (50 + 2) * 0.1 * cq
    got = oracle.qtransform(t, [0], 1)[0]
    assert np.allclose(got, expect, rtol=1e-5, atol=1e-6)
    assert np.allclose(mn.qtransform_completed_by_mix_value(mn_tree(t), np.array([0]), np.array([0]))[0], expect,
                       rtol=1e-5, atol=1e-6)


def mn_tree(t):
    n = mn.Tree(t.B, t.N, t.A, t.E)
    for k, v in t.arrays().items():
        setattr(n, k, v.copy())
    return n
