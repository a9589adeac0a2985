"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Bar: bit-exact on every integer array of the tree (indices, visit counts, parents, actions) and --
because both sides implement the same MZ-F32 arithmetic spec -- equal floats as well (checked with
==, and again with the 1e-5 tolerance the north star states).
"""
import numpy as np
import pytest
import torch

from helpers import assert_trees_equal, make_case

pytestmark = pytest.mark.gpu

F32 = np.float32


def _fused(case, tiebreak, key, max_depth=None, temperature=1.0, use_gumbel=True, use_noise=True,
           pred_on="child", global_batch=None, root_offset=0, rows=None, discount=0.99):
    from muax_amd import MuZeroSearch, SearchConfig
    sl = slice(None) if rows is None else rows
    B = case["obs"][sl].shape[0]
    cfg = SearchConfig(case["A"], case["S"], case["E"], max_depth=max_depth, tiebreak=tiebreak,
                       global_batch=global_batch, root_offset=root_offset)
    s = MuZeroSearch(B, cfg)
    s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"],
                      case["support"], discount, pred_on)
    out = s.act_mlp(torch.from_numpy(case["obs"][sl]), key,
                    dirichlet_noise=torch.from_numpy(case["noise"][sl]) if use_noise else None,
                    invalid_actions=None if case["invalid"] is None else torch.from_numpy(case["invalid"][sl]),
                    temperature=temperature,
                    gumbel=torch.from_numpy(case["gumbel"][sl]) if use_gumbel else None, with_tree=True)
    torch.cuda.synchronize()
    return s, out


def _oracle(oracle, case, tiebreak, key, max_depth=0, temperature=1.0, use_gumbel=True, use_noise=True,
            pred_on=0, discount=0.99):
    mlp = oracle.Mlp(case["w"], case["obs_dim"], case["E"], case["A"], case["F"], support_size=case["support"],
                     recurrent_pred_on=pred_on, discount=discount)
    cfg = oracle.SearchCfg(case["S"], max_depth=max_depth or 0, tiebreak=int(tiebreak))
    return oracle.act_mlp(mlp, cfg, case["obs"], key, case["noise"] if use_noise else None, 0.25, case["invalid"],
                          temperature, case["gumbel"] if use_gumbel else None)


def _compare(ref, s, out):
    assert np.array_equal(ref["action"], out.action.cpu().numpy())
    assert np.array_equal(ref["action_weights"], out.action_weights.cpu().numpy())
    assert np.array_equal(ref["root_value"], s.root_value.cpu().numpy())
    assert np.array_equal(ref["depth_sum"], s.depth_sum.cpu().numpy().astype(np.int64))
    assert np.array_equal(ref["tree"].node_values[:, 0], s.search_value.cpu().numpy())
    assert_trees_equal(ref["tree"], out.search_tree, exact_floats=True)
    assert_trees_equal(ref["tree"], out.search_tree, exact_floats=False)


@pytest.mark.parametrize("tiebreak", [False, True])
@pytest.mark.parametrize("A,E,S,B", [(2, 8, 50, 333), (3, 8, 32, 77), (4, 8, 50, 130), (4, 32, 50, 100), (2, 8, 63, 150), (3, 8, 50, 90), (2, 16, 50, 70), (4, 16, 40, 50), (2, 8, 100, 60), (2, 8, 127, 30), (4, 32, 100, 24), (2, 10, 50, 100), (4, 10, 50, 60), (6, 8, 50, 45), (8, 8, 50, 40), (2, 32, 50, 50), (4, 64, 50, 40), (2, 64, 30, 20)])
def test_fused_matches_oracle(oracle, A, E, S, B, tiebreak):
    case = make_case(oracle, 10 * A + E, B, 4 if E == 8 else 8, E, A, S)
    key = [123, 456 + A]
    s, out = _fused(case, tiebreak, key)
    _compare(_oracle(oracle, case, tiebreak, key), s, out)


def test_fused_invalid_actions_and_key_gumbel(oracle):
    case = make_case(oracle, 5, 200, 4, 8, 4, 40, invalid_frac=0.3)
    key = [9, 8]
    s, out = _fused(case, True, key, use_gumbel=False)
    ref = _oracle(oracle, case, True, key, use_gumbel=False)
    _compare(ref, s, out)
    a = out.action.cpu().numpy()
    inv = case["invalid"]
    ok = inv.sum(1) < inv.shape[1]
    assert (inv[np.arange(len(a)), a][ok] == 0).all()  # never samples an invalid action


@pytest.mark.parametrize("max_depth", [1, 3])
def test_fused_max_depth_reexpansion(oracle, max_depth):
    case = make_case(oracle, 6, 64, 4, 8, 2, 30)
    s, out = _fused(case, True, [1, 2], max_depth=max_depth)
    ref = _oracle(oracle, case, True, [1, 2], max_depth=max_depth)
    _compare(ref, s, out)
    assert (out.search_tree.node_visits.cpu().numpy()[:, 1:].max(axis=1) > 1).any() or max_depth > 1


def test_fused_temperature_zero_no_noise_parent_quirk(oracle):
    case = make_case(oracle, 7, 96, 4, 8, 2, 25)
    s, out = _fused(case, False, 3, temperature=0.0, use_noise=False, pred_on="parent")
    ref = _oracle(oracle, case, False, [0, 3], temperature=0.0, use_noise=False, pred_on=1)
    _compare(ref, s, out)
    vc = out.search_tree.children_visits.cpu().numpy()[:, 0]
    a = out.action.cpu().numpy()
    assert (vc[np.arange(len(a)), a] == vc.max(axis=1)).all()


@pytest.mark.parametrize("A,E,S", [(2, 8, 50), (4, 8, 30), (3, 8, 24)])
def test_fused_all_ties_noise_decides(oracle, A, E, S):
    """Degenerate nets (all-zero weights): uniform priors, zero values and rewards, so EVERY selection
    is an exact tie and mctx's 1e-7*uniform tie-break noise decides it.  Exercises the kernel's
    near-tie path (on-demand threefry key walk) at every level; the result must still be bit-exact,
    and it must differ from the no-noise search."""
    case = make_case(oracle, 77, 48, 4, E, A, S)
    case["w"] = {k: np.zeros_like(v) for k, v in case["w"].items()}
    key = [11, 22]
    s, out = _fused(case, True, key, use_noise=False)
    ref = _oracle(oracle, case, True, key, use_noise=False)
    _compare(ref, s, out)
    s0, out0 = _fused(case, False, key, use_noise=False)
    _compare(_oracle(oracle, case, False, key, use_noise=False), s0, out0)
    assert not torch.equal(out.search_tree.children_index, out0.search_tree.children_index)


def test_fused_small_margins(oracle):
    """Tiny weights: score margins straddle the 1e-7 noise scale, mixing cached and noisy decisions."""
    case = make_case(oracle, 78, 64, 4, 8, 2, 40)
    case["w"] = {k: (v * 1e-4).astype(np.float32) for k, v in case["w"].items()}
    s, out = _fused(case, True, [5, 5], use_noise=False)
    _compare(_oracle(oracle, case, True, [5, 5], use_noise=False), s, out)


@pytest.mark.parametrize("A,E", [(2, 8), (4, 32), (3, 8)])
def test_fused_rare_division_paths(oracle, A, E):
    """The two shared-reciprocal divisions of the fused kernel hand over to the IEEE division behind a wave-uniform
    range test; random networks never trip it, these do.  (a) Second-layer weights x 60: support logits spread over
    hundreds, so softmax terms land in (0, 2^-100) and below exp's cut at -87 (exact zeros).  (b) An all-zero reward
    head and a discount of 1e-30: q = discount * value, so the value-score numerators q - min q are ~ 1e-31, inside
    (0, 2^-100).  Both must stay bit-exact (tools: a build with either IEEE branch deliberately wrong fails exactly
    this test and none of the others)."""
    case = make_case(oracle, 900 + A, 96, 4 if E == 8 else 8, E, A, 40)
    w = dict(case["w"])
    for k in ("pv_w2", "dr_w2"):
        w[k] = (w[k] * 60.0).astype(F32)
    sharp = dict(case, w=w)
    s, out = _fused(sharp, True, [9, A])
    ref = _oracle(oracle, sharp, True, [9, A])
    _compare(ref, s, out)
    assert np.abs(ref["tree"].children_rewards).max() > 1.0  # (the decode saw one-hot-like distributions)
    w = dict(case["w"])
    for k in ("dr_w1", "dr_b1", "dr_w2", "dr_b2"):
        w[k] = np.zeros_like(w[k])  # rewards exactly 0 (uniform softmax decodes to 0): q = discount * value
    flat = dict(case, w=w)
    for disc in (1e-30, 3e-33):
        s, out = _fused(flat, False, [A, 9], discount=disc, use_noise=False)
        _compare(_oracle(oracle, flat, False, [A, 9], discount=disc, use_noise=False), s, out)


@pytest.mark.parametrize("A,E,support,S,B", [(2, 8, 8, 30, 40), (2, 8, 15, 50, 40), (2, 8, 16, 50, 70), (2, 8, 31, 20, 33),
                                             (6, 8, 20, 50, 45), (4, 32, 20, 50, 50), (6, 8, 12, 30, 20)])
def test_fused_support_sizes(oracle, A, E, support, S, B):
    """support_size is a constructor argument of the reference (muax/model.py:48-49): F = 2 support + 1 is a
    run-time parameter of the fused kernel, 17..32 logits in two lane slots, 33..63 in four -- e.g. the
    (A = 6, E = 8, support = 20) shape through mzs_act_mlp."""
    case = make_case(oracle, 100 * A + support, B, 4 if E == 8 else 8, E, A, S, support=support)
    key = [support, A]
    s, out = _fused(case, True, key)
    _compare(_oracle(oracle, case, True, key), s, out)
    s, out = _fused(case, False, key, temperature=0.25, use_noise=False)
    _compare(_oracle(oracle, case, False, key, temperature=0.25, use_noise=False), s, out)


def test_fused_sharding_invariance(oracle):
    """A shard (global_batch, root_offset) reproduces the same roots of the full batch bit for bit."""
    case = make_case(oracle, 8, 96, 4, 8, 2, 20)
    s_full, full = _fused(case, True, 77, use_gumbel=False)
    s_half, half = _fused(case, True, 77, use_gumbel=False, global_batch=96, root_offset=48, rows=slice(48, 96))
    assert torch.equal(full.action[48:], half.action)
    for f in full.search_tree._fields:
        assert torch.equal(getattr(full.search_tree, f)[48:], getattr(half.search_tree, f)), f


def test_fused_small_batch_and_plumbing_config(oracle):
    """BASELINE config 1: B=1, S=10 (latency/plumbing point)."""
    case = make_case(oracle, 9, 1, 4, 8, 2, 10)
    s, out = _fused(case, True, [0, 42])
    _compare(_oracle(oracle, case, True, [0, 42]), s, out)


@pytest.mark.parametrize("B,obs_dim,E,A", [(4096, 4, 8, 2), (8192, 8, 32, 4)])
def test_fused_full_size_properties(oracle, B, obs_dim, E, A):
    """BASELINE configs 2 and 3 at full size (4096 CartPole-shaped / 8192 LunarLander-shaped roots, S=50):
    structural invariants that do not need the oracle, plus exact agreement with the oracle on the whole batch."""
    case = make_case(oracle, 0, B, obs_dim, E, A, 50, bias_scale=0.0)
    s, out = _fused(case, True, [0, 0], use_gumbel=False)
    t = out.search_tree
    nv, cv, ci, par, afp = (x.cpu().numpy() for x in
                            (t.node_visits, t.children_visits, t.children_index, t.parents, t.action_from_parent))
    S = case["S"]
    assert (nv[:, 0] == S + 1).all()
    assert (cv.sum(-1) == nv - 1)[nv > 0].all()
    assert (cv[:, 0].sum(-1) == S).all()
    b, n, a = np.nonzero(ci >= 0)
    c = ci[b, n, a]
    assert (par[b, c] == n).all() and (afp[b, c] == a).all() and (cv[b, n, a] == nv[b, c]).all()
    assert (np.sort(c.reshape(B, S), axis=1) == np.arange(1, S + 1)).all()  # every node expanded exactly once
    w = out.action_weights.cpu().numpy()
    assert np.allclose(w.sum(1), 1, atol=1e-6) and np.array_equal(w, (cv[:, 0] / F32(S)).astype(F32))
    # depth_sum equals the sum of node depths recomputed from the parents array
    depth = np.zeros_like(par)
    for k in range(1, S + 1):
        depth[:, k] = depth[np.arange(B), par[:, k]] + 1
    assert np.array_equal(depth.sum(1), s.depth_sum.cpu().numpy())
    # ... and the whole batch against the oracle (OpenMP: a fraction of a second), every tree array
    mlp = oracle.Mlp(case["w"], obs_dim, E, A, 21)
    ref = oracle.act_mlp(mlp, oracle.SearchCfg(S, tiebreak=1), case["obs"], [0, 0], case["noise"], 0.25, None, 1.0,
                         None, nthreads=8)
    assert np.array_equal(ref["action"], out.action.cpu().numpy())
    assert np.array_equal(ref["depth_sum"], s.depth_sum.cpu().numpy().astype(np.int64))
    assert_trees_equal(ref["tree"], out.search_tree, exact_floats=True)


def _torch_recurrent(case):
    """The default MLP trio in plain torch (any plugin net would do): the step-wise path's nets."""
    w = {k: torch.from_numpy(v).cuda() for k, v in case["w"].items()}
    A, sup = case["A"], case["support"]
    bins = torch.arange(-sup, sup + 1, dtype=torch.float32, device="cuda")

    def minmax(s):
        mn, mx = s.min(1, keepdim=True).values, s.max(1, keepdim=True).values
        sc = mx - mn
        sc = torch.where(sc < 1e-5, sc + 1e-5, sc)
        return (s - mn) / sc

    def inv(x):
        e = 1e-3
        return torch.sign(x) * (((torch.sqrt(1 + 4 * e * (x.abs() + 1 + e)) - 1) / (2 * e)) ** 2 - 1)

    def mlp(x, a, b, c, d):
        return torch.nn.functional.elu(x @ a + b) @ c + d

    def pred(s):
        v = mlp(s, w["pv_w1"], w["pv_b1"], w["pv_w2"], w["pv_b2"])
        pl = mlp(s, w["pp_w1"], w["pp_b1"], w["pp_w2"], w["pp_b2"])
        return pl, inv((torch.softmax(v, -1) * bins).sum(-1))

    def root(obs):
        s = minmax(obs @ w["repr_w"] + w["repr_b"])
        pl, v = pred(s)
        return pl, v, s

    def rec(action, emb):
        sa = torch.cat([emb, torch.nn.functional.one_hot(action.long(), A).float()], 1)
        r = inv((torch.softmax(mlp(sa, w["dr_w1"], w["dr_b1"], w["dr_w2"], w["dr_b2"]), -1) * bins).sum(-1))
        ns = minmax(mlp(sa, w["dn_w1"], w["dn_b1"], w["dn_w2"], w["dn_b2"]))
        pl, v = pred(ns)
        return r, torch.full_like(r, 0.99), pl, v, ns

    return root, rec


@pytest.mark.parametrize("tiebreak", [False, True])
@pytest.mark.parametrize("A,E,S,B", [(2, 8, 20, 70), (5, 8, 24, 33), (18, 40, 30, 40), (18, 300, 12, 21)])  # E >= 256: wide rows
@pytest.mark.parametrize("walk,fused_select", [(False, False), (True, False), (False, True), (True, True)])
def test_stepwise_matches_oracle(oracle, A, E, S, B, tiebreak, walk, fused_select, monkeypatch):
    """Plugin-net path: torch nets between mzs_select and mzs_expand_backup; the oracle is fed the very
    same net outputs, so every tree array must agree exactly.  Both sets of tree kernels: cached decisions
    (mz_step_jump.cuh, the default) and the level-by-level walk (mz_step.cuh, MZS_STEP_WALK=1); and both call
    shapes: mzs_select + mzs_expand_backup per simulation, or mzs_expand_backup_select (the next simulation's selection
    as the tail of the expand + backward launch)."""
    if walk:
        monkeypatch.setenv("MZS_STEP_WALK", "1")
    from muax_amd import MuZeroSearch, SearchConfig
    case = make_case(oracle, 20 + A, B, 6, E, A, S, invalid_frac=0.2 if A > 2 else 0.0)
    root, rec = _torch_recurrent(case)
    key = [4, 5]
    obs = torch.from_numpy(case["obs"]).cuda()
    pl, v, emb = root(obs)
    s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=tiebreak))
    inv = None if case["invalid"] is None else torch.from_numpy(case["invalid"])
    s.root(pl, v, emb, key, inv, torch.from_numpy(case["noise"]), 0.25)
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S, tiebreak=int(tiebreak))
    oracle.tree_init(tree, oracle.root_prior(pl.cpu().numpy(), case["noise"], 0.25, case["invalid"]),
                     v.cpu().numpy(), emb.cpu().numpy(), case["invalid"])
    k_sample, _, sims = oracle.sim_keys_from_act_key(key, S)
    nxt = None
    for sim in range(S):
        action, pemb = nxt if fused_select and sim > 0 else s.select(sim)
        p_ref, a_ref, _ = oracle.step_select(tree, cfg, sim, sims[sim])
        assert np.array_equal(a_ref, action.cpu().numpy()), sim
        assert np.array_equal(tree.embeddings[np.arange(B), p_ref], pemb.cpu().numpy())
        outs = rec(action, pemb)
        if fused_select:
            nxt = s.expand_backup_select(sim, *outs)
            assert (nxt is None) == (sim == S - 1)
        else:
            s.expand_backup(sim, *outs)
        oracle.step_expand_backup(tree, sim, p_ref, a_ref, *[o.cpu().numpy() for o in outs])
    out = s.finish(1.0, None, with_tree=True)
    g = oracle.gumbel(k_sample, B * A).reshape(B, A)
    a_ref, w_ref = oracle.summary_sample(tree, 1.0, g)
    assert np.array_equal(a_ref, out.action.cpu().numpy())
    assert np.array_equal(w_ref, out.action_weights.cpu().numpy())
    assert np.array_equal(np.asarray([tree.parents[b, 1:].size for b in range(B)]), np.full(B, S))
    assert_trees_equal(tree, out.search_tree, exact_floats=True)
    assert int(s.depth_sum.sum()) > 0


def test_stepwise_equals_fused_when_fed_oracle_nets(oracle):
    """The two product paths agree with each other when the step-wise nets are the oracle's MLPs."""
    from muax_amd import MuZeroSearch, SearchConfig
    case = make_case(oracle, 31, 50, 4, 8, 2, 16)
    mlp = oracle.Mlp(case["w"], 4, 8, 2, 21)
    _, fused = _fused(case, True, [2, 2], use_gumbel=False)
    s = MuZeroSearch(50, SearchConfig(2, 16, 8, tiebreak=True))
    pl, v, emb = oracle.root_inference(mlp, case["obs"])
    s.root(torch.from_numpy(pl), torch.from_numpy(v), torch.from_numpy(emb), [2, 2], None,
           torch.from_numpy(case["noise"]), 0.25)
    for sim in range(16):
        action, pemb = s.select(sim)
        outs = oracle.recurrent_inference(mlp, action.cpu().numpy(), pemb.cpu().numpy())
        s.expand_backup(sim, *[torch.from_numpy(o) for o in outs])
    out = s.finish(1.0, None, with_tree=True)
    assert torch.equal(out.action, fused.action)
    for f in out.search_tree._fields:
        assert torch.equal(getattr(out.search_tree, f), getattr(fused.search_tree, f)), f


def test_error_behaviour():
    from muax_amd import MuZeroSearch, SearchConfig
    with pytest.raises(ValueError):
        MuZeroSearch(0, SearchConfig(2, 5, 8))
    s = MuZeroSearch(4, SearchConfig(2, 5, 8))
    with pytest.raises(ValueError):
        s.act_mlp(torch.zeros(4, 4), 0)  # weights not set
    with pytest.raises(ValueError):
        s.select(0)  # root() not called
    s7 = MuZeroSearch(4, SearchConfig(7, 5, 8))
    w = {k: torch.zeros(1) for k in ["repr_w"]}
    with pytest.raises((ValueError, KeyError)):
        s7.set_mlp_weights(w, 4)


# ---------------------------------------------------------------- edge cases of the domain

def _stepwise_vs_oracle(oracle, case, S, tiebreak, max_depth=None, key=(4, 5), fused_select=False):
    """Drive the step-wise kernels and the oracle with the same (torch) net outputs; compare everything.
    fused_select: mzs_expand_backup_select (the next selection in the expand + backward launch) instead of two calls."""
    from muax_amd import MuZeroSearch, SearchConfig
    B, A, E = case["B"], case["A"], case["E"]
    root, rec = _torch_recurrent(case)
    obs = torch.from_numpy(case["obs"]).cuda()
    pl, v, emb = root(obs)
    s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=tiebreak, max_depth=max_depth))
    inv = None if case["invalid"] is None else torch.from_numpy(case["invalid"])
    s.root(pl, v, emb, list(key), inv, torch.from_numpy(case["noise"]), 0.25)
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S, tiebreak=int(tiebreak), max_depth=max_depth or 0)
    oracle.tree_init(tree, oracle.root_prior(pl.cpu().numpy(), case["noise"], 0.25, case["invalid"]),
                     v.cpu().numpy(), emb.cpu().numpy(), case["invalid"])
    k_sample, _, sims = oracle.sim_keys_from_act_key(list(key), S)
    nxt = None
    for sim in range(S):
        action, pemb = nxt if fused_select and sim > 0 else s.select(sim)
        p_ref, a_ref, _ = oracle.step_select(tree, cfg, sim, sims[sim])
        assert np.array_equal(a_ref, action.cpu().numpy()), sim
        outs = rec(action, pemb)
        if fused_select:
            nxt = s.expand_backup_select(sim, *outs)
        else:
            s.expand_backup(sim, *outs)
        oracle.step_expand_backup(tree, sim, p_ref, a_ref, *[o.cpu().numpy() for o in outs])
    out = s.finish(1.0, None, with_tree=True)
    g = oracle.gumbel(k_sample, B * A).reshape(B, A)
    a_ref, w_ref = oracle.summary_sample(tree, 1.0, g)
    assert np.array_equal(a_ref, out.action.cpu().numpy())
    assert np.array_equal(w_ref, out.action_weights.cpu().numpy())
    assert_trees_equal(tree, out.search_tree, exact_floats=True)
    return tree


def test_single_action_and_single_simulation(oracle):
    """A = 1 (every walk is a chain, visit probabilities are [1]) and S = 1."""
    case = make_case(oracle, 41, 20, 4, 8, 1, 12)
    tree = _stepwise_vs_oracle(oracle, case, 12, True)
    assert (tree.children_visits[:, 0, 0] == 12).all()
    depth = np.zeros((20, 13), int)
    for k in range(1, 13):
        depth[:, k] = depth[np.arange(20), tree.parents[:, k]] + 1
    assert (depth[:, 12] == 12).all()  # a single chain of depth S
    case1 = make_case(oracle, 42, 30, 4, 8, 2, 1)
    s, out = _fused(case1, True, [7, 7])
    _compare(_oracle(oracle, case1, True, [7, 7]), s, out)


def test_deep_chains_beyond_two_backup_chunks(oracle):
    """Trees deeper than 32 levels: three 16-entry chunks in the backup phase, jump words across chunks."""
    case = make_case(oracle, 43, 24, 4, 8, 2, 50)
    # make one action overwhelmingly likely so the search digs a single deep line
    case["w"]["pp_b2"] = np.array([6.0, -6.0], np.float32)
    key = [3, 1]
    s, out = _fused(case, True, key)
    ref = _oracle(oracle, case, True, key)
    _compare(ref, s, out)
    par = ref["tree"].parents
    depth = np.zeros_like(par)
    for k in range(1, 51):
        depth[:, k] = depth[np.arange(24), par[:, k]] + 1
    assert depth.max() > 32


def test_stepwise_many_simulations_and_wide_actions(oracle):
    """S above the fused path's 50-node trees, A = 64 (the step-wise maximum), ragged batch."""
    case = make_case(oracle, 44, 19, 6, 8, 64, 80, invalid_frac=0.5)
    _stepwise_vs_oracle(oracle, case, 80, True)
    case2 = make_case(oracle, 45, 5, 6, 8, 3, 300)
    _stepwise_vs_oracle(oracle, case2, 300, False, max_depth=7)


@pytest.mark.gpu
def test_stepwise_single_deep_line_beyond_64_levels(oracle):
    """A search that digs one line 100 levels deep on the step-wise kernels: more levels than the 64 rows a
    1024-thread expand-backup workgroup has in flight (two rounds of decisions, one long JUMP chain)."""
    case = make_case(oracle, 46, 7, 4, 8, 2, 120)
    case["w"]["pp_b2"] = np.array([7.0, -7.0], np.float32)
    _stepwise_vs_oracle(oracle, case, 120, True)
    _stepwise_vs_oracle(oracle, case, 120, False)


def _fused_any(case, tiebreak, key, route, policy="muzero", **kw):
    """_fused for a shape without a listed instance: `route` = "jit" (build one on demand) or "generic"."""
    from muax_amd import MuZeroSearch, SearchConfig, _jit
    cfg = SearchConfig(case["A"], case["S"], case["E"], tiebreak=tiebreak, policy=policy, **kw)
    s = MuZeroSearch(case["B"], cfg)
    s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], case["support"], 0.99,
                      kw.get("pred_on", "child"))
    args = dict(invalid_actions=None if case["invalid"] is None else torch.from_numpy(case["invalid"]), with_tree=True,
                gumbel=torch.from_numpy(case["gumbel"]))
    if policy == "muzero":
        args["dirichlet_noise"] = torch.from_numpy(case["noise"])
    with pytest.raises(ValueError, match="no fused kernel instance"):
        s.act_mlp(torch.from_numpy(case["obs"]), key, **args)
    if route == "jit":
        assert _jit.ensure_instance(case["A"], case["E"], case["F"], case["S"])
    else:
        assert _jit.plan(case["A"], case["E"], case["F"], case["S"]) is None  # no instance can exist for it
        s.allow_generic()
    out = s.act_mlp(torch.from_numpy(case["obs"]), key, **args)
    torch.cuda.synchronize()
    return s, out


@pytest.mark.parametrize("A,E,S,B,support", [(5, 12, 30, 90, 10), (2, 8, 50, 70, 20), (7, 24, 80, 33, 10), (2, 24, 20, 40, 10),
                                             (4, 24, 52, 59, 10), (3, 40, 60, 20, 12), (1, 56, 25, 17, 10),
                                             # round 5: nine to sixteen actions (all of a node's scores in one lane) ...
                                             (9, 8, 50, 70, 10), (12, 16, 40, 37, 10), (16, 8, 50, 45, 10), (16, 32, 30, 21, 20),
                                             # ... and 128 to 255 simulations (FusedCfg::LONG: the nodes' root paths in HBM,
                                             # up to eight path words per lane), also where HBM paths only buy roots per CU
                                             (2, 8, 160, 40, 10), (2, 8, 255, 23, 10), (4, 32, 200, 18, 10), (6, 8, 100, 26, 10),
                                             (10, 8, 100, 19, 10), (3, 8, 128, 33, 12),
                                             # embeddings above 16 that are no multiple of 8 (packed-fma first layers)
                                             (4, 20, 40, 31, 10), (2, 50, 30, 17, 10), (3, 17, 50, 29, 12), (5, 33, 100, 12, 10)])
def test_fused_instance_built_on_demand_matches_oracle(oracle, A, E, S, B, support):
    """Shapes mz_instances.def does not list (5 actions x 12-wide embedding; support_size 20 with F = 41 at CartPole
    widths is listed, 7 actions x 24 at 80 simulations is not ...): mzs_act_mlp refuses, muax_amd/_jit.py compiles ONE
    translation unit for the shape with the hipcc of this box, registers it, and the same call then runs as one launch --
    every tree array equal to the oracle's.  E = 24 is the shape that exposed a DPP read hazard inside the first-layer
    asm blocks (the second slot register read right behind the select that masks its lanes: mz_spec.cuh)."""
    from muax_amd import _jit
    if _jit.plan(A, E, 2 * support + 1, S) is None:
        pytest.skip("outside the fused kernel's limits")
    case = make_case(oracle, 300 + A + E, B, 6, E, A, S, support=support, invalid_frac=0.2 if A > 2 else 0.0)
    key = [31, A]
    try:
        s, out = _fused_any(case, True, key, "jit")
    except BaseException as e:  # (a listed shape does not raise "no fused kernel instance": nothing to build)
        if "DID NOT RAISE" in str(e):
            pytest.skip("the library already has an instance for this shape")
        raise
    _compare(_oracle(oracle, case, True, key), s, out)
    if S >= 100:  # the same long search cut at max_depth (re-expansions: the cut reads the path array in HBM)
        from muax_amd import MuZeroSearch, SearchConfig
        s.close()
        s2 = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True, max_depth=9))
        s2.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], support, 0.99)
        out2 = s2.act_mlp(torch.from_numpy(case["obs"]), key, dirichlet_noise=torch.from_numpy(case["noise"]), with_tree=True,
                          invalid_actions=None if case["invalid"] is None else torch.from_numpy(case["invalid"]),
                          gumbel=torch.from_numpy(case["gumbel"]))
        torch.cuda.synchronize()
        _compare(_oracle(oracle, case, True, key, max_depth=9), s2, out2)


@pytest.mark.parametrize("A,E,S,B", [(2, 8, 63, 77), (2, 8, 127, 40), (2, 8, 200, 21), (4, 32, 100, 19), (12, 8, 50, 33), (2, 8, 50, 64)])
def test_fused_without_tree_export_long_instances(oracle, A, E, S, B):
    """act() WITHOUT a tree export on the instances that keep their embeddings (and root paths) in HBM: the search then
    runs on the handle's own scratch arrays (allocated when the dispatcher asks for them: kNeedEmbScratch /
    kNeedPathScratch), not on the caller's export buffers -- actions, weights, values and depth sums equal to the
    oracle's, twice in a row on the same handle (the scratch is reused).  (2, 8, 50) is the plain record, for contrast."""
    from muax_amd import MuZeroSearch, SearchConfig, _jit
    case = make_case(oracle, 640 + A + S, B, 6, E, A, S, invalid_frac=0.2 if A > 2 else 0.0)
    _jit.ensure_instance(A, E, case["F"], S)
    s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
    s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], 10, 0.99)
    inv = None if case["invalid"] is None else torch.from_numpy(case["invalid"])
    for key in ([5, S], [6, A]):
        out = s.act_mlp(torch.from_numpy(case["obs"]), key, dirichlet_noise=torch.from_numpy(case["noise"]), invalid_actions=inv,
                        gumbel=torch.from_numpy(case["gumbel"]))
        torch.cuda.synchronize()
        ref = _oracle(oracle, case, True, key)
        assert out.search_tree is None
        assert np.array_equal(ref["action"], out.action.cpu().numpy())
        assert np.array_equal(ref["action_weights"], out.action_weights.cpu().numpy())
        assert np.array_equal(ref["root_value"], s.root_value.cpu().numpy())
        assert np.array_equal(ref["depth_sum"], s.depth_sum.cpu().numpy().astype(np.int64))
        assert np.array_equal(ref["tree"].node_values[:, 0], s.search_value.cpu().numpy())
    s.close()


@pytest.mark.parametrize("qt", ["qtransform_completed_by_mix_value", "qtransform_by_parent_and_siblings"])
@pytest.mark.parametrize("A,E,S,B,maxc", [(12, 8, 50, 40, 16), (16, 8, 30, 25, 5), (2, 8, 160, 30, 16), (4, 8, 200, 12, 3)])
def test_gumbel_fused_wide_and_long_instances_built_on_demand(oracle, A, E, S, B, maxc, qt):
    """Gumbel MuZero on the round-5 instances: more than eight actions (sequential halving over up to 16 considered
    actions, in-lane sums of 16 terms in the canonical butterfly order) and more than 127 simulations (root paths in
    HBM, root Gumbel noise behind the tree) -- built on demand, every tree array equal to the oracle's."""
    from muax_amd import MuZeroSearch, SearchConfig, _jit
    kind = 1 if qt.endswith("mix_value") else 0
    case = make_case(oracle, 170 + A + S, B, 5, E, A, S, invalid_frac=0.3 if A > 2 else 0.0)
    key = [23, S]
    assert _jit.ensure_instance(A, E, case["F"], S)
    s = MuZeroSearch(B, SearchConfig(A, S, E, policy="gumbel", qtransform=qt, max_num_considered_actions=maxc, tiebreak=False))
    s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], 10, 0.99)
    out = s.act_mlp(torch.from_numpy(case["obs"]), key, gumbel=torch.from_numpy(case["gumbel"]), with_tree=True,
                    invalid_actions=None if case["invalid"] is None else torch.from_numpy(case["invalid"]))
    torch.cuda.synchronize()
    _compare(_gumbel_oracle_act(oracle, case, key, kind, maxc, gumbel=case["gumbel"]), s, out)


@pytest.mark.parametrize("A,E,S,B,policy", [(18, 8, 50, 130, "muzero"), (2, 8, 300, 40, "muzero"), (4, 100, 40, 25, "muzero"),
                                            (33, 20, 70, 21, "muzero"), (18, 8, 40, 50, "gumbel"), (2, 8, 260, 9, "gumbel"),
                                            (18, 8, 30, 2100, "muzero"), (5, 70, 20, 2200, "muzero"), (33, 12, 20, 2100, "muzero"),
                                            (1, 70, 130, 6, "muzero"), (1, 70, 70, 2100, "muzero"),
                                            (20, 8, 20, 2100, "gumbel")])
def test_generic_one_launch_search_matches_oracle(oracle, A, E, S, B, policy):
    """What no instance of the fused kernel can serve -- 18 / 33 actions, 260 / 300 simulations, a 100-wide embedding --
    through mzs_act_mlp's generic route (mz_mlp_generic.cuh: the trio with run-time shapes, tree in HBM, one launch for
    all simulations): every tree array, actions, weights, values and depth sums equal to the oracle's, both policies.
    More than 2048 roots of a short search take the kernel's 128-register build (four wavefronts per SIMD, weights through
    32-bit offsets, a dozen loop invariants in scratch): one case per variant of it (two / one / any action slots, Gumbel).
    One action: the tree is a chain, paths of 70 / 130 levels (the return chain's 63-level chunks, pointer jumping in LDS)."""
    case = make_case(oracle, 500 + A + E, B, 6, E, A, S, invalid_frac=0.2 if A > 2 else 0.0)
    key = [77, S]
    if policy == "muzero":
        s, out = _fused_any(case, True, key, "generic")
        _compare(_oracle(oracle, case, True, key), s, out)
        return
    s, out = _fused_any(case, False, key, "generic", policy="gumbel", qtransform="qtransform_completed_by_mix_value")
    mlp = oracle.Mlp(case["w"], case["obs_dim"], E, A, case["F"])
    pl, v, emb = oracle.root_inference(mlp, case["obs"])
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S)
    oracle.tree_init(tree, oracle.mask_root_logits(pl, case["invalid"]), v, emb, case["invalid"])
    for sim in range(S):
        p_, a_, _ = oracle.gumbel_step_select(tree, cfg, case["gumbel"], 1, 16)
        oracle.step_expand_backup(tree, sim, p_, a_, *oracle.recurrent_inference(mlp, a_, tree.embeddings[np.arange(B), p_]))
    action, weights = oracle.gumbel_finish(tree, case["gumbel"], 1)
    assert np.array_equal(action, out.action.cpu().numpy()) and np.array_equal(weights, out.action_weights.cpu().numpy())
    assert np.array_equal(v, s.root_value.cpu().numpy())
    assert_trees_equal(tree, out.search_tree, exact_floats=True)


def test_generic_route_at_4096_roots_x_300_simulations(oracle, monkeypatch):
    """B (S + 1)^2 cached path words beyond the slab budget used to end in MZS_E_UNSUPPORTED (and the
    step-wise fall-back, 70-270x).  The budget is 8 GiB of the 288 GB now (4096 x 300 = 1.5 GB fits undivided) and a
    tree beyond it is searched in chunks of roots that fit the slab (roots never interact; same kernels, same per-root
    PRNG streams) -- forced here with a 64 MB budget, i.e. 23 chunks of 184 roots: both equal to the oracle's tree."""
    A, E, S, B = 2, 8, 300, 4096
    case = make_case(oracle, 901, B, 4, E, A, S)
    key = [5, S]
    ref = _oracle(oracle, case, True, key)
    for budget in (None, "64"):
        if budget:
            monkeypatch.setenv("MZS_JUMP_BUDGET_MB", budget)
        s, out = _fused_any(case, True, key, "generic")
        _compare(ref, s, out)
        # a second act() on the same handle (the slab is reused chunk after chunk, act after act), without an export
        out2 = s.act_mlp(torch.from_numpy(case["obs"]), [6, S], dirichlet_noise=torch.from_numpy(case["noise"]),
                         gumbel=torch.from_numpy(case["gumbel"]))
        torch.cuda.synchronize()
        ref2 = _oracle(oracle, case, True, [6, S])
        assert np.array_equal(ref2["action"], out2.action.cpu().numpy())
        assert np.array_equal(ref2["action_weights"], out2.action_weights.cpu().numpy())
        assert np.array_equal(ref2["depth_sum"], s.depth_sum.cpu().numpy().astype(np.int64))
        s.close()


@pytest.mark.parametrize("A,E,S,B,policy", [(18, 8, 60, 97, "muzero"), (3, 8, 270, 50, "gumbel")])
def test_generic_route_in_chunks_matches_oracle(oracle, monkeypatch, A, E, S, B, policy):
    """The chunked generic route with invalid actions, ragged last chunk (97 roots in chunks of 4 / 50 in chunks of 3),
    both policies: equal to the oracle like the undivided launch."""
    monkeypatch.setenv("MZS_JUMP_BUDGET_MB", "1")
    case = make_case(oracle, 640 + A, B, 6, E, A, S, invalid_frac=0.2)
    key = [78, S]
    if policy == "muzero":
        s, out = _fused_any(case, True, key, "generic")
        _compare(_oracle(oracle, case, True, key), s, out)
        return
    s, out = _fused_any(case, False, key, "generic", policy="gumbel", qtransform="qtransform_completed_by_mix_value")
    _compare(_gumbel_oracle_act(oracle, case, key, 1, 16, gumbel=case["gumbel"]), s, out)


@pytest.mark.parametrize("A,E,S,B", [(2, 8, 160, 70), (16, 8, 50, 45), (9, 8, 50, 33)])
def test_muzero_only_instances_planned_per_policy(oracle, A, E, S, B):
    """the planner sized every on-demand instance for the Gumbel modes' five-word children.  Planned
    per policy the MuZero policy gets 16 roots per workgroup at 160 simulations (12 before), 8 at 16 actions (4), 12 at 9
    actions (8): such an instance (-DMZ_FUSED_MUZERO_ONLY=1) declines a Gumbel handle, is tried before the all-modes ones
    for a MuZero handle, and gives the oracle's tree."""
    from muax_amd import MuZeroSearch, SearchConfig, _jit
    F = 21
    assert _jit.plan(A, E, F, S, gumbel=False)[2] > _jit.plan(A, E, F, S, gumbel=True)[2]
    assert _jit.ensure_instance(A, E, F, S, gumbel=False)
    case = make_case(oracle, 730 + A + S, B, 5, E, A, S, invalid_frac=0.25 if A > 2 else 0.0)
    for tiebreak in (True, False):
        key = [41, S + tiebreak]
        s, out = _fused(case, tiebreak, key)
        _compare(_oracle(oracle, case, tiebreak, key), s, out)
        s.close()
    # a Gumbel handle of the shape is NOT served by it: without an all-modes instance the library has none
    if (A, E, 2, *_jit.plan(A, E, F, S, True)[1:], True) not in _jit._loaded:
        g = MuZeroSearch(B, SearchConfig(A, S, E, policy="gumbel", tiebreak=False))
        g.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], 10, 0.99)
        with pytest.raises(ValueError, match="no fused kernel instance"):
            g.act_mlp(torch.from_numpy(case["obs"]), [1, 2], gumbel=torch.from_numpy(case["gumbel"]))
        g.close()


def test_model_act_above_the_fused_kernels_limits_takes_the_generic_route(oracle):
    """The reference's act() takes any num_simulations (muax/model.py:82-96): 300 simulations on the default trio (no
    instance possible) go through the library's generic one-launch search -- no step-wise policy adapter, no torch
    modules -- and give the oracle's actions, weights and values for the same key; so does an 18-action trio."""
    import muax_amd as mx
    # ((2, 127) / (2, 200) / (12, 50): LONG instances -- listed, and built on demand -- through the NumPy entry point, i.e.
    # without a tree export, on the handle's own path / embedding scratch)
    for A, S in ((2, 300), (18, 50), (2, 127), (2, 200), (12, 50)):
        g = torch.Generator().manual_seed(0)
        net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(A, 21, generator=g),
                              mx.nn.Dynamic(8, A, 21, generator=g))
        m = mx.MuZero(net)
        m.init(0, np.zeros((1, 4)))
        obs = np.random.default_rng(0).uniform(-1, 1, (10, 4)).astype(F32)
        a, pi, v = m.act(2, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=S)
        assert a.shape == (10,) and np.allclose(pi.sum(1), 1, atol=1e-6) and np.allclose(pi * S, np.round(pi * S), atol=1e-4)
        assert len(m._policy._handles) == 0  # NOT through the step-wise policy adapter
        w = {k: p.detach().cpu().numpy() for k, p in mx.nn.mlp_trio_weights(m.network).items()}
        mlp = oracle.Mlp(w, 4, 8, A, 21)
        noise = oracle.dirichlet(oracle.split([0, 2], 3)[1], 0.3, 10, A)
        ref = oracle.act_mlp(mlp, oracle.SearchCfg(S, tiebreak=1), obs, [0, 2], noise, 0.25, None, 1.0)
        assert np.array_equal(ref["action"], a) and np.array_equal(ref["action_weights"], pi) and np.array_equal(ref["root_value"], v)


def test_bad_arguments_raise_value_error():
    from muax_amd import MuZeroSearch, SearchConfig
    with pytest.raises(ValueError):
        MuZeroSearch(4, SearchConfig(65, 5, 8))      # more than 64 actions
    with pytest.raises(ValueError):
        MuZeroSearch(4, SearchConfig(2, 0, 8))       # no simulations
    s = MuZeroSearch(4, SearchConfig(2, 5, 8, global_batch=8, root_offset=4))
    with pytest.raises(ValueError):
        MuZeroSearch(4, SearchConfig(2, 5, 8, global_batch=6, root_offset=4))  # shard past the batch
    w = {k: torch.zeros(s) for k, s in {"repr_w": (4, 8)}.items()}
    with pytest.raises(KeyError):
        s.set_mlp_weights(w, 4)
    with pytest.raises(ValueError):
        s.act_mlp(torch.zeros(4, 4), 0)              # weights never set


# ---------------------------------------------------------------- Gumbel MuZero (muax/policy.py:33-47)

@pytest.mark.parametrize("qt", ["qtransform_completed_by_mix_value", "qtransform_by_parent_and_siblings"])
@pytest.mark.parametrize("A,E,S,B,maxc", [(4, 8, 32, 70, 16), (18, 24, 40, 33, 5), (2, 8, 50, 48, 16), (4, 260, 10, 9, 4)])
@pytest.mark.parametrize("walk", [False, True])
def test_gumbel_stepwise_matches_oracle(oracle, A, E, S, B, maxc, qt, walk, monkeypatch):
    """mctx.gumbel_muzero_policy on the step-wise kernels (both sets): same torch net outputs fed to both
    sides, Gumbel noise from the key; trees, chosen actions and the completed-Q policy target must agree
    exactly."""
    if walk:
        monkeypatch.setenv("MZS_STEP_WALK", "1")
    from muax_amd import MuZeroSearch, SearchConfig
    kind = 1 if qt.endswith("mix_value") else 0
    case = make_case(oracle, 60 + A, B, 6, E, A, S, invalid_frac=0.25 if A > 2 else 0.0)
    root, rec = _torch_recurrent(case)
    key = [8, 9]
    pl, v, emb = root(torch.from_numpy(case["obs"]).cuda())
    s = MuZeroSearch(B, SearchConfig(A, S, E, policy="gumbel", qtransform=qt, max_num_considered_actions=maxc,
                                     tiebreak=False))
    inv = None if case["invalid"] is None else torch.from_numpy(case["invalid"])
    s.root_gumbel(pl, v, emb, key, inv)
    g = oracle.gumbel(oracle.split(key, 2)[1], B * A).reshape(B, A)
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S)
    oracle.tree_init(tree, oracle.mask_root_logits(pl.cpu().numpy(), case["invalid"]), v.cpu().numpy(),
                     emb.cpu().numpy(), case["invalid"])
    for sim in range(S):
        action, pemb = s.select(sim)
        p_ref, a_ref, _ = oracle.gumbel_step_select(tree, cfg, g, kind, maxc)
        assert np.array_equal(a_ref, action.cpu().numpy()), sim
        outs = rec(action, pemb)
        s.expand_backup(sim, *outs)
        oracle.step_expand_backup(tree, sim, p_ref, a_ref, *[o.cpu().numpy() for o in outs])
    out = s.finish(with_tree=True)
    a_ref, w_ref = oracle.gumbel_finish(tree, g, kind)
    assert np.array_equal(a_ref, out.action.cpu().numpy())
    assert np.array_equal(w_ref, out.action_weights.cpu().numpy())
    assert_trees_equal(tree, out.search_tree, exact_floats=True)
    if case["invalid"] is not None:
        ok = case["invalid"].sum(1) < A
        assert (case["invalid"][np.arange(B), a_ref][ok] == 0).all()


def test_gumbel_policy_through_model_act(oracle):
    """MuZero(policy_class=GumbelMuZeroPolicy).act(): the reference's contract (muax/policy.py:33-47 behind
    muax/model.py:82-179), checked against mctx.gumbel_muzero_policy as the oracle restates it, driven with the
    same key: actions and action weights equal, both qtransforms, batched and unbatched."""
    import muax_amd as mx
    from muax_amd import prng
    g = torch.Generator().manual_seed(1)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(4, 21, generator=g),
                          mx.nn.Dynamic(8, 4, 21, generator=g))
    m = mx.MuZero(net, policy_class=mx.GumbelMuZeroPolicy)
    m.init(0, np.zeros((1, 6)))
    obs = np.random.default_rng(3).uniform(-1, 1, (40, 6)).astype(F32)
    w = {k: v.detach().cpu().numpy() for k, v in mx.nn.mlp_trio_weights(m.network).items()}

    def want(rows, kind, maxc):
        case = {"B": rows.shape[0], "A": 4, "E": 8, "S": 24, "w": w, "obs_dim": 6, "F": 21, "obs": rows, "invalid": None}
        return _gumbel_oracle_act(oracle, case, [int(x) for x in prng.as_key(11)], kind, maxc)

    a, pi, v = m.act(11, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=24)
    assert a.shape == (40,) and pi.shape == (40, 4) and np.allclose(pi.sum(1), 1, atol=1e-5)
    ref = want(obs, 0, 16)  # muax/model.py:230-231 forces by_parent_and_siblings on every policy
    assert np.array_equal(a, ref["action"]) and np.array_equal(pi, ref["action_weights"])
    assert np.array_equal(v, ref["root_value"])
    a2, pi2 = m.act(11, obs, with_pi=True, obs_from_batch=True, num_simulations=24,
                    qtransform="qtransform_completed_by_mix_value", max_num_considered_actions=2)
    ref2 = want(obs, 1, 2)
    assert np.array_equal(a2, ref2["action"]) and np.array_equal(pi2, ref2["action_weights"])
    # unbatched: python int, and the one-root search of the same key (a batch of one draws its own Gumbel row)
    a1, pi1, v1 = m.act(11, obs[0], with_pi=True, with_value=True, num_simulations=24)
    ref1 = want(obs[:1], 0, 16)
    assert isinstance(a1, int) and a1 == int(ref1["action"][0])
    assert pi1.shape == (1, 4) and np.array_equal(pi1, ref1["action_weights"]) and v1 == float(ref1["root_value"][0])
    m2 = mx.MuZero(net.representation_fn, net.prediction_fn, net.dynamic_fn, policy="gumbel")
    m2.init(0, np.zeros((1, 6)))
    a3 = m2.act(11, obs, obs_from_batch=True, num_simulations=24)
    assert np.array_equal(a3, a)


def _gumbel_oracle_act(oracle, case, key, kind, maxc, gumbel=None, max_depth=0):
    """mctx.gumbel_muzero_policy around the oracle's MLP trio, composed from the oracle's pieces."""
    B, A, E, S = case["B"], case["A"], case["E"], case["S"]
    mlp = oracle.Mlp(case["w"], case["obs_dim"], E, A, case["F"])
    pl, v, emb = oracle.root_inference(mlp, case["obs"])
    g = gumbel if gumbel is not None else oracle.gumbel(oracle.split(key, 2)[1], B * A).reshape(B, A)
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S, max_depth=max_depth)
    oracle.tree_init(tree, oracle.mask_root_logits(pl, case["invalid"]), v, emb, case["invalid"])
    dsum = np.zeros(B, np.int64)
    for sim in range(S):
        p_, a_, d_ = oracle.gumbel_step_select(tree, cfg, g, kind, maxc)
        dsum += d_
        oracle.step_expand_backup(tree, sim, p_, a_, *oracle.recurrent_inference(mlp, a_, tree.embeddings[np.arange(B), p_]))
    action, weights = oracle.gumbel_finish(tree, g, kind)
    return {"action": action, "action_weights": weights, "root_value": v, "depth_sum": dsum, "tree": tree}


@pytest.mark.parametrize("qt", ["qtransform_completed_by_mix_value", "qtransform_by_parent_and_siblings"])
@pytest.mark.parametrize("A,E,S,B,maxc", [(2, 8, 50, 200, 16), (4, 8, 50, 90, 3), (4, 32, 40, 50, 16), (3, 8, 30, 40, 2),
                                          (2, 8, 50, 4096, 16)])  # the last: BASELINE config 5's acting half at full size
def test_gumbel_fused_matches_oracle(oracle, A, E, S, B, maxc, qt):
    """The whole Gumbel MuZero act() in the fused kernel (sequential halving at the root, cached
    deterministic interior decisions through the JUMP words) against the oracle: bit-exact."""
    from muax_amd import MuZeroSearch, SearchConfig
    kind = 1 if qt.endswith("mix_value") else 0
    case = make_case(oracle, 70 + A + E, B, 4 if E == 8 else 8, E, A, S, invalid_frac=0.3 if A > 2 else 0.0)
    key = [21, 22]
    s = MuZeroSearch(B, SearchConfig(A, S, E, policy="gumbel", qtransform=qt, max_num_considered_actions=maxc,
                                     tiebreak=False))
    s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], 10, 0.99)
    out = s.act_mlp(torch.from_numpy(case["obs"]), key,
                    invalid_actions=None if case["invalid"] is None else torch.from_numpy(case["invalid"]),
                    with_tree=True)
    torch.cuda.synchronize()
    _compare(_gumbel_oracle_act(oracle, case, key, kind, maxc), s, out)


def test_gumbel_fused_injected_noise_and_max_depth(oracle):
    from muax_amd import MuZeroSearch, SearchConfig
    case = make_case(oracle, 91, 64, 4, 8, 2, 30)
    g = np.random.default_rng(1).gumbel(size=(64, 2)).astype(F32)
    s = MuZeroSearch(64, SearchConfig(2, 30, 8, policy="gumbel", qtransform="qtransform_completed_by_mix_value",
                                      max_depth=4, tiebreak=False))
    s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, 4, 10, 0.99)
    out = s.act_mlp(torch.from_numpy(case["obs"]), 5, gumbel=torch.from_numpy(g), with_tree=True)
    torch.cuda.synchronize()
    _compare(_gumbel_oracle_act(oracle, case, [0, 5], 1, 16, gumbel=g, max_depth=4), s, out)


@pytest.mark.parametrize("B,tiebreak", [(4097, True), (8192, True), (10000, False)])
def test_fused_packed_record_lunarlander_above_one_workgroup_per_cu(oracle, B, tiebreak):
    """A = 4, E = 32 above 4096 roots: the PACKED compact record (child indices and visit counts as bytes, first-layer
    matrices in LDS, paths and embeddings in HBM, two workgroups per CU).  Same bits as the oracle on every tree array
    -- BASELINE config 3's whole 8192-root batch included --, with masks, a max_depth cut, no export (the handle's own
    embedding scratch), and against a shard of the same rows run on the plain instance."""
    case = make_case(oracle, 79, B, 8, 32, 4, 50, invalid_frac=0.1)
    key = [B, 5]
    s, out = _fused(case, tiebreak, key)
    _compare(_oracle(oracle, case, tiebreak, key), s, out)
    ref = _oracle(oracle, case, tiebreak, key, max_depth=5, temperature=0.5, use_gumbel=False)
    s, out = _fused(case, tiebreak, key, max_depth=5, temperature=0.5, use_gumbel=False)
    _compare(ref, s, out)
    # no tree export: embeddings in the handle's scratch
    from muax_amd import MuZeroSearch, SearchConfig
    s3 = MuZeroSearch(B, SearchConfig(4, 50, 32, max_depth=5, tiebreak=tiebreak))
    s3.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, 8, 10, 0.99)
    o3 = s3.act_mlp(torch.from_numpy(case["obs"]), key, dirichlet_noise=torch.from_numpy(case["noise"]),
                    invalid_actions=torch.from_numpy(case["invalid"]), temperature=0.5)
    torch.cuda.synchronize()
    assert np.array_equal(ref["action"], o3.action.cpu().numpy())
    assert np.array_equal(ref["action_weights"], o3.action_weights.cpu().numpy())
    assert np.array_equal(ref["depth_sum"], s3.depth_sum.cpu().numpy().astype(np.int64))
    sl = slice(B - 200, B)
    s2, o2 = _fused(case, tiebreak, key, max_depth=5, temperature=0.5, use_gumbel=False, global_batch=B,
                    root_offset=B - 200, rows=sl)
    assert torch.equal(o2.action, out.action[sl])
    for f in out.search_tree._fields:
        assert torch.equal(getattr(o2.search_tree, f), getattr(out.search_tree, f)[sl]), f


@pytest.mark.parametrize("B,tiebreak", [(4097, True), (9000, True), (16384, False)])
def test_fused_compact_record_above_one_workgroup_per_cu(oracle, B, tiebreak):
    """More 16-root workgroups than CUs (> 4096 roots on MI355X): mzs_act_mlp takes the compact-record instance
    (root paths in HBM, two workgroups per CU).  Same bits as the oracle on every tree array, with masks, a
    max_depth cut (the overshoot branch reads the parent's path from HBM) and against a shard of the same rows
    run on the plain instance."""
    case = make_case(oracle, 77, B, 4, 8, 2, 50, invalid_frac=0.1)
    key = [B, 3]
    s, out = _fused(case, tiebreak, key)
    _compare(_oracle(oracle, case, tiebreak, key), s, out)
    s, out = _fused(case, tiebreak, key, max_depth=6, temperature=0.5, use_gumbel=False)
    _compare(_oracle(oracle, case, tiebreak, key, max_depth=6, temperature=0.5, use_gumbel=False), s, out)
    sl = slice(B - 300, B)
    s2, o2 = _fused(case, tiebreak, key, max_depth=6, temperature=0.5, use_gumbel=False, global_batch=B,
                    root_offset=B - 300, rows=sl)
    assert torch.equal(o2.action, out.action[sl])
    for f in out.search_tree._fields:
        assert torch.equal(getattr(o2.search_tree, f), getattr(out.search_tree, f)[sl]), f
