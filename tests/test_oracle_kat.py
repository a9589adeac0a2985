"""Known-answer tests that pin the CPU oracle.

The reference has no tests on this path and jax/mctx cannot be imported here, so
these hand-derivable cases -- plus the JAX-published PRNG vectors -- are what
the oracle is pinned by ("parity unpinned" otherwise).
"""
import numpy as np
import pytest

from oracle import mz_numpy as mn

F32 = np.float32


# ---- (7) threefry / JAX PRNG ------------------------------------------------

def test_threefry_random123_vectors(oracle):
    # Random123 / jax tests/random_test.py::testThreefry2x32
    assert [hex(int(x)) for x in oracle.threefry2x32([0, 0], 0, 0)] == ["0x6b200159", "0x99ba4efe"]
    assert [hex(int(x)) for x in oracle.threefry2x32([0xFFFFFFFF] * 2, 0xFFFFFFFF, 0xFFFFFFFF)] == \
        ["0x1cb996fc", "0xbb002be7"]
    assert [hex(int(x)) for x in oracle.threefry2x32([0x13198A2E, 0x03707344], 0x243F6A88, 0x85A308D3)] == \
        ["0xc4923a9c", "0x483df7a0"]


def test_jax_documented_split_and_uniform(oracle):
    # jax.random.split(PRNGKey(0)) and jax.random.uniform(PRNGKey(0)) as printed in the JAX docs
    assert oracle.split([0, 0], 2).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert oracle.uniform([0, 0], 1)[0] == F32(0.41845703)


def test_jax_documented_normal_and_the_tutorial_key_walk(oracle):
    """Values printed in JAX's own documentation (reproduced by the oracle to the last printed digit, i.e. to the bit: a
    float32's shortest decimal form identifies it): the "Pseudorandom numbers" tutorial walks PRNGKey(42) --
    `random.normal(key)` -0.18471177; `new_key, subkey = random.split(key)` -> [2465931498 3679230171] and
    [255383827 267815257], `random.normal(subkey)` 1.3694694 -- and the jax.random module docs print
    `random.normal(PRNGKey(0))` as -0.20584226.  They pin split(), the [-1, 1) uniform and the erf_inv-based normal that
    the Dirichlet / gamma sampler restatement (mz_oracle.c, DESIGN.md section 2 "Root noise") is built on."""
    assert oracle.split([0, 42], 2).tolist() == [[2465931498, 3679230171], [255383827, 267815257]]
    assert oracle.normal([0, 42]) == F32(-0.18471177)
    assert oracle.normal([255383827, 267815257]) == F32(1.3694694)
    assert oracle.normal([0, 0]) == F32(-0.20584226)
    # the JAX quickstart / README: key = random.PRNGKey(0); x = random.normal(key, (10,)); print(x)
    quickstart = [-0.3721109, 0.26423115, -0.18252768, -0.7368197, -0.44030377, -0.1521442, -0.67135346, -0.5908641,
                  0.73168886, 0.5673026]
    assert oracle.normal([0, 0], 10).tolist() == [float(F32(v)) for v in quickstart]


def test_random_bits_layout_odd_and_even(oracle):
    key = np.array([7, 9], np.uint32)
    # even: first half of iota hashed against second half
    b4 = oracle.random_bits(key, 4)
    t0, t1 = oracle.threefry2x32(key, 0, 2), oracle.threefry2x32(key, 1, 3)
    assert b4.tolist() == [t0[0], t1[0], t0[1], t1[1]]
    # odd: zero padded, last output dropped
    b3 = oracle.random_bits(key, 3)
    t0, t1 = oracle.threefry2x32(key, 0, 2), oracle.threefry2x32(key, 1, 0)
    assert b3.tolist() == [t0[0], t1[0], t0[1]]
    u = oracle.uniform(key, 64)
    assert (u >= 0).all() and (u < 1).all()
    g = oracle.gumbel(key, 64)
    assert np.isfinite(g).all()


# ---- (4) codec ----------------------------------------------------------------

def test_math_accuracy(oracle):
    x = np.linspace(-87, 0, 4001).astype(F32)
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(oracle.exp(x) - ref) / ref) < 2e-7
    assert oracle.exp(np.array([-100.0], F32))[0] == 0.0
    x = np.concatenate([np.geomspace(1.2e-38, 1e38, 4001), 1 + np.linspace(-1e-3, 1e-2, 1001)]).astype(F32)
    ref = np.log(x.astype(np.float64))
    assert np.max(np.abs(oracle.log(x) - ref) / np.maximum(np.abs(ref), 1e-30)) < 2e-7
    x = np.linspace(-20, 5, 4001).astype(F32)
    ref = np.where(x > 0, x, np.expm1(np.minimum(x, 0).astype(np.float64)))
    assert np.max(np.abs(oracle.elu(x) - ref)) < 1e-7


def test_inv_scaling_roundtrip(oracle):
    x = np.linspace(-300, 300, 2001).astype(F32)
    y = oracle.inv_scaling(mn.scaling(x))
    assert np.max(np.abs(y - x) / np.maximum(1, np.abs(x))) < 2e-3  # f32 cancellation in the formula
    assert oracle.inv_scaling(np.zeros(1, F32))[0] == 0.0
    # against the same formula in float64
    xs = np.linspace(-10, 10, 401).astype(F32)
    e = 1e-3
    ref = np.sign(xs) * (((np.sqrt(1 + 4 * e * (np.abs(xs.astype(np.float64)) + 1 + e)) - 1) / (2 * e)) ** 2 - 1)
    assert np.max(np.abs(oracle.inv_scaling(xs) - ref)) < 2e-3


def test_support_roundtrip(oracle):
    for x in [-50.0, -3.3, -0.2, 0.0, 0.7, 12.0, 99.0]:
        p = mn.scalar_to_support(np.array([x], F32), 10)[0]
        assert abs(p.sum() - 1) < 1e-6
        y = oracle.support_to_scalar(p, 10)
        assert abs(y - x) <= 2e-3 * max(1, abs(x))


def test_sum16_is_the_documented_tree(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 2, 4, 5, 16, 21, 33, 40):
        x = rng.standard_normal(n).astype(F32)
        p = np.zeros(16, F32)
        for l in range(16):
            idx = list(range(l, n, 16))
            if idx:
                acc = x[idx[0]]
                for i in idx[1:]:
                    acc = F32(acc + x[i])
                p[l] = acc
        for m in (1, 2, 4, 8):
            p = np.array([F32(p[l] + p[l ^ m]) for l in range(16)], F32)
        assert oracle.sum16(x) == p[0]
        assert abs(oracle.sum16(x) - x.astype(np.float64).sum()) < 1e-5


def test_softmax_and_minmax(oracle):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(21).astype(F32) * 3
    assert np.allclose(oracle.softmax(x), mn.softmax(x), atol=2e-7)
    s = rng.standard_normal((1, 8)).astype(F32)
    assert np.allclose(oracle.min_max_normalize(s[0]), mn.min_max_normalize(s)[0], atol=1e-7)
    c = np.full(8, 0.3, F32)  # degenerate range -> scale + 1e-5 (muax/nn.py:42)
    assert np.all(oracle.min_max_normalize(c) == 0)


# ---- search KATs (1)(2)(3)(5)(6) ----------------------------------------------

def _const_model(A, E, value, reward=0.0, logits=None):
    def rec(action, emb):
        B = len(action)
        pl = np.zeros((B, A), F32) if logits is None else np.tile(np.asarray(logits, F32), (B, 1))
        return (np.full(B, reward, F32), np.full(B, 0.99, F32), pl, np.full(B, value, F32),
                np.zeros((B, E), F32))
    return rec


def _run_c(oracle, B, A, E, S, root_logits, root_value, rec, max_depth=0, invalid=None,
           tiebreak=0, keys=None):
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S, max_depth=max_depth, tiebreak=tiebreak)
    oracle.tree_init(tree, np.tile(np.asarray(root_logits, F32), (B, 1)), np.full(B, root_value, F32),
                     np.zeros((B, E), F32), invalid)
    for s in range(S):
        p, a, d = oracle.step_select(tree, cfg, s, None if keys is None else keys[s])
        r, disc, pl, v, ne = rec(a, tree.embeddings[np.arange(B), p])
        oracle.step_expand_backup(tree, s, p, a, r, disc, pl, v, ne)
    return tree


def test_kat_single_simulation(oracle):
    """(1) S=1: node 1 = argmax prior child; visits [2,1]; root value (v0 + r + g*v1)/2."""
    A, E = 3, 2
    rec = _const_model(A, E, value=0.5, reward=0.25)
    tree = _run_c(oracle, 1, A, E, 1, [0.1, 2.0, -1.0], 1.0, rec)
    assert tree.children_index[0, 0].tolist() == [-1, 1, -1]
    assert tree.node_visits[0].tolist() == [2, 1]
    assert tree.parents[0].tolist() == [-1, 0] and tree.action_from_parent[0].tolist() == [-1, 1]
    expect = F32((F32(1.0) * F32(1) + (F32(0.25) + F32(0.99) * F32(0.5))) / F32(2))
    assert tree.node_values[0, 0] == expect
    assert tree.children_visits[0, 0].tolist() == [0, 1, 0]
    assert tree.children_values[0, 0, 1] == F32(0.5) and tree.children_rewards[0, 0, 1] == F32(0.25)


def test_kat_breadth_first_fill(oracle):
    """(2) uniform priors, zero reward, constant value, no noise => breadth-first in action order."""
    A, E, S = 3, 1, 12
    tree = _run_c(oracle, 1, A, E, S, [0.0] * A, 0.7, _const_model(A, E, value=0.7))
    # value_score == 0 everywhere (q == node value would need r + g*v == v; use visit pattern only)
    vc = tree.children_visits[0, 0]
    assert vc.sum() == S and vc.max() - vc.min() <= 1
    assert tree.children_index[0, 0].tolist() == [1, 2, 3]


def test_kat_max_depth_one(oracle):
    """(3) max_depth=1: every simulation (re)expands a root child; sum of root child visits == S."""
    A, E, S = 2, 1, 9
    tree = _run_c(oracle, 2, A, E, S, [0.3, -0.3], 0.0, _const_model(A, E, value=0.1, reward=0.05),
                  max_depth=1)
    assert (tree.children_visits[:, 0].sum(axis=1) == S).all()
    assert (tree.node_visits[:, 0] == S + 1).all()
    # only root children exist; re-expanded nodes accumulate node_visits (mctx update_tree_node)
    kids = tree.children_index[0, 0]
    assert min(kids) == 1 and (kids > 0).all()  # a child first expanded at simulation i is node i+1
    assert tree.node_visits[0, kids].sum() == S
    others = np.setdiff1d(np.arange(1, S + 1), kids)
    assert (tree.parents[0, others] == -1).all() and (tree.node_visits[0, others] == 0).all()


def test_kat_invariants_and_numpy_agreement(oracle):
    """(6) structural invariants on a random MLP search, and C == NumPy restatement."""
    B, obs_dim, E, A, Fs, S = 64, 4, 8, 3, 21, 30
    w = oracle.random_mlp_weights(3, obs_dim, E, A, Fs, bias_scale=0.1)
    rng = np.random.default_rng(5)
    obs = rng.uniform(-1, 1, (B, obs_dim)).astype(F32)
    noise = rng.dirichlet([0.3] * A, B).astype(F32)
    gum = rng.gumbel(size=(B, A)).astype(F32)
    mlp = oracle.Mlp(w, obs_dim, E, A, Fs)
    rc = oracle.act_mlp(mlp, oracle.SearchCfg(S), obs, [0, 1], noise, 0.25, None, 1.0, gum)
    t = rc["tree"]
    assert (t.node_visits[:, 0] == S + 1).all()
    for b in range(B):
        for n in range(S + 1):
            assert t.children_visits[b, n].sum() == t.node_visits[b, n] - 1
            for a in range(A):
                c = t.children_index[b, n, a]
                if c >= 0:
                    assert t.parents[b, c] == n and t.action_from_parent[b, c] == a
                    assert t.children_visits[b, n, a] == t.node_visits[b, c]
    rn = mn.act_mlp(w, obs, S, A, E, dirichlet_noise=noise, gumbel=gum)
    good = rn["min_margin"] > 1e-4
    assert good.sum() > B // 2
    tn = rn["tree"].arrays()
    for k, a in t.arrays().items():
        if a.dtype == np.int32:
            assert (a[good] == tn[k][good]).all(), k
        else:
            assert np.allclose(a[good], tn[k][good], rtol=2e-3, atol=2e-3), k
    assert (rc["action"][good] == rn["action"][good]).all()
    assert np.array_equal(rc["depth_sum"], rn["depth_sum"]) or (rc["depth_sum"][good] == rn["depth_sum"][good]).all()


def test_kat_temperature_zero_is_argmax(oracle):
    """(5) temperature -> 0: action == argmax visit count whatever the gumbel draw."""
    B, A, E, S = 8, 4, 1, 20
    tree = _run_c(oracle, B, A, E, S, [0.5, 1.5, -0.5, 0.0], 0.2, _const_model(A, E, 0.3, 0.1, [0.2, 0.1, 0.0, -0.1]))
    g = np.random.default_rng(0).gumbel(size=(B, A)).astype(F32)
    action, w = oracle.summary_sample(tree, 0.0, g)
    vc = tree.children_visits[:, 0]
    assert np.allclose(w, vc / vc.sum(1, keepdims=True))
    for b in range(B):
        assert vc[b, action[b]] == vc[b].max()
    a1, _ = oracle.summary_sample(tree, 1.0, g)
    an, _ = mn.summary_sample(_as_np_tree(tree), 1.0, g)
    assert (a1 == an).all()


def _as_np_tree(t):
    n = mn.Tree(t.B, t.N, t.A, t.E)
    for k, v in t.arrays().items():
        setattr(n, k, v.copy())
    return n


def test_kat_invalid_actions_masked_at_root_only(oracle):
    A, E, S, B = 3, 1, 15, 2
    invalid = np.array([[1, 0, 0], [0, 0, 1]], np.uint8)
    logits = oracle.root_prior(np.zeros((B, A), F32), None, 0.0, invalid)
    assert logits[0, 0] == np.finfo(F32).min and logits[1, 2] == np.finfo(F32).min
    assert logits[0, 1] == 0 and logits[0, 2] == 0
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S)
    oracle.tree_init(tree, logits, np.zeros(B, F32), np.zeros((B, E), F32), invalid)
    rec = _const_model(A, E, 0.1, 0.0)
    for s in range(S):
        p, a, d = oracle.step_select(tree, cfg, s)
        oracle.step_expand_backup(tree, s, p, a, *rec(a, None))
    assert tree.children_visits[0, 0, 0] == 0 and tree.children_visits[1, 0, 2] == 0
    # interior nodes are free to pick any action
    assert (tree.children_visits[:, 1:, :].sum(axis=(0, 1)) > 0).all()


def test_tiebreak_stream_is_sharding_invariant(oracle):
    """Per-root keys come from (global batch, global index): a shard reproduces the full run."""
    B, obs_dim, E, A, Fs, S = 16, 4, 8, 2, 21, 12
    w = oracle.random_mlp_weights(7, obs_dim, E, A, Fs)
    obs = np.random.default_rng(2).uniform(-1, 1, (B, obs_dim)).astype(F32)
    mlp = oracle.Mlp(w, obs_dim, E, A, Fs)
    full = oracle.act_mlp(mlp, oracle.SearchCfg(S, tiebreak=1), obs, [3, 4], None, 0.0)
    half = oracle.act_mlp(mlp, oracle.SearchCfg(S, tiebreak=1, global_batch=B, root_offset=8), obs[8:], [3, 4],
                          None, 0.0)
    assert np.array_equal(full["action"][8:], half["action"])
    for k, a in full["tree"].arrays().items():
        assert np.array_equal(a[8:], half["tree"].arrays()[k]), k


def test_tiebreak_noise_matches_numpy_path(oracle):
    """Tie-break noise drawn by the C oracle == the documented key walk, fed to the NumPy search."""
    B, obs_dim, E, A, Fs, S = 8, 4, 8, 4, 21, 10
    w = oracle.random_mlp_weights(11, obs_dim, E, A, Fs)
    obs = np.random.default_rng(4).uniform(-1, 1, (B, obs_dim)).astype(F32)
    key = [5, 6]
    rc = oracle.act_mlp(oracle.Mlp(w, obs_dim, E, A, Fs), oracle.SearchCfg(S, tiebreak=1), obs, key, None, 0.0)
    k_sample, _, sims = oracle.sim_keys_from_act_key(key, S)
    cache = {}

    def noise_fn(sim, level, rows):
        out = np.zeros((len(rows), A), F32)
        for i, b in enumerate(rows):
            if (sim, b) not in cache:
                cache[(sim, b)] = oracle.split(sims[sim], B)[b]
            if level == 0:
                cache[(sim, b, "k")] = cache[(sim, b)]
            k2 = oracle.split(cache[(sim, b, "k")], 2)
            cache[(sim, b, "k")] = k2[0]
            out[i] = F32(1e-7) * oracle.uniform(k2[1], A)
        return out

    g = oracle.gumbel(k_sample, B * A).reshape(B, A)
    rn = mn.act_mlp(w, obs, S, A, E, dirichlet_noise=None, gumbel=g, noise_fn=noise_fn)
    good = rn["min_margin"] > 1e-4
    assert good.sum() >= B // 2
    assert (rc["tree"].children_index[good] == rn["tree"].children_index[good]).all()
    assert (rc["action"][good] == rn["action"][good]).all()


def test_markstein_small_integer_division_equals_true_division():
    """muax_amd/csrc/mz_fused.cuh (div_small) divides by visit counts with the correctly rounded
    reciprocal and two fma corrections; the oracle writes x / d.  Every binary32 mantissa, both signs,
    d = 1..1030 (round 6: the one-launch searches keep the 1 / n table for every visit count a 1023-simulation tree
    reaches; 300 until then), at two exponents (powers of two scale both sides alike): no mismatch."""
    from oracle import pyoracle as po
    assert po.markstein_mismatches(1030, 0) == 0
    assert po.markstein_mismatches(64, -40) == 0


FROZEN_DIRICHLET = [[1061768092, 1046139280], [1026641503, 1064626970], [1065042773, 1016567145]]


def test_dirichlet_restatement_statistics_shards_and_frozen_vector(oracle):
    """oracle/mz_oracle.c restates jax.random.dirichlet (loggamma by Marsaglia-Tsang rejection on the threefry key
    walk, log-space boost for alpha < 1, softmax).  No jax here to pin the float bits (spec-to-confirm): what CAN be
    pinned is the distribution (moments of Dir(alpha) and of Gamma(alpha)), the building blocks against scipy, the
    shard contract, and -- against accidental change -- one frozen vector of this restatement."""
    import math

    from scipy.special import erfinv
    L = oracle.lib()
    xs = np.linspace(-0.9999, 0.9999, 1001)
    assert max(abs(L.mzo_erf_inv(float(x)) - erfinv(float(np.float32(x)))) / max(1e-3, abs(erfinv(float(x)))) for x in xs) < 1e-5
    us = np.random.default_rng(0).uniform(1e-7, 0.999, 500).astype(np.float32)
    assert max(abs(L.mzo_log1p(-float(u)) - math.log1p(-float(u))) / abs(math.log1p(-float(u))) for u in us) < 3e-7
    d = oracle.dirichlet([0, 42], 0.3, 20000, 2)
    assert np.allclose(d.sum(1), 1, atol=1e-6) and d.min() >= 0
    assert abs(d[:, 0].mean() - 0.5) < 0.01 and abs(d[:, 0].var() - 0.25 / 1.6) < 0.005
    d18 = oracle.dirichlet([1, 2], 0.3, 4000, 18)
    assert abs(d18.mean() - 1 / 18) < 1e-6 and abs(d18.var(0).mean() - (1 / 18) * (17 / 18) / (18 * 0.3 + 1)) < 5e-4
    assert np.array_equal(oracle.dirichlet([1, 2], 0.3, 100, 18, global_batch=4000, root_offset=700), d18[700:800])
    for alpha in (0.3, 1.0, 2.5):  # Gamma(alpha): mean = var = alpha
        keys = oracle.split([3, 4], 4000)
        g = np.exp([L.mzo_loggamma_one(oracle._p(np.ascontiguousarray(k, np.uint32), oracle._u32p), alpha) for k in keys])
        assert abs(g.mean() - alpha) < 0.06 * max(1, alpha) and abs(g.var() - alpha) < 0.2 * max(1, alpha), alpha
    frozen = oracle.dirichlet([0, 42], 0.3, 3, 2)
    assert np.array_equal(frozen.view(np.uint32), np.array(FROZEN_DIRICHLET, np.uint32)), frozen.view(np.uint32).tolist()


def test_division_by_two_eps_and_clamped_elu_equal_the_spec(oracle):
    """Two instruction-count savings of the HIP kernels that must not change a bit: (d - 1) / 0.002f of
    _inv_scaling (muax/utils.py:70-76) as a 3-op Markstein sequence -- compared with the IEEE division for EVERY
    binary32 value with exponent 2^-27 .. 2^13 (344 M cases; _inv_scaling produces 2^-9 .. 2^-3) -- and ELU without
    its `x < -87 -> -1` select (the clamp at -87 already gives exp(-87) - 1 = -1 exactly)."""
    L = oracle.lib()
    assert L.mzo_div2eps_mismatches(100, 140) == 0
    for x in (-87.0, -87.000008, -88.0, -100.0, -1e4, -3e38, float("-inf")):
        assert L.mzo_elu_clamped(x) == -1.0 and L.mzo_elu(x) == -1.0
    xs = np.concatenate([np.linspace(-90, 3, 20001), -np.logspace(-30, 1.9, 500)]).astype(np.float32)
    assert all(L.mzo_elu_clamped(float(x)) == L.mzo_elu(float(x)) for x in xs)
