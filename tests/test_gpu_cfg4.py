"""BASELINE config 4 at its real workload, on the GPU: 84x84x4 frames, the reference's ResNet nets
(muax/nn.py:313-395), A = 18, E = 6*6*64 = 2304, num_simulations = 200, 128 roots = one GPU's shard of the
1024-root batch.

* the tree kernels against the oracle fed the very same net outputs (every tree array exact), plus the
  size-independent structural invariants;
* launch-shape independence: the recurrent kernel has two launch shapes (<= 128 roots: two workgroups per
  root, above: one) and both must give the same bits, so that rows 0..127 of a 1024-root act equal the
  128-root shard's act;
* accuracy of the fp32-MFMA recurrent kernel against an fp64 evaluation of the same torch modules, next to
  what MIOpen's fp32 gives on the same inputs;
* the lost-rendezvous branch of pair mode: status word set -> pair mode dropped, search repeated, same bits.
"""
import copy

import numpy as np
import pytest
import torch

import muax_amd as mx
from helpers import assert_trees_equal

pytestmark = pytest.mark.gpu
F32 = np.float32
A, S_FULL, SUPPORT = 18, 200, 10


def _nets(seed=0, A=A):
    g = torch.Generator().manual_seed(seed)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(A, 2 * SUPPORT + 1, generator=g),
            mx.nn.ResNetDynamic(A, 2 * SUPPORT + 1, generator=g))
    m = mx.MuZero(*mods)
    m.init(0, np.zeros((1, 84, 84, 4), F32))
    with torch.no_grad():  # biases and LayerNorm parameters away from their init values
        for mod in mods[1:]:
            for p in mod.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g).to(p.device))
    m.weights_changed()
    return m, mods


def _bench_nets(B=128):
    """tools/bench_atari.py's nets and frames, draw for draw (the frames BEFORE init: the lazy Linear layers draw from the
    same generator): trees ~44 levels deep on average at 200 simulations, where _nets' stay under 20."""
    g = torch.Generator().manual_seed(0)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(A, 2 * SUPPORT + 1, generator=g),
            mx.nn.ResNetDynamic(A, 2 * SUPPORT + 1, generator=g))
    obs = torch.randint(0, 256, (B, 84, 84, 4), generator=g).float().cuda()
    m = mx.MuZero(*mods)
    m.init(0, np.zeros((1, 84, 84, 4), F32))
    return m, mods, obs


def _frames(B, seed=1):
    return np.random.default_rng(seed).integers(0, 256, (B, 84, 84, 4)).astype(F32)


@pytest.fixture(autouse=True)
def _pair_mode_on():
    mx.nn.ResNetDynamic.use_pair_tower = True
    yield
    mx.nn.ResNetDynamic.use_pair_tower = True


def test_config4_workload_tree_kernels_against_the_oracle(oracle):
    """128 roots x 200 simulations x 18 actions x 2304-float embeddings, ResNet nets through the one-launch HIP
    recurrent_fn (pair mode at this batch).  The oracle's tree is driven with the same net outputs."""
    B, S = 128, S_FULL
    m, mods = _nets()
    obs = torch.from_numpy(_frames(B)).cuda()
    pl, v, emb = m._root_inference(None, None, obs)
    E = emb[0].numel()
    assert E == 2304 and pl.shape == (B, A)
    rng = np.random.default_rng(3)
    noise = rng.dirichlet([0.3] * A, B).astype(F32)
    invalid = (rng.uniform(size=(B, A)) < 0.1).astype(np.uint8)
    invalid[np.arange(B), rng.integers(0, A, B)] = 0
    key = [7, 11]
    s = mx.MuZeroSearch(B, mx.SearchConfig(A, S, E, tiebreak=True))
    s.root(pl, v, emb.reshape(B, -1), key, torch.from_numpy(invalid), torch.from_numpy(noise), 0.25)
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S, tiebreak=1)
    oracle.tree_init(tree, oracle.root_prior(pl.cpu().numpy(), noise, 0.25, invalid), v.cpu().numpy(),
                     emb.reshape(B, -1).cpu().numpy(), invalid)
    k_sample, _, sims = oracle.sim_keys_from_act_key(key, S)
    for sim in range(S):
        action, pemb = s.select(sim)
        p_ref, a_ref, _ = oracle.step_select(tree, cfg, sim, sims[sim])
        assert np.array_equal(a_ref, action.cpu().numpy()), sim
        (r, disc, logits, val), ns = m._recurrent_inference(None, None, action, pemb.reshape(B, 6, 6, 64))
        outs = (r, disc, logits, val, ns.reshape(B, -1))
        s.expand_backup(sim, *outs)
        oracle.step_expand_backup(tree, sim, p_ref, a_ref, *[o.cpu().numpy() for o in outs])
    out = s.finish(1.0, None, with_tree=True)
    g = oracle.gumbel(k_sample, B * A).reshape(B, A)
    a_ref, w_ref = oracle.summary_sample(tree, 1.0, g)
    assert np.array_equal(a_ref, out.action.cpu().numpy())
    assert np.array_equal(w_ref, out.action_weights.cpu().numpy())
    assert_trees_equal(tree, out.search_tree, exact_floats=True)
    assert mods[2]._pair_scratch and not mods[2].pair_lost()  # the recurrent kernel ran in pair mode, undisturbed
    # size-independent structure (as test_fused_full_size_properties)
    t = out.search_tree
    nv, cv, ci, par, afp = (x.cpu().numpy() for x in
                            (t.node_visits, t.children_visits, t.children_index, t.parents, t.action_from_parent))
    assert (nv[:, 0] == S + 1).all() and (cv[:, 0].sum(-1) == S).all()
    assert (cv.sum(-1) == nv - 1)[nv > 0].all()
    b, n, a = np.nonzero(ci >= 0)
    c = ci[b, n, a]
    assert (par[b, c] == n).all() and (afp[b, c] == a).all() and (cv[b, n, a] == nv[b, c]).all()
    assert (np.sort(c.reshape(B, S), axis=1) == np.arange(1, S + 1)).all()
    assert (cv[:, 0][invalid.astype(bool)] == 0).all()  # masked root actions are never visited
    depth = np.zeros_like(par)
    for k in range(1, S + 1):
        depth[:, k] = depth[np.arange(B), par[:, k]] + 1
    assert np.array_equal(depth.sum(1), s.depth_sum.cpu().numpy())


@pytest.mark.parametrize("scale", [1.0, 0.1, 5.0])
@pytest.mark.parametrize("B", [1, 9, 128])
def test_recurrent_kernel_launch_shapes_give_the_same_bits(B, scale):
    """Pair mode (two workgroups per root) == one workgroup per root, bit for bit, on every output -- also on
    ill-conditioned inputs (a channel that is nearly constant over the map: min_max_normalize2d divides by its
    range), where round 1's two moment orders differed by up to 8.7e-2."""
    m, mods = _nets(11 + B)
    d, pred = mods[2], mods[1]
    g = torch.Generator().manual_seed(5 + B)
    s = torch.rand(B, 6, 6, 64, generator=g) * scale
    s[:, :, :, 3] = 0.25 + 1e-6 * torch.rand(B, 6, 6, generator=g)  # nearly constant channel
    s = s.cuda()
    a = torch.randint(0, A, (B,), generator=g).cuda()
    d.use_pair_tower = False
    ref = d.hip_recurrent(pred, s, a, SUPPORT)
    assert not d._pair_scratch
    d.use_pair_tower = True
    out = None
    for _ in range(5):
        out = d.hip_recurrent(pred, s, a, SUPPORT)
    torch.cuda.synchronize()
    assert len(d._pair_scratch) == 1 and d.pair_status() == 0
    for k, (x, y) in enumerate(zip(ref, out)):
        assert torch.equal(x, y), (k, float((x - y).abs().max()))
    # two launches deep (the second one on the first one's next state)
    d.use_pair_tower = False
    ref2 = d.hip_recurrent(pred, ref[3], a, SUPPORT)
    d.use_pair_tower = True
    out2 = d.hip_recurrent(pred, out[3], a, SUPPORT)
    for x, y in zip(ref2, out2):
        assert torch.equal(x, y)


def test_rows_of_a_1024_root_act_equal_the_128_root_shard_act():
    """Sharding invariance of config 4: rows 0..127 and 896..1023 of the 1024-root search (one workgroup per
    root) against the 128-root shards run with (global_batch=1024, root_offset) (pair mode).  Root inference
    is evaluated once for the whole batch and sliced (the representation net runs on MIOpen, whose algorithm
    choice may depend on the batch size; the search kernels are what is under test)."""
    Bg, Bs, S = 1024, 128, S_FULL
    m, mods = _nets(2)
    obs = torch.from_numpy(_frames(Bg, seed=4)).cuda()
    pl, v, emb = [], [], []
    for i in range(0, Bg, 128):  # (chunks: activation memory of the 84x84 stages)
        a_, b_, c_ = m._root_inference(None, None, obs[i:i + 128])
        pl.append(a_); v.append(b_); emb.append(c_)
    pl, v, emb = torch.cat(pl), torch.cat(v), torch.cat(emb)
    noise = torch.from_numpy(np.random.default_rng(5).dirichlet([0.3] * A, Bg).astype(F32)).cuda()

    def rec_of(B):
        def rec(action, flat):
            (r, disc, logits, val), ns = m._recurrent_inference(None, None, action, flat.reshape(B, 6, 6, 64))
            return r, disc, logits, val, ns.reshape(B, -1)
        return rec

    full = mx.MuZeroSearch(Bg, mx.SearchConfig(A, S, 2304, tiebreak=True))
    o_full = full.search((pl, v, emb.reshape(Bg, -1)), rec_of(Bg), key=[3, 9], dirichlet_noise=noise, with_tree=True)
    assert not mods[2]._pair_scratch  # 1024 roots: one workgroup per root
    ints = ("node_visits", "parents", "action_from_parent", "children_index", "children_visits")
    for off in (0, Bg - Bs):
        sl = slice(off, off + Bs)
        shard = mx.MuZeroSearch(Bs, mx.SearchConfig(A, S, 2304, tiebreak=True, global_batch=Bg, root_offset=off))
        o = shard.search((pl[sl], v[sl], emb.reshape(Bg, -1)[sl]), rec_of(Bs), key=[3, 9], dirichlet_noise=noise[sl],
                         with_tree=True)
        assert mods[2]._pair_scratch and not mods[2].pair_lost()  # 128 roots: pair mode
        assert torch.equal(o.action, o_full.action[sl]) and torch.equal(o.action_weights, o_full.action_weights[sl])
        for f in o.search_tree._fields:
            x, y = getattr(o.search_tree, f), getattr(o_full.search_tree, f)[sl]
            assert torch.equal(x, y), (off, f, int((x != y).sum()))
        assert all(getattr(o.search_tree, f).dtype == torch.int32 for f in ints)
        shard.close()


def test_recurrent_kernel_accuracy_against_fp64():
    """mzs_resnet_tower with heads (both launch shapes) against an fp64 CPU evaluation of the SAME torch
    modules; MIOpen's fp32 evaluation of them is measured beside it (tools/tower_accuracy.py prints the table
    kept in profiles/r02_tower_accuracy.txt).  A floating-point kernel 24 convolutions and LayerNorms deep.  Bars:
    the HIP kernel's MEAN error is no larger than the fp32 library path's (x 1.1) on every output, its MAX error
    within 3 x the library's (a maximum over 55 000 elements is decided by the one channel min_max_normalize2d
    conditions worst) and within 1e-4 x max(1, max |x|) of fp64.  reward / value: both fp32 paths carry the SAME
    ~1e-4 error against fp64 -- it is the conditioning of the reference's own _inv_scaling formula in fp32
    (muax/utils.py:70-76, sqrt(1 + 0.004 (|x| + 1.001)) - 1 cancels), not a difference between implementations:
    the logits they decode agree with fp64 to 1e-6."""
    m, mods = _nets(21)
    d, pred = mods[2], mods[1]
    g = torch.Generator().manual_seed(9)
    B = 24
    s = torch.rand(B, 6, 6, 64, generator=g)
    a = torch.randint(0, A, (B,), generator=g)
    d64, p64 = copy.deepcopy(d).cpu().double(), copy.deepcopy(pred).cpu().double()
    d64.use_hip_tower = False
    with torch.no_grad():
        r_l, ns64 = d64(s.double(), a)
        v_l, lg64 = p64(ns64)
        r64 = mx.utils.support_to_scalar(torch.softmax(r_l, -1), SUPPORT).flatten()
        v64 = mx.utils.support_to_scalar(torch.softmax(v_l, -1), SUPPORT).flatten()
    ref = (r64, v64, lg64, ns64)
    sc, ac = s.cuda(), a.cuda()
    d.use_hip_tower = False
    (r0, _, lg0, v0), ns0 = m._recurrent_inference(None, None, ac, sc)
    d.use_hip_tower = True
    lib = (r0, v0, lg0, ns0)
    names = ("reward", "value", "prior_logits", "next_state")
    outs = {}
    for pair in (False, True):
        d.use_pair_tower = pair
        outs[pair] = d.hip_recurrent(pred, sc, ac, SUPPORT)
    for x, y in zip(outs[False], outs[True]):
        assert torch.equal(x, y)
    for n, h, l, x in zip(names, outs[True], lib, ref):
        eh, el = (h.double().cpu() - x).abs(), (l.double().cpu() - x).abs()
        mag = float(x.abs().max())
        print(f"{n}: HIP max {float(eh.max()):.2e} mean {float(eh.mean()):.2e} | MIOpen fp32 max {float(el.max()):.2e} "
              f"mean {float(el.mean()):.2e} | max |x| {mag:.2f}")
        assert float(eh.mean()) <= 1.1 * float(el.mean()) + 1e-9, (n, float(eh.mean()), float(el.mean()))
        assert float(eh.max()) <= 3.0 * float(el.max()) + 1e-6, (n, float(eh.max()), float(el.max()))
        assert float(eh.max()) <= 1e-4 * max(1.0, mag), (n, float(eh.max()), mag)


def test_lost_pair_rendezvous_drops_pair_mode_and_repeats_the_search():
    """include/mzsearch.h: a root whose two workgroups lost each other sets its status word and the launch's
    results are invalid.  Force it (status word poisoned: 'once lost, never wait again', so the halves read
    stale messages) and check MuZero.act(): warns, disables pair mode, repeats the search with one workgroup
    per root and returns exactly what the undisturbed search returns."""
    B, S = 16, 12
    m, mods = _nets(31)
    d = mods[2]
    obs = _frames(B, seed=8)
    kw = dict(with_pi=True, with_value=True, obs_from_batch=True, num_simulations=S)
    a0, pi0, v0 = m.act(5, obs, **kw)  # pair mode, undisturbed
    assert len(d._pair_scratch) == 1 and not d.pair_lost()
    scratch = next(iter(d._pair_scratch.values()))
    scratch.view(-1)[-4 * B:].view(-1, 4)[3, 3] = 1  # root 3: lost
    assert d.pair_lost()
    with pytest.warns(RuntimeWarning, match="lost a rendezvous"):
        a1, pi1, v1 = m.act(5, obs, **kw)
    assert mx.nn.ResNetDynamic.use_pair_tower is False and not d._pair_scratch
    assert np.array_equal(a0, a1) and np.array_equal(pi0, pi1) and np.array_equal(v0, v1)
    a2, pi2, v2 = m.act(5, obs, **kw)  # later acts stay on one workgroup per root, silently
    assert np.array_equal(a0, a2) and np.array_equal(pi0, pi2)
    # the same through a captured graph: the graph held pair-mode launches and must be re-captured
    mx.nn.ResNetDynamic.use_pair_tower = True
    mg = mx.MuZero(*mods, capture_graph=True)
    mg.init(0, obs[:1])
    mg._params = m._params
    b0 = mg.act(5, obs, **kw)
    for scratch in d._pair_scratch.values():
        scratch.view(-1)[-4 * B:].view(-1, 4)[0, 3] = 2  # XCC mismatch code
    with pytest.warns(RuntimeWarning, match="lost a rendezvous"):
        b1 = mg.act(5, obs, **kw)
    for x, y in zip(b0, b1):
        assert np.array_equal(x, y)
    assert np.array_equal(b0[1], pi0)


def test_ez_nets_drive_the_stepwise_search_eager_and_captured():
    """The reference's EfficientZero-style nets (muax/nn.py:180-309) as plugin nets: step-wise search with torch
    modules between the tree kernels; with capture_graph=True both the root inference and the simulation loop are
    hipGraphs and give the same search."""
    g = torch.Generator().manual_seed(6)
    mods = (mx.nn.EZRepresentation(32, generator=g), mx.nn.EZPrediction(A, 21, 1.0, generator=g),
            mx.nn.EZDynamic(32, A, 21, 1.0, generator=g))
    obs = _frames(6, seed=12)
    eager, graph = mx.MuZero(*mods), mx.MuZero(*mods, capture_graph=True)
    eager.init(0, obs[:1])
    graph.init(0, obs[:1])
    kw = dict(with_pi=True, with_value=True, obs_from_batch=True, num_simulations=10)
    a1, pi1, v1 = eager.act(3, obs, **kw)
    for _ in range(2):  # second call: pure replay of both graphs
        a2, pi2, v2 = graph.act(3, obs, **kw)
        assert np.array_equal(a1, a2) and np.array_equal(pi1, pi2) and np.allclose(v1, v2, rtol=1e-5, atol=1e-6)
    assert pi1.shape == (6, A) and np.allclose(pi1.sum(1), 1) and len(graph._root_graphs) == 1
    a3, _, _ = eager.act(4, _frames(6, seed=13), **kw)
    assert a3.shape == (6,) and a3.dtype == np.int32


@pytest.mark.parametrize("shape", [(5, 42, 42, 32), (3, 21, 21, 64), (7, 11, 11, 64), (4, 6, 6, 64), (2, 6, 6, 16), (6, 32)])
def test_fused_layernorm_chains_match_the_torch_expressions(shape):
    """mzs_layernorm_act (muax_amd/csrc/mz_norm.cuh) against hk.LayerNorm as the torch modules spell it
    (muax/nn.py:118-148: statistics over the whole sample, scale / offset per channel, eps 1e-5), alone and with what
    follows it in the residual blocks: + LN(projected shortcut) or + identity shortcut, relu.  Also against an fp64
    evaluation: the fused moments are accumulated in fp64, so it is at least as close as the fp32 expression."""
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 3 + 1.5).cuda()
    x2 = (torch.randn(shape, generator=g) * 0.2 - 4).cuda()
    res = torch.randn(shape, generator=g).cuda()
    axis = (-1,) if len(shape) == 2 else (-3, -2, -1)
    ln, ln2 = mx.nn.HkLayerNorm(axis), mx.nn.HkLayerNorm(axis)
    for m, t in ((ln, x), (ln2, x2)):
        m.materialize(t)
        with torch.no_grad():
            m.scale.copy_(torch.rand(shape[-1], generator=g).cuda() + 0.5)
            m.offset.copy_(torch.randn(shape[-1], generator=g).cuda())

    def ref(dtype):
        def one(m, t):
            t = t.to(dtype)
            dims = tuple(range(1, t.dim()))
            mean, var = t.mean(dims, keepdim=True), t.var(dims, keepdim=True, unbiased=False)
            return (t - mean) * torch.rsqrt(var + 1e-5) * m.scale.to(dtype) + m.offset.to(dtype)
        return one

    with torch.no_grad():
        for kw in ({}, {"relu": True}, {"relu": True, "add_ln": (x2, ln2)}, {"relu": True, "residual": res},
                   {"add_ln": (x2, ln2), "residual": res}):
            assert ln.fused_ok(x)
            got = mx.nn.ln_act(x, ln, **kw)
            outs = []
            for dtype in (torch.float32, torch.float64):
                y = ref(dtype)(ln, x)
                if "add_ln" in kw:
                    y = ref(dtype)(ln2, x2) + y
                if "residual" in kw:
                    y = res.to(dtype) + y
                outs.append(torch.relu(y) if kw.get("relu") else y)
            assert torch.allclose(got, outs[0], rtol=1e-5, atol=2e-5), kw
            e_fused, e_torch = (got.double() - outs[1]).abs().max(), (outs[0].double() - outs[1]).abs().max()
            assert e_fused <= 2 * e_torch + 1e-6, (kw, float(e_fused), float(e_torch))
        ln.use_hip = False
        assert torch.equal(mx.nn.ln_act(x, ln, relu=True), torch.relu(ref(torch.float32)(ln, x)))


def test_conv_nets_with_fused_layernorm_equal_the_torch_modules():
    """Root inference of the ResNet and EZ nets (muax/nn.py:180-331) with the fused LayerNorm chains against the same
    modules evaluated with the torch expressions: embeddings and head outputs agree to fp32 rounding through the
    8- / 5-block encoders, and act() returns the same search."""
    g = torch.Generator().manual_seed(21)
    obs = _frames(6, seed=5)
    for mods in ((mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(A, 21, generator=g),
                  mx.nn.ResNetDynamic(A, 21, generator=g)),
                 (mx.nn.EZRepresentation(32, generator=g), mx.nn.EZPrediction(A, 21, 1.0, generator=g),
                  mx.nn.EZDynamic(32, A, 21, 1.0, generator=g))):
        m = mx.MuZero(*mods)
        m.init(0, obs[:1])
        x = torch.as_tensor(obs).cuda()
        with torch.no_grad():
            s1 = mods[0](x)
            v1, p1 = mods[1](s1)
            mx.nn.HkLayerNorm.use_hip = False
            try:
                s0 = mods[0](x)
                v0, p0 = mods[1](s0)
            finally:
                mx.nn.HkLayerNorm.use_hip = True
        assert torch.allclose(s1, s0, rtol=1e-4, atol=1e-4), float((s1 - s0).abs().max())
        assert torch.allclose(v1, v0, rtol=1e-3, atol=1e-3) and torch.allclose(p1, p0, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("C,B", [(32, 1), (32, 9), (32, 128), (64, 5), (64, 40)])
def test_ez_recurrent_kernel_matches_the_torch_modules(C, B):
    """mzs_ez_recurrent (muax_amd/csrc/mz_ez.cuh): the whole recurrent_fn of the EZ nets (EZDynamic + EZPrediction with
    pre-activation blocks, muax/nn.py:151-178,221-309; decodes of muax/model.py:273-274) in one launch against the torch
    modules it replaces (MIOpen convolutions, the LayerNorm expressions), LayerNorm parameters and biases away from
    their init values.  fp32 kernel on fp32 modules: 3e-4, like the ResNet recurrent kernel."""
    g = torch.Generator().manual_seed(100 + C + B)
    rep, pred, dy = (mx.nn.EZRepresentation(C, generator=g), mx.nn.EZPrediction(A, 21, 1.0, generator=g),
                     mx.nn.EZDynamic(C, A, 21, 1.0, generator=g))
    m = mx.MuZero(rep, pred, dy)
    m.init(0, np.zeros((1, 84, 84, 4), F32))
    with torch.no_grad():
        for mod in (pred, dy):
            for p in mod.parameters():
                if p.dim() == 1:
                    p.add_(0.2 * torch.randn(p.shape, generator=g).to(p.device))
    m.weights_changed()
    s = (torch.randn(B, 6, 6, C, generator=g) * 0.7).cuda()
    a = torch.randint(0, A, (B,), generator=g).cuda()
    got = dy.hip_recurrent(pred, s, a, SUPPORT)
    assert got is not None
    with torch.no_grad():
        mx.nn.HkLayerNorm.use_hip = False
        try:
            r_logits, ns = dy(s, a)
            v_logits, pi = pred(ns)
        finally:
            mx.nn.HkLayerNorm.use_hip = True
        rew = mx.utils.support_to_scalar(torch.softmax(r_logits, -1), SUPPORT).flatten()
        val = mx.utils.support_to_scalar(torch.softmax(v_logits, -1), SUPPORT).flatten()
    assert torch.allclose(got[3], ns, rtol=1e-4, atol=3e-4), float((got[3] - ns).abs().max())
    assert torch.allclose(got[2], pi, rtol=1e-4, atol=3e-4), float((got[2] - pi).abs().max())
    assert torch.allclose(got[0], rew, rtol=1e-3, atol=3e-4), float((got[0] - rew).abs().max())
    assert torch.allclose(got[1], val, rtol=1e-3, atol=3e-4), float((got[1] - val).abs().max())
    # through MuZero._recurrent_inference: the one-launch route is the one that runs
    (r2, d2, l2, v2), n2 = m._recurrent_inference(m.params, None, a, s)
    assert torch.equal(n2, got[3]) and torch.equal(l2, got[2]) and float(d2[0]) == pytest.approx(0.99)
    dy.use_hip_recurrent = False
    (r3, _, l3, v3), n3 = m._recurrent_inference(m.params, None, a, s)
    assert torch.allclose(n3, n2, rtol=1e-4, atol=3e-4) and torch.allclose(r3, r2, rtol=1e-3, atol=3e-4)


def test_fused_layernorm_in_place():
    """include/mzsearch.h: `y` may alias x, x2 or the residual (every element is read and written by the same thread,
    after the moments launch): the in-place call gives the bits of the out-of-place one."""
    import ctypes as C

    from muax_amd import _lib
    L = _lib.load()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 11, 11, 64, generator=g).cuda()
    res = torch.randn(5, 11, 11, 64, generator=g).cuda()
    ln = mx.nn.HkLayerNorm()
    ln.materialize(x)
    with torch.no_grad():
        ln.offset.add_(0.3)
        want = mx.nn.ln_act(x, ln, relu=True, residual=res)
    for target in ("x", "residual"):
        xx, rr = x.clone(), res.clone()
        a = _lib.MzsLayerNormArgs()
        a.struct_size = C.sizeof(_lib.MzsLayerNormArgs)
        a.device, a.batch, a.n, a.channels, a.relu, a.eps = 0, 5, 11 * 11 * 64, 64, 1, 1e-5
        a.x, a.scale, a.offset, a.residual = xx.data_ptr(), ln.scale.data_ptr(), ln.offset.data_ptr(), rr.data_ptr()
        out = xx if target == "x" else rr
        a.y = out.data_ptr()
        ws = torch.empty(L.mzs_layernorm_workspace_bytes(5, a.n) // 8, dtype=torch.float64, device="cuda")
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 8
        _lib.check(L.mzs_layernorm_act(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        assert torch.equal(out, want), target


@pytest.mark.parametrize("B,S,policy,max_depth", [(128, S_FULL, "muzero", None), (9, 40, "muzero", None),
                                                  (160, 40, "muzero", None), (33, 48, "muzero", 6),
                                                  (16, 40, "gumbel", None), (130, 24, "gumbel", None)])
def test_one_launch_search_equals_the_per_simulation_launches(B, S, policy, max_depth):
    """mzs_resnet_search -- the whole simulation loop as ONE launch, every root advanced by its own workgroup(s)
    (pair mode up to 128 roots, one workgroup per root above), next states written straight into the tree's rows --
    against the loop of per-simulation launches it replaces (recurrent kernel + mzs_expand_backup_select): every tree
    array, the actions, the weights and the depth sums, bit for bit; both policies, invalid root actions, and a
    `max_depth` cut (simulations that RE-expand an existing node and overwrite its row)."""
    m, mods = _nets(31 + B)
    dy, pred = mods[2], mods[1]
    obs = torch.from_numpy(_frames(B, seed=B)).cuda()
    pl, v, emb = m._root_inference(None, None, obs)
    rng = np.random.default_rng(B)
    noise = torch.from_numpy(rng.dirichlet([0.3] * A, B).astype(F32)).cuda()
    invalid = (rng.uniform(size=(B, A)) < 0.1).astype(np.uint8)
    invalid[np.arange(B), rng.integers(0, A, B)] = 0
    invalid = torch.from_numpy(invalid).cuda()

    def rec(action, flat):
        (r, disc, logits, val), ns = m._recurrent_inference(None, None, action, flat.reshape(B, 6, 6, 64))
        return r, disc, logits, val, ns.reshape(B, -1)

    def native(handle, b, e):
        dy.hip_search(pred, handle, SUPPORT, 0.99, b, e)

    assert dy.hip_search_ok(pred, (6, 6, 64), SUPPORT)
    outs = []
    for loop in (None, native):
        cfg = mx.SearchConfig(A, S, 2304, tiebreak=policy == "muzero", policy=policy, max_depth=max_depth,
                              qtransform="qtransform_by_parent_and_siblings")
        s = mx.MuZeroSearch(B, cfg)
        kw = dict(dirichlet_noise=noise) if policy == "muzero" else {}
        o = s.search((pl, v, emb.reshape(B, -1)), rec, key=[5, B], invalid_actions=invalid, with_tree=True,
                     native_loop=loop, **kw)
        torch.cuda.synchronize()
        outs.append((o.action.clone(), o.action_weights.clone(), s.depth_sum.clone(),
                     {f: getattr(o.search_tree, f).clone() for f in o.search_tree._fields}))
        s.close()
    if B <= 128:
        assert dy._pair_scratch and not dy.pair_lost()
    (a0, w0, d0, t0), (a1, w1, d1, t1) = outs
    assert torch.equal(a0, a1) and torch.equal(w0, w1) and torch.equal(d0, d1)
    for f in t0:
        assert torch.equal(t0[f], t1[f]), (f, int((t0[f] != t1[f]).sum()))
    if max_depth:
        assert int(t1["node_visits"][:, 1:].max()) > 1 and int((t1["parents"][:, 1:] == -1).sum()) > 0  # re-expansions happened


@pytest.mark.parametrize("A_,B,S,lds_tree", [(6, 20, 50, "1"), (6, 150, 30, "1"), (16, 9, 40, "1"), (18, 12, 60, "0"), (6, 20, 50, "0"),
                                            (32, 7, 30, "1"), (33, 7, 30, "1")])
def test_one_launch_search_tree_in_lds_for_any_action_count(monkeypatch, A_, B, S, lds_tree):
    """Round 6: the one-launch search keeps the statistics its tree step reads and rewrites in LDS (TreeView) and
    specialises the decision refresh on the 16-lane slots the action count fills -- one slot (6, 16 actions), two (18, 32),
    the general code with the HBM tree beyond 32 actions or with MZS_SEARCH_LDS_TREE=0 -- in pair mode and with one
    workgroup per root (150 roots).  Each against the loop of per-simulation launches (HBM tree, general code): every tree
    array, actions, weights, depth sums, bit for bit."""
    monkeypatch.setenv("MZS_SEARCH_LDS_TREE", lds_tree)
    m, mods = _nets(70 + A_, A=A_)
    dy, pred = mods[2], mods[1]
    obs = torch.from_numpy(_frames(B, seed=B + A_)).cuda()
    pl, v, emb = m._root_inference(None, None, obs)
    rng = np.random.default_rng(B + A_)
    noise = torch.from_numpy(rng.dirichlet([0.3] * A_, B).astype(F32)).cuda()
    invalid = (rng.uniform(size=(B, A_)) < 0.15).astype(np.uint8)
    invalid[np.arange(B), rng.integers(0, A_, B)] = 0
    invalid = torch.from_numpy(invalid).cuda()

    def rec(action, flat):
        (r, disc, logits, val), ns = m._recurrent_inference(None, None, action, flat.reshape(B, 6, 6, 64))
        return r, disc, logits, val, ns.reshape(B, -1)

    def native(handle, b, e):
        dy.hip_search(pred, handle, SUPPORT, 0.99, b, e)

    assert dy.hip_search_ok(pred, (6, 6, 64), SUPPORT)
    outs = []
    for loop in (None, native):
        s = mx.MuZeroSearch(B, mx.SearchConfig(A_, S, 2304, tiebreak=True))
        o = s.search((pl, v, emb.reshape(B, -1)), rec, key=[9, B], invalid_actions=invalid, with_tree=True,
                     native_loop=loop, dirichlet_noise=noise)
        torch.cuda.synchronize()
        outs.append((o.action.clone(), o.action_weights.clone(), s.depth_sum.clone(),
                     {f: getattr(o.search_tree, f).clone() for f in o.search_tree._fields}))
        s.close()
    (a0, w0, d0, t0), (a1, w1, d1, t1) = outs
    assert torch.equal(a0, a1) and torch.equal(w0, w1) and torch.equal(d0, d1)
    for f in t0:
        assert torch.equal(t0[f], t1[f]), (f, int((t0[f] != t1[f]).sum()))


def test_one_launch_search_on_paths_deeper_than_a_wavefront():
    """The bench's own nets (_bench_nets: trees ~44 levels deep on average): paths of 64 and more levels take the return chain
    in chunks of 63 levels, and pair mode runs that chain on its second wavefront beside the expansion with the last
    edge's reward / discount patched in from registers (mz_step_jump.cuh, return_chain).  A path of EXACTLY 64 levels
    puts the last edge alone in the upper chunk, where lane 63 of the lower chunk has the same level index: the case a
    first version of the patch got wrong with every other test green.  Pair mode against the loop of per-simulation
    launches (no prefetch, chain after the expansion) and against the one launch with one workgroup per root: every tree
    array, actions, weights, depth sums, bit for bit."""
    B, S = 128, S_FULL
    m, mods, obs = _bench_nets(B)
    dy, pred = mods[2], mods[1]
    pl, v, emb = m._root_inference(None, None, obs)
    noise = torch.from_numpy(np.random.default_rng(7).dirichlet([0.3] * A, B).astype(F32)).cuda()

    def rec(action, flat):
        (r, disc, logits, val), ns = m._recurrent_inference(None, None, action, flat.reshape(B, 6, 6, 64))
        return r, disc, logits, val, ns.reshape(B, -1)

    def native(handle, b, e):
        dy.hip_search(pred, handle, SUPPORT, 0.99, b, e)

    assert dy.hip_search_ok(pred, (6, 6, 64), SUPPORT)
    outs = []
    for loop, pair in ((None, True), (native, True), (native, False)):  # the last: the one launch with one workgroup per root
        dy.use_pair_tower = pair
        if not pair:
            dy._pair_scratch.clear()
        s = mx.MuZeroSearch(B, mx.SearchConfig(A, S, 2304, tiebreak=True))
        o = s.search((pl, v, emb.reshape(B, -1)), rec, key=[12, B], with_tree=True, native_loop=loop, dirichlet_noise=noise)
        torch.cuda.synchronize()
        if loop is not None:
            assert not getattr(s, "_native_loop_unusable", False)
            assert bool(dy._pair_scratch) == pair and not dy.pair_lost()
        outs.append((o.action.clone(), o.action_weights.clone(), s.depth_sum.clone(),
                     {f: getattr(o.search_tree, f).clone() for f in o.search_tree._fields}))
        s.close()
    dy.__dict__.pop("use_pair_tower", None)
    (a0, w0, d0, t0), (a1, w1, d1, t1), (a2, w2, d2, t2) = outs
    assert torch.equal(d1, d2) and torch.equal(a1, a2) and torch.equal(w1, w2)
    for f in t1:
        assert torch.equal(t1[f], t2[f]), ("one workgroup per root", f, int((t1[f] != t2[f]).sum()))
    par = t0["parents"].cpu().numpy()
    depth = np.zeros_like(par)
    for k in range(1, S + 1):
        depth[:, k] = depth[np.arange(B), par[:, k]] + 1
    assert (depth == 64).any() and depth.max() > 64, depth.max()  # the workload reaches the case this test is for
    assert torch.equal(d0, d1), int((d0 != d1).sum())
    for f in t0:
        assert torch.equal(t0[f], t1[f]), (f, int((t0[f] != t1[f]).sum()))
    assert torch.equal(a0, a1) and torch.equal(w0, w1)


@pytest.mark.parametrize("B,S,deep", [(128, S_FULL, False), (144, 60, False), (128, S_FULL, True)])
def test_one_launch_search_replayed_in_the_oracle(oracle, B, S, deep):
    """mz_resnet_search_kernel (the bench line's dominant config-4 kernel) against the ORACLE directly, not through
    the per-simulation HIP route: the tree the one launch leaves is replayed -- for every simulation the oracle's own
    `step_select` (its tree, the same simulation key) must name the (parent, action) the kernel stored for node
    sim + 1, then the oracle's `step_expand_backup` is fed that node's own stored reward / prior logits / raw value /
    embedding -- and at the end every tree array, the sampled actions and the weights must be equal, bit for bit.
    128 roots x 200 simulations x 18 actions is config 4's shard (pair mode); 144 roots take one workgroup per root.
    A regression in the tree-step body the two HIP routes share (mz_step_jump.cuh) fails HERE.
    `deep`: the bench's own nets and frames, whose paths pass 64 levels (the chain's chunks, pointer jumping's LDS form)."""
    if deep:
        m, mods, obs = _bench_nets(B)
    else:
        m, mods = _nets(5)
        obs = torch.from_numpy(_frames(B, seed=40 + B)).cuda()
    dy, pred = mods[2], mods[1]
    pl, v, emb = m._root_inference(None, None, obs)
    E = 2304
    rng = np.random.default_rng(B + S)
    noise = rng.dirichlet([0.3] * A, B).astype(F32)
    invalid = (rng.uniform(size=(B, A)) < 0.1).astype(np.uint8)
    invalid[np.arange(B), rng.integers(0, A, B)] = 0
    key = [3, 1000 + B]
    assert dy.hip_search_ok(pred, (6, 6, 64), SUPPORT)

    def native(handle, b, e):
        dy.hip_search(pred, handle, SUPPORT, 0.99, b, e)

    s = mx.MuZeroSearch(B, mx.SearchConfig(A, S, E, tiebreak=True))
    out = s.search((pl, v, emb.reshape(B, -1)), None, key=key, invalid_actions=torch.from_numpy(invalid).cuda(),
                   dirichlet_noise=torch.from_numpy(noise).cuda(), with_tree=True, native_loop=native)
    torch.cuda.synchronize()
    assert not getattr(s, "_native_loop_unusable", False)  # the one launch really ran
    if B <= 128:
        assert dy._pair_scratch and not dy.pair_lost()
    t = out.search_tree
    par, afp, rew, logit, raw, embs = (getattr(t, f).cpu().numpy() for f in
                                       ("parents", "action_from_parent", "children_rewards", "children_prior_logits",
                                        "raw_values", "embeddings"))
    tree = oracle.Tree(B, S + 1, A, E)
    cfg = oracle.SearchCfg(S, tiebreak=1)
    oracle.tree_init(tree, oracle.root_prior(pl.cpu().numpy(), noise, 0.25, invalid), v.cpu().numpy(),
                     emb.reshape(B, -1).cpu().numpy(), invalid)
    k_sample, _, sims = oracle.sim_keys_from_act_key(key, S)
    rows = np.arange(B)
    disc = np.full(B, 0.99, F32)
    for sim in range(S):
        n = sim + 1
        p_ref, a_ref, _ = oracle.step_select(tree, cfg, sim, sims[sim])
        assert np.array_equal(p_ref, par[:, n]) and np.array_equal(a_ref, afp[:, n]), sim
        oracle.step_expand_backup(tree, sim, p_ref, a_ref, rew[rows, p_ref, a_ref], disc, logit[:, n], raw[:, n], embs[:, n])
    g = oracle.gumbel(k_sample, B * A).reshape(B, A)
    a_ref, w_ref = oracle.summary_sample(tree, 1.0, g)
    assert np.array_equal(a_ref, out.action.cpu().numpy())
    assert np.array_equal(w_ref, out.action_weights.cpu().numpy())
    assert_trees_equal(tree, out.search_tree, exact_floats=True)
    depth = np.zeros_like(par)
    for k in range(1, S + 1):
        depth[:, k] = depth[rows, par[:, k]] + 1
    assert np.array_equal(depth.sum(1), s.depth_sum.cpu().numpy())
    assert not deep or ((depth == 64).any() and depth.max() > 64)
    s.close()


def test_one_launch_search_in_two_halves_and_through_act():
    """[0, S/2) and [S/2, S) as two launches == one launch (the second continues from the selection the first one's
    tail left in the handle), and MuZero.act() takes the one-launch route by itself: same actions, weights and
    values as with MZS_RESNET_SEARCH=0 (the per-simulation launches), batched NumPy round trip."""
    B, S = 24, 32
    m, mods = _nets(77)
    dy, pred = mods[2], mods[1]
    obs_np = _frames(B, seed=8)
    obs = torch.from_numpy(obs_np).cuda()
    pl, v, emb = m._root_inference(None, None, obs)
    noise = torch.from_numpy(np.random.default_rng(1).dirichlet([0.3] * A, B).astype(F32)).cuda()
    trees = []
    for parts in ((S,), (S // 2, S)):
        def native(handle, b, e):
            lo = b
            for hi in parts:
                dy.hip_search(pred, handle, SUPPORT, 0.99, lo, hi)
                lo = hi
        s = mx.MuZeroSearch(B, mx.SearchConfig(A, S, 2304, tiebreak=True))
        o = s.search((pl, v, emb.reshape(B, -1)), None, key=[1, 2], dirichlet_noise=noise, with_tree=True, native_loop=native)
        torch.cuda.synchronize()
        trees.append({f: getattr(o.search_tree, f).clone() for f in o.search_tree._fields})
        s.close()
    for f in trees[0]:
        assert torch.equal(trees[0][f], trees[1][f]), f
    got = m.act(9, obs_np, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=S)
    import os
    os.environ["MZS_RESNET_SEARCH"] = "0"
    try:
        m2, _ = _nets(77)
        want = m2.act(9, obs_np, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=S)
    finally:
        del os.environ["MZS_RESNET_SEARCH"]
    for x, y in zip(got, want):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("C,H,W,B", [(64, 21, 21, 9), (64, 11, 11, 16), (64, 6, 6, 5), (32, 21, 21, 7), (32, 11, 11, 3), (32, 6, 6, 130),
                                     (64, 13, 29, 2), (32, 32, 32, 2), (64, 1, 1, 3), (16, 42, 42, 6), (16, 11, 5, 3), (16, 1, 1, 2)])
def test_representation_conv3x3_against_fp64_and_the_library(C, H, W, B):
    """mzs_conv3x3_nhwc (the C -> C 3x3 stride-1 convolutions of the representation nets' residual blocks at their 21 x 21
    / 11 x 11 / 6 x 6 stages: muax/nn.py:118-178 inside :180-207, :291-310) against an fp64 evaluation of the same
    hk.Conv2D(SAME, no bias), with MIOpen's fp32 result beside it: a floating-point kernel, 9 C terms per output.  Bars:
    within 2e-6 x sqrt(9 C) x max|y| of fp64 and no further from it than 2 x the library's fp32 result + that floor;
    odd sizes (runs of pixels ending mid-row, a 1 x 1 map) included."""
    g = torch.Generator().manual_seed(C + H + W)
    conv = mx.nn.HkConv2D(C, 3, 1, in_channels=C, generator=g).cuda()
    x = (torch.rand(B, H, W, C, generator=g) * 2 - 1).cuda()
    with torch.no_grad():
        assert conv._hip_ok(x)
        y = conv(x)
        conv.use_hip = False
        y_lib = conv(x)
        conv.use_hip = True
        y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), conv.w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    e_hip, e_lib = float((y.double() - y64).abs().max()), float((y_lib.double() - y64).abs().max())
    floor = 2e-6 * (9 * C) ** 0.5 * float(y64.abs().max())
    assert y.shape == x.shape and e_hip <= floor and e_hip <= 2 * e_lib + floor, (e_hip, e_lib, floor)


@pytest.mark.parametrize("cin,cout,H,W,B,div", [(4, 32, 84, 84, 5, 255.0), (32, 64, 42, 42, 7, None), (4, 32, 96, 96, 2, 255.0),
                                                (32, 64, 21, 21, 3, None), (4, 32, 13, 29, 4, None), (32, 64, 2, 2, 3, None),
                                                (4, 32, 1, 1, 2, 255.0),
                                                # round 5: the EZ encoder at embedding_dim 32 (muax/nn.py:189: 4 -> 16; :196: 16 -> 32)
                                                (4, 16, 84, 84, 5, 255.0), (16, 32, 42, 42, 6, None), (4, 16, 13, 29, 3, None),
                                                (16, 32, 7, 9, 2, None)])
def test_stem_conv3x3_stride2_against_fp64_and_the_library(cin, cout, H, W, B, div):
    """mzs_conv3x3_stride2_nhwc (the stems of the representation nets: hk.Conv2D(32 | 64, 3, stride=2, 'SAME', no bias) on
    raw frame stacks / on the 32-channel map, muax/nn.py:189,299,303, with the observations / 255 in front and the relu
    behind fused) against an fp64 evaluation of [relu](conv(x / 255)) and the module's library path: odd sizes (haiku's
    SAME padding puts the odd pixel after), a 1 x 1 image."""
    g = torch.Generator().manual_seed(cin + H + W)
    conv = mx.nn.HkConv2D(cout, 3, 2, in_channels=cin, generator=g).cuda()
    x = (torch.rand(B, H, W, cin, generator=g) * (255.0 if div else 2.0) - (0.0 if div else 1.0)).cuda()
    with torch.no_grad():
        assert conv._hip_ok(x)
        y = conv.scaled(x, div, relu=True)
        conv.use_hip = False
        y_lib = conv.scaled(x, div, relu=True)
        conv.use_hip = True
        (ht, hb), (wl, wr) = mx.nn._same_pad(H, 3, 2), mx.nn._same_pad(W, 3, 2)
        xd = (x.double() / div if div else x.double()).permute(0, 3, 1, 2)
        y64 = torch.relu(torch.nn.functional.conv2d(torch.nn.functional.pad(xd, (wl, wr, ht, hb)), conv.w.double().permute(3, 2, 0, 1),
                                                     stride=2)).permute(0, 2, 3, 1)
    e_hip, e_lib = float((y.double() - y64).abs().max()), float((y_lib.double() - y64).abs().max())
    floor = 2e-6 * (9 * cin) ** 0.5 * max(1e-30, float(y64.abs().max()))
    assert y.shape == y64.shape and e_hip <= floor and e_hip <= 2 * e_lib + floor, (e_hip, e_lib, floor)


def test_representation_net_takes_the_hip_convolutions():
    """Root inference of the ResNet nets (muax/model.py:251-263) with its 8 residual blocks (24 C -> C convolutions at
    42 x 42 x 32, 21 x 21 x 64, 11 x 11 x 64 and their 24 LayerNorms) as mzs_resblock_v1 calls, against the same blocks on
    single mzs_conv3x3_nhwc / mzs_layernorm_act calls and against the library's convolutions: embeddings (after
    min_max_normalize2d), prior logits and values agree to 1e-3 of their largest entries (fp32 summation orders through 26
    convolutions and 17 LayerNorms: 2.2e-4 measured)."""
    m, mods = _nets(5)
    obs = torch.from_numpy(_frames(6, seed=3)).cuda()
    m._root_inference(None, None, obs)  # (lazily built layers: the first call takes the module path)
    calls, convs = [], []
    orig, orig_c = mx.nn.ResidualConvBlockV1._forward_hip, mx.nn.HkConv2D._conv_hip
    mx.nn.ResidualConvBlockV1._forward_hip = lambda self, x: (calls.append(tuple(x.shape[1:])), orig(self, x))[1]
    mx.nn.HkConv2D._conv_hip = lambda self, x, **kw: (convs.append(tuple(x.shape[1:])), orig_c(self, x, **kw))[1]
    try:
        pl, v, emb = m._root_inference(None, None, obs)
        assert calls.count((42, 42, 32)) == 2 and calls.count((21, 21, 64)) == 3 and calls.count((11, 11, 64)) == 3 and len(calls) == 8
        assert convs == [(84, 84, 4), (42, 42, 32)]  # the two stride-2 stems; the other 24 convolutions inside the blocks
        mx.nn.ResidualConvBlockV1.use_hip = False
        pl1, v1, emb1 = m._root_inference(None, None, obs)
        assert convs.count((42, 42, 32)) == 8 and convs.count((21, 21, 64)) == 9 and convs.count((11, 11, 64)) == 9 and len(convs) == 28
        mx.nn.HkConv2D.use_hip = False
        pl0, v0, emb0 = m._root_inference(None, None, obs)
    finally:
        mx.nn.ResidualConvBlockV1._forward_hip, mx.nn.HkConv2D._conv_hip = orig, orig_c
        mx.nn.HkConv2D.use_hip = True
        mx.nn.ResidualConvBlockV1.use_hip = True
    for ref in ((pl1, v1, emb1), (pl0, v0, emb0)):
        for a, b in zip((pl, v, emb), ref):
            assert float((a - b).abs().max()) <= 1e-3 * max(1.0, float(b.abs().max()))


def test_root_inference_tail_in_one_launch():
    """mzs_resnet_root_tail (the last AvgPool of ResNetRepresentation, min_max_normalize2d, ResNetPrediction and the value's
    support decode: muax/nn.py:308-341, muax/model.py:251-263) against the modules on the same 11 x 11 x 64 maps: the
    embedding to 2e-6 (a mean of <= 9 values and one division), value and prior logits to 1e-5 of their largest entries; the
    root inference of the model takes it."""
    m, mods = _nets(5)
    obs = torch.from_numpy(_frames(7, seed=4)).cuda()
    m._root_inference(None, None, obs)  # builds the layers
    taken = []
    orig = mx.nn.ResNetRepresentation.hip_root
    mx.nn.ResNetRepresentation.hip_root = lambda self, *a: (lambda out: (taken.append(out is not None), out)[1])(orig(self, *a))
    try:
        pl, v, emb = m._root_inference(None, None, obs)
        assert taken == [True]
        mx.nn.ResNetRepresentation.use_hip_root = False
        pl0, v0, emb0 = m._root_inference(None, None, obs)
        assert taken == [True, False]
    finally:
        mx.nn.ResNetRepresentation.hip_root = orig
        mx.nn.ResNetRepresentation.use_hip_root = True
    assert emb.shape == emb0.shape == (7, 6, 6, 64) and pl.shape == pl0.shape and v.shape == v0.shape
    assert float((emb - emb0).abs().max()) <= 2e-6
    assert float((v - v0).abs().max()) <= 1e-5 * max(1.0, float(v0.abs().max()))
    assert float((pl - pl0).abs().max()) <= 1e-5 * max(1.0, float(pl0.abs().max()))


@pytest.mark.parametrize("C,H,W,B,proj", [(64, 21, 21, 5, True), (64, 11, 11, 9, True), (32, 42, 42, 3, True), (64, 21, 21, 4, False),
                                         (32, 13, 29, 2, True), (64, 10, 10, 130, False)])
def test_residual_block_in_three_launches_against_fp64(C, H, W, B, proj):
    """mzs_resblock_v1 (ResidualConvBlockV1, muax/nn.py:118-148, projected and identity shortcut) against an fp64
    evaluation of the block and against the module path (single convolution / LayerNorm calls): a floating-point kernel --
    within 2e-5 of the fp64 result's largest entry (outputs are O(1): three LayerNorms) and no further from it than twice
    the module path + that floor."""
    g = torch.Generator().manual_seed(C + H + B)
    blk = mx.nn.ResidualConvBlockV1(C, 1, proj, generator=g)
    x = (torch.rand(B, H, W, C, generator=g) * 2 - 1).cuda()
    with torch.no_grad():
        blk.use_hip = False
        blk(x[:1].cpu())  # builds the layers
        blk.cuda()
        for ln in ([blk.ln_0, blk.ln_1] + ([blk.proj_ln] if proj else [])):  # non-trivial scales and offsets
            ln.scale.copy_(torch.rand(C, generator=g) + 0.5)
            ln.offset.copy_(torch.rand(C, generator=g) - 0.5)
        y_mod = blk(x)
        blk.use_hip = True
        assert blk._hip_ok(x)
        y = blk(x)
        mx.nn.HkConv2D.use_hip = mx.nn.HkLayerNorm.use_hip = False
        blk.use_hip = False
        try:
            y64 = blk.double()(x.double())
        finally:
            mx.nn.HkConv2D.use_hip = mx.nn.HkLayerNorm.use_hip = True
    e_hip, e_mod = float((y.double() - y64).abs().max()), float((y_mod.double() - y64).abs().max())
    floor = 2e-5 * max(1.0, float(y64.abs().max()))
    assert y.shape == x.shape and e_hip <= floor and e_hip <= 2 * e_mod + floor, (e_hip, e_mod, floor)


@pytest.mark.parametrize("C,H,W,B", [(32, 42, 42, 3), (64, 21, 21, 5), (64, 11, 11, 9), (64, 6, 6, 128), (32, 13, 29, 2), (64, 10, 10, 130),
                                     (16, 42, 42, 5), (16, 9, 14, 3), (16, 6, 6, 20)])
def test_preactivation_block_in_three_launches_against_fp64(C, H, W, B):
    """mzs_resblock_v2 (ResidualConvBlockV2 with the identity shortcut, muax/nn.py:151-178: the EZ encoder's block,
    :180-207) -- moments of x, conv_0 normalising x on the way in, conv_1 normalising conv_0's outputs on the way in and
    adding x -- against an fp64 evaluation of the block and against the module path (single LayerNorm / convolution calls
    + a framework add): within 2e-5 of the fp64 result's largest entry and no further from it than twice the module path
    + that floor.  Also: the error returns of the entry point."""
    g = torch.Generator().manual_seed(C + H + B)
    blk = mx.nn.ResidualConvBlockV2(C, 1, False, generator=g)
    x = (torch.rand(B, H, W, C, generator=g) * 2 - 1).cuda()
    with torch.no_grad():
        blk.use_hip = False
        blk(x[:1].cpu())  # builds the layers
        blk.cuda()
        for ln in (blk.ln_0, blk.ln_1):  # non-trivial scales and offsets
            ln.scale.copy_(torch.rand(C, generator=g) + 0.5)
            ln.offset.copy_(torch.rand(C, generator=g) - 0.5)
        y_mod = blk(x)
        blk.use_hip = True
        assert blk._hip_ok(x)
        y = blk(x)
        mx.nn.HkConv2D.use_hip = mx.nn.HkLayerNorm.use_hip = False
        blk.use_hip = False
        try:
            y64 = blk.double()(x.double())
        finally:
            mx.nn.HkConv2D.use_hip = mx.nn.HkLayerNorm.use_hip = True
    e_hip, e_mod = float((y.double() - y64).abs().max()), float((y_mod.double() - y64).abs().max())
    floor = 2e-5 * max(1.0, float(y64.abs().max()))
    assert y.shape == x.shape and e_hip <= floor and e_hip <= 2 * e_mod + floor, (e_hip, e_mod, floor)
    if C == 64 and H == 11:
        import ctypes as Ct

        from muax_amd import _lib
        L = _lib.load()
        b = _lib.MzsResblockArgs()
        assert L.mzs_resblock_v2(Ct.byref(b), None) == _lib.MZS_E_INVALID  # struct_size not set
        b.struct_size, b.batch, b.height, b.width, b.channels, b.eps = Ct.sizeof(b), B, H, W, C, 1e-5
        b.x = b.w0 = b.w1 = b.workspace = x.data_ptr()
        b.y = y.data_ptr()
        with pytest.raises((ValueError, RuntimeError), match="scale and offset"):
            _lib.check(L.mzs_resblock_v2(Ct.byref(b), None))
        b.ln0_scale = b.ln0_offset = b.ln1_scale = b.ln1_offset = x.data_ptr()
        b.workspace_bytes = 16
        with pytest.raises((ValueError, RuntimeError), match="workspace too small"):
            _lib.check(L.mzs_resblock_v2(Ct.byref(b), None))
        b.w_proj = x.data_ptr()
        with pytest.raises((ValueError, RuntimeError), match="identity shortcut only"):
            _lib.check(L.mzs_resblock_v2(Ct.byref(b), None))
        assert L.mzs_resblock_v2_workspace_bytes(2, 12, 12, 48) == 0
        assert L.mzs_resblock_v2_workspace_bytes(2, 12, 12, 64) > 2 * 12 * 12 * 64 * 4


def test_c_abi_rejects_bad_representation_arguments():
    """Error behaviour of round 4's root-inference entry points (mzs_conv3x3_nhwc, mzs_conv3x3_stride2_nhwc,
    mzs_resblock_v1, mzs_resnet_root_tail): negative status + message, mapped to ValueError / RuntimeError -- never a launch
    on bad pointers or unsupported shapes."""
    import ctypes as C

    from muax_amd import _lib
    L = _lib.load()
    x = torch.zeros(2, 12, 12, 64, device="cuda")
    a = _lib.MzsConv3x3Args()
    assert L.mzs_conv3x3_nhwc(C.byref(a), None) == _lib.MZS_E_INVALID  # struct_size not set
    a.struct_size, a.batch, a.height, a.width, a.channels = C.sizeof(a), 2, 12, 12, 48
    a.x = a.w_packed = a.y = x.data_ptr()
    with pytest.raises((ValueError, RuntimeError), match="channels must be 16, 32 or 64"):
        _lib.check(L.mzs_conv3x3_nhwc(C.byref(a), None))
    a.channels, a.width = 64, 4096
    with pytest.raises((ValueError, RuntimeError), match="too wide"):
        _lib.check(L.mzs_conv3x3_nhwc(C.byref(a), None))
    s = _lib.MzsConv3x3sArgs()
    assert L.mzs_conv3x3_stride2_nhwc(C.byref(s), None) == _lib.MZS_E_INVALID
    s.struct_size, s.batch, s.height, s.width, s.in_channels, s.out_channels = C.sizeof(s), 2, 12, 12, 8, 32
    s.x = s.w_packed = s.y = x.data_ptr()
    with pytest.raises((ValueError, RuntimeError), match=r"\(16, 32\) or \(32, 64\)"):
        _lib.check(L.mzs_conv3x3_stride2_nhwc(C.byref(s), None))
    b = _lib.MzsResblockArgs()
    assert L.mzs_resblock_v1(C.byref(b), None) == _lib.MZS_E_INVALID
    b.struct_size, b.batch, b.height, b.width, b.channels, b.eps = C.sizeof(b), 2, 12, 12, 64, 1e-5
    b.x = b.w0 = b.w1 = b.y = b.workspace = x.data_ptr()
    with pytest.raises((ValueError, RuntimeError), match="scale and offset"):
        _lib.check(L.mzs_resblock_v1(C.byref(b), None))
    b.ln0_scale = b.ln0_offset = b.ln1_scale = b.ln1_offset = x.data_ptr()
    b.workspace_bytes = 16
    with pytest.raises((ValueError, RuntimeError), match="workspace too small"):
        _lib.check(L.mzs_resblock_v1(C.byref(b), None))
    assert L.mzs_resblock_workspace_bytes(2, 12, 12, 48) == 0 and L.mzs_resblock_workspace_bytes(2, 12, 12, 64) > 3 * 2 * 12 * 12 * 64 * 4
    t = _lib.MzsRootTailArgs()
    assert L.mzs_resnet_root_tail(C.byref(t), None) == _lib.MZS_E_INVALID
    t.struct_size, t.batch, t.height, t.width, t.num_actions, t.support_size = C.sizeof(t), 2, 21, 21, 18, 10
    t.x = t.embedding = t.value = t.prior_logits = x.data_ptr()
    with pytest.raises((ValueError, RuntimeError), match="pooled map must be 6 x 6"):
        _lib.check(L.mzs_resnet_root_tail(C.byref(t), None))
    t.height = t.width = 11
    with pytest.raises((ValueError, RuntimeError), match="11 weight arrays"):
        _lib.check(L.mzs_resnet_root_tail(C.byref(t), None))
