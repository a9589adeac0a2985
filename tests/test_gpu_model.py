"""GPU tests of the host mirror: MuZero.act()'s contract (muax/model.py:82-179), the plugin-net path,
the fit/test inner loops, and the golden fixtures through the C-ABI."""
import os

import numpy as np
import pytest
import torch

import muax_amd as mx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def _model(E=8, A=2, obs_dim=4, seed=0, **kw):
    g = torch.Generator().manual_seed(seed)
    net = mx.nn.MZNetwork(mx.nn.Representation(E, generator=g), mx.nn.Prediction(A, 21, generator=g),
                          mx.nn.Dynamic(E, A, 21, generator=g))
    m = mx.MuZero(net, **kw)
    m.init(mx.prng.PRNGKey(seed), np.zeros((1, obs_dim)))
    return m


@pytest.mark.parametrize("name", ["cartpole_s10", "cartpole_s50", "lunarlander_s50"])
def test_fused_path_reproduces_golden_fixtures(name):
    """Frozen oracle vectors (tests/golden/make_golden.py) through the C-ABI: bit-exact."""
    g = np.load(os.path.join(ROOT, "tests", "golden", f"act_mlp_{name}.npz"))
    B, obs_dim, E, A, S, tb = (int(x) for x in g["meta"])
    s = mx.MuZeroSearch(B, mx.SearchConfig(A, S, E, tiebreak=bool(tb)))
    s.set_mlp_weights({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_")}, obs_dim)
    out = s.act_mlp(torch.from_numpy(g["obs"]), g["key"], dirichlet_noise=torch.from_numpy(g["dirichlet_noise"]),
                    with_tree=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.action.cpu().numpy(), g["action"])
    assert np.array_equal(out.action_weights.cpu().numpy(), g["action_weights"])
    assert np.array_equal(s.root_value.cpu().numpy(), g["root_value"])
    assert np.array_equal(s.depth_sum.cpu().numpy(), g["depth_sum"])
    for f in out.search_tree._fields:
        assert np.array_equal(getattr(out.search_tree, f).cpu().numpy(), g["tree_" + f]), f


def test_fused_gumbel_reproduces_its_golden_fixture():
    """Frozen oracle vectors of the Gumbel MuZero act through the C-ABI (fused MODE 3 kernel): bit-exact."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "act_gumbel_cartpole_s32.npz"))
    B, obs_dim, E, A, S, maxc = (int(x) for x in g["meta"])
    s = mx.MuZeroSearch(B, mx.SearchConfig(A, S, E, policy="gumbel", qtransform="qtransform_completed_by_mix_value",
                                           max_num_considered_actions=maxc, tiebreak=False))
    s.set_mlp_weights({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_")}, obs_dim)
    out = s.act_mlp(torch.from_numpy(g["obs"]), g["key"], with_tree=True)
    torch.cuda.synchronize()
    assert np.array_equal(out.action.cpu().numpy(), g["action"])
    assert np.array_equal(out.action_weights.cpu().numpy(), g["action_weights"])
    assert np.array_equal(s.depth_sum.cpu().numpy(), g["depth_sum"])
    for f in out.search_tree._fields:
        assert np.array_equal(getattr(out.search_tree, f).cpu().numpy(), g["tree_" + f]), f


def test_act_return_conventions():
    """muax/model.py:173-179: unbatched -> (int, [1,A] weights, float); batched -> arrays; flag order."""
    m = _model()
    obs = np.random.default_rng(0).uniform(-1, 1, 4).astype(F32)
    a = m.act(0, obs, num_simulations=10)
    assert isinstance(a, int) and a in (0, 1)
    a, pi, v = m.act(0, obs, with_pi=True, with_value=True, num_simulations=10)
    assert isinstance(a, int) and isinstance(v, float) and pi.shape == (1, 2) and abs(pi.sum() - 1) < 1e-6
    a2, v2 = m.act(0, obs, with_value=True, num_simulations=10)
    a3, pi3 = m.act(0, obs, with_pi=True, num_simulations=10)
    assert (a2, v2) == (a, v) and a3 == a and np.array_equal(pi3, pi)  # same key -> same result
    batch = np.random.default_rng(1).uniform(-1, 1, (33, 4)).astype(F32)
    ab, pib, vb = m.act(1, batch, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=10)
    assert isinstance(ab, np.ndarray) and ab.shape == (33,) and pib.shape == (33, 2) and vb.shape == (33,)
    ad, pid, vd = m.act(1, batch, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=10,
                        device_outputs=True)
    assert ad.is_cuda and np.array_equal(ad.cpu().numpy(), ab) and np.array_equal(vd.cpu().numpy(), vb)
    # root_value is the network's value of the root, not the search value (muax/model.py:243)
    with torch.no_grad():
        s = m.repr_func(torch.from_numpy(batch).cuda())
        vl, _ = m.pred_func(s)
        v_net = mx.utils.support_to_scalar(torch.softmax(vl, -1), 10).cpu().numpy()
    assert np.allclose(vb, v_net, rtol=2e-3, atol=2e-3)
    # the default of act() is num_simulations=5 (muax/model.py:86)
    _, pi5 = m.act(0, obs, with_pi=True)
    assert abs(pi5[0, 0] * 5 - round(pi5[0, 0] * 5)) < 1e-5


def test_root_noise_is_jax_dirichlet_restated_on_the_device(oracle):
    """Default act(key): the Dirichlet root noise is jax.random.dirichlet's sampler restated on the device
    (mzs_dirichlet) from the dirichlet sub-key: bit-equal to the oracle's restatement (rejection loops on the
    threefry key walk included), for any batch size, action count and shard; statistics of Dir(0.3)."""
    from muax_amd.model import _dirichlet
    for B, A, key, alpha in ((1, 2, (0, 42), 0.3), (33, 2, (5, 6), 0.3), (2048, 2, (3, 4), 0.3), (300, 18, (9, 9), 0.3),
                             (64, 4, (1, 1), 1.5), (1000, 64, (2, 7), 0.03),
                             # 1.6 M elements: some exhaust the four speculative lanes of the device sampler (~6e-6 each)
                             (200000, 8, (11, 12), 0.3), (50000, 4, (13, 14), 2.5)):
        d = _dirichlet(key, alpha, (B, A), "cuda")
        ref = oracle.dirichlet(key, alpha, B, A)
        assert d.is_cuda and d.dtype == torch.float32 and np.array_equal(ref, d.cpu().numpy()), (B, A)
        assert torch.allclose(d.sum(1), torch.ones(B, device="cuda"), atol=1e-6) and float(d.min()) >= 0.0
    full = _dirichlet((3, 4), 0.3, (2048, 2), "cuda")
    part = _dirichlet((3, 4), 0.3, (100, 2), "cuda", global_batch=2048, root_offset=700)
    assert torch.equal(part, full[700:800])
    # Dir(0.3, 0.3): symmetric, mass near the corners -- mean 1/2, variance 1/(4 (2 alpha + 1)) = 0.15625
    assert abs(float(full[:, 0].mean()) - 0.5) < 0.03 and abs(float(full[:, 0].var()) - 0.15625) < 0.02
    m = _model()
    batch = np.random.default_rng(4).uniform(-1, 1, (2048, 4)).astype(F32)
    a1, p1, v1 = m.act(7, batch, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=20)
    a2, p2, v2 = m.act(7, batch, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=20)
    a3, p3, _ = m.act(8, batch, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=20)
    assert np.array_equal(a1, a2) and np.array_equal(p1, p2) and np.array_equal(v1, v2)
    assert not np.array_equal(p1, p3)


def _model_with(w, E, A, obs_dim, **kw):
    """A MuZero on the default MLP trio carrying the oracle-side weight dict `w` (haiku layouts)."""
    m = _model(E, A, obs_dim, **kw)
    with torch.no_grad():
        for k, p in mx.nn.mlp_trio_weights(m.network).items():
            p.copy_(torch.from_numpy(w[k]))
    m.weights_changed()
    return m


@pytest.mark.parametrize("B,A,E,obs_dim,S", [(1, 2, 8, 4, 10), (33, 2, 8, 4, 50), (33, 4, 32, 8, 20)])
def test_muzero_act_equals_the_oracle_for_the_same_key(oracle, B, A, E, obs_dim, S):
    """MuZero.act() itself (muax/model.py:82-179), default arguments: Dirichlet root noise from split(key, 3)[1],
    tie-break noise and the categorical's Gumbel from the key -- against the oracle driven with the same key.
    Batched, unbatched (python int / [1, A] / python float) and with an injected noise array."""
    w = oracle.random_mlp_weights(40 + B + A, obs_dim, E, A, 21, bias_scale=0.1)
    m = _model_with(w, E, A, obs_dim)
    rng = np.random.default_rng(B)
    obs = rng.uniform(-1, 1, (B, obs_dim)).astype(F32)
    mlp, cfg = oracle.Mlp(w, obs_dim, E, A, 21), oracle.SearchCfg(S, tiebreak=1)
    key = mx.prng.PRNGKey(1234 + B)
    noise = oracle.dirichlet(oracle.split(key, 3)[1], 0.3, B, A)
    ref = oracle.act_mlp(mlp, cfg, obs, key, noise, 0.25, None, 1.0, None)
    a, pi, v = m.act(key, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=S)
    assert a.dtype == np.int32 and np.array_equal(a, ref["action"]) and np.array_equal(pi, ref["action_weights"])
    assert np.array_equal(v, ref["root_value"])
    # (NumPy observations take the one-call host round trip mzs_act_mlp_host; device tensors the launch-only path)
    ad, pid, vd = m.act(key, torch.from_numpy(obs).cuda(), with_pi=True, with_value=True, obs_from_batch=True,
                        num_simulations=S, device_outputs=True)
    assert ad.is_cuda and np.array_equal(ad.cpu().numpy(), a) and np.array_equal(pid.cpu().numpy(), pi)
    assert np.array_equal(vd.cpu().numpy(), v)
    inv = np.zeros((B, A), np.uint8)
    inv[::2, 0] = 1
    ref_m = oracle.act_mlp(mlp, cfg, obs, key, noise, 0.25, inv, 1.0, None)
    am, pim = m.act(key, obs, with_pi=True, obs_from_batch=True, num_simulations=S, invalid_actions=inv)
    assert np.array_equal(am, ref_m["action"]) and np.array_equal(pim, ref_m["action_weights"]) and (pim[::2, 0] == 0).all()
    # an injected noise array replaces the draw; temperature and the other act() keywords reach the search
    inj = rng.dirichlet([0.3] * A, B).astype(F32)
    ref2 = oracle.act_mlp(mlp, cfg, obs, key, inj, 0.25, None, 0.5, None)
    a2, pi2 = m.act(key, obs, with_pi=True, obs_from_batch=True, num_simulations=S, temperature=0.5, dirichlet_noise=inj)
    assert np.array_equal(a2, ref2["action"]) and np.array_equal(pi2, ref2["action_weights"])
    # unbatched call on the first observation: the batch of one has its own PRNG layout (global batch 1)
    n1 = oracle.dirichlet(oracle.split(key, 3)[1], 0.3, 1, A)
    ref1 = oracle.act_mlp(mlp, cfg, obs[:1], key, n1, 0.25, None, 1.0, None)
    a1, pi1, v1 = m.act(key, obs[0], with_pi=True, with_value=True, num_simulations=S)
    assert isinstance(a1, int) and isinstance(v1, float) and pi1.shape == (1, A)
    assert a1 == int(ref1["action"][0]) and np.array_equal(pi1, ref1["action_weights"]) and F32(v1) == ref1["root_value"][0]


def test_rollout_reproduces_the_fit_loop_trace_fixture():
    """muax_amd.rollout() -- the inner loop of muax.fit (muax/train.py:153-170: one key split
    and one unbatched act() per environment step) -- against the committed oracle trace of 20 CartPole steps at
    num_simulations=10, batch 1 (tests/golden/rollout_cartpole_s10.npz, generator beside it): every sub-key,
    observation, action, policy target and value."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    from cartpole_env import CartPole
    g = np.load(os.path.join(ROOT, "tests", "golden", "rollout_cartpole_s10.npz"))
    seed, env_seed, S, steps, obs_dim, E, A = (int(x) for x in g["meta"])
    m = _model_with({k[2:]: g[k] for k in g.files if k.startswith("w_")}, E, A, obs_dim)
    traj, key = mx.rollout(m, CartPole(seed=env_seed), g["key"], num_simulations=S, temperature=1.0, max_steps=steps)
    assert len(traj) == steps
    for t, (obs, a, r, done, v, pi) in enumerate(traj):
        assert np.array_equal(obs, g["obs"][t]) and a == int(g["a"][t]), t
        assert pi.shape == (1, A) and np.array_equal(pi, g["pi"][t]) and F32(v) == g["v"][t], t
    k = g["key"]
    for _ in range(steps):
        k, _sub = mx.prng.split(k)
    assert np.array_equal(np.asarray(key, np.uint32), np.asarray(k, np.uint32))  # the advanced key
    # greedy evaluation (muax/test.py:25-41): temperature 0 picks the most visited action of the same search
    a0, pi0 = m.act(g["subkey"][0], g["obs"][0], with_pi=True, num_simulations=S, temperature=0.)
    assert np.array_equal(pi0, g["pi"][0]) and a0 == int(np.argmax(pi0[0]))


def test_act_greedy_and_invalid_actions():
    m = _model(A=4, obs_dim=6)
    batch = np.random.default_rng(2).uniform(-1, 1, (64, 6)).astype(F32)
    a, pi = m.act(5, batch, with_pi=True, obs_from_batch=True, num_simulations=20, temperature=0.)
    assert (pi[np.arange(64), a] == pi.max(1)).all()
    inv = np.zeros((64, 4), np.uint8)
    inv[:, 2] = 1
    a, pi = m.act(5, batch, with_pi=True, obs_from_batch=True, num_simulations=20, invalid_actions=inv)
    assert (a != 2).all() and (pi[:, 2] == 0).all()


def test_plugin_nets_go_through_stepwise_path_and_agree_with_fused():
    """Any torch module is a plugin net.  Wrapping the default trio hides it from the fused kernel; the
    step-wise path (torch nets between the HIP kernels) must then produce the same search up to the
    float rounding of torch's own matmul/exp (visit counts may differ only where scores nearly tie)."""
    m = _model(seed=3)

    class Wrap(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner
            for k in ("num_actions", "embedding_dim"):
                if hasattr(inner, k):
                    setattr(self, k, getattr(inner, k))

        def forward(self, *a):
            return self.inner(*a)

    net = m.network
    m2 = mx.MuZero(Wrap(net.representation_fn), Wrap(net.prediction_fn), Wrap(net.dynamic_fn), policy="muzero")
    m2.init(0, np.zeros((1, 4)))
    assert not mx.nn.is_default_mlp_trio(m2.network)
    batch = np.random.default_rng(4).uniform(-1, 1, (128, 4)).astype(F32)
    noise = np.random.default_rng(5).dirichlet([0.3, 0.3], 128).astype(F32)
    kw = dict(with_pi=True, with_value=True, obs_from_batch=True, num_simulations=25, dirichlet_noise=noise,
              tiebreak=False, temperature=0.)
    a1, pi1, v1 = m.act(7, batch, **kw)
    a2, pi2, v2 = m2.act(7, batch, **kw)
    assert np.allclose(v1, v2, rtol=2e-3, atol=2e-3)
    same = (pi1 == pi2).all(1)
    assert same.mean() > 0.9 and (a1[same] == a2[same]).all()


def test_custom_plugin_network_with_2d_embedding():
    """A non-MLP plugin (conv-shaped embedding [B, 2, 3, 4], A=5): the embedding is opaque to the search."""
    torch.manual_seed(0)

    class Rep(torch.nn.Module):
        def forward(self, obs):
            return torch.tanh(obs[:, :24]).reshape(-1, 2, 3, 4)

    class Pred(torch.nn.Module):
        num_actions = 5

        def __init__(self):
            super().__init__()
            self.v, self.p = torch.nn.Linear(24, 21), torch.nn.Linear(24, 5)

        def forward(self, s):
            f = s.reshape(s.shape[0], -1)
            return self.v(f), self.p(f)

    class Dyn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.r, self.n = torch.nn.Linear(29, 21), torch.nn.Linear(29, 24)

        def forward(self, s, a):
            x = torch.cat([s.reshape(s.shape[0], -1), torch.nn.functional.one_hot(a.long(), 5).float()], 1)
            return self.r(x), torch.sigmoid(self.n(x)).reshape(-1, 2, 3, 4)

    m = mx.MuZero(Rep(), Pred(), Dyn())
    m.init(0, np.zeros((1, 30)))
    obs = np.random.default_rng(6).normal(size=(40, 30)).astype(F32)
    a, pi, v = m.act(3, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=16)
    assert a.shape == (40,) and pi.shape == (40, 5) and np.allclose(pi.sum(1), 1, atol=1e-6)
    assert (pi * 16 == np.round(pi * 16)).all()
    a1 = m.act(3, obs[0], num_simulations=16)
    assert isinstance(a1, int)


class _ToyEnv:
    """CartPole-shaped toy (4 floats, 2 actions): enough to drive the reference's loops."""

    class spec:
        max_episode_steps = 12

    def __init__(self):
        self.rng = np.random.default_rng(0)

    def reset(self):
        self.t, self.x = 0, self.rng.uniform(-0.05, 0.05, 4).astype(F32)
        return self.x, {}

    def step(self, a):
        self.t += 1
        self.x = (self.x + (0.1 if a else -0.1) * np.array([1, 0.5, -0.5, 1], F32)).astype(F32)
        return self.x, 1.0, abs(self.x[0]) > 0.5, False, {}


def test_fit_and_test_inner_loops():
    """muax/train.py:153-170 and muax/test.py:25-41: one key split per step, act contract, greedy eval."""
    m = _model(seed=9)
    traj, key = mx.rollout(m, _ToyEnv(), mx.prng.PRNGKey(1), num_simulations=8,
                           temperature=mx._temperature_fn(100, 0))
    assert 1 <= len(traj) <= 12
    obs, a, r, done, v, pi = traj[0]
    assert isinstance(a, int) and isinstance(v, float) and pi.shape == (1, 2)
    k = mx.prng.PRNGKey(1)
    for _ in range(len(traj)):
        k = mx.prng.split(k)[0]
    assert np.array_equal(k, key)
    g = mx.test(m, _ToyEnv(), mx.prng.PRNGKey(2), num_simulations=8, num_test_episodes=2)
    assert 1.0 <= g <= 12.0


def test_stepwise_search_reuses_handle_and_errors():
    s = mx.MuZeroSearch(8, mx.SearchConfig(3, 6, 5))
    with pytest.raises(ValueError):
        s.root(torch.zeros(8, 2), torch.zeros(8), torch.zeros(8, 5))  # wrong A
    with pytest.raises(ValueError):
        s.expand_backup(0, torch.zeros(8), torch.zeros(8), torch.zeros(8, 3), torch.zeros(8), torch.zeros(8, 5))
    for rep in range(2):  # the handle's tree storage is reused across searches
        out = s.search((torch.zeros(8, 3), torch.zeros(8), torch.zeros(8, 5)),
                       lambda a, e: (torch.zeros(8), torch.ones(8), torch.zeros(8, 3), torch.zeros(8), e),
                       key=rep, with_tree=True)
        assert int(out.search_tree.node_visits[:, 0].min()) == 7


@pytest.mark.gpu
def test_graph_captured_simulation_loop_equals_eager_loop():
    """capture_graph=True replays the S x (select, recurrent_fn, expand_backup) loop as ONE hipGraph; the
    search must be identical to the eager loop, call after call, with fresh keys/observations each call
    (keys live in device memory / the un-captured root and finish), for both policies."""
    for policy in ("muzero", "gumbel"):
        g = torch.Generator().manual_seed(11)
        mk = lambda: mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),  # noqa: E731
                                     mx.nn.Dynamic(8, 2, 21, generator=g))
        net = mk()

        class Wrap(torch.nn.Module):  # hides the default trio from the fused kernel -> step-wise path
            def __init__(self, inner):
                super().__init__()
                self.inner = inner

            def forward(self, *a):
                return self.inner(*a)

        mods = [Wrap(x) for x in net]
        eager = mx.MuZero(*mods, policy=policy)
        graph = mx.MuZero(*mods, policy=policy, capture_graph=True)
        eager.init(0, np.zeros((1, 4)))
        graph.init(0, np.zeros((1, 4)))
        rng = np.random.default_rng(12)
        for call in range(4):
            obs = rng.uniform(-1, 1, (96, 4)).astype(F32)
            kw = dict(with_pi=True, with_value=True, obs_from_batch=True, num_simulations=20)
            if policy == "muzero":
                kw["dirichlet_noise"] = rng.dirichlet([0.3, 0.3], 96).astype(F32)
            a1, pi1, v1 = eager.act(100 + call, obs, **kw)
            a2, pi2, v2 = graph.act(100 + call, obs, **kw)
            assert (a1 == a2).all() and (pi1 == pi2).all() and (v1 == v2).all(), (policy, call)
        handles = list(graph._policy._handles.values())
        assert len(handles) == 1 and len(handles[0]._graphs) == 1  # captured once, replayed afterwards


@pytest.mark.gpu
def test_resnet_plugin_nets_drive_the_stepwise_search():
    """Conv plugin nets (NHWC embedding [B,h,w,c], A=18): the embedding is opaque to the tree kernels; the
    hipGraph-captured loop reproduces the eager one."""
    g = torch.Generator().manual_seed(0)
    mods = (mx.nn.ResNetRepresentation(4, generator=g), mx.nn.ResNetPrediction(18, 21, 8, generator=g),
            mx.nn.ResNetDynamic(18, 21, output_channels=8, generator=g))
    obs = np.random.default_rng(0).integers(0, 256, (6, 32, 32, 2)).astype(F32)
    out = []
    for cap in (False, True):
        m = mx.MuZero(*mods, capture_graph=cap)
        m.init(0, obs[:1])
        for call in range(2):
            out.append(m.act(5 + call, obs, with_pi=True, with_value=True, obs_from_batch=True, num_simulations=12))
    for (a, pi, v), (a2, pi2, v2) in zip(out[:2], out[2:]):
        assert a.shape == (6,) and pi.shape == (6, 18) and v.shape == (6,)
        assert (pi * 12 == np.round(pi * 12)).all() and np.allclose(pi.sum(1), 1, atol=1e-6)
        assert np.allclose(v, v2, rtol=1e-4, atol=1e-5) and (pi == pi2).mean() > 0.9


@pytest.mark.gpu
def test_fit_runs_end_to_end_on_the_hip_paths(tmp_path):
    """muax.fit's whole loop: HIP search for acting, tracer/buffer on the host, fused HIP training step."""

    class Env(_ToyEnv):
        class observation_space:
            @staticmethod
            def sample():
                return np.zeros(4, F32)

        def reset(self, seed=None):
            return super().reset()

    g = torch.Generator().manual_seed(0)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                          mx.nn.Dynamic(8, 2, 21, generator=g))
    model = mx.MuZero(net, optimizer=mx.optimizers.create_optimizer("adam", 5e-3))
    rows = []
    path = mx.fit(model, env=Env(), test_env=Env(), tracer=mx.PNStep(5, 0.99, 0.5), buffer=mx.TrajectoryReplayBuffer(50),
                  max_episodes=3, test_interval=1, num_test_episodes=2, max_training_steps=1000, num_simulations=8,
                  k_steps=4, buffer_warm_up=4, num_trajectory=8, sample_per_trajectory=2, num_update_per_episode=10,
                  model_save_path=str(tmp_path), metrics=rows)
    assert path is not None and os.path.exists(path) and len(rows) == 3
    assert all(np.isfinite(r["loss"]) and "test_G" in r for r in rows) and rows[-1]["training_step"] == 30
    assert rows[-1]["loss"] < rows[0]["loss"] and model._fused_train is not None  # the HIP training kernel ran
    model.load(path)


@pytest.mark.gpu
def test_hip_resnet_tower_matches_the_torch_modules():
    """mzs_resnet_tower (conv1x1 stem + 8 x ResidualConvBlockV1 + min_max_normalize2d, fp32 MFMA) against
    the torch modules it replaces (MIOpen convolutions + LayerNorm): a floating-point kernel, tolerance
    2e-4 absolute on outputs normalised to [0, 1] (24 convolutions and LayerNorms deep)."""
    g = torch.Generator().manual_seed(3)
    d = mx.nn.ResNetDynamic(18, 21, generator=g)
    s = torch.rand(5, 6, 6, 64, generator=g)
    a = torch.tensor([0, 3, 17, 9, 1])
    with torch.no_grad():
        d(s, a)  # materialise on the host
        for blk in d.ns_blocks:  # non-trivial LayerNorm parameters
            for ln in (blk.proj_ln, blk.ln_0, blk.ln_1):
                ln.scale.add_(0.2 * torch.randn(64, generator=g))
                ln.offset.add_(0.2 * torch.randn(64, generator=g))
        d.cuda()
        s, a = s.cuda(), a.cuda()
        d.use_hip_tower = False
        r_ref, ns_ref = d(s, a)
        d.use_hip_tower = True
        assert d._hip_tower_ok(s)
        r_hip, ns_hip = d(s, a)
    assert torch.equal(r_ref, r_hip) and ns_hip.shape == (5, 6, 6, 64)
    assert float((ns_hip - ns_ref).abs().max()) < 2e-4, float((ns_hip - ns_ref).abs().max())
    assert float(ns_hip.amin()) == 0.0 and float(ns_hip.amax()) == 1.0


@pytest.mark.gpu
def test_hip_resnet_recurrent_fn_matches_the_torch_modules():
    """One launch for recurrent_fn of the ResNet nets (reward head + tower + prediction heads + decodes)
    against the torch modules: reward / value to 2e-4 relative (+1e-4), logits and next state to 3e-4."""
    g = torch.Generator().manual_seed(5)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    m = mx.MuZero(*mods)
    m.init(0, np.zeros((1, 84, 84, 4), F32))
    with torch.no_grad():
        for mod in mods[1:]:  # biases away from zero
            for p in mod.parameters():
                if p.dim() == 1 and p.shape[0] in (16, 18, 21, 64):
                    p.add_(0.1 * torch.randn(p.shape, generator=g).to(p.device))
    s = torch.rand(7, 6, 6, 64, generator=g).cuda()
    a = torch.tensor([0, 17, 3, 9, 12, 1, 5]).cuda()
    hip = mods[2].hip_recurrent(mods[1], s, a, 10)
    assert hip is not None
    (r1, d1, lg1, v1), ns1 = m._recurrent_inference(None, None, a, s)
    mods[2].use_hip_tower = False
    (r0, d0, lg0, v0), ns0 = m._recurrent_inference(None, None, a, s)
    mods[2].use_hip_tower = True
    assert torch.equal(hip[0], r1) and torch.equal(d0, d1)
    for x1, x0, tol in ((r1, r0, 2e-4), (v1, v0, 2e-4), (lg1, lg0, 3e-4), (ns1, ns0, 3e-4)):
        assert x1.shape == x0.shape and float((x1 - x0).abs().max()) <= tol * float(x0.abs().max()) + 1e-4, \
            (float((x1 - x0).abs().max()), float(x0.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 9, 128])
def test_hip_tower_pair_mode_equals_one_workgroup_per_root(B):
    """<= 128 roots: two workgroups per root split the pixels of the map and swap boundary pixels and
    LayerNorm moments through L2 (mz_conv.cuh, pair mode).  Same recurrent_fn as one workgroup per root, bit
    for bit (both launch shapes take the LayerNorm moments per pixel half and merge them the same way);
    launches keep their message numbering across calls (epochs); no root's halves ever lose each other."""
    g = torch.Generator().manual_seed(11 + B)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    m = mx.MuZero(*mods)
    m.init(0, np.zeros((1, 84, 84, 4), F32))
    with torch.no_grad():
        for mod in mods[1:]:
            for p in mod.parameters():
                if p.dim() == 1 and p.shape[0] in (16, 18, 21, 64):
                    p.add_(0.1 * torch.randn(p.shape, generator=g).to(p.device))
    d, pred = mods[2], mods[1]
    s = torch.rand(B, 6, 6, 64, generator=g).cuda()
    a = torch.randint(0, 18, (B,), generator=g).cuda()
    d.use_pair_tower = False
    ref = d.hip_recurrent(pred, s, a, 10)
    assert not d._pair_scratch
    d.use_pair_tower = True
    first = d.hip_recurrent(pred, s, a, 10)
    for _ in range(30):
        out = d.hip_recurrent(pred, s, a, 10)
    torch.cuda.synchronize()
    assert len(d._pair_scratch) == 1 and d.pair_status() == 0
    for x, y, z in zip(ref, out, first):
        assert torch.equal(y, z)  # deterministic across launches
        assert torch.equal(x, y), float((x - y).abs().max())
    # other inputs through the same scratch (the exchange slots hold stale data of the previous launch)
    s2 = torch.rand(B, 6, 6, 64, generator=g).cuda()
    d.use_pair_tower = False
    ref2 = d.hip_recurrent(pred, s2, a, 10)
    d.use_pair_tower = True
    out2 = d.hip_recurrent(pred, s2, a, 10)
    for x, y in zip(ref2, out2):
        assert torch.equal(x, y), float((x - y).abs().max())
    assert d.pair_status() == 0


@pytest.mark.gpu
def test_fit_batched_acts_on_all_environments_with_one_launch_per_step():
    class Env(_ToyEnv):
        class observation_space:
            @staticmethod
            def sample():
                return np.zeros(4, F32)

        def __init__(self, seed):
            self.rng = np.random.default_rng(seed)

    g = torch.Generator().manual_seed(0)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                          mx.nn.Dynamic(8, 2, 21, generator=g))
    model = mx.MuZero(net, optimizer=mx.optimizers.create_optimizer("adam", 5e-3))
    rows = []
    envs = [Env(s) for s in range(24)]
    mx.fit_batched(model, envs, Env(99), tracer_factory=lambda: mx.PNStep(5, 0.99, 0.5), iterations=3, num_simulations=8,
                   k_steps=4, num_trajectory=16, sample_per_trajectory=2, num_update_per_iteration=8, test_interval=2,
                   num_test_episodes=2, metrics=rows)
    assert len(rows) == 3 and all(r["env_steps"] >= 24 * 4 for r in rows) and rows[-1]["training_step"] == 24
    assert "test_G" in rows[0] and "test_G" in rows[2] and all(np.isfinite(r["loss"]) for r in rows)
    trajs, _ = mx.collect_batched(model, envs[:5], [mx.NStep(3, 0.9) for _ in range(5)], mx.prng.PRNGKey(1), 8, 1.0)
    assert len(trajs) == 5 and all(len(t) >= 1 and t.batched_transitions.pi.shape[-2:] == (1, 2) for t in trajs)


@pytest.mark.gpu
def test_captured_graph_follows_weight_updates():
    """The hipGraph of the plugin loop freezes host-side work of recurrent_fn (the ResNet dynamics re-packs its
    convolution weights for the HIP tower): after the weights change the graph must be re-captured."""
    g = torch.Generator().manual_seed(2)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    obs = np.random.default_rng(1).integers(0, 256, (4, 84, 84, 4)).astype(F32)
    eager, graph = mx.MuZero(*mods), mx.MuZero(*mods, capture_graph=True)
    eager.init(0, obs[:1])
    graph.init(0, obs[:1])
    kw = dict(with_pi=True, with_value=True, obs_from_batch=True, num_simulations=6)
    for step in range(2):
        a1, pi1, v1 = eager.act(9, obs, **kw)
        a2, pi2, v2 = graph.act(9, obs, **kw)
        assert (pi1 == pi2).all() and np.allclose(v1, v2, rtol=1e-5), step
        with torch.no_grad():  # a "training step": every tower weight moves
            for p in mods[2].ns_blocks.parameters():
                p.mul_(1.05)
        eager.weights_changed()
        graph.weights_changed()


@pytest.mark.gpu
def test_update_trains_the_resnet_plugin_nets_through_autograd():
    """Plugin nets have no fused training kernel: update() takes the torch autograd route (and refuses
    backend='hip'); acting afterwards uses the refreshed weights in the HIP recurrent kernel."""
    g = torch.Generator().manual_seed(4)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(18, 21, generator=g),
            mx.nn.ResNetDynamic(18, 21, generator=g))
    m = mx.MuZero(*mods, optimizer=mx.optimizers.create_optimizer("adam", 1e-3))
    rng = np.random.default_rng(0)
    obs = rng.integers(0, 256, (3, 2, 84, 84, 4)).astype(F32)
    m.init(0, obs[:1, 0])
    batch = mx.Transition(obs=obs, a=rng.integers(0, 18, (3, 2)), r=rng.uniform(0, 1, (3, 2)).astype(F32),
                          Rn=rng.uniform(0, 5, (3, 2)).astype(F32), pi=rng.dirichlet(np.ones(18), (3, 2)).astype(F32))
    with pytest.raises(ValueError):
        m.update(batch, backend="hip")
    a0, pi0, v0 = m.act(1, obs[:, 0], with_pi=True, with_value=True, obs_from_batch=True, num_simulations=6)
    losses = [m.update(batch)["loss"] for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    a1, pi1, v1 = m.act(1, obs[:, 0], with_pi=True, with_value=True, obs_from_batch=True, num_simulations=6)
    assert not np.array_equal(v0, v1)  # the search sees the updated weights
