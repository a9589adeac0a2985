"""CPU tests: host-side logic, the C-ABI library's exports, golden fixtures vs the oracle."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

import muax_amd as mx
from muax_amd import _build, _lib
from oracle import mz_numpy as mn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


def test_prng_matches_oracle(oracle):
    assert mx.prng.split(mx.prng.PRNGKey(0)).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    for key, n in (([7, 9], 5), ([0xFFFFFFFF, 1], 3), ([123, 456], 2)):
        assert np.array_equal(mx.prng.split(key, n), oracle.split(key, n))
    for size in (1, 2, 3, 8, 9):
        assert np.array_equal(mx.prng.random_bits(np.array([3, 4], np.uint32), size), oracle.random_bits([3, 4], size))
        assert np.array_equal(mx.prng.uniform([3, 4], size), oracle.uniform([3, 4], size))
    assert mx.prng.PRNGKey((5 << 32) | 6).tolist() == [5, 6]
    assert mx.key_words(7) == (0, 7) and mx.key_words(np.array([1, 2], np.uint32)) == (1, 2)
    with pytest.raises(ValueError):
        mx.prng.as_key([1, 2, 3])


def test_abi_library_exports_every_declared_symbol():
    """Every entry point declared in include/mzsearch.h is exported by the built library (no compute)."""
    header = open(os.path.join(ROOT, "include", "mzsearch.h")).read()
    declared = set(re.findall(r"\b(mzs_[a-z0-9_]+)\s*\(", header))
    assert {"mzs_create", "mzs_act_mlp", "mzs_select", "mzs_expand_backup", "mzs_finish"} <= declared
    _build.build()
    lib = ctypes.CDLL(_build.LIB_PATH)
    for sym in declared:
        getattr(lib, sym)
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    lib.mzs_abi_version.restype = ctypes.c_int
    assert lib.mzs_abi_version() == 1
    # struct sizes seen by ctypes == what the C compiler sees (checked by the library itself through
    # struct_size at run time); here: no CPU fallback behind the ABI
    cfg = _lib.MzsConfig()
    cfg.struct_size = ctypes.sizeof(_lib.MzsConfig)
    cfg.batch, cfg.num_actions, cfg.num_simulations, cfg.embed_dim = 4, 2, 5, 8
    h = ctypes.c_void_p()
    lib.mzs_create.argtypes = [ctypes.POINTER(_lib.MzsConfig), ctypes.POINTER(ctypes.c_void_p)]
    lib.mzs_last_error.restype = ctypes.c_char_p
    lib.mzs_last_error.argtypes = [ctypes.c_void_p]
    if not torch.cuda.is_available():
        assert lib.mzs_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.MZS_E_NODEVICE
        assert b"no CPU fallback" in lib.mzs_last_error(None) or b"gfx950" in lib.mzs_last_error(None)
    cfg.struct_size = 3
    assert lib.mzs_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.MZS_E_INVALID


def test_product_never_imports_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "muax_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
            src = open(path, errors="ignore").read()
            assert "oracle" not in src.replace("CPU oracle", "").lower() or "pyoracle" not in src, path
            assert "import oracle" not in src and "from oracle" not in src, path


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        mx.MuZeroSearch(4, mx.SearchConfig(2, 5, 8))
    net = mx.create_muzero_network(mx.nn.Representation, mx.nn.Prediction, mx.nn.Dynamic, 8, 2, 21)
    m = mx.MuZero(net, device="cpu")
    m.init(0, np.zeros((1, 4)))
    with pytest.raises(RuntimeError):
        m.act(0, np.zeros(4))


def test_nn_plugin_surface_and_haiku_layout():
    g = torch.Generator().manual_seed(0)
    rep = mx.nn.Representation(8, generator=g)
    pred = mx.nn.Prediction(2, 21, generator=g)
    dyn = mx.nn.Dynamic(8, 2, 21, generator=g)
    obs = torch.rand(5, 4)
    s = rep(obs)
    assert s.shape == (5, 8) and float(s.min()) == 0.0 and float(s.max()) == 1.0  # min_max_normalize rows
    v, pi = pred(s)
    r, ns = dyn(s, torch.tensor([0, 1, 0, 1, 1]))
    assert v.shape == (5, 21) and pi.shape == (5, 2) and r.shape == (5, 21) and ns.shape == (5, 8)
    net = mx.nn.MZNetwork(rep, pred, dyn)
    assert mx.nn.is_default_mlp_trio(net)
    w = {k: t.detach().numpy() for k, t in mx.nn.mlp_trio_weights(net).items()}
    assert w["repr_w"].shape == (4, 8) and w["dr_w1"].shape == (10, 16) and w["pp_w2"].shape == (16, 2)
    # same nets restated in NumPy from the haiku-layout arrays
    pl, val, emb = mn.root_inference(w, obs.numpy(), 10)
    assert np.allclose(emb, s.detach().numpy(), atol=1e-6) and np.allclose(pl, pi.detach().numpy(), atol=1e-5)
    rr, dd, pl2, v2, ns2 = mn.recurrent_inference(w, np.array([0, 1, 0, 1, 1]), emb, 10, 0.99, 2)
    assert np.allclose(ns2, ns.detach().numpy(), atol=1e-5)
    # haiku default init: TruncatedNormal(1/sqrt(fan_in)) weights, zero biases
    big = mx.nn.HkLinear(64, 256, torch.Generator().manual_seed(1))
    assert float(big.b.abs().max()) == 0 and abs(float(big.w.std()) * 8 - 0.88) < 0.05
    assert float(big.w.abs().max()) <= 2.0 / 8 + 1e-6
    net2 = mx.create_muzero_network(mx.nn.Representation, mx.nn.Prediction, mx.nn.Dynamic, 8, 2, 21)
    assert isinstance(net2, mx.MZNetwork) and net2.representation_fn.embedding_dim == 8


def test_codec_matches_reference_formulas(oracle):
    x = torch.linspace(-30, 30, 41)
    p = mx.utils.scalar_to_support(x, 10)
    assert torch.allclose(p.sum(-1), torch.ones(41)) and p.shape == (41, 21)
    y = mx.utils.support_to_scalar(p, 10)
    assert torch.allclose(y, x, rtol=3e-3, atol=3e-3)
    assert np.allclose(mx.utils._inv_scaling(x).numpy(), oracle.inv_scaling(x.numpy()), rtol=1e-4, atol=1e-3)
    g = torch.ones(3, requires_grad=True)
    (mx.utils.scale_gradient(g, 0.5) * 2).sum().backward()
    assert torch.allclose(g.grad, torch.ones(3))


def test_model_constructor_shapes_and_errors():
    net = mx.create_muzero_network(mx.nn.Representation, mx.nn.Prediction, mx.nn.Dynamic, 8, 2, 21)
    m1 = mx.MuZero(net, device="cpu")
    m2 = mx.MuZero(net.representation_fn, net.prediction_fn, net.dynamic_fn, policy="muzero", device="cpu")
    assert m1.network == m2.network
    with pytest.raises(ValueError):
        mx.MuZero(net.representation_fn, device="cpu")
    assert isinstance(mx.MuZero(net, policy="gumbel", device="cpu")._policy, mx.GumbelMuZeroPolicy)
    with pytest.raises(NotImplementedError):
        mx.MuZero(net, policy="stochastic", device="cpu")
    with pytest.raises(TypeError):
        mx.MuZero(net, policy_class=dict, device="cpu")
    with pytest.raises(ValueError):
        m1.act(0, np.zeros(4))  # init() not called
    params = m1.init(mx.prng.PRNGKey(3), np.zeros((1, 4)))
    assert isinstance(params, mx.MZNetworkParams) and "repr_func.w" in params.representation


def test_temperature_schedule_and_sharding():
    t = mx._temperature_fn
    assert [t(100, s) for s in (0, 49, 50, 74, 75, 100)] == [1.0, 1.0, 0.5, 0.5, 0.25, 0.25]
    for world in (1, 2, 3, 8):
        spans = [mx.shard_roots(4099, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == 4099
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert mx.shard_roots(4096, 8, 3) == (1536, 512)


@pytest.mark.parametrize("name", ["cartpole_s10", "cartpole_s50", "lunarlander_s50"])
def test_oracle_reproduces_golden_fixtures(oracle, name):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"act_mlp_{name}.npz"))
    B, obs_dim, E, A, S, tb = (int(x) for x in g["meta"])
    w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
    out = oracle.act_mlp(oracle.Mlp(w, obs_dim, E, A, 21), oracle.SearchCfg(S, tiebreak=tb), g["obs"],
                         g["key"].tolist(), g["dirichlet_noise"], 0.25)
    assert np.array_equal(out["action"], g["action"]) and np.array_equal(out["action_weights"], g["action_weights"])
    assert np.array_equal(out["root_value"], g["root_value"]) and np.array_equal(out["depth_sum"], g["depth_sum"])
    for k, a in out["tree"].arrays().items():
        assert np.array_equal(a, g["tree_" + k]), k
    # independent NumPy restatement agrees with the frozen integers wherever no argmax was a near tie
    rn = mn.act_mlp(w, g["obs"], S, A, E, dirichlet_noise=g["dirichlet_noise"])
    good = rn["min_margin"] > 1e-4
    assert good.any()
    assert np.array_equal(rn["tree"].children_index[good], g["tree_children_index"][good])
    assert np.array_equal(rn["tree"].children_visits[good], g["tree_children_visits"][good])


def test_reference_checkpoint_reader_on_a_synthetic_file_of_the_believed_layout(tmp_path):
    """muax_amd/checkpoint.py reads `jnp.save`d checkpoints without jax.  No real file exists here, so this
    writes one of the layout the reader assumes -- jax Arrays reduced through jax._src.array._reconstruct_array,
    muax.nn.MZNetworkParams, an optax-like state -- with stand-in modules, removes them, and reads it back."""
    import sys
    import types
    from collections import namedtuple

    import numpy as np

    import muax_amd as mx
    from muax_amd import checkpoint

    fake = {n: types.ModuleType(n) for n in ("jax", "jax._src", "jax._src.array", "muax", "muax.nn", "optax", "optax._src",
                                             "optax._src.transform")}

    class Array:  # pickles like jax's ArrayImpl
        def __init__(self, v):
            self.v = np.asarray(v)

        def __reduce__(self):
            fun, args, state = self.v.__reduce__()
            return fake["jax._src.array"]._reconstruct_array, (fun, args, state, {"weak_type": False})

    def _reconstruct_array(fun, args, arr_state, aval_state):
        raise AssertionError("the real reconstructor must not be needed")

    _reconstruct_array.__module__ = "jax._src.array"
    _reconstruct_array.__qualname__ = "_reconstruct_array"
    fake["jax._src.array"]._reconstruct_array = _reconstruct_array
    Params = namedtuple("MZNetworkParams", "representation prediction dynamic")
    Params.__module__, Params.__qualname__ = "muax.nn", "MZNetworkParams"
    fake["muax.nn"].MZNetworkParams = Params
    State = namedtuple("ScaleByAdamState", "count mu nu")
    State.__module__, State.__qualname__ = "optax._src.transform", "ScaleByAdamState"
    fake["optax._src.transform"].ScaleByAdamState = State

    rng = np.random.default_rng(0)
    lin = lambda i, o: {"w": Array(rng.normal(size=(i, o)).astype(np.float32)), "b": Array(rng.normal(size=o).astype(np.float32))}  # noqa: E731
    rep = {"representation/~/linear": lin(4, 8)}
    pred = {"prediction/~/linear": lin(8, 16), "prediction/~/linear_1": lin(16, 21),
            "prediction/~/linear_2": lin(8, 16), "prediction/~/linear_3": lin(16, 2)}
    dyn = {"dynamic/~/linear": lin(10, 16), "dynamic/~/linear_1": lin(16, 8),
           "dynamic/~/linear_2": lin(10, 16), "dynamic/~/linear_3": lin(16, 21)}
    path = str(tmp_path / "model_params.npy")
    sys.modules.update(fake)
    try:
        np.save(path, {"params": Params(rep, pred, dyn), "optimizer_state": (State(Array(3), rep, rep),)})
    finally:
        for n in fake:
            sys.modules.pop(n, None)

    saved = checkpoint.read_reference_checkpoint(path)
    assert isinstance(saved["params"], mx.MZNetworkParams)
    assert np.array_equal(saved["params"].prediction["prediction/~/linear_3"]["w"], pred["prediction/~/linear_3"]["w"].v)
    g = torch.Generator().manual_seed(0)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                          mx.nn.Dynamic(8, 2, 21, generator=g))
    m = mx.MuZero(net, device="cpu")
    m.init(0, np.zeros((1, 4)))
    checkpoint.load_reference_params(m, path)
    w = mx.nn.mlp_trio_weights(m.network)
    for ours, theirs in (("repr_w", rep["representation/~/linear"]["w"]), ("pv_w2", pred["prediction/~/linear_1"]["w"]),
                         ("pp_b2", pred["prediction/~/linear_3"]["b"]), ("dn_w1", dyn["dynamic/~/linear"]["w"]),
                         ("dr_b2", dyn["dynamic/~/linear_3"]["b"])):
        assert np.array_equal(w[ours].detach().numpy(), theirs.v), ours
    bad = mx.MuZero(mx.nn.MZNetwork(mx.nn.Representation(16), mx.nn.Prediction(2, 21), mx.nn.Dynamic(16, 2, 21)), device="cpu")
    bad.init(0, np.zeros((1, 4)))
    with pytest.raises(ValueError):
        checkpoint.load_reference_params(bad, path)


def test_reference_checkpoint_reader_resolves_no_global_outside_its_allow_list(tmp_path):
    """A crafted .npy must not reach builtins.eval / os.system / numpy helpers through the Unpickler: every
    global that is not on the allow-list becomes an inert stand-in."""
    import pickle

    from muax_amd import checkpoint

    class Evil:
        def __reduce__(self):
            return (eval, ("__import__('os').environ.__setitem__('MUAX_PWNED', '1')",))

    class Evil2:
        def __reduce__(self):
            return (np.load, ("/nonexistent",))

    path = str(tmp_path / "evil.npy")
    for payload in (Evil(), Evil2()):
        with open(path, "wb") as f:
            np.lib.format.write_array_header_1_0(f, {"descr": "|O", "fortran_order": False, "shape": ()})
            pickle.dump({"params": payload}, f, protocol=2)
        os.environ.pop("MUAX_PWNED", None)
        saved = checkpoint.read_reference_checkpoint(path)
        assert "MUAX_PWNED" not in os.environ
        assert isinstance(saved["params"], checkpoint._Stub)


def test_oracle_reproduces_the_gumbel_golden_fixture(oracle):
    """tests/golden/act_gumbel_cartpole_s32.npz freezes the oracle's Gumbel MuZero act (generator beside it)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    g = np.load(os.path.join(ROOT, "tests", "golden", "act_gumbel_cartpole_s32.npz"))
    B, obs_dim, E, A, S, maxc = (int(x) for x in g["meta"])
    w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
    out = mg.gumbel_act(w, obs_dim, E, A, S, g["obs"], tuple(int(x) for x in g["key"]), 1, maxc)
    assert np.array_equal(out["action"], g["action"]) and np.array_equal(out["action_weights"], g["action_weights"])
    for k, a in out["tree"].arrays().items():
        assert np.array_equal(a, g["tree_" + k]), k
    assert (g["tree_children_visits"][:, 0].sum(-1) == S).all()


def test_oracle_reproduces_the_fit_loop_trace_fixture(oracle):
    """tests/golden/rollout_cartpole_s10.npz (SURVEY.md 8 row a10's pin): 20 CartPole steps of the reference's
    fit() inner loop at num_simulations=10, batch 1, frozen from the oracle with its generator beside it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_rollout_trace",
                                                  os.path.join(ROOT, "tests", "golden", "make_rollout_trace.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    g = np.load(os.path.join(ROOT, "tests", "golden", "rollout_cartpole_s10.npz"))
    seed, env_seed, S, steps = (int(x) for x in g["meta"][:4])
    w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
    rows = mt.oracle_rollout(w, g["key"], mt.CartPole(seed=env_seed), steps, S)
    assert len(rows) == steps == 20 and S == 10
    for f in ("subkey", "obs", "noise", "a", "pi", "v"):
        assert np.array_equal(np.stack([np.asarray(r[f]) for r in rows]), g[f]), f
    # the sub-keys are the reference loop's `key, subkey = jax.random.split(key)` chain (muax/train.py:154)
    from muax_amd import prng
    key = g["key"]
    for t in range(steps):
        key, subkey = prng.split(key)
        assert np.array_equal(np.asarray(subkey, np.uint32), g["subkey"][t])
    assert g["pi"].shape == (20, 1, 2) and np.allclose(g["pi"].sum(-1), 1)  # pi keeps its leading 1 (muax/model.py:176)


def test_reference_checkpoint_reader_accepts_other_plausible_writers(tmp_path):
    """No real muax checkpoint exists here, so the reader is exercised on layouts a different writer could
    plausibly produce (all UNVERIFIED against a real file): params as a plain tuple instead of MZNetworkParams,
    haiku FlatMapping containers, module prefixes other than the constructor names, float64 leaves, plain NumPy
    leaves instead of jax Arrays, and extra top-level keys."""
    import sys
    import types

    import muax_amd as mx
    from muax_amd import checkpoint

    class FlatMapping(dict):
        def __reduce__(self):
            return FlatMapping, (dict(self),)

    fake = types.ModuleType("haiku._src.data_structures")
    FlatMapping.__module__, FlatMapping.__qualname__ = "haiku._src.data_structures", "FlatMapping"
    fake.FlatMapping = FlatMapping
    rng = np.random.default_rng(3)
    lin = lambda i, o, dt=np.float32: {"w": rng.normal(size=(i, o)).astype(dt), "b": rng.normal(size=o).astype(dt)}  # noqa: E731
    rep = FlatMapping({"mz/representation/~/linear": FlatMapping(lin(4, 8, np.float64))})
    pred = FlatMapping({"mz/prediction/~/linear": FlatMapping(lin(8, 16)), "mz/prediction/~/linear_1": FlatMapping(lin(16, 21)),
                        "mz/prediction/~/linear_2": FlatMapping(lin(8, 16)), "mz/prediction/~/linear_3": FlatMapping(lin(16, 2))})
    dyn = {"dynamic/~/linear_3": lin(16, 21), "dynamic/~/linear": lin(10, 16), "dynamic/~/linear_2": lin(10, 16),
           "dynamic/~/linear_1": lin(16, 8)}  # written out of order
    path = str(tmp_path / "other_writer.npy")
    sys.modules["haiku"] = types.ModuleType("haiku")
    sys.modules["haiku._src"] = types.ModuleType("haiku._src")
    sys.modules["haiku._src.data_structures"] = fake
    try:
        np.save(path, {"params": (rep, pred, dyn), "optimizer_state": None, "step": 7})
    finally:
        for n in ("haiku", "haiku._src", "haiku._src.data_structures"):
            sys.modules.pop(n, None)
    g = torch.Generator().manual_seed(0)
    m = mx.MuZero(mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                                  mx.nn.Dynamic(8, 2, 21, generator=g)), device="cpu")
    m.init(0, np.zeros((1, 4)))
    v0 = m._weights_version
    m.load(path)
    w = mx.nn.mlp_trio_weights(m.network)
    assert w["repr_w"].dtype == torch.float32 and np.allclose(w["repr_w"].detach().numpy(), rep["mz/representation/~/linear"]["w"])
    assert np.array_equal(w["pp_w2"].detach().numpy(), pred["mz/prediction/~/linear_3"]["w"])
    assert np.array_equal(w["dn_w2"].detach().numpy(), dyn["dynamic/~/linear_1"]["w"])   # haiku order: ns_func first
    assert np.array_equal(w["dr_b2"].detach().numpy(), dyn["dynamic/~/linear_3"]["b"]) and m._weights_version > v0
    with pytest.raises(ValueError):  # not a pickled-object .npy
        np.save(str(tmp_path / "plain.npy"), np.zeros(3))
        checkpoint.read_reference_checkpoint(str(tmp_path / "plain.npy"))


def test_small_reference_utilities():
    """muax/utils.py:37-68,104-223 and the tracer / buffer base classes (muax/episode_tracer.py:58-111,
    muax/replay_buffer.py:122-147): the diff-transform matrix of the reference's docstring, a hand-computed strided
    2-step return, the slicing deque, the interfaces fit() relies on."""
    from muax_amd import utils
    from muax_amd.episode_tracer import BaseTracer, NStep, PNStep, Transition, flatten_transition_func, unflatten_transition_func
    from muax_amd.replay_buffer import BaseReplayBuffer, TrajectoryReplayBuffer
    assert np.array_equal(utils.diff_transform_matrix(4),
                          np.array([[-1, 0, 0, 0], [3, 1, 0, 0], [-3, -2, -1, 0], [1, 1, 1, 1]], np.float32))
    x = np.arange(8, dtype=np.float32).reshape(2, 4) ** 2  # frames t-3 .. t on the last axis
    d = utils.diff_transform(x)
    assert np.allclose(d[:, 3], x[:, 3]) and np.allclose(d[:, 2], x[:, 3] - x[:, 2])
    assert np.allclose(d[:, 1], x[:, 3] - 2 * x[:, 2] + x[:, 1])
    q = utils.sliceable_deque(range(6), maxlen=6)
    assert list(q[1:4]) == [1, 2, 3] and isinstance(q[1:4], utils.sliceable_deque) and q[2] == 2
    G = utils.n_step_bootstrapped_returns(np.array([1., 2, 3, 4]), np.full(4, 0.9), np.array([10., 20, 30, 40]), 2)
    assert np.allclose(G, [1 + .9 * (2 + .9 * 20), 2 + .9 * (3 + .9 * 30), 3 + .9 * (4 + .9 * 40), 4 + .9 * 40])
    G1 = utils.n_step_bootstrapped_returns(np.array([1., 2, 3]), np.full(3, 0.5), np.array([4., 5, 6]), 1, lambda_t=0.3)
    assert np.allclose(G1, [1 + .5 * 4, 2 + .5 * 5, 3 + .5 * 6])  # n = 1: one bootstrap step whatever lambda
    assert utils.action2plane(np.float32(0.25), (3, 6, 6, 1)).shape == (3, 6, 6, 1)
    assert np.allclose(utils.min_max(np.array([1., 2., 3.]), 1., 3.), [0, .5, 1])
    assert issubclass(NStep, BaseTracer) and issubclass(PNStep, BaseTracer)
    assert issubclass(TrajectoryReplayBuffer, BaseReplayBuffer)
    t = Transition(obs=1, a=2, r=3., done=False, Rn=4., v=5., pi=6., w=1.)
    leaves, _ = flatten_transition_func(t)
    assert unflatten_transition_func(None, list(leaves)) == t


def test_new_entry_points_validate_their_arguments_without_a_gpu():
    """mzs_layernorm_act / mzs_ez_recurrent / mzs_expand_backup_select (round 3): ABI struct sizes as the C compiler sees
    them, argument errors reported before any device is touched, and no CPU fallback behind them."""
    L = _lib.load()
    a = _lib.MzsLayerNormArgs()
    assert L.mzs_layernorm_act(ctypes.byref(a), None) == _lib.MZS_E_INVALID  # struct_size 0: ABI mismatch
    a.struct_size = ctypes.sizeof(_lib.MzsLayerNormArgs)
    a.batch, a.n, a.channels, a.eps = 2, 30, 6, 1e-5
    assert L.mzs_layernorm_act(ctypes.byref(a), None) == _lib.MZS_E_UNSUPPORTED  # n, channels: multiples of 4
    a.n, a.channels = 32, 8
    assert L.mzs_layernorm_act(ctypes.byref(a), None) == _lib.MZS_E_INVALID  # null tensors
    assert L.mzs_layernorm_workspace_bytes(128, 56448) == 2 * 128 * 13 * 2 * 8 and L.mzs_layernorm_workspace_bytes(0, 8) == 0
    x = np.zeros((2, 32), np.float32)
    so = np.ones(8, np.float32)
    ws = np.zeros(64, np.float64)
    a.x = a.y = x.ctypes.data
    a.scale = a.offset = so.ctypes.data
    a.workspace, a.workspace_bytes = ws.ctypes.data, 8
    assert L.mzs_layernorm_act(ctypes.byref(a), None) == _lib.MZS_E_INVALID  # workspace too small
    a.workspace_bytes = ws.nbytes
    if not torch.cuda.is_available():
        assert L.mzs_layernorm_act(ctypes.byref(a), None) == _lib.MZS_E_NODEVICE
    e = _lib.MzsEzArgs()
    assert L.mzs_ez_recurrent(ctypes.byref(e), None) == _lib.MZS_E_INVALID
    e.struct_size = ctypes.sizeof(_lib.MzsEzArgs)
    e.batch, e.channels, e.num_actions, e.support_size = 4, 48, 18, 10
    assert L.mzs_ez_recurrent(ctypes.byref(e), None) == _lib.MZS_E_UNSUPPORTED  # 32 or 64 channels
    e.channels, e.support_size = 32, 40
    assert L.mzs_ez_recurrent(ctypes.byref(e), None) == _lib.MZS_E_UNSUPPORTED  # 2 support + 1 <= 64
    e.support_size = 10
    assert L.mzs_ez_recurrent(ctypes.byref(e), None) == _lib.MZS_E_INVALID  # null tensors
    assert L.mzs_expand_backup_select(None, 0, None, None, None, None, None, None, None, None) == _lib.MZS_E_INVALID
    # the torch expressions stay the CPU route of the LayerNorm chains (training, tests): same values as the modules'
    ln = mx.nn.HkLayerNorm()
    t = torch.randn(3, 6, 6, 8)
    assert not ln.fused_ok(t)
    y = mx.nn.ln_act(t, ln, relu=True, residual=t)
    assert torch.equal(y, torch.relu(t + ln(t)))


def test_bench_refuses_to_run_without_the_devices_it_was_asked_for():
    """`python bench.py --gpus N` on a box with fewer than N ROCm devices (here: none) stops with a message and no JSON
    line -- it never prints a line for fewer ranks than it was asked for, and there is no CPU fallback to measure."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MUAX_BENCH_SINGLE_DEVICE")}
    for n in ("1", "2"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", n, "--steps", "2", "--warmup", "1"],
                             env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and "{" not in out.stdout
        assert "ROCm" in out.stderr


def test_jit_planner_agrees_with_the_kernels_own_limits(tmp_path):
    """muax_amd/_jit.py::plan restates FusedCfg's LDS arithmetic in Python; a disagreement would make an on-demand build
    fail its static_asserts on the user's machine (act() would then fall to the generic route: silent, 5x slower).  The
    plan table's invariants, then hipcc (it cross-compiles without a GPU) on boundary shapes of both record kinds: the
    widest action set on the plain record, the longest search, a LONG instance that just fits four roots, the E = 32 one."""
    import subprocess
    from muax_amd import _jit
    for A in (1, 2, 4, 8, 9, 12, 16):
        for S in (1, 50, 63, 100, 127, 128, 200, 255):
            for E in (8, 32):
                pl = _jit.plan(A, E, 21, S)
                if pl is None:
                    continue
                FS, NMAX, W, LONG = pl
                assert FS == 2 and NMAX >= S + 1 and 1 <= W <= 4 and _jit.lds_bytes(A, E, NMAX, W, LONG) <= 160 * 1024
                assert LONG or NMAX <= 128  # beyond 127 simulations the root paths cannot live in LDS
    assert _jit.plan(17, 8, 21, 50) is None and _jit.plan(2, 8, 21, 256) is None and _jit.plan(2, 8, 65, 50) is None
    assert _jit.plan(2, 8, 21, 255) == (2, 256, 2, True) and _jit.plan(2, 8, 41, 50)[0] == 4
    try:
        cc = _build.hipcc()
    except RuntimeError:
        pytest.skip("no hipcc")
    procs = []
    # (round 6: instances planned for the MuZero policy's modes alone -- four words per child -- hold more roots per
    # workgroup at 160 simulations and for 9 / 16 actions; their LDS arithmetic must pass the same static_asserts)
    assert _jit.plan(2, 8, 21, 160, gumbel=False)[2] == 4 > _jit.plan(2, 8, 21, 160)[2]
    assert _jit.plan(16, 8, 21, 50, gumbel=False)[2] == 2 > _jit.plan(16, 8, 21, 50)[2]
    shapes = [(16, 8, 63, True), (2, 8, 255, True), (6, 8, 160, True), (4, 32, 200, True),
              (2, 8, 160, False), (16, 8, 50, False), (9, 8, 50, False)]
    for k, (A, E, S, gumbel) in enumerate(shapes):
        FS, NMAX, W, LONG = _jit.plan(A, E, 21, S, gumbel)
        assert _jit.lds_bytes(A, E, NMAX, W, LONG, gumbel) <= 160 * 1024
        deff = tmp_path / f"inst{k}.def"
        deff.write_text(f"MZS_INST(100, {A}, {E}, {FS}, {NMAX}, {W}, {'2' if LONG else 'false'})\n")
        cmd = [cc] + _build.FLAGS + [f'-DMZ_INSTANCES_FILE="{deff}"', "-DMZ_FUSED_GROUP=100",
                                     f"-DMZ_FUSED_MUZERO_ONLY={0 if gumbel else 1}", "-shared",
                                     os.path.join(_build.CSRC, "mz_fused_jit.hip"), "-o", str(tmp_path / f"inst{k}.so")]
        procs.append(((A, E, S, gumbel), subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    # ... and one on-demand instance of the training-step kernel (mz_train_jit.hip), with the entry points _jit binds
    cmd = [cc] + _build.FLAGS + ["-DMZ_TRAIN_A=5", "-DMZ_TRAIN_E=12", "-DMZ_TRAIN_F=25", "-shared",
                                 os.path.join(_build.CSRC, "mz_train_jit.hip"), "-o", str(tmp_path / "train.so")]
    procs.append((("train", 5, 12, 25), subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for shape, p in procs:
        _, err = p.communicate(timeout=600)
        assert p.returncode == 0, (shape, err[-1500:])
    side = ctypes.CDLL(str(tmp_path / "train.so"))
    a, e, f = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    side.mzs_jit_train_shape(ctypes.byref(a), ctypes.byref(e), ctypes.byref(f))
    assert (a.value, e.value, f.value) == (5, 12, 25) and side.mzs_jit_train_launch
    L = ctypes.CDLL(_build.LIB_PATH)
    assert side.mzs_jit_train_abi() == L.mzs_train_jit_abi()
    assert not _jit.ensure_train_instance(17, 8, 21) and not _jit.ensure_train_instance(2, 8, 65)



def test_jit_cache_key_names_everything_the_planner_decided(monkeypatch):
    """the cache file of an on-demand instance is named after (A, E, FS, NMAX, W) AND the record kind
    (LONG) AND the policy class, and its hash covers muax_amd/_jit.py (the planner) and PLAN_VERSION: a cache directory
    that survives a planner change cannot serve an instance of another record kind."""
    from muax_amd import _jit
    t = _jit.instance_tag(2, 8, 2, 161, 4, True, False)
    assert "_l1_" in t and "_p0-" in t
    assert t != _jit.instance_tag(2, 8, 2, 161, 4, False, False) and t != _jit.instance_tag(2, 8, 2, 161, 4, True, True)
    f0 = _jit.instance_file(4, 8, 21, 100)
    real = _jit.plan

    def flipped(A, E, F, S, gumbel=True):  # the planner changes its mind about the record kind of this one shape
        pl = real(A, E, F, S, gumbel)
        return (pl[0], pl[1], pl[2], not pl[3]) if (A, S) == (4, 100) and pl else pl

    monkeypatch.setattr(_jit, "plan", flipped)
    f1 = _jit.instance_file(4, 8, 21, 100)
    assert f0 != f1 and f0.replace("_l1_", "_l0_") == f1 or f0.replace("_l0_", "_l1_") == f1
    monkeypatch.setattr(_jit, "plan", real)
    h0 = _jit._source_hash()
    monkeypatch.setattr(_jit, "PLAN_VERSION", _jit.PLAN_VERSION + 1)
    assert _jit._source_hash() != h0
    # a policy-specific instance only where it buys roots per workgroup: otherwise ONE file serves every mode
    assert _jit.instance_file(2, 8, 21, 255, gumbel=False) == _jit.instance_file(2, 8, 21, 255, gumbel=True)
    assert _jit.instance_file(2, 8, 21, 160, gumbel=False) != _jit.instance_file(2, 8, 21, 160, gumbel=True)


def test_jit_failed_build_leaves_a_compiler_log():
    """A failed on-demand build is no longer silent: hipcc's output stays in muax_amd/lib/jit/mzfused_<tag>.log and
    build_log_tail() returns its last lines (MuZero puts them into its warning)."""
    from muax_amd import _jit
    try:
        _build.hipcc()
    except RuntimeError:
        pytest.skip("no hipcc")
    assert not _jit.ensure_instance(3, 12, 21, 20, extra_flags=("-DMZ_FUSED_GROUP=this_is_not_a_number",))
    log = _jit.last_build_log
    assert log and os.path.exists(log) and log.endswith(".log")
    text = open(log).read()
    assert "hipcc" in text.splitlines()[0] and "error" in text
    assert "error" in _jit.build_log_tail(40)
    os.remove(log)
    for f in os.listdir(_jit.JIT_DIR):  # (the one-line instance list of the failed build)
        if f.startswith("inst_a3_e12_") and "-x" in f:
            os.remove(os.path.join(_jit.JIT_DIR, f))
