"""GPU tests of the fused training step (mzs_mlp_loss_grad, muax_amd/csrc/mz_train.cuh): a floating-point
kernel, checked against torch autograd on the SAME formula (muax_amd/loss.py restating muax/loss.py:10-88)
-- fp32 on the GPU and fp64 on the CPU -- with the tolerance stated here: 2e-4 of the largest gradient entry
per array (fp32 accumulation over B*L terms in different orders), loss 1e-5 relative."""
import numpy as np
import pytest
import torch

import muax_amd as mx

pytestmark = pytest.mark.gpu
F32 = np.float32


def _model(A, E, obs_dim, seed, support=10):
    g = torch.Generator().manual_seed(seed)
    F = 2 * support + 1
    net = mx.nn.MZNetwork(mx.nn.Representation(E, generator=g), mx.nn.Prediction(A, F, generator=g),
                          mx.nn.Dynamic(E, A, F, generator=g))
    m = mx.MuZero(net, optimizer=mx.optimizers.create_optimizer("adam", 1e-2), support_size=support)
    m.init(0, np.zeros((1, obs_dim)))
    with torch.no_grad():  # non-zero biases so that every gradient path is exercised
        for p in [p for mod in m.network for p in mod.parameters()]:
            if p.dim() == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g).to(p.device))
    return m


def _batch(B, L, A, obs_dim, seed):
    rng = np.random.default_rng(seed)
    return mx.Transition(obs=rng.uniform(-1, 1, (B, L, obs_dim)).astype(F32), a=rng.integers(0, A, (B, L)),
                         r=rng.uniform(-2, 3, (B, L)).astype(F32), Rn=rng.uniform(-30, 60, (B, L)).astype(F32),
                         pi=rng.dirichlet(np.ones(A), (B, L)).astype(F32).reshape(B, L, 1, A))


def _autograd(m, b, dtype, device, **kw):
    import copy
    mods = [copy.deepcopy(x).to(device=device, dtype=dtype) for x in m.network]
    m2 = mx.MuZero(mx.nn.MZNetwork(*mods), device=device)
    m2._params, m2._support_size = True, m._support_size
    bb = mx.Transition(**{k: (torch.as_tensor(v).to(dtype) if isinstance(v, np.ndarray) and v.dtype == F32 else v)
                          for k, v in b.__dict__.items()})
    orig = mx.loss.default_loss_fn

    def loss64(inst, batch, **k2):  # the restated loss casts to float32; redo it in `dtype`
        dev = inst.device
        t = lambda x, dt=dtype: torch.as_tensor(x, device=dev).to(dt)  # noqa: E731
        a = t(batch.a, torch.long)
        B, L = a.shape[:2]
        S = inst._support_size
        r_t = mx.utils.scalar_to_support(t(batch.r).reshape(B, L), S)
        Rn_t = mx.utils.scalar_to_support(t(batch.Rn).reshape(B, L), S)
        pi = t(batch.pi).reshape(B, L, -1)
        s = inst.repr_func(t(batch.obs)[:, 0])
        loss = 0
        for i in range(L):
            v, lg = inst.pred_func(s)
            s = mx.utils.scale_gradient(s, 0.5)
            r, ns = inst.dy_func(s, a[:, i])
            ce = mx.loss.softmax_cross_entropy
            loss = loss + ce(r, r_t[:, i]).mean() + ce(v, Rn_t[:, i]).mean() + ce(lg, pi[:, i]).mean()
            s = ns
        if k2.get("divide_by_length"):
            loss = loss / L
        return loss + 1e-4 * 0.5 * sum((p ** 2).sum() for mod in inst.network for p in mod.parameters())

    loss = (loss64 if dtype == torch.float64 else orig)(m2, bb if dtype == torch.float64 else b, **kw)
    loss.backward()
    w = mx.nn.mlp_trio_weights(m2.network)
    from muax_amd._lib import MLP_WEIGHT_NAMES
    return float(loss.detach()), [w[n].grad.detach().cpu().double().numpy() for n in MLP_WEIGHT_NAMES]


@pytest.mark.parametrize("A,E,obs_dim,B,L,kw", [
    (2, 8, 4, 50, 5, {}), (2, 8, 4, 4096, 10, {}), (2, 8, 4, 7, 1, {}), (4, 32, 8, 33, 3, {}),
    (3, 8, 5, 16, 4, {"divide_by_length": True}), (4, 16, 6, 40, 3, {}), (2, 16, 4, 20, 2, {}), (4, 8, 8, 24, 3, {}),
    (2, 10, 4, 30, 3, {}), (4, 10, 8, 21, 2, {}), (6, 8, 6, 18, 3, {}), (8, 8, 6, 17, 2, {}), (2, 32, 8, 19, 3, {})])
def test_fused_loss_and_gradients_match_autograd(A, E, obs_dim, B, L, kw):
    m, b = _model(A, E, obs_dim, seed=A + E), _batch(B, L, A, obs_dim, seed=B)
    fused = mx.loss.FusedLossGrad(m)
    loss, flat = fused(b, **kw)
    loss, views = float(loss.item()), [v.detach().cpu().double().numpy() for v in fused.views]
    l32, g32 = _autograd(m, b, torch.float32, "cuda", **kw)
    l64, g64 = _autograd(m, b, torch.float64, "cpu", **kw)
    assert abs(loss - l64) <= 1e-5 * abs(l64), (loss, l64, l32)
    from muax_amd._lib import MLP_WEIGHT_NAMES
    for n, gh, gt, gd in zip(MLP_WEIGHT_NAMES, views, g32, g64):
        tol = 2e-4 * max(np.abs(gd).max(), 1e-6)
        assert gh.shape == gd.shape and np.abs(gh - gd).max() <= tol, (n, np.abs(gh - gd).max(), tol)
        assert np.abs(gt - gd).max() <= 5 * tol  # the torch fp32 route is no closer to fp64 than the kernel
    loss2, flat2 = fused(b, **kw)  # fixed-order reduction: bit-reproducible
    assert float(loss2.item()) == loss and torch.equal(flat2, flat)


@pytest.mark.parametrize("support", [15, 20])
def test_fused_loss_and_gradients_other_support_sizes(support):
    """support_size is a constructor argument of the reference (muax/model.py:48-49): kernel instances for 15 and 20."""
    m, b = _model(2, 8, 4, seed=support, support=support), _batch(40, 4, 2, 4, seed=support)
    fused = mx.loss.FusedLossGrad(m)
    loss, _ = fused(b)
    views = [v.detach().cpu().double().numpy() for v in fused.views]
    l64, g64 = _autograd(m, b, torch.float64, "cpu")
    assert abs(float(loss.item()) - l64) <= 1e-5 * abs(l64)
    for gh, gd in zip(views, g64):
        assert np.abs(gh - gd).max() <= 2e-4 * max(np.abs(gd).max(), 1e-6)


@pytest.mark.parametrize("A,E,obs_dim,support,B,L", [(3, 16, 5, 12, 40, 4), (5, 12, 6, 10, 33, 3), (12, 8, 4, 10, 24, 3),
                                                     (2, 24, 8, 10, 19, 2), (7, 40, 8, 15, 16, 2)])
def test_fused_training_instance_built_on_demand(A, E, obs_dim, support, B, L):
    """Shapes of the default trio mzs_mlp_loss_grad lists no instance for (round 5: act() serves them through on-demand
    instances of the search kernel; the training step follows): the first call returns "no kernel instance",
    muax_amd/_jit.py::ensure_train_instance compiles mz_train_jit.hip for the triple with this box's hipcc and registers it,
    and the same call then gives loss and gradients that agree with fp64 autograd to the tolerance of the listed
    instances; MuZero.update() takes that route by itself and steps like the torch route."""
    from muax_amd import _jit
    m, b = _model(A, E, obs_dim, seed=A + E, support=support), _batch(B, L, A, obs_dim, seed=B)
    fused = mx.loss.FusedLossGrad(m)
    if ("train", A, E, 2 * support + 1) not in _jit._loaded:
        with pytest.raises(ValueError, match="no kernel instance"):
            fused(b)
    assert _jit.ensure_train_instance(A, E, 2 * support + 1)
    loss, flat = fused(b)
    loss, views = float(loss.item()), [v.detach().cpu().double().numpy() for v in fused.views]
    l64, g64 = _autograd(m, b, torch.float64, "cpu")
    assert abs(loss - l64) <= 1e-5 * abs(l64), (loss, l64)
    from muax_amd._lib import MLP_WEIGHT_NAMES
    for n, gh, gd in zip(MLP_WEIGHT_NAMES, views, g64):
        assert gh.shape == gd.shape and np.abs(gh - gd).max() <= 2e-4 * max(np.abs(gd).max(), 1e-6), n
    loss2, flat2 = fused(b)
    assert float(loss2.item()) == loss and torch.equal(flat2, flat)
    out = {}
    for backend in ("auto", "torch"):
        m2 = _model(A, E, obs_dim, seed=A + E, support=support)
        losses = [m2.update(b, backend=backend)["loss"] for _ in range(10)]
        out[backend] = losses
        assert (m2._fused_train is not None) == (backend == "auto")
    assert np.allclose(out["auto"], out["torch"], rtol=5e-4)


def test_update_hip_and_torch_routes_take_the_same_step():
    b = _batch(256, 6, 2, 4, seed=3)
    out = {}
    for backend in ("hip", "torch"):
        m = _model(2, 8, 4, seed=5)
        losses = [m.update(b, backend=backend)["loss"] for _ in range(25)]
        out[backend] = (losses, torch.cat([p.detach().reshape(-1) for mod in m.network for p in mod.parameters()]).cpu())
        assert losses[-1] < losses[0] - 0.2
    assert np.allclose(out["hip"][0], out["torch"][0], rtol=2e-4)
    assert torch.allclose(out["hip"][1], out["torch"][1], rtol=5e-3, atol=5e-4)
    m = _model(2, 8, 4, seed=5)
    assert np.isfinite(m.update(b)["loss"]) and m._fused_train is not None  # auto -> HIP route
    custom = mx.MuZero(m.network, loss_fn=mx.default_loss_fn)  # a user loss_fn is opaque to the kernel
    custom.init(0, np.zeros((1, 4)))
    with pytest.raises(ValueError):
        custom.update(b, backend="hip")
    assert np.isfinite(custom.update(b)["loss"]) and custom._fused_train is None


def test_c_abi_rejects_bad_training_and_tower_arguments():
    """Error behaviour of the next-tier entry points: negative status + message, mapped to ValueError."""
    import ctypes as C

    from muax_amd import _lib
    L = _lib.load()
    w = _lib.MzsMlpWeights()
    a = _lib.MzsTrainArgs()
    assert L.mzs_mlp_loss_grad(C.byref(w), C.byref(a), None) == _lib.MZS_E_INVALID  # struct_size not set
    w.struct_size, a.struct_size = C.sizeof(w), C.sizeof(a)
    with pytest.raises(ValueError, match="null weight pointer"):
        _lib.check(L.mzs_mlp_loss_grad(C.byref(w), C.byref(a), None))
    m = _model(2, 8, 4, seed=1)
    f = mx.loss.FusedLossGrad(m)
    with pytest.raises(ValueError, match="features"):
        f(_batch(4, 2, 2, 5, seed=0))  # obs width does not match the network
    t = _lib.MzsTowerArgs()
    assert L.mzs_resnet_tower(C.byref(t), None) == _lib.MZS_E_INVALID
    t.struct_size, t.batch, t.blocks = C.sizeof(t), 2, 1
    with pytest.raises(ValueError, match="null tensor pointer"):
        _lib.check(L.mzs_resnet_tower(C.byref(t), None))
    x = torch.zeros(2, 6, 6, 64, device="cuda")
    t.x = t.y = t.conv_w = t.ln = x.data_ptr()
    t.stem_w = x.data_ptr()
    with pytest.raises(ValueError, match="stem needs actions"):
        _lib.check(L.mzs_resnet_tower(C.byref(t), None))
    assert L.mzs_mlp_num_params(4, 8, 2, 10) == 4 * 8 + 8 + 2 * (8 * 16 + 16) + 16 * 21 + 21 + 16 * 2 + 2 + 2 * (10 * 16 + 16) + 16 * 21 + 21 + 16 * 8 + 8


@pytest.mark.gpu
def test_device_selftest_of_the_shortened_sqrt_and_division():
    """mzs_selftest: sqrt_normal (v_sqrt_f32 + two exact residual checks) against the IEEE sqrt for every binary32 in
    [1, 4), the 3-op division by 0.002f against the IEEE division over 2^-9 .. 2^-2, and the shared-reciprocal division
    of the support decode and of the value scores against n / d (2^24 denominators in [1, 64) x 12 numerators, 2^24 in
    2^-27 .. 2^41 x 8), ON the device."""
    import ctypes as C

    from muax_amd import _lib
    out = (C.c_int64 * 4)(-1, -1, -1, -1)
    _lib.check(_lib.load().mzs_selftest(0, C.byref(out)))
    assert list(out) == [0, 0, 0, 0], list(out)
