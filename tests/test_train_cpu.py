"""CPU tests of the interim training step: the loss restates muax/loss.py:10-88,
update() descends it, and the data-parallel gradient mean (one flat all-reduce) equals the single-process
gradient of the concatenated batch (world_size-2 gloo)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import muax_amd as mx
from oracle import mz_numpy as mn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(seed=0):
    g = torch.Generator().manual_seed(seed)
    net = mx.nn.MZNetwork(mx.nn.Representation(8, generator=g), mx.nn.Prediction(2, 21, generator=g),
                          mx.nn.Dynamic(8, 2, 21, generator=g))
    m = mx.MuZero(net, device="cpu")
    m.init(0, np.zeros((1, 4)))
    return m


def _batch(B=6, L=5, seed=0):
    rng = np.random.default_rng(seed)
    pi = rng.dirichlet([1, 1], (B, L)).astype(np.float32)
    return mx.Transition(obs=rng.uniform(-1, 1, (B, L, 4)).astype(np.float32), a=rng.integers(0, 2, (B, L)),
                         r=rng.uniform(0, 1, (B, L)).astype(np.float32), Rn=rng.uniform(0, 20, (B, L)).astype(np.float32),
                         pi=pi.reshape(B, L, 1, 2))  # the reference stores pi per step as [1, A]


def test_loss_matches_numpy_restatement_of_reference_formula():
    m, b = _model(), _batch()
    loss = float(mx.default_loss_fn(m, b))
    w = {k: v.detach().numpy() for k, v in mx.nn.mlp_trio_weights(m.network).items()}
    B, L = b.a.shape
    r_t, Rn_t = mn.scalar_to_support(b.r, 10), mn.scalar_to_support(b.Rn, 10)
    s = mn.min_max_normalize(b.obs[:, 0] @ w["repr_w"] + w["repr_b"])
    ce = lambda lg, y: float(np.mean(-(y * (lg - np.log(np.exp(lg - lg.max(-1, keepdims=True)).sum(-1, keepdims=True))
                                            - lg.max(-1, keepdims=True))).sum(-1)))  # noqa: E731
    ref = 0.0
    for i in range(L):
        v, lg = mn.prediction(w, s)
        sa = np.concatenate([s, np.eye(2, dtype=np.float32)[b.a[:, i]]], 1)
        r = mn._mlp2(sa, w["dr_w1"], w["dr_b1"], w["dr_w2"], w["dr_b2"])
        s = mn.min_max_normalize(mn._mlp2(sa, w["dn_w1"], w["dn_b1"], w["dn_w2"], w["dn_b2"]))
        ref += ce(r, r_t[:, i]) + ce(v, Rn_t[:, i]) + ce(lg, b.pi[:, i, 0])
    ref += 1e-4 * 0.5 * sum(float((x ** 2).sum()) for x in w.values())
    assert abs(loss - ref) < 1e-4 * max(1, abs(ref))
    assert abs(float(mx.default_loss_fn(m, b, divide_by_length=True)) - ((ref - 1e-4 * 0.5 * sum(float((x ** 2).sum()) for x in w.values())) / L
                                                                          + 1e-4 * 0.5 * sum(float((x ** 2).sum()) for x in w.values()))) < 1e-4
    assert float(mx.default_loss_fn(m, b, pi_all_pairs=True)) != loss  # the reference's [B,1,A] broadcast quirk


def test_scale_gradient_halves_the_state_gradient():
    """muax/loss.py:60-61: with L=1 the reward loss reaches the representation weights through
    scale_gradient(s, 0.5), the value/policy losses do not: grad = g_pred + 0.5 * g_dyn (+ L2)."""
    m, b = _model(), _batch(B=3, L=1)
    mx.default_loss_fn(m, b).backward()
    w = m.repr_func.repr_func.w
    got = w.grad.clone()
    s = m.repr_func(torch.as_tensor(b.obs[:, 0]))
    v, lg = m.pred_func(s)
    r, _ = m.dy_func(s, torch.as_tensor(b.a[:, 0]))
    ce = mx.loss.softmax_cross_entropy
    S = m._support_size
    g_pred = torch.autograd.grad(ce(v, mx.utils.scalar_to_support(torch.as_tensor(b.Rn[:, 0]), S)).mean()
                                 + ce(lg, torch.as_tensor(b.pi[:, 0, 0])).mean(), w, retain_graph=True)[0]
    g_dyn = torch.autograd.grad(ce(r, mx.utils.scalar_to_support(torch.as_tensor(b.r[:, 0]), S)).mean(), w)[0]
    assert torch.allclose(got, g_pred + 0.5 * g_dyn + 1e-4 * w.detach(), rtol=1e-4, atol=1e-7)
    assert float(g_dyn.abs().sum()) > 0


def test_update_descends_the_loss_and_refreshes_weights():
    m, b = _model(1), _batch(B=16, L=5, seed=1)
    m2 = mx.MuZero(m.network, optimizer=mx.optimizers.create_optimizer("adam", 1e-2), device="cpu")
    m2.init(0, np.zeros((1, 4)))
    v0 = m2._weights_version
    losses = [m2.update(b)["loss"] for _ in range(30)]
    assert losses[-1] < losses[0] - 0.3 and m2._weights_version == v0 + 30
    assert m2.optimizer_state is not None
    coax = mx.optimizers.optimizer(warmup_steps=5, transition_steps=10)
    m3 = mx.MuZero(m.network, optimizer=coax, device="cpu")
    m3.init(0, np.zeros((1, 4)))
    assert np.isfinite(m3.update(b)["loss"])
    with pytest.raises(ValueError):
        mx.optimizers.create_optimizer("lion")


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import muax_amd as mx2
    m, b = _model(2), _batch(B=8, L=4, seed=2)
    half = mx2.Transition(**{k: (v[rank * 4:(rank + 1) * 4] if isinstance(v, np.ndarray) else v)
                             for k, v in b.__dict__.items()})
    params = [p for mod in m.network for p in mod.parameters()]
    mx2.default_loss_fn(m, half).backward()
    mx2.allreduce_mean_flat([p.grad for p in params])
    got = torch.cat([p.grad.reshape(-1) for p in params])
    if rank == 0:
        m_full = _model(2)
        pf = [p for mod in m_full.network for p in mod.parameters()]
        mx2.default_loss_fn(m_full, b).backward()
        want = torch.cat([p.grad.reshape(-1) for p in pf])
        q.put(bool(torch.allclose(got, want, rtol=1e-4, atol=1e-6)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_gradient_mean_equals_full_batch_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_save_load_restores_the_optimiser_state_and_the_schedule(tmp_path):
    """muax/model.py:203-212 restores opt_state on load; resumed training must continue the Adam moments, the
    step counts and the learning-rate schedule -- bit for bit the same trajectory as an uninterrupted run."""
    b = _batch(B=8, L=3, seed=4)

    def fresh():
        m = _model(3)
        m2 = mx.MuZero(m.network, optimizer=mx.optimizers.optimizer(warmup_steps=4, transition_steps=6), device="cpu")
        m2.init(0, np.zeros((1, 4)))
        return m2

    ref = fresh()
    ref_losses = [ref.update(b)["loss"] for _ in range(6)]
    first = fresh()
    losses = [first.update(b)["loss"] for _ in range(3)]
    path = str(tmp_path / "ckpt.pt")
    first.save(path)
    resumed = fresh()
    resumed.load(path)  # before the optimiser is bound to parameters: applied at the first update()
    losses += [resumed.update(b)["loss"] for _ in range(3)]
    assert losses == ref_losses
    state = resumed._optimizer.opt.state_dict()["state"]
    assert all(int(s["step"]) == 6 for s in state.values())
    assert resumed._optimizer.sched.last_epoch == 6
    assert resumed._optimizer.opt.param_groups[0]["lr"] == ref._optimizer.opt.param_groups[0]["lr"]
    for p, q in zip((p for mod in resumed.network for p in mod.parameters()), (p for mod in ref.network for p in mod.parameters())):
        assert torch.equal(p, q)
    # loading into a model whose optimiser is already bound applies the state at once
    again = fresh()
    again.update(b)
    again.load(path)
    assert all(int(s["step"]) == 3 for s in again._optimizer.opt.state_dict()["state"].values())
    assert again._optimizer.sched.last_epoch == 3
