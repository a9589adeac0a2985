"""Host plumbing of fit(): n-step tracer and trajectory replay (muax/episode_tracer.py:118-249,
muax/replay_buffer.py:38-262) against hand-computed values."""
import numpy as np
import pytest

import muax_amd as mx


def test_nstep_returns_bootstrap_and_truncate_like_the_reference():
    n, g = 3, 0.5
    tr = mx.NStep(n, g)
    rewards, values = [1.0, 2.0, 3.0, 4.0, 5.0], [10.0, 20.0, 30.0, 40.0, 50.0]
    out = []
    for t, (r, v) in enumerate(zip(rewards, values)):
        tr.add(np.array([t], np.float32), t % 2, r, t == 4, v=v, pi=np.array([[0.5, 0.5]]))
        assert bool(tr) == (t == 4 or len(tr) > n)
        while tr:
            out.append(tr.pop())
    assert [o.r for o in out] == rewards and [int(o.obs[0]) for o in out] == [0, 1, 2, 3, 4]
    # t=0: 1 + .5*2 + .25*3 + .125*v3 ; t=1: 2 + .5*3 + .25*4 + .125*v4 ; then no bootstrap, truncated sums
    want = [1 + 1 + .75 + .125 * 40, 2 + 1.5 + 1 + .125 * 50, 3 + 2 + 1.25, 4 + 2.5, 5.0]
    assert np.allclose([o.Rn for o in out], want)
    assert [o.done for o in out] == [False, False, True, True, True]
    ptr = mx.PNStep(n, g, alpha=0.5)
    ptr.add(0, 0, 1.0, True, v=5.0)
    assert abs(ptr.pop().w - abs(5.0 - 1.0) ** 0.5) < 1e-12


def _trajectory(T, seed):
    rng = np.random.default_rng(seed)
    traj = mx.Trajectory()
    for t in range(T):
        traj.add(mx.Transition(obs=rng.normal(size=4).astype(np.float32), a=int(t % 2), r=float(t), done=t == T - 1,
                               Rn=float(10 * seed + t), v=0.5, pi=np.array([[0.25, 0.75]], np.float32), w=1.0 + t))
    traj.finalize()
    return traj


def test_trajectory_buffer_batches_k_consecutive_steps():
    t0 = _trajectory(12, 1)
    b = t0.batched_transitions
    assert b.obs.shape == (1, 12, 4) and b.a.shape == (1, 12) and b.pi.shape == (1, 12, 1, 2) and b.w.shape == (1, 12)
    assert _trajectory(5, 0).sample(3, k_steps=5) == []  # too short: nothing
    buf = mx.TrajectoryReplayBuffer(3, random_seed=0)
    for s in (1, 2, 3, 4):
        tj = _trajectory(12, s)
        buf.add(tj, tj.batched_transitions.w.mean())
    assert len(buf) == 3 and buf.capacity == 3  # ring: the first trajectory fell out
    batch = buf.sample(num_trajectory=4, sample_per_trajectory=2, k_steps=5)
    assert batch.obs.shape == (8, 5, 4) and batch.a.shape == (8, 5) and batch.pi.shape == (8, 5, 1, 2)
    assert (np.diff(batch.r, axis=1) == 1).all()  # consecutive steps of one episode
    assert set((batch.Rn[:, 0] - batch.r[:, 0]) // 10 * 10) <= {20.0, 30.0, 40.0}
    again = mx.TrajectoryReplayBuffer(3, random_seed=0)
    for s in (2, 3, 4):
        tj = _trajectory(12, s)
        again.add(tj, tj.batched_transitions.w.mean())
    b32 = buf.sample(batch_size=32, k_steps=3)
    assert b32.obs.shape == (32, 3, 4)
    with pytest.raises(ValueError):
        buf.sample(batch_size=None)


def test_fit_argument_errors_match_the_reference():
    with pytest.raises(ValueError):
        mx.fit(None)
    with pytest.raises(ValueError):
        mx.fit(None, env_id="CartPole-v1", env=object())
    with pytest.raises(ValueError):
        mx.fit(None, env=object())
