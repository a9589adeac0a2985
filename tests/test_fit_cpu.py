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


# ---- vectorised tracer / collector (muax_amd/vector.py) against the per-step tracers ----
def _trace_with(tracer, obs, a, r, v, pi):
    tracer.reset()
    out = []
    T = len(r)
    for t in range(T):
        tracer.add(obs[t], int(a[t]), float(r[t]), t == T - 1, v=float(v[t]), pi=pi[t:t + 1])
        while tracer:
            out.append(tracer.pop())
    return out


@pytest.mark.parametrize("T", [1, 3, 5, 6, 23])
@pytest.mark.parametrize("alpha", [None, 0.5])
def test_vectorised_nstep_equals_the_per_step_tracer(T, alpha):
    import muax_amd as mx
    rng = np.random.default_rng(T)
    n, gamma = 5, 0.97
    obs = rng.normal(size=(T, 4)).astype(np.float32)
    a, r, v = rng.integers(0, 3, T), rng.normal(size=T), rng.normal(size=T)
    pi = rng.dirichlet(np.ones(3), T).astype(np.float32)
    ref = _trace_with(mx.NStep(n, gamma) if alpha is None else mx.PNStep(n, gamma, alpha), obs, a, r, v, pi)
    tr = mx.episode_trajectory(obs, a, r, v, pi, n, gamma, alpha)
    assert len(tr) == len(ref) == T
    for t, want in enumerate(ref):
        got = tr[t]
        assert np.array_equal(got.obs, want.obs) and int(got.a) == want.a and float(got.r) == want.r
        assert bool(got.done) == want.done and float(got.v) == want.v and np.array_equal(got.pi, want.pi)
        assert abs(float(got.Rn) - want.Rn) <= 1e-12 * max(1.0, abs(want.Rn))
        assert abs(float(got.w) - want.w) <= 1e-9
    # the batch a sample produces has the shapes of a trajectory filled by add()
    listed = mx.Trajectory()
    for tt in ref:
        listed.add(tt)
    listed.finalize()
    for x, y in zip(tr.batched_transitions, listed.batched_transitions):
        assert x.shape == y.shape
    if T > 3:
        s = tr.sample(num_samples=4, k_steps=3)
        assert len(s) == 4 and s[0].obs.shape == (1, 3, 4) and s[0].pi.shape == (1, 3, 1, 3)
    with pytest.raises(ValueError):
        tr.add(ref[0])


class _ScriptedVecEnv:
    """Episode lengths scripted per environment; observation = (env id, episode, step)."""

    def __init__(self, lengths):
        self.lengths, self.n = lengths, len(lengths)

    def _obs(self):
        return np.stack([np.array([e, self.ep[e], self.t[e]], np.float32) for e in range(self.n)])

    def reset(self):
        self.ep, self.t = [0] * self.n, [0] * self.n
        return self._obs()

    def step(self, actions):
        r, d = np.zeros(self.n), np.zeros(self.n, bool)
        for e in range(self.n):
            r[e] = 1.0 + 0.1 * e + 0.01 * self.t[e] + float(actions[e])
            self.t[e] += 1
            if self.t[e] == self.lengths[e][self.ep[e] % len(self.lengths[e])]:
                d[e], self.ep[e], self.t[e] = True, self.ep[e] + 1, 0
        return self._obs(), r, d


class _FakeModel:
    """act() as a pure function of the observation, so two collection schemes can be compared."""

    def act(self, key, obs, with_pi=False, with_value=False, obs_from_batch=False, **kw):
        obs = np.asarray(obs, np.float32)
        a = (obs.sum(1) % 2).astype(np.int64)
        pi = np.stack([0.25 + 0.5 * (a == 0), 0.25 + 0.5 * (a == 1)], 1).astype(np.float32)
        v = obs.sum(1).astype(np.float64) * 0.1
        return (a, pi, v) if (with_pi and with_value) else a


def test_vector_collector_cuts_the_stream_into_the_same_episodes_as_per_env_tracers():
    import muax_amd as mx
    lengths = [[3, 7], [12], [1, 2, 30], [9, 4]]
    n, gamma, alpha = 4, 0.9, 0.5
    col = mx.VectorCollector(_ScriptedVecEnv(lengths), n, gamma, alpha)
    key = mx.prng.PRNGKey(0)
    got = []
    for steps in (5, 11, 8, 17):  # episodes straddle the call boundaries
        trajs, key, count = col.collect(_FakeModel(), key, steps, num_simulations=4)
        assert count == steps * 4
        got += trajs
    # the same stream through one PNStep per environment
    env, model = _ScriptedVecEnv(lengths), _FakeModel()
    obs = env.reset()
    tracers = [mx.PNStep(n, gamma, alpha) for _ in lengths]
    want = {e: [] for e in range(len(lengths))}
    cur = {e: mx.Trajectory() for e in range(len(lengths))}
    for _ in range(5 + 11 + 8 + 17):
        a, pi, v = model.act(None, obs, with_pi=True, with_value=True)
        nxt, r, d = env.step(a)
        for e in range(len(lengths)):
            tracers[e].add(obs[e], int(a[e]), float(r[e]), bool(d[e]), v=float(v[e]), pi=pi[e:e + 1])
            while tracers[e]:
                cur[e].add(tracers[e].pop())
            if d[e]:
                cur[e].finalize()
                want[e].append(cur[e])
                cur[e] = mx.Trajectory()
                tracers[e].reset()
        obs = nxt
    by_env = {e: [] for e in range(len(lengths))}
    for tr in got:
        by_env[int(tr[0].obs[0])].append(tr)
    for e in by_env:
        assert len(by_env[e]) == len(want[e]) > 0
        for x, y in zip(by_env[e], want[e]):
            assert len(x) == len(y)
            bx, by = x.batched_transitions, y.batched_transitions
            assert np.array_equal(bx.obs, by.obs) and np.array_equal(bx.a, by.a) and np.array_equal(bx.done, by.done)
            assert np.array_equal(bx.pi, by.pi) and np.allclose(bx.Rn, by.Rn, rtol=1e-12, atol=0)
            assert np.allclose(bx.w, by.w, rtol=1e-9, atol=1e-12) and np.array_equal(bx.r, by.r)


def test_vector_greedy_test_counts_the_first_episode_of_every_environment():
    import muax_amd as mx
    env = _ScriptedVecEnv([[3, 7], [12], [1, 2, 30], [9, 4]])
    env.spec = type("S", (), {"max_episode_steps": 50})()
    got = mx.test_vector(_FakeModel(), env, mx.prng.PRNGKey(0), num_simulations=4)
    # replay the first episodes by hand
    ref_env, model = _ScriptedVecEnv([[3, 7], [12], [1, 2, 30], [9, 4]]), _FakeModel()
    obs, G, live = ref_env.reset(), np.zeros(4), np.ones(4, bool)
    for _ in range(12):
        a = model.act(None, obs)
        obs, r, d = ref_env.step(a)
        G += np.where(live, r, 0.0)
        live &= ~d
    assert not live.any() and abs(got - G.mean()) < 1e-12
