"""Vectorised tracer + collector: the data plumbing that feeds the batched act().

The reference traces one environment with a Python deque per step (`muax/episode_tracer.py:118-249`,
driven from `muax/train.py:150-173`).  With thousands of roots searched per launch that bookkeeping is the
bottleneck, so here the n-step returns and priority weights of whole episodes are computed as array
operations, and a vector environment is stepped with ONE batched act() per step.  Values are those of
`NStep` / `PNStep` (tests/test_fit_cpu.py compares them transition by transition).  Host-side NumPy:
no arithmetic of the hot path lives here.

Vector-environment protocol (that of `rollout_batched`): `reset() -> obs [N, ...]`,
`step(actions [N]) -> (obs [N, ...], reward [N], done [N])`, where `done` marks the LAST step of an
episode (terminated or truncated) and the returned observation of a finished environment is already the
first observation of its next episode (auto-reset).
"""
from __future__ import annotations

import time

import numpy as np

from . import prng
from .replay_buffer import Trajectory


def nstep_returns(r, v, n: int, gamma: float):
    """`NStep` over one COMPLETE episode at once (muax/episode_tracer.py:161-195).
    r, v: [T].  Returns (Rn [T] float64, done [T] bool): Rn[t] = sum_{i<n, t+i<T} gamma^i r[t+i]
    + gamma^n v[t+n] when step t+n exists; `done` marks the transitions that could not bootstrap."""
    r = np.asarray(r, np.float64).reshape(-1)
    v = np.asarray(v, np.float64).reshape(-1)
    T = r.shape[0]
    rp = np.concatenate([r, np.zeros(n)])
    Rn = np.zeros(T)
    for i in range(n):  # n is ~10: a loop over the horizon, vectorised over the episode
        Rn += (gamma ** i) * rp[i:i + T]
    boot = np.arange(T) + n < T
    vb = np.concatenate([v, np.zeros(n)])[n:n + T]
    Rn = Rn + np.where(boot, vb * (gamma ** n), 0.0)
    return Rn, ~boot


def episode_trajectory(obs, a, r, v, pi, n: int, gamma: float, alpha=None) -> Trajectory:
    """One finished episode (arrays [T, ...]) -> an array-backed Trajectory with the fields `NStep`
    (alpha None: w = 1) or `PNStep` (w = |v - Rn| ** alpha, muax/episode_tracer.py:198-249) would emit."""
    Rn, done = nstep_returns(r, v, n, gamma)
    v64 = np.asarray(v, np.float64).reshape(-1)
    w = np.ones_like(Rn) if alpha is None else np.abs(v64 - Rn) ** alpha
    pi = np.asarray(pi)
    if pi.ndim == 2:  # the reference keeps act()'s [1, A] row per step (muax/model.py:176, train.py:164)
        pi = pi[:, None, :]
    return Trajectory.from_arrays(np.asarray(obs), np.asarray(a).astype(np.int64), np.asarray(r, np.float64), done,
                                  Rn, v64, pi, w)


class VectorCollector:
    """Steps a vector environment with one batched act() per step and cuts the stream into episodes.
    Unfinished episodes are carried into the next `collect` call, so every trajectory handed out is a
    complete episode, exactly what the per-environment tracers of `fit` produce."""

    def __init__(self, venv, n: int, gamma: float, alpha=0.5):
        self.venv, self.n, self.gamma, self.alpha = venv, int(n), float(gamma), alpha
        self._obs = None
        self._pending = None  # per environment: (obs, a, r, v, pi) of the episode still open, or None

    def collect(self, model, key, steps: int, num_simulations: int = 50, temperature: float = 1.0, **act_kwargs):
        """`steps` lock-step environment steps.  Returns (finished trajectories, advanced key, env steps)."""
        if self._obs is None:
            self._obs = np.asarray(self.venv.reset())
            self._pending = [None] * self._obs.shape[0]
        N = self._obs.shape[0]
        obs_l, a_l, r_l, d_l, v_l, pi_l = [], [], [], [], [], []
        obs = self._obs
        for _ in range(steps):
            key, subkey = prng.split(key)
            a, pi, v = model.act(subkey, obs, with_pi=True, with_value=True, obs_from_batch=True,
                                 num_simulations=num_simulations, temperature=temperature, **act_kwargs)
            nxt, r, done = self.venv.step(a)
            obs_l.append(obs), a_l.append(np.asarray(a)), r_l.append(np.asarray(r, np.float64))
            d_l.append(np.asarray(done, bool)), v_l.append(np.asarray(v, np.float64)), pi_l.append(np.asarray(pi))
            obs = np.asarray(nxt)
        self._obs = obs
        # env-major flat streams: for every environment its carried-over open episode, then the new steps
        fields = [np.stack(x) for x in (obs_l, a_l, r_l, v_l, pi_l)]  # [T, N, ...]
        D = np.stack(d_l)
        carried = np.array([0 if p is None else len(p[0]) for p in self._pending])
        flat = []
        for k, f in enumerate(fields):
            f = np.swapaxes(f, 0, 1)  # [N, T, ...]
            if carried.any():
                f = np.concatenate([x for e in range(N) for x in
                                    ((self._pending[e][k], f[e]) if carried[e] else (f[e],))])
            else:
                f = f.reshape((N * steps,) + f.shape[2:])
            flat.append(f)
        O, A, R, V, P = flat
        first = np.concatenate([[0], np.cumsum(carried + steps)])  # stream offsets per environment
        done = np.zeros(first[-1], bool)  # carried steps are never episode ends
        done[(first[:-1] + carried)[:, None] + np.arange(steps)[None, :]] = D.T
        ends = np.flatnonzero(done)
        # closed part of every stream = up to its last episode end; the rest is carried over
        last_end = np.full(N, -1)
        env_of_end = np.searchsorted(first, ends, side="right") - 1
        last_end[env_of_end] = ends  # ends ascend, so the last one per environment stays
        M = first[-1]
        pos = np.arange(M)
        nxt = ends[np.minimum(np.searchsorted(ends, pos), max(len(ends) - 1, 0))] if len(ends) else pos
        left = nxt - pos + 1  # steps to the end of the episode, this one included (garbage on open tails)
        n, g = self.n, self.gamma
        Rp, Vp = np.concatenate([R, np.zeros(n)]), np.concatenate([V, np.zeros(n)])
        Rn = np.zeros(M)
        for i in range(n):
            Rn += np.where(i < left, (g ** i) * Rp[i:i + M], 0.0)
        boot = left > n
        Rn = Rn + np.where(boot, Vp[n:n + M] * (g ** n), 0.0)
        W = np.ones(M) if self.alpha is None else np.abs(V - Rn) ** self.alpha
        if P.ndim == 2:
            P = P[:, None, :]
        A = A.astype(np.int64)
        nb = ~boot
        out = []
        starts = np.concatenate([[0], ends[:-1] + 1]) if len(ends) else np.zeros(0, int)
        starts = np.maximum(starts, first[env_of_end]) if len(ends) else starts
        for s0, e0 in zip(starts.tolist(), (ends + 1).tolist()):
            out.append(Trajectory.from_arrays(O[s0:e0], A[s0:e0], R[s0:e0], nb[s0:e0], Rn[s0:e0], V[s0:e0],
                                              P[s0:e0], W[s0:e0], _checked=True))
        for e in range(N):
            s0 = first[e] if last_end[e] < 0 else last_end[e] + 1
            e0 = first[e + 1]
            self._pending[e] = (O[s0:e0], A[s0:e0], R[s0:e0], V[s0:e0], P[s0:e0, 0]) if s0 < e0 else None
        return out, key, steps * N


def test_vector(model, venv, key, num_simulations: int, max_steps=None):
    """Greedy evaluation (muax/test.py:5-48: temperature 0, mean undiscounted return) on a vector
    environment: the FIRST episode of each of its N environments, one batched act() per step."""
    obs = np.asarray(venv.reset())
    N = obs.shape[0]
    G, live = np.zeros(N), np.ones(N, bool)
    steps = max_steps if max_steps is not None else venv.spec.max_episode_steps
    for _ in range(steps):
        key, subkey = prng.split(key)
        a = model.act(subkey, obs, obs_from_batch=True, num_simulations=num_simulations, temperature=0.)
        obs, r, done = venv.step(a)
        obs = np.asarray(obs)
        G += np.where(live, np.asarray(r, np.float64), 0.0)
        live &= ~np.asarray(done, bool)
        if not live.any():
            break
    return float(G.mean())


def fit_vector(model, venv, test_env, n_step: int = 10, gamma: float = 0.997, alpha=0.5, buffer=None,
               iterations: int = 100, steps_per_iteration: int = 64, num_simulations: int = 50, k_steps: int = 10,
               num_trajectory: int = 32, sample_per_trajectory: int = 1, num_update_per_iteration: int = 50,
               max_training_steps: int = 10000, test_interval: int = 10, num_test_episodes: int = 10,
               random_seed: int = 42, temperature_fn=None, metrics=None, trajectory_weight: str = "mean"):
    """The reference's fit() loop (muax/train.py:175-241: temperature schedule, buffer sampling, update,
    greedy test) with the acting half on a vector environment: per iteration `steps_per_iteration`
    batched act() calls -> finished episodes -> buffer, then `num_update_per_iteration` updates.
    `trajectory_weight`: "mean" is the reference's buffer weight (mean priority of the episode,
    muax/train.py:171,203); "sum" weights an episode by its total priority, which undoes the bias of a
    fixed collection window towards short episodes (many short episodes finish while one long one runs)."""
    if trajectory_weight not in ("mean", "sum"):
        raise ValueError("trajectory_weight must be 'mean' or 'sum'")
    from .replay_buffer import TrajectoryReplayBuffer
    from .train import _temperature_fn, test
    temperature_fn = temperature_fn or _temperature_fn
    buffer = buffer if buffer is not None else TrajectoryReplayBuffer(500)
    collector = VectorCollector(venv, n_step, gamma, alpha)
    key = prng.PRNGKey(random_seed)
    key, test_key, subkey = prng.split(key, 3)
    model.init(subkey, np.asarray(venv.reset())[:1].astype(float))
    training_step = 0
    for it in range(iterations):
        temperature = temperature_fn(max_training_steps=max_training_steps, training_steps=training_step)
        t0 = time.perf_counter()
        trajs, key, env_steps = collector.collect(model, key, steps_per_iteration, num_simulations, temperature)
        collect_s = time.perf_counter() - t0
        for tr in trajs:
            if len(tr) >= k_steps:
                buffer.add(tr, tr.weights.mean() if trajectory_weight == "mean" else tr.weights.sum())
        row = {"iteration": it, "env_steps": env_steps, "episodes": len(trajs), "collect_s": collect_s,
               "G": float(np.mean([float(np.sum(t.rewards)) for t in trajs])) if trajs else float("nan")}
        if len(buffer):
            loss = 0.0
            for _ in range(num_update_per_iteration):
                loss += model.update(buffer.sample(num_trajectory=num_trajectory,
                                                   sample_per_trajectory=sample_per_trajectory, k_steps=k_steps))["loss"]
                training_step += 1
            row["loss"] = loss / num_update_per_iteration
        row["training_step"] = training_step
        if it % test_interval == 0:
            if hasattr(test_env, "observation_space"):  # a gym-style environment: the reference's test()
                row["test_G"] = test(model, test_env, test_key, num_simulations=num_simulations,
                                     num_test_episodes=num_test_episodes)
            else:  # a vector environment: all its episodes in lock step
                row["test_G"] = test_vector(model, test_env, test_key, num_simulations)
        if metrics is not None:
            metrics.append(row)
        if training_step >= max_training_steps:
            break
    return model
