"""A dependency-free cart-pole environment with the gymnasium calling convention, so the reference's
README recipe (`muax.fit(model, 'CartPole-v1', ...)`) can be run in an image without gymnasium:
`muax.fit(model, env=CartPole(), test_env=CartPole(), ...)`.

Dynamics: the Barto-Sutton-Anderson (1983) cart-pole, explicit Euler at 50 Hz, +1 reward per step, the
episode ends when |x| > 2.4 m or |theta| > 12 degrees, truncated after 500 steps.
"""
import math
from types import SimpleNamespace

import numpy as np


class _Box:
    def __init__(self, high, rng):
        self.high, self.low, self.shape, self._rng = high, -high, high.shape, rng

    def sample(self):
        return self._rng.uniform(-1.0, 1.0, self.shape).astype(np.float32) * np.minimum(self.high, 10.0)


class _Discrete:
    def __init__(self, n, rng):
        self.n, self._rng = n, rng

    def sample(self):
        return int(self._rng.integers(self.n))


class CartPole:
    GRAVITY, M_CART, M_POLE, HALF_LEN, FORCE, DT = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT = 2.4, 12 * 2 * math.pi / 360

    def __init__(self, max_episode_steps=500, seed=None):
        self.spec = SimpleNamespace(id="CartPole-v1", max_episode_steps=max_episode_steps)
        self._rng = np.random.default_rng(seed)
        high = np.array([2 * self.X_LIMIT, np.finfo(np.float32).max, 2 * self.THETA_LIMIT, np.finfo(np.float32).max],
                        np.float32)
        self.observation_space = _Box(high, self._rng)
        self.action_space = _Discrete(2, self._rng)
        self._state, self._t = None, 0

    def reset(self, seed=None, options=None):
        if seed is not None:
            self._rng = np.random.default_rng(seed)
        self._state = self._rng.uniform(-0.05, 0.05, 4)
        self._t = 0
        return self._state.astype(np.float32), {}

    def step(self, action):
        x, x_dot, th, th_dot = self._state
        f = self.FORCE if int(action) == 1 else -self.FORCE
        m_total, pm_l = self.M_CART + self.M_POLE, self.M_POLE * self.HALF_LEN
        c, s = math.cos(th), math.sin(th)
        tmp = (f + pm_l * th_dot * th_dot * s) / m_total
        th_acc = (self.GRAVITY * s - c * tmp) / (self.HALF_LEN * (4.0 / 3.0 - self.M_POLE * c * c / m_total))
        x_acc = tmp - pm_l * th_acc * c / m_total
        x, x_dot = x + self.DT * x_dot, x_dot + self.DT * x_acc
        th, th_dot = th + self.DT * th_dot, th_dot + self.DT * th_acc
        self._state = np.array([x, x_dot, th, th_dot])
        self._t += 1
        terminated = bool(abs(x) > self.X_LIMIT or abs(th) > self.THETA_LIMIT)
        truncated = bool(self._t >= self.spec.max_episode_steps)
        return self._state.astype(np.float32), 1.0, terminated, truncated, {}


class VectorCartPole:
    """N cart-poles stepped as arrays, with auto-reset: `step` returns (obs, reward, done) where `done`
    marks the last step of an episode and the observation of a finished cart-pole is already the first one
    of its next episode (the protocol of muax_amd.rollout_batched / VectorCollector)."""

    def __init__(self, n, max_episode_steps=500, seed=None):
        self.n = int(n)
        self.spec = SimpleNamespace(id="CartPole-v1", max_episode_steps=max_episode_steps)
        self._rng = np.random.default_rng(seed)
        self._state = np.zeros((self.n, 4))
        self._t = np.zeros(self.n, np.int64)

    def reset(self):
        self._state = self._rng.uniform(-0.05, 0.05, (self.n, 4))
        self._t[:] = 0
        return self._state.astype(np.float32)

    def step(self, actions):
        c = CartPole
        x, x_dot, th, th_dot = self._state.T
        f = np.where(np.asarray(actions).reshape(-1) == 1, c.FORCE, -c.FORCE)
        m_total, pm_l = c.M_CART + c.M_POLE, c.M_POLE * c.HALF_LEN
        cos, sin = np.cos(th), np.sin(th)
        tmp = (f + pm_l * th_dot * th_dot * sin) / m_total
        th_acc = (c.GRAVITY * sin - cos * tmp) / (c.HALF_LEN * (4.0 / 3.0 - c.M_POLE * cos * cos / m_total))
        x_acc = tmp - pm_l * th_acc * cos / m_total
        x, x_dot = x + c.DT * x_dot, x_dot + c.DT * x_acc
        th, th_dot = th + c.DT * th_dot, th_dot + c.DT * th_acc
        self._state = np.stack([x, x_dot, th, th_dot], 1)
        self._t += 1
        done = (np.abs(x) > c.X_LIMIT) | (np.abs(th) > c.THETA_LIMIT) | (self._t >= self.spec.max_episode_steps)
        k = int(done.sum())
        if k:
            self._state[done] = self._rng.uniform(-0.05, 0.05, (k, 4))
            self._t[done] = 0
        return self._state.astype(np.float32), np.ones(self.n), done
