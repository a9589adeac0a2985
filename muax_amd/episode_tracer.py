"""n-step tracers of the reference (muax/episode_tracer.py:41-249) on plain Python lists / NumPy.

Host-side data plumbing for fit(): no arithmetic of the hot path lives here.  The
protocol is the reference's: `add` one environment step, `while tracer: tracer.pop()` yields transitions
whose n-step bootstrapped return can already be computed (or all of them once the episode is done)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Any

import numpy as np


@dataclass
class Transition:
    """muax/episode_tracer.py:41-56; batched instances hold [B, L, ...] arrays in every field."""
    obs: Any = 0.
    a: Any = 0
    r: Any = 0.
    done: Any = False
    Rn: Any = 0.
    v: Any = 0.
    pi: Any = 0.
    w: Any = 1.

    def __iter__(self):
        return iter((self.obs, self.a, self.r, self.done, self.Rn, self.v, self.pi, self.w))

    def __getitem__(self, index):
        return Transition(*(x[index] for x in self))


def flatten_transition_func(transition):
    """muax/episode_tracer.py:58-59 (the pytree flattening of a Transition): (leaves, treedef)."""
    return iter(dataclasses.astuple(transition) if dataclasses.is_dataclass(transition) else transition), None


def unflatten_transition_func(treedef, leaves):
    """muax/episode_tracer.py:61-62."""
    return Transition(*leaves)


class BaseTracer:
    """muax/episode_tracer.py:71-111: what fit() needs from a tracer."""

    def reset(self):
        raise NotImplementedError

    def add(self, obs, a, r, done, v=0.0, pi=0.0, w=1.0):
        raise NotImplementedError

    def pop(self):
        raise NotImplementedError


class NStep(BaseTracer):
    """muax/episode_tracer.py:118-195: Rn = sum_{i<n} gamma^i r_{t+i} + gamma^n v_{t+n}; towards the end
    of an episode the sum is truncated and nothing is bootstrapped (done=True)."""

    def __init__(self, n, gamma, transition_class=Transition):
        self.n, self.gamma, self.transition_class = int(n), float(gamma), transition_class
        self.reset()

    def reset(self):
        self._steps, self._rewards, self._done = [], [], False
        self._gammas = np.power(self.gamma, np.arange(self.n))
        self._gamman = np.power(self.gamma, self.n)

    def add(self, obs, a, r, done, v=0.0, pi=0.0, w=1.0):
        self._steps.append((obs, a, v, pi, w))
        self._rewards.append(r)
        self._done = bool(done)

    def __len__(self):
        return len(self._steps)

    def __bool__(self):
        return bool(len(self)) and (self._done or len(self) > self.n)

    def _pop(self):
        obs, a, v, pi, w = self._steps.pop(0)
        rs = np.asarray(self._rewards[:self.n], dtype=np.float64)
        Rn = float(np.sum(self._gammas[:len(rs)] * rs))
        r = self._rewards.pop(0)
        if len(self) >= self.n:  # the state n steps ahead is in the cache: bootstrap from its value
            v_next, done, gamman = self._steps[self.n - 1][2], False, self._gamman
        else:
            v_next, done, gamman = 0.0, True, self._gammas[len(rs) - 1]
        return obs, a, r, done, Rn + float(v_next) * float(gamman), v, pi, w

    def pop(self):
        obs, a, r, done, Rn, v, pi, w = self._pop()
        return self.transition_class(obs=obs, a=a, r=r, done=done, Rn=Rn, v=v, pi=pi, w=w)


class PNStep(NStep):
    """muax/episode_tracer.py:198-249: priority weight w = |v - Rn| ** alpha."""

    def __init__(self, n, gamma, alpha: float = 0.5, transition_class=Transition):
        self.alpha = float(alpha)
        super().__init__(n, gamma, transition_class)

    def pop(self):
        obs, a, r, done, Rn, v, pi, _ = self._pop()
        return self.transition_class(obs=obs, a=a, r=r, done=done, Rn=Rn, v=v, pi=pi,
                                     w=abs(float(np.asarray(v).reshape(-1)[0]) - Rn) ** self.alpha)
