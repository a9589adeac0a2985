"""Trajectory replay of the reference (muax/replay_buffer.py:38-262) on NumPy: episodes are kept whole,
a training sample is k consecutive transitions drawn inside an episode with the transitions' priority
weights, episodes are drawn with their mean weight.  Batches come out as Transition([B, L, ...])."""
from __future__ import annotations

import random
from collections import deque
from itertools import accumulate, chain

import numpy as np

from .episode_tracer import Transition


def _stack(transitions, transition_class):
    cols = list(zip(*(tuple(t) for t in transitions)))
    return transition_class(*(np.stack([np.asarray(x) for x in col]) for col in cols))


class Trajectory:
    """muax/replay_buffer.py:38-125."""

    def __init__(self, transition_class=Transition):
        self.trajectory, self._transition_weight = [], []
        self.transition_class = transition_class
        self._batched_transitions = None
        self._rows, self._length, self._cum = None, 0, None

    @classmethod
    def from_arrays(cls, obs, a, r, done, Rn, v, pi, w, transition_class=Transition, _checked=False):
        """A whole episode handed over as arrays [T, ...] (the vectorised tracer of muax_amd/vector.py):
        same sampling behaviour as a trajectory filled by `add`, without T Python objects."""
        self = cls(transition_class)
        cols = (obs, a, r, done, Rn, v, pi, w) if _checked else [np.asarray(x) for x in (obs, a, r, done, Rn, v, pi, w)]
        if not _checked and len({len(c) for c in cols}) != 1:
            raise ValueError("Trajectory.from_arrays: all fields need the same leading length")
        self._rows = cols  # the Transition views are built on first use: most episodes of a big collection
        self._length = len(cols[0])  # are dropped by the buffer before anything samples them
        self._transition_weight = None
        return self

    def add(self, transition):
        if self._rows is not None:
            raise ValueError("an array-backed trajectory is complete: it cannot be extended")
        self.trajectory.append(transition)
        self._transition_weight.append(transition.w)

    def _materialise(self):
        if self._batched_transitions is None:
            self._transition_weight = self._rows[7].reshape(self._length, -1)[:, 0].tolist()
            self._batched_transitions = self.transition_class(*(c[None] for c in self._rows))

    @property
    def weights(self):
        """Priority weights of the transitions, [T]."""
        if self._rows is not None:
            return self._rows[7].reshape(self._length, -1)[:, 0]
        return np.asarray(self._transition_weight, dtype=np.float64).reshape(len(self), -1)[:, 0]

    @property
    def rewards(self):
        return self._rows[2] if self._rows is not None else np.asarray([t.r for t in self.trajectory])

    def finalize(self):
        """Every field becomes one array [1, T, ...]."""
        if self._rows is not None:
            self._materialise()
            return
        b = _stack(self.trajectory, self.transition_class)
        self._batched_transitions = self.transition_class(*(np.expand_dims(x, 0) for x in b))

    @property
    def batched_transitions(self):
        if self._rows is not None:
            self._materialise()
        return self._batched_transitions

    def _sample_cols(self, num_samples, k_steps):
        """sample() as tuples of arrays [1, k, ...] (no Transition objects).  Same draws as
        random.choices(range(max_idx), weights=w[:max_idx]): the prefix sums it would build are cached."""
        if len(self) <= k_steps:
            return []
        if self._batched_transitions is None:
            self.finalize()
        max_idx = len(self) - k_steps
        if self._cum is None or len(self._cum) != len(self._transition_weight):
            self._cum = list(accumulate(self._transition_weight))
        idxes = random.choices(range(max_idx), cum_weights=self._cum[:max_idx], k=num_samples)
        cols = tuple(self._batched_transitions)
        return [tuple(c[:, i:i + k_steps] for c in cols) for i in idxes]

    def sample(self, num_samples: int = 1, k_steps: int = 5):
        return [self.transition_class(*cols) for cols in self._sample_cols(num_samples, k_steps)]

    def __getitem__(self, index):
        if self._rows is not None:
            return self.transition_class(*(c[index] for c in self._rows))
        return self.trajectory[index]

    def __len__(self):
        return self._length if self._rows is not None else len(self.trajectory)

    def __repr__(self):
        return f"{type(self)}(len={len(self)})"


class BaseReplayBuffer:
    """muax/replay_buffer.py:122-147: the interface fit() uses (capacity, add, sample, clear, len, bool)."""

    @property
    def capacity(self):
        raise NotImplementedError

    def add(self, transition_batch):
        raise NotImplementedError

    def sample(self, batch_size=32):
        raise NotImplementedError

    def clear(self):
        raise NotImplementedError

    def __len__(self):
        raise NotImplementedError

    def __bool__(self):
        return len(self) > 0


class TrajectoryReplayBuffer(BaseReplayBuffer):
    """muax/replay_buffer.py:161-262: ring buffer of trajectories."""

    def __init__(self, capacity, random_seed=None, transition_class=Transition):
        self._capacity = int(capacity)
        random.seed(random_seed)
        self._random_state = random.getstate()
        self.transition_class = transition_class
        self.clear()

    @property
    def capacity(self):
        return self._capacity

    def add(self, trajectory, w=1.):
        self._storage.append(trajectory)
        self._trajectory_weight.append(w)
        self._view = None

    def sample(self, batch_size=32, num_trajectory: int = None, k_steps: int = 5, sample_per_trajectory: int = 1):
        if batch_size is None and num_trajectory is None:
            raise ValueError("Either num_trajectory or batch_size need to be given.")
        elif batch_size is not None and num_trajectory is None:
            num_trajectory, sample_per_trajectory = batch_size, 1
        if self._view is None:  # list copy (a deque indexes in O(n)) + the prefix sums random.choices would build
            self._view = (list(self._storage), list(accumulate(self._trajectory_weight)))
        random.setstate(self._random_state)
        trajectories = random.choices(self._view[0], cum_weights=self._view[1], k=num_trajectory)
        batch = list(chain.from_iterable(t._sample_cols(sample_per_trajectory, k_steps) for t in trajectories))
        self._random_state = random.getstate()
        return self.transition_class(*(np.concatenate(c) for c in zip(*batch)))

    def clear(self):
        self._storage = deque([], maxlen=self.capacity)
        self._trajectory_weight = deque([], maxlen=self.capacity)
        self._view = None

    def __len__(self):
        return len(self._storage)

    def __bool__(self):
        return bool(len(self))

    def __iter__(self):
        return iter(self._storage)
