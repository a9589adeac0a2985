"""Trajectory replay of the reference (muax/replay_buffer.py:38-262) on NumPy: episodes are kept whole,
a training sample is k consecutive transitions drawn inside an episode with the transitions' priority
weights, episodes are drawn with their mean weight.  Batches come out as Transition([B, L, ...])."""
from __future__ import annotations

import random
from collections import deque
from itertools import chain

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

    def add(self, transition):
        self.trajectory.append(transition)
        self._transition_weight.append(transition.w)

    def finalize(self):
        """Every field becomes one array [1, T, ...]."""
        b = _stack(self.trajectory, self.transition_class)
        self._batched_transitions = self.transition_class(*(np.expand_dims(x, 0) for x in b))

    @property
    def batched_transitions(self):
        return self._batched_transitions

    def sample(self, num_samples: int = 1, k_steps: int = 5):
        if len(self) <= k_steps:
            return []
        max_idx = len(self) - k_steps
        idxes = random.choices(range(max_idx), weights=self._transition_weight[:max_idx], k=num_samples)
        if self._batched_transitions is None:
            self.finalize()
        return [self._batched_transitions[:, i:i + k_steps] for i in idxes]

    def __getitem__(self, index):
        return self.trajectory[index]

    def __len__(self):
        return len(self.trajectory)

    def __repr__(self):
        return f"{type(self)}(len={len(self)})"


class TrajectoryReplayBuffer:
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

    def sample(self, batch_size=32, num_trajectory: int = None, k_steps: int = 5, sample_per_trajectory: int = 1):
        if batch_size is None and num_trajectory is None:
            raise ValueError("Either num_trajectory or batch_size need to be given.")
        elif batch_size is not None and num_trajectory is None:
            num_trajectory, sample_per_trajectory = batch_size, 1
        random.setstate(self._random_state)
        trajectories = random.choices(self._storage, weights=self._trajectory_weight, k=num_trajectory)
        batch = list(chain.from_iterable(t.sample(num_samples=sample_per_trajectory, k_steps=k_steps)
                                         for t in trajectories))
        self._random_state = random.getstate()
        cols = list(zip(*(tuple(t) for t in batch)))
        return self.transition_class(*(np.vstack(c) for c in cols))

    def clear(self):
        self._storage = deque([], maxlen=self.capacity)
        self._trajectory_weight = deque([], maxlen=self.capacity)

    def __len__(self):
        return len(self._storage)

    def __bool__(self):
        return bool(len(self))

    def __iter__(self):
        return iter(self._storage)
