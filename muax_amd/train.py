"""Callers of the hot path: the act/step inner loops of muax.fit and muax.test
(muax/train.py:16-23,148-201; muax/test.py:5-48) without the replay/learning side.

What the reference's loops pin and this module reproduces: one `split` of the key per environment
step, the act() call contract, the temperature schedule, greedy evaluation at temperature 0.  The
batched variants (`rollout_batched`) are what the metric measures: B environments stepped in lock step,
one batched act() per step.  The learning half of fit() (tracer, replay buffer, loss, optimiser) is the
next tier.
"""
from __future__ import annotations

import numpy as np

from . import prng


def _temperature_fn(max_training_steps, training_steps):
    r"""muax/train.py:16-23: 1.0 / 0.5 / 0.25 below 50% / 75% / otherwise of max_training_steps."""
    if training_steps < 0.5 * max_training_steps:
        return 1.0
    elif training_steps < 0.75 * max_training_steps:
        return 0.5
    else:
        return 0.25


def rollout(model, env, key, num_simulations: int = 50, temperature: float = 1.0, max_steps=None):
    """One episode exactly as the inner loop of muax/train.py:153-170: returns the list of
    (obs, a, r, done, v, pi) transitions and the advanced key."""
    obs, info = env.reset()
    steps = max_steps if max_steps is not None else env.spec.max_episode_steps
    traj = []
    for t in range(steps):
        key, subkey = prng.split(key)
        a, pi, v = model.act(subkey, obs, with_pi=True, with_value=True, obs_from_batch=False,
                             num_simulations=num_simulations, temperature=temperature)
        obs_next, r, done, truncated, info = env.step(a)
        traj.append((obs, a, r, done or truncated, v, pi))
        if done or truncated:
            break
        obs = obs_next
    return traj, key


def test(model, env, key, num_simulations: int, num_test_episodes: int = 10, max_steps=None):
    """muax/test.py:5-48: greedy evaluation (temperature=0.), mean undiscounted episode return."""
    total_rewards = np.zeros(num_test_episodes)
    for episode in range(num_test_episodes):
        obs, info = env.reset()
        steps = max_steps if max_steps is not None else env.spec.max_episode_steps
        for t in range(steps):
            key, subkey = prng.split(key)
            a = model.act(subkey, obs, num_simulations=num_simulations, temperature=0.)
            obs, r, done, truncated, info = env.step(a)
            total_rewards[episode] += r
            if done or truncated:
                break
    return float(np.mean(total_rewards))


def rollout_batched(model, envs, key, steps: int, num_simulations: int = 50, temperature: float = 1.0):
    """B environments in lock step, one batched act(obs_from_batch=True) per step: the capability the
    reference exposes (muax/model.py:160-161,173-174) but never exercises, and the metric's unit.
    `envs` needs reset() -> obs [B, ...] and step(actions [B]) -> (obs, reward, done) arrays.
    Returns total env-steps taken and the per-env returns."""
    obs = envs.reset()
    returns = np.zeros(obs.shape[0])
    for t in range(steps):
        key, subkey = prng.split(key)
        a = model.act(subkey, obs, obs_from_batch=True, num_simulations=num_simulations, temperature=temperature)
        obs, r, done = envs.step(a)
        returns += r
    return steps * obs.shape[0], returns
