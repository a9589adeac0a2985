"""Callers of the hot path: the act/step inner loops of muax.fit and muax.test
(muax/train.py:16-23,148-201; muax/test.py:5-48) without the replay/learning side.

What the reference's loops pin and this module reproduces: one `split` of the key per environment
step, the act() call contract, the temperature schedule, greedy evaluation at temperature 0.  The
batched variants (`rollout_batched`) are what the metric measures: B environments stepped in lock step,
one batched act() per step.  `fit` is the reference's whole loop (muax/train.py:26-241): acting through the HIP
search, the n-step tracer and trajectory buffer on the host, `model.update` on the fused HIP training kernel.
"""
from __future__ import annotations

import os
import time

import numpy as np

from . import prng
from .episode_tracer import PNStep
from .replay_buffer import Trajectory, TrajectoryReplayBuffer


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


def test(model, env, key, num_simulations: int, num_test_episodes: int = 10, random_seed=None, max_steps=None):
    """muax/test.py:5-48: greedy evaluation (temperature=0.), mean undiscounted episode return.  `random_seed` goes
    to env.reset(seed=...) at every episode as in the reference (None: the environment's own stream)."""
    total_rewards = np.zeros(num_test_episodes)
    for episode in range(num_test_episodes):
        obs, info = env.reset() if random_seed is None else env.reset(seed=random_seed)
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


def _episode(model, env, tracer, key, num_simulations, temperature, reset_kwargs=None, on_transition=None):
    """Acting half of one fit() episode (muax/train.py:150-173,181-204): one key split and one act() per
    environment step, tracer -> trajectory."""
    obs, info = env.reset(**(reset_kwargs or {}))
    tracer.reset()
    trajectory = Trajectory()
    for t in range(env.spec.max_episode_steps):
        key, subkey = prng.split(key)
        a, pi, v = model.act(subkey, obs, with_pi=True, with_value=True, obs_from_batch=False,
                             num_simulations=num_simulations, temperature=temperature)
        obs_next, r, done, truncated, info = env.step(a)
        tracer.add(obs, a, r, done or truncated, v=v, pi=pi)
        while tracer:
            trans = tracer.pop()
            trajectory.add(trans)
            if on_transition is not None:
                on_transition(trans)
        if done or truncated:
            break
        obs = obs_next
    trajectory.finalize()
    return trajectory, key


def fit(model, env_id=None, env=None, test_env=None, tracer=None, buffer=None, max_episodes: int = 1000,
        test_interval: int = 10, num_test_episodes: int = 10, max_training_steps: int = 10000,
        save_every_n_epochs: int = 1, num_simulations: int = 50, k_steps: int = 10, buffer_warm_up: int = 128,
        num_trajectory: int = 32, sample_per_trajectory: int = 10, name: str = None, tensorboard_dir=None,
        model_save_path=None, save_name=None, random_seed: int = 42, temperature_fn=_temperature_fn,
        log_all_metrics=False, num_update_per_episode: int = 50, metrics=None):
    """muax/train.py:26-241 with the same arguments, defaults and return value (the path of the best
    checkpoint).  The TrainMonitor / tensorboard side of the reference is not rebuilt (`name`,
    `tensorboard_dir`, `log_all_metrics` are accepted and ignored); pass a list as `metrics` to receive one
    dict per episode (`loss`, `G`, `training_step`, `test_G`)."""
    if env_id is None and env is None:
        raise ValueError("You must provide either `env_id` or `env`.")
    if env_id is not None and env is not None:
        raise ValueError("You can only provide either `env_id` or `env`, not both.")
    if env is None:
        try:
            import gymnasium as gym
        except ImportError:
            try:
                import gym
            except ImportError:
                raise ValueError("`env_id` needs gymnasium (or gym), which is not installed: pass `env` and `test_env`") from None
        env = gym.make(env_id, render_mode="rgb_array")
        if test_env is None:
            test_env = gym.make(env_id, render_mode="rgb_array")
    if test_env is None:
        raise ValueError("You must provide `test_env` when using a custom `env`.")
    tracer = tracer if tracer is not None else PNStep(50, 0.997, 0.5)
    buffer = buffer if buffer is not None else TrajectoryReplayBuffer(500)
    save_name = save_name or "model_params"
    model_dir = model_save_path or os.path.join("models", time.strftime("%Y-%m-%d_%H-%M-%S"))

    sample_input = np.expand_dims(np.asarray(env.observation_space.sample()), 0).astype(float)
    key = prng.PRNGKey(random_seed)
    key, test_key, subkey = prng.split(key, 3)
    model.init(subkey, sample_input)

    training_step, best_test_G, model_path = 0, -float("inf"), None
    while len(buffer) < buffer_warm_up:  # buffer warm up (muax/train.py:148-173)
        temperature = temperature_fn(max_training_steps=max_training_steps, training_steps=training_step)
        trajectory, key = _episode(model, env, tracer, key, num_simulations, temperature)
        if len(trajectory) >= k_steps:
            buffer.add(trajectory, trajectory.batched_transitions.w.mean())

    for ep in range(max_episodes):
        temperature = temperature_fn(max_training_steps=max_training_steps, training_steps=training_step)
        G = [0.0]

        def on_transition(trans):
            G[0] += float(trans.r)

        trajectory, key = _episode(model, env, tracer, key, num_simulations, temperature,
                                   {"seed": random_seed}, on_transition)
        if len(trajectory) >= k_steps:
            buffer.add(trajectory, trajectory.batched_transitions.w.mean())
        train_loss = 0.0
        for _ in range(num_update_per_episode):
            batch = buffer.sample(num_trajectory=num_trajectory, sample_per_trajectory=sample_per_trajectory,
                                  k_steps=k_steps)
            train_loss += model.update(batch)["loss"]
            training_step += 1
        train_loss /= num_update_per_episode
        row = {"episode": ep, "loss": train_loss, "G": G[0], "training_step": training_step}
        if ep % save_every_n_epochs == 0:
            folder = os.path.join(model_dir, f"epoch_{ep:04d}_loss_{train_loss:.8f}")
            os.makedirs(folder, exist_ok=True)
            cur_path = os.path.join(folder, save_name)
            model.save(cur_path)
            if not model_path:
                model_path = cur_path
        if training_step >= max_training_steps:
            if metrics is not None:
                metrics.append(row)
            return model_path
        if ep % test_interval == 0:
            test_G = test(model, test_env, test_key, num_simulations=num_simulations,
                          num_test_episodes=num_test_episodes)
            row["test_G"] = test_G
            if test_G >= best_test_G:
                best_test_G = test_G
                folder = os.path.join(model_dir, f"epoch_{ep:04d}_test_G_{test_G:.8f}")
                os.makedirs(folder, exist_ok=True)
                model_path = os.path.join(folder, save_name)
                model.save(model_path)
        if metrics is not None:
            metrics.append(row)
    return model_path


def collect_batched(model, envs, tracers, key, num_simulations: int = 50, temperature: float = 1.0, max_steps=None):
    """Acting half of fit() for a LIST of gym-style environments stepped in lock step: one batched
    act(obs_from_batch=True) per step for all of them -- the use the batched HIP search is built for; the
    reference's fit() acts on one environment at a time.  Each environment keeps its own tracer
    (muax/train.py:150-173 per environment).  Returns the finished trajectories and the advanced key."""
    n = len(envs)
    obs = [e.reset()[0] for e in envs]
    for t in tracers:
        t.reset()
    trajs = [Trajectory() for _ in range(n)]
    live = list(range(n))
    steps = max_steps if max_steps is not None else envs[0].spec.max_episode_steps
    for _ in range(steps):
        if not live:
            break
        key, subkey = prng.split(key)
        batch = np.stack([np.asarray(obs[i], np.float32) for i in live])
        a, pi, v = model.act(subkey, batch, with_pi=True, with_value=True, obs_from_batch=True,
                             num_simulations=num_simulations, temperature=temperature)
        still = []
        for row, i in enumerate(live):
            obs_next, r, done, truncated, info = envs[i].step(int(a[row]))
            tracers[i].add(obs[i], int(a[row]), r, done or truncated, v=float(v[row]), pi=pi[row:row + 1])
            while tracers[i]:
                trajs[i].add(tracers[i].pop())
            if not (done or truncated):
                obs[i] = obs_next
                still.append(i)
        live = still
    for tr in trajs:
        tr.finalize()
    return trajs, key


def fit_batched(model, envs, test_env, tracer_factory=None, buffer=None, iterations: int = 100,
                num_simulations: int = 50, k_steps: int = 10, num_trajectory: int = 32, sample_per_trajectory: int = 10,
                num_update_per_iteration: int = 50, max_training_steps: int = 10000, test_interval: int = 10,
                num_test_episodes: int = 10, random_seed: int = 42, temperature_fn=_temperature_fn, metrics=None):
    """fit() with the acting half batched over `envs` (see collect_batched); losses, buffer sampling, the
    temperature schedule and the greedy test are the reference's (muax/train.py:175-241).  Every iteration:
    one lock-step episode of all environments -> buffer, `num_update_per_iteration` updates."""
    tracer_factory = tracer_factory or (lambda: PNStep(50, 0.997, 0.5))
    buffer = buffer if buffer is not None else TrajectoryReplayBuffer(500)
    tracers = [tracer_factory() for _ in envs]
    sample_input = np.expand_dims(np.asarray(envs[0].observation_space.sample()), 0).astype(float)
    key = prng.PRNGKey(random_seed)
    key, test_key, subkey = prng.split(key, 3)
    model.init(subkey, sample_input)
    training_step = 0
    for it in range(iterations):
        temperature = temperature_fn(max_training_steps=max_training_steps, training_steps=training_step)
        trajs, key = collect_batched(model, envs, tracers, key, num_simulations, temperature)
        for tr in trajs:
            if len(tr) >= k_steps:
                buffer.add(tr, tr.batched_transitions.w.mean())
        row = {"iteration": it, "env_steps": int(sum(len(t) for t in trajs)),
               "G": float(np.mean([float(np.sum(t.batched_transitions.r)) for t in trajs]))}
        if len(buffer):
            loss = 0.0
            for _ in range(num_update_per_iteration):
                loss += model.update(buffer.sample(num_trajectory=num_trajectory,
                                                   sample_per_trajectory=sample_per_trajectory, k_steps=k_steps))["loss"]
                training_step += 1
            row["loss"] = loss / num_update_per_iteration
        row["training_step"] = training_step
        if it % test_interval == 0:
            row["test_G"] = test(model, test_env, test_key, num_simulations=num_simulations,
                                 num_test_episodes=num_test_episodes)
        if metrics is not None:
            metrics.append(row)
        if training_step >= max_training_steps:
            break
    return model
