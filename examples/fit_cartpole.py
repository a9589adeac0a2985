"""The reference README's CartPole recipe, line for line, on muax_amd (README.md:45-98 of bwfbowen/muax):
same nets, tracer, buffer, optimiser schedule and `fit` arguments; only the environment object is passed
in (`examples/cartpole_env.py`) because gymnasium is not in this image.

    python examples/fit_cartpole.py [--episodes 1000] [--steps 10000] [--out gpurun_out/fit_cartpole.json]

Prints one line per tested episode and a JSON summary (wall time, env steps, updates, test returns).
Needs a ROCm GPU: act() and update() run the HIP kernels, there is no CPU path.
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import muax_amd as muax  # noqa: E402
from muax_amd import nn  # noqa: E402
from cartpole_env import CartPole  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--episodes", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=10000, help="max_training_steps")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    support_size, embedding_size, discount, num_actions = 10, 8, 0.99, 2
    full_support_size = int(support_size * 2 + 1)
    repr_fn = nn._init_representation_func(nn.Representation, embedding_size)
    pred_fn = nn._init_prediction_func(nn.Prediction, num_actions, full_support_size)
    dy_fn = nn._init_dynamic_func(nn.Dynamic, embedding_size, num_actions, full_support_size)
    tracer = muax.PNStep(10, discount, 0.5)
    buffer = muax.TrajectoryReplayBuffer(500)
    gradient_transform = muax.model.optimizer(init_value=0.02, peak_value=0.02, end_value=0.002,
                                              warmup_steps=5000, transition_steps=5000)
    model = muax.MuZero(repr_fn, pred_fn, dy_fn, policy="muzero", discount=discount,
                        optimizer=gradient_transform, support_size=support_size)

    metrics = []
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as tmp:
        model_path = muax.fit(model, env=CartPole(seed=args.seed), test_env=CartPole(seed=args.seed + 1),
                              max_episodes=args.episodes, max_training_steps=args.steps, tracer=tracer,
                              buffer=buffer, k_steps=10, sample_per_trajectory=1, num_trajectory=32,
                              model_save_path=tmp, save_name="cartpole_model_params", random_seed=args.seed,
                              metrics=metrics)
        wall = time.perf_counter() - t0
        model.load(model_path)
        t1 = time.perf_counter()
        final = muax.test(model, CartPole(seed=args.seed + 2), muax.prng.PRNGKey(0), num_simulations=50,
                          num_test_episodes=20)
        test_wall = time.perf_counter() - t1
    for row in metrics:
        if "test_G" in row:
            print(f"episode {row['episode']:4d}  updates {row['training_step']:6d}  G {row['G']:6.1f}  "
                  f"loss {row['loss']:.4f}  test_G {row['test_G']:.1f}", flush=True)
    summary = {"recipe": "reference README CartPole (fit, S=50, k_steps=10, 32x1 batch, 50 updates/episode)",
               "episodes": len(metrics), "updates": metrics[-1]["training_step"] if metrics else 0,
               "env_steps_trained_episodes": int(sum(r["G"] for r in metrics)), "fit_wall_s": round(wall, 1),
               "test_G_curve": [[r["episode"], r["test_G"]] for r in metrics if "test_G" in r],
               "best_checkpoint_test_G_20_episodes": final, "final_test_wall_s": round(test_wall, 1)}
    print(json.dumps(summary), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
