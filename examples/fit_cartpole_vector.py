"""CartPole on a vector environment: `muax.fit_vector` steps N cart-poles as arrays with ONE batched
act() (N roots, one fused HIP launch) per step; whole episodes are traced as array operations
(muax_amd/vector.py) and the reference's update loop runs on the fused training kernel.

    python examples/fit_cartpole_vector.py [--envs 1024] [--steps 128] [--iterations 100] [--updates 300]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import muax_amd as muax  # noqa: E402
from muax_amd import nn  # noqa: E402
from cartpole_env import VectorCartPole  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=128, help="lock-step env steps per iteration")
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--updates", type=int, default=300, help="updates per iteration")
    ap.add_argument("--buffer", type=int, default=4000, help="trajectories kept")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--traj-weight", default="sum", choices=["mean", "sum"])
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    support_size, embedding_size, discount, num_actions = 10, 8, 0.99, 2
    net = muax.create_muzero_network(nn.Representation, nn.Prediction, nn.Dynamic, embedding_dim=embedding_size,
                                     num_actions=num_actions, full_support_size=2 * support_size + 1)
    total = args.iterations * args.updates
    opt = muax.model.optimizer(init_value=0.02, peak_value=0.02, end_value=0.002, warmup_steps=total // 6,
                               transition_steps=total // 6)
    model = muax.MuZero(net, discount=discount, optimizer=opt, support_size=support_size)
    metrics, marks = [], [time.perf_counter()]

    class Timed(list):  # wall-clock mark per iteration
        def append(self, row):
            marks.append(time.perf_counter())
            row["wall_s"] = round(marks[-1] - marks[0], 2)
            super().append(row)

    metrics = Timed()
    muax.fit_vector(model, VectorCartPole(args.envs, seed=args.seed), VectorCartPole(16, seed=10_000 + args.seed), n_step=10,
                    gamma=discount, alpha=0.5, buffer=muax.TrajectoryReplayBuffer(args.buffer),
                    iterations=args.iterations, steps_per_iteration=args.steps, k_steps=10, num_trajectory=32,
                    sample_per_trajectory=1, num_update_per_iteration=args.updates, max_training_steps=total,
                    test_interval=5, random_seed=args.seed, metrics=metrics, trajectory_weight=args.traj_weight)
    wall = time.perf_counter() - marks[0]
    for r in metrics:
        print(f"iteration {r['iteration']:3d}  t {r['wall_s']:6.1f}s  episodes {r['episodes']:5d}  mean G {r['G']:6.1f}  "
              f"loss {r.get('loss', float('nan')):.4f}  updates {r['training_step']:6d}"
              + (f"  test_G {r['test_G']:.1f}" if "test_G" in r else ""), flush=True)
    env_steps = int(sum(r["env_steps"] for r in metrics))
    summary = {"recipe": f"fit_vector, {args.envs} cart-poles as arrays, {args.steps} steps/iteration, S=50, k_steps=10, "
                         f"buffer {args.buffer}, trajectory weight {args.traj_weight}",
               "iterations": len(metrics), "updates": metrics[-1]["training_step"], "env_steps": env_steps,
               "wall_s": round(wall, 1), "env_steps_per_s_whole_loop": round(env_steps / wall),
               "env_steps_per_s_acting_half": round(env_steps / sum(r["collect_s"] for r in metrics)),
               "test_G_curve": [[r["iteration"], r["test_G"]] for r in metrics if "test_G" in r]}
    print(json.dumps(summary), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
