"""CartPole with the acting half batched: `muax.fit_batched` steps N environments in lock step with ONE
batched act() (N roots, one fused HIP launch) per step, then runs the reference's update loop.  Same nets,
tracer, buffer and optimiser as examples/fit_cartpole.py.

    python examples/fit_cartpole_batched.py [--envs 64] [--iterations 60] [--updates 500]
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
from cartpole_env import CartPole  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--iterations", type=int, default=60)
    ap.add_argument("--updates", type=int, default=500, help="updates per iteration")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()

    support_size, embedding_size, discount, num_actions = 10, 8, 0.99, 2
    net = muax.create_muzero_network(nn.Representation, nn.Prediction, nn.Dynamic, embedding_dim=embedding_size,
                                     num_actions=num_actions, full_support_size=2 * support_size + 1)
    total = args.iterations * args.updates
    opt = muax.model.optimizer(init_value=0.02, peak_value=0.02, end_value=0.002, warmup_steps=total // 6,
                               transition_steps=total // 6)
    model = muax.MuZero(net, discount=discount, optimizer=opt, support_size=support_size)
    envs = [CartPole(seed=args.seed + i) for i in range(args.envs)]
    metrics = []
    t0 = time.perf_counter()
    muax.fit_batched(model, envs, CartPole(seed=10_000 + args.seed), tracer_factory=lambda: muax.PNStep(10, discount, 0.5),
                     buffer=muax.TrajectoryReplayBuffer(500), iterations=args.iterations, k_steps=10,
                     num_trajectory=32, sample_per_trajectory=1, num_update_per_iteration=args.updates,
                     max_training_steps=total, test_interval=5, random_seed=args.seed, metrics=metrics)
    wall = time.perf_counter() - t0
    for r in metrics:
        print(f"iteration {r['iteration']:3d}  env_steps {r['env_steps']:6d}  mean G {r['G']:6.1f}  "
              f"loss {r.get('loss', float('nan')):.4f}  updates {r['training_step']:6d}"
              + (f"  test_G {r['test_G']:.1f}" if "test_G" in r else ""), flush=True)
    summary = {"recipe": f"fit_batched, {args.envs} CartPole envs in lock step, S=50, k_steps=10, 32x1 batch",
               "iterations": len(metrics), "updates": metrics[-1]["training_step"],
               "env_steps": int(sum(r["env_steps"] for r in metrics)), "wall_s": round(wall, 1),
               "test_G_curve": [[r["iteration"], r["test_G"]] for r in metrics if "test_G" in r]}
    print(json.dumps(summary), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main()
