"""Randomised parity stress (not part of the suite): the fused kernel against the C oracle over many seeds,
shapes, simulation counts, depth cuts, invalid-action masks, weight scales.  Run on the GPU box:
    python tools/stress_parity.py [cases] [seed] [compact]
`compact`: launches with more workgroups than CUs (4097 .. 12000 roots) on the two shapes that have a compact-record
instance -- CartPole's (paths in HBM) and LunarLander's packed one (byte-sized child fields, first-layer matrices in
LDS) -- support sizes 8 .. 15."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_trees_equal, make_case  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
compact = len(sys.argv) > 3 and sys.argv[3] == "compact"
rng = np.random.default_rng(seed)
shapes = [(2, 8, 4), (4, 32, 8)] if compact else [(2, 8, 4), (3, 8, 6), (4, 8, 5), (2, 16, 4), (4, 16, 8), (4, 32, 8), (2, 10, 4), (4, 10, 8), (6, 8, 6),
          (8, 8, 6), (2, 32, 8), (4, 64, 8), (2, 64, 8)]
wide_support = {(2, 8), (4, 32), (6, 8)}  # instances with four support slots (support_size 16..31)
bad = 0
for c in range(n):
    A, E, obs_dim = shapes[rng.integers(len(shapes))]
    S = int(rng.integers(1, 51))
    B = int(rng.integers(4097, 12000)) if compact else int(rng.integers(1, 200))
    tiebreak = bool(rng.integers(2))
    max_depth = None if rng.random() < 0.6 else int(rng.integers(1, S + 1))
    support = int(rng.integers(8, 32 if (A, E) in wide_support and not compact else 16)) if rng.random() < 0.5 else 10
    case = make_case(oracle, 1000 + c + 100003 * abs(seed - 2024), B, obs_dim, E, A, S, support=support,
                     invalid_frac=0.3 if (A > 2 and rng.random() < 0.4) else 0.0)
    scale = float(rng.choice([0.3, 1.0, 3.0]))  # sharper / flatter heads: other depths and tie patterns
    case["w"] = {k: (v * scale).astype(np.float32) if k.endswith(("w1", "w2")) else v for k, v in case["w"].items()}
    temperature = float(rng.choice([0.0, 0.5, 1.0]))
    key = [int(rng.integers(2 ** 31)), int(rng.integers(2 ** 31))]
    try:
        s, out = tp._fused(case, tiebreak, key, max_depth=max_depth, temperature=temperature)
        if compact:  # thousands of roots: the oracle on all host threads
            mlp = oracle.Mlp(case["w"], case["obs_dim"], case["E"], case["A"], case["F"], support_size=case["support"])
            ref = oracle.act_mlp(mlp, oracle.SearchCfg(case["S"], max_depth=max_depth or 0, tiebreak=int(tiebreak)),
                                 case["obs"], key, case["noise"], 0.25, case["invalid"], temperature, case["gumbel"],
                                 nthreads=len(os.sched_getaffinity(0)))
        else:
            ref = tp._oracle(oracle, case, tiebreak, key, max_depth=max_depth or 0, temperature=temperature)
        tp._compare(ref, s, out)
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH case {c}: A={A} E={E} support={support} S={S} B={B} tb={tiebreak} md={max_depth} scale={scale} T={temperature}: {str(e)[:200]}")
print(f"seed {seed}: {n} cases, {bad} mismatches; mean depth of the last case {float(s.depth_sum.float().mean()) / S:.2f}")
