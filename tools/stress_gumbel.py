"""Randomised parity stress of the fused kernel's Gumbel MuZero modes against the C oracle (not part of the suite):
shapes, simulation counts, considered-action counts, both q-transforms, depth cuts, invalid-action masks.
    python tools/stress_gumbel.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_case  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import test_gpu_parity as tp  # noqa: E402
from muax_amd import MuZeroSearch, SearchConfig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
shapes = [(2, 8, 4), (3, 8, 6), (4, 8, 5), (2, 16, 4), (4, 16, 8), (4, 32, 8), (6, 8, 6), (8, 8, 6), (4, 64, 8)]
bad = 0
for c in range(n):
    A, E, obs_dim = shapes[rng.integers(len(shapes))]
    S = int(rng.integers(1, 51))
    B = int(rng.integers(1, 160))
    qt = ["qtransform_by_parent_and_siblings", "qtransform_completed_by_mix_value"][int(rng.integers(2))]
    maxc = int(rng.integers(1, 17))
    max_depth = None if rng.random() < 0.6 else int(rng.integers(1, S + 1))
    case = make_case(oracle, 5000 + c + 7919 * seed, B, obs_dim, E, A, S,
                     invalid_frac=0.3 if (A > 2 and rng.random() < 0.4) else 0.0)
    scale = float(rng.choice([0.3, 1.0, 3.0]))
    case["w"] = {k: (v * scale).astype(np.float32) if k.endswith(("w1", "w2")) else v for k, v in case["w"].items()}
    key = [int(rng.integers(2 ** 31)), int(rng.integers(2 ** 31))]
    try:
        s = MuZeroSearch(B, SearchConfig(A, S, E, policy="gumbel", qtransform=qt, max_num_considered_actions=maxc,
                                         max_depth=max_depth, tiebreak=False))
        s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], 10, 0.99)
        out = s.act_mlp(torch.from_numpy(case["obs"]), key,
                        invalid_actions=None if case["invalid"] is None else torch.from_numpy(case["invalid"]),
                        with_tree=True)
        torch.cuda.synchronize()
        ref = tp._gumbel_oracle_act(oracle, case, key, 1 if qt.endswith("mix_value") else 0, maxc, max_depth=max_depth or 0)
        tp._compare(ref, s, out)
        s.close()
    except AssertionError as e:
        bad += 1
        print("MISMATCH", c, (A, E, S, B, qt, maxc, max_depth, scale), str(e)[:200])
print(f"seed {seed}: {n} gumbel cases, {bad} mismatches")
