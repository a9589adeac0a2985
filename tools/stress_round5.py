"""Randomised parity stress of round 5's on-demand instances (not part of the suite): 9 .. 16 actions and 100 .. 255
simulations (FusedCfg::LONG: root paths in HBM), MuZero policy with / without tie-break noise and Gumbel MuZero with
both qtransforms, depth cuts, invalid-action masks, weight scales, support sizes -- every tree array against the C
oracle.  A small fixed set of shapes (each is one on-demand build, ~3 s), many random cases per shape.
    python tools/stress_round5.py [cases] [seed]"""
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
from muax_amd import MuZeroSearch, SearchConfig, _jit  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2025
rng = np.random.default_rng(seed)
# (A, E, obs_dim, largest S of the instance): wide action sets, long searches, both at once where an instance exists
shapes = [(9, 8, 4, 50), (12, 8, 5, 50), (16, 8, 4, 50), (13, 16, 6, 63), (16, 32, 8, 50), (10, 8, 4, 100),
          (2, 8, 4, 160), (2, 8, 4, 255), (4, 32, 8, 200), (6, 8, 6, 160), (3, 8, 4, 128), (4, 8, 5, 200),
          (4, 20, 6, 50), (3, 17, 5, 100), (2, 50, 8, 40)]  # (embeddings above 16 that are no multiple of 8)
bad = total = 0
for c in range(n):
    A, E, obs_dim, smax = shapes[rng.integers(len(shapes))]
    nmax = _jit.plan(A, E, 21, smax)[1]
    S = int(rng.integers(max(1, (nmax * 2) // 3), smax + 1))  # simulation counts that plan onto the same instance mostly
    B = int(rng.integers(1, 70))
    policy = "gumbel" if rng.random() < 0.35 else "muzero"
    tiebreak = policy == "muzero" and bool(rng.integers(2))
    max_depth = None if rng.random() < 0.6 else int(rng.integers(1, min(S, 40) + 1))
    support = int(rng.integers(8, 16)) if (policy == "muzero" and rng.random() < 0.5) else 10  # (the Gumbel oracle helper: 10)
    case = make_case(oracle, 5000 + c + 100003 * abs(seed - 2025), B, obs_dim, E, A, S, support=support,
                     invalid_frac=0.3 if rng.random() < 0.4 else 0.0)
    scale = float(rng.choice([0.3, 1.0, 3.0]))
    case["w"] = {k: (v * scale).astype(np.float32) if k.endswith(("w1", "w2")) else v for k, v in case["w"].items()}
    key = [int(rng.integers(2 ** 31)), int(rng.integers(2 ** 31))]
    if not _jit.ensure_instance(A, E, case["F"], S):
        print(f"no instance for A={A} E={E} S={S}")
        continue
    total += 1
    inv = None if case["invalid"] is None else torch.from_numpy(case["invalid"])
    try:
        if policy == "muzero":
            temperature = float(rng.choice([0.0, 0.5, 1.0]))
            s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=tiebreak, max_depth=max_depth))
            s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, obs_dim, support, 0.99)
            out = s.act_mlp(torch.from_numpy(case["obs"]), key, dirichlet_noise=torch.from_numpy(case["noise"]),
                            invalid_actions=inv, temperature=temperature, gumbel=torch.from_numpy(case["gumbel"]), with_tree=True)
            torch.cuda.synchronize()
            ref = tp._oracle(oracle, case, tiebreak, key, max_depth=max_depth or 0, temperature=temperature)
        else:
            qt = str(rng.choice(["qtransform_completed_by_mix_value", "qtransform_by_parent_and_siblings"]))
            maxc = int(rng.choice([2, 5, 16]))
            s = MuZeroSearch(B, SearchConfig(A, S, E, policy="gumbel", qtransform=qt, max_num_considered_actions=maxc,
                                             tiebreak=False, max_depth=max_depth))
            s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, obs_dim, support, 0.99)
            out = s.act_mlp(torch.from_numpy(case["obs"]), key, gumbel=torch.from_numpy(case["gumbel"]), invalid_actions=inv,
                            with_tree=True)
            torch.cuda.synchronize()
            ref = tp._gumbel_oracle_act(oracle, case, key, 1 if qt.endswith("mix_value") else 0, maxc, gumbel=case["gumbel"],
                                        max_depth=max_depth or 0)
        tp._compare(ref, s, out)
        s.close()
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH case {c}: A={A} E={E} support={support} S={S} B={B} policy={policy} tb={tiebreak} md={max_depth} "
              f"scale={scale}: {str(e)[:200]}")
print(f"seed {seed}: {total} cases on {len(shapes)} on-demand shapes, {bad} mismatches")
