"""Randomised stress of the step-wise kernels (cached decisions, mz_step_jump.cuh) against the C oracle: random
action counts up to 64, simulation counts, depth cuts, tie-break on/off, masks, weight scales (deep chains /
broad trees / many exact ties).  Not part of the suite:   python tools/stress_stepwise.py [cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_case  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
bad = 0
for c in range(n):
    A = int(rng.choice([1, 2, 3, 5, 8, 18, 33, 64]))
    E = int(rng.choice([8, 24, 40]))
    S = int(rng.integers(1, 90)) if rng.random() < 0.8 else int(rng.integers(90, 280))  # long searches: paths beyond 64 levels
    B = int(rng.integers(1, 40))
    fused_select = bool(rng.integers(2))  # mzs_expand_backup_select or the two separate calls
    tiebreak = bool(rng.integers(2))
    max_depth = None if rng.random() < 0.5 else int(rng.integers(1, S + 1))
    case = make_case(oracle, 5000 + c + 100003 * abs(seed - 7), B, 6, E, A, S, invalid_frac=0.3 if (A > 2 and rng.random() < 0.4) else 0.0)
    scale = float(rng.choice([0.0, 0.3, 1.0, 4.0]))  # 0: every score ties exactly, the noise decides
    case["w"] = {k: (v * scale).astype(np.float32) for k, v in case["w"].items()}
    try:
        tp._stepwise_vs_oracle(oracle, case, S, tiebreak, max_depth=max_depth, key=(int(rng.integers(2 ** 31)), c),
                                fused_select=fused_select)
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH case {c}: A={A} E={E} S={S} B={B} tb={tiebreak} md={max_depth} scale={scale} fused_select={fused_select}: {str(e)[:160]}")
print(f"seed {seed}: {n} step-wise cases, {bad} mismatches")
