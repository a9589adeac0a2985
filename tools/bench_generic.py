"""Per-act time of the three routes a default-trio act() can take, on one shape: the tuned fused instance, the generic
one-launch search (MZS_FORCE_GENERIC=1 on a handle with allow_generic), and -- for shapes without an instance -- the
generic route alone.   python tools/bench_generic.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import haiku_style_weights  # noqa: E402
from muax_amd import MuZeroSearch, SearchConfig  # noqa: E402
from muax_amd.utils import warm_runtime  # noqa: E402

warm_runtime()  # (the runtime's signal pool grown before anything is timed: tools/diag_stall.py)


def run(B, obs_dim, E, A, S, force_generic):
    w = haiku_style_weights(0, obs_dim, E, A, 21)
    s = MuZeroSearch(B, SearchConfig(A, S, E, tiebreak=True))
    s.set_mlp_weights(w, obs_dim, 10, 0.99)
    s.allow_generic()
    if force_generic:
        os.environ["MZS_FORCE_GENERIC"] = "1"
    else:
        os.environ.pop("MZS_FORCE_GENERIC", None)
    obs = (torch.rand(B, obs_dim) * 2 - 1).cuda()
    noise = torch.distributions.Dirichlet(torch.full((A,), 0.3)).sample((B,)).cuda()
    for i in range(5):
        s.act_mlp(obs, (0, i), dirichlet_noise=noise)
    torch.cuda.synchronize()
    n = 20
    ts = []
    for i in range(n):  # the MEDIAN of per-act times: one host stall of the runtime (tens of ms, tools/diag_stall.py) in 20
        t0 = time.perf_counter()  # acts is otherwise the whole figure
        s.act_mlp(obs, (0, 100 + i), dirichlet_noise=noise)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[n // 2]
    depth = float(s.depth_sum.float().mean()) / S
    s.close()
    os.environ.pop("MZS_FORCE_GENERIC", None)
    return dt, depth


def run_jit(B, obs_dim, E, A, S, gumbel=True):
    """The same shape through an instance built on demand (muax_amd/_jit.py), or None when no instance can exist.
    gumbel=False: an instance planned for the MuZero policy's record alone (round 6)."""
    from muax_amd import _jit
    if _jit.plan(A, E, 21, S, gumbel) is None or not _jit.ensure_instance(A, E, 21, S, gumbel=gumbel):
        return None
    return run(B, obs_dim, E, A, S, False) + (_jit.plan(A, E, 21, S, gumbel),)


def guarded(label, fn):
    """A shape that fails is a ROW of the table (profiles/r05_generic_route.txt committed a raw traceback instead)."""
    try:
        return fn()
    except Exception as e:  # noqa: BLE001
        print(f"{label}: FAILED -- {type(e).__name__}: {str(e)[:200]}")
        return None


if __name__ == "__main__":
    if "--round6" in sys.argv:
        # round 6: (a) instances planned per policy (the MuZero policy's four-word children: more roots per workgroup),
        # per simulation against the metric's listed instance; (b) the generic route where B (S + 1)^2 path words used to
        # exceed the 1 GiB slab (4096 roots x 300 simulations), undivided (8 GiB budget) and in chunks (forced)
        t_ref, _ = run(4096, 4, 8, 2, 50, False)
        print(f"reference: 4096 roots, A=2, E=8, S=50, listed instance {t_ref * 1e3:.3f} ms = {t_ref / 50 * 1e6:.2f} us/sim")
        for (B, od, E, A, S) in ((4096, 4, 8, 2, 160), (4096, 4, 8, 9, 50), (4096, 4, 8, 12, 50), (4096, 4, 8, 16, 50),
                                 (4096, 4, 8, 18, 50)):
            label = f"{B} roots, A={A}, E={E}, S={S}"
            os.environ["MUAX_AMD_JIT"] = "1"
            for gumbel in (True, False):
                r = guarded(label, lambda: run_jit(B, od, E, A, S, gumbel))
                if r is None:
                    print(f"{label}: no {'all-modes' if gumbel else 'MuZero-only'} instance")
                    continue
                t_j, d, pl = r
                print(f"{label} (mean depth {d:.1f}): {'all-modes  ' if gumbel else 'MuZero-only'} instance (NMAX {pl[1]}, "
                      f"{4 * pl[2]} roots/workgroup, paths in {'HBM' if pl[3] else 'LDS'}) {t_j * 1e3:8.3f} ms = "
                      f"{t_j / S * 1e6:6.2f} us/sim = x{(t_j / S) / (t_ref / 50) * 4096 / B:.1f} the listed instance's")
            os.environ["MUAX_AMD_JIT"] = "0"
            r = guarded(label, lambda: run(B, od, E, A, S, True))
            os.environ["MUAX_AMD_JIT"] = "1"
            if r:
                print(f"{label}: generic route {r[0] * 1e3:8.3f} ms = {r[0] / S * 1e6:6.2f} us/sim = "
                      f"x{(r[0] / S) / (t_ref / 50) * 4096 / B:.1f}")
        for budget in (None, "256"):
            if budget:
                os.environ["MZS_JUMP_BUDGET_MB"] = budget
            r = guarded("4096 roots, A=2, E=8, S=300", lambda: run(4096, 4, 8, 2, 300, False))
            if r:
                print(f"4096 roots, A=2, E=8, S=300 (mean depth {r[1]:.1f}): generic only, "
                      f"{'slab budget ' + budget + ' MB (chunks of roots)' if budget else 'undivided (8 GiB slab budget)'} "
                      f"{r[0] * 1e3:8.3f} ms = {r[0] / 300 * 1e6:6.2f} us/sim")
        os.environ.pop("MZS_JUMP_BUDGET_MB", None)
        sys.exit(0)
    if "--round5" in sys.argv:
        # round 5: shapes beyond round 4's instance limits (A <= 8, S <= 127) -- generic one-launch search against the
        # instance now built on demand (A <= 16; 128..255 simulations with the root paths in HBM), per simulation, next
        # to the metric's own instance (4096 x (2, 8) x 50: the reference for "a listed instance's per-simulation cost")
        t_ref, _ = run(4096, 4, 8, 2, 50, False)
        print(f"reference: 4096 roots, A=2, E=8, S=50, listed instance {t_ref * 1e3:.3f} ms = {t_ref / 50 * 1e6:.2f} us/sim")
        for (B, od, E, A, S) in ((4096, 4, 8, 2, 160), (4096, 4, 8, 2, 255), (4096, 4, 8, 9, 50), (4096, 4, 8, 12, 50),
                                 (4096, 4, 8, 16, 50), (4096, 8, 32, 4, 200), (4096, 4, 8, 6, 100), (1024, 4, 8, 16, 50),
                                 (1024, 4, 8, 2, 255)):
            os.environ["MUAX_AMD_JIT"] = "0"
            t_g, d = run(B, od, E, A, S, True)
            os.environ["MUAX_AMD_JIT"] = "1"
            r = run_jit(B, od, E, A, S)
            if r is None:
                print(f"{B} roots, A={A}, E={E}, S={S}: generic {t_g * 1e3:8.3f} ms = {t_g / S * 1e6:6.2f} us/sim | no instance possible")
                continue
            t_j, _, pl = r
            print(f"{B} roots, A={A}, E={E}, S={S} (mean depth {d:.1f}): generic {t_g * 1e3:8.3f} ms = {t_g / S * 1e6:6.2f} us/sim | "
                  f"on-demand instance (NMAX {pl[1]}, {4 * pl[2]} roots/workgroup, paths in {'HBM' if pl[3] else 'LDS'}) "
                  f"{t_j * 1e3:8.3f} ms = {t_j / S * 1e6:6.2f} us/sim = x{(t_j / S) / (t_ref / 50) * 4096 / B:.1f} the listed instance's per-root-simulation cost "
                  f"(generic: x{(t_g / S) / (t_ref / 50) * 4096 / B:.1f})")
        sys.exit(0)
    for (B, od, E, A, S) in ((4096, 4, 8, 2, 50), (8192, 8, 32, 4, 50)):
        t_f, d = run(B, od, E, A, S, False)
        t_g, _ = run(B, od, E, A, S, True)
        print(f"{B} roots, A={A}, E={E}, S={S} (mean depth {d:.1f}): fused instance {t_f * 1e3:8.3f} ms = {t_f / S * 1e6:6.2f} us/sim | "
              f"generic {t_g * 1e3:8.3f} ms = {t_g / S * 1e6:6.2f} us/sim | x{t_g / t_f:.1f}")
    for (B, od, E, A, S) in ((4096, 4, 8, 2, 300), (4096, 4, 8, 18, 50), (1024, 8, 100, 4, 50)):  # (no instance can serve these)
        r = guarded(f"{B} roots, A={A}, E={E}, S={S}", lambda: run(B, od, E, A, S, False))
        if r is None:
            continue
        t_g, d = r
        print(f"{B} roots, A={A}, E={E}, S={S} (mean depth {d:.1f}): generic only {t_g * 1e3:8.3f} ms = {t_g / S * 1e6:6.2f} us/sim")
