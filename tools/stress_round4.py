"""Randomised parity stress of round 4's routes (not part of the suite).  On the GPU box:
    python tools/stress_round4.py generic [cases] [seed]   # mzs_act_mlp's generic one-launch search vs the C oracle
    python tools/stress_round4.py jit [cases] [seed]       # fused instances compiled on demand vs the C oracle
    python tools/stress_round4.py search [cases] [seed]    # one-launch ResNet search vs the per-simulation launches
    python tools/stress_round4.py repr [cases] [seed]      # representation convolutions / residual blocks vs fp64
generic: num_actions 1..64, embeddings 1..120, num_simulations 1..260, support sizes 8..31, both recurrent_pred_on modes,
depth cuts, masks, weight scales (0 = every score ties: the noise decides), MuZero and Gumbel MuZero policies.
search: 1..200 roots (pair mode up to 128), 1..60 simulations, depth cuts, masks, both policies."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_trees_equal, make_case  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import test_gpu_parity as tp  # noqa: E402
import muax_amd as mx  # noqa: E402
from muax_amd import MuZeroSearch, SearchConfig, _jit  # noqa: E402

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
rng = np.random.default_rng(seed)
bad = 0


def run_mlp(case, tiebreak, key, route, max_depth, temperature, pred_on, policy="muzero"):
    cfg = SearchConfig(case["A"], case["S"], case["E"], tiebreak=tiebreak, max_depth=max_depth, policy=policy,
                       qtransform="qtransform_completed_by_mix_value" if policy == "gumbel" else "qtransform_by_parent_and_siblings")
    s = MuZeroSearch(case["B"], cfg)
    s.set_mlp_weights({k: torch.from_numpy(v) for k, v in case["w"].items()}, case["obs_dim"], case["support"], 0.99, pred_on)
    if route == "generic":
        s.allow_generic()
        os.environ["MZS_FORCE_GENERIC"] = "1"
    else:
        assert _jit.ensure_instance(case["A"], case["E"], case["F"], case["S"])
    args = dict(invalid_actions=None if case["invalid"] is None else torch.from_numpy(case["invalid"]), with_tree=True,
                gumbel=torch.from_numpy(case["gumbel"]), temperature=temperature)
    if policy == "muzero":
        args["dirichlet_noise"] = torch.from_numpy(case["noise"])
    out = s.act_mlp(torch.from_numpy(case["obs"]), key, **args)
    torch.cuda.synchronize()
    os.environ.pop("MZS_FORCE_GENERIC", None)
    return s, out


if mode in ("generic", "jit"):
    for c in range(n):
        if mode == "generic":
            A = int(rng.choice([1, 2, 3, 5, 8, 9, 18, 33, 64]))
            E = int(rng.choice([1, 5, 8, 24, 40, 77, 120]))
            S = int(rng.integers(1, 90)) if rng.random() < 0.8 else int(rng.integers(90, 261))
            support = int(rng.integers(8, 32)) if rng.random() < 0.5 else 10
        else:
            A = int(rng.integers(1, 9))
            E = int(rng.choice([3, 5, 7, 8, 9, 12, 16, 24, 40, 48, 56]))
            S = int(rng.integers(1, 128))
            support = int(rng.integers(8, 32)) if rng.random() < 0.5 else 10
            if _jit.plan(A, E, 2 * support + 1, S) is None:
                continue
        B = int(rng.integers(1, 70))
        policy = "gumbel" if (mode == "generic" and rng.random() < 0.3) else "muzero"
        tiebreak = bool(rng.integers(2)) and policy == "muzero"
        max_depth = None if rng.random() < 0.6 else int(rng.integers(1, S + 1))
        pred_on = "parent" if rng.random() < 0.25 else "child"
        case = make_case(oracle, 9000 + c + 100003 * seed, B, int(rng.integers(1, 12)), E, A, S, support=support,
                         invalid_frac=0.3 if (A > 2 and rng.random() < 0.4) else 0.0)
        scale = float(rng.choice([0.0, 0.3, 1.0, 3.0]))
        case["w"] = {k: (v * scale).astype(np.float32) if k.endswith(("w1", "w2")) else v for k, v in case["w"].items()}
        temperature = float(rng.choice([0.0, 0.5, 1.0]))
        key = [int(rng.integers(2 ** 31)), int(rng.integers(2 ** 31))]
        try:
            s, out = run_mlp(case, tiebreak, key, mode, max_depth, temperature, pred_on, policy)
            mlp = oracle.Mlp(case["w"], case["obs_dim"], E, A, case["F"], support_size=support, recurrent_pred_on=int(pred_on == "parent"))
            if policy == "muzero":
                ref = oracle.act_mlp(mlp, oracle.SearchCfg(S, max_depth=max_depth or 0, tiebreak=int(tiebreak)), case["obs"], key,
                                     case["noise"], 0.25, case["invalid"], temperature, case["gumbel"])
                tp._compare(ref, s, out)
            else:
                pl, v, emb = oracle.root_inference(mlp, case["obs"])
                tree = oracle.Tree(B, S + 1, A, E)
                cfg = oracle.SearchCfg(S, max_depth=max_depth or 0)
                oracle.tree_init(tree, oracle.mask_root_logits(pl, case["invalid"]), v, emb, case["invalid"])
                for sim in range(S):
                    p_, a_, _ = oracle.gumbel_step_select(tree, cfg, case["gumbel"], 1, 16)
                    oracle.step_expand_backup(tree, sim, p_, a_, *oracle.recurrent_inference(mlp, a_, tree.embeddings[np.arange(B), p_]))
                action, weights = oracle.gumbel_finish(tree, case["gumbel"], 1)
                assert np.array_equal(action, out.action.cpu().numpy()) and np.array_equal(weights, out.action_weights.cpu().numpy())
                assert_trees_equal(tree, out.search_tree, exact_floats=True)
        except AssertionError as e:
            bad += 1
            print(f"MISMATCH case {c}: A={A} E={E} support={support} S={S} B={B} policy={policy} tb={tiebreak} md={max_depth} "
                  f"scale={scale} T={temperature} pred_on={pred_on}: {str(e)[:200]}")
    print(f"{mode} seed {seed}: {n} cases, {bad} mismatches")
elif mode == "repr":
    # mzs_conv3x3_nhwc / mzs_resblock_v1 against an fp64 evaluation with the module path beside them: random C, odd
    # heights / widths (runs of pixels ending mid-row, maps narrower than a staging sweep), batches, both shortcuts
    g = torch.Generator().manual_seed(seed)
    worst = 0.0
    for c in range(n):
        C = int(rng.choice([32, 64]))
        H, W = int(rng.integers(1, 49)), int(rng.integers(1, 49))
        if C == 64 and H * W > 1200:
            H = W = 24
        B = int(rng.integers(1, 24))
        proj = bool(rng.integers(2))
        blk = mx.nn.ResidualConvBlockV1(C, 1, proj, generator=g)
        x = (torch.rand(B, H, W, C, generator=g) * 2 - 1) * float(rng.choice([0.1, 1.0, 10.0]))
        with torch.no_grad():
            blk.use_hip = False
            blk(x[:1])
            blk.cuda()
            x = x.cuda()
            for ln in ([blk.ln_0, blk.ln_1] + ([blk.proj_ln] if proj else [])):
                ln.scale.copy_(torch.rand(C, generator=g) + 0.5)
                ln.offset.copy_(torch.rand(C, generator=g) - 0.5)
            y_mod = blk(x)
            yc = blk.conv_0(x)
            blk.use_hip = True
            ok_path = blk._hip_ok(x)
            y = blk(x)
            mx.nn.HkConv2D.use_hip = mx.nn.HkLayerNorm.use_hip = False
            blk.use_hip = False
            yc_lib = blk.conv_0(x)
            blk.double()
            y64 = blk(x.double())
            yc64 = blk.conv_0(x.double())
            mx.nn.HkConv2D.use_hip = mx.nn.HkLayerNorm.use_hip = True
        e, em = float((y.double() - y64).abs().max()), float((y_mod.double() - y64).abs().max())
        ec, el = float((yc.double() - yc64).abs().max()), float((yc_lib.double() - yc64).abs().max())
        floor = 2e-5 * max(1.0, float(y64.abs().max()))
        floor_c = 2e-6 * (9 * C) ** 0.5 * max(1e-30, float(yc64.abs().max()))
        worst = max(worst, e / floor)
        if not ok_path or not (e <= floor and e <= 2 * em + floor) or not (ec <= floor_c and ec <= 2 * el + floor_c):
            bad += 1
            print(f"MISMATCH case {c}: C={C} H={H} W={W} B={B} proj={proj} hip_ok={ok_path}: block {e:.3g} (module {em:.3g}, floor {floor:.3g}) "
                  f"conv {ec:.3g} (library {el:.3g}, floor {floor_c:.3g})")
    print(f"repr seed {seed}: {n} cases, {bad} outside the bars (largest block error / bar {worst:.2f})")
else:
    A_, SUP = 18, 10
    g = torch.Generator().manual_seed(seed)
    mods = (mx.nn.ResNetRepresentation(32, generator=g), mx.nn.ResNetPrediction(A_, 21, generator=g), mx.nn.ResNetDynamic(A_, 21, generator=g))
    m = mx.MuZero(*mods)
    m.init(0, np.zeros((1, 84, 84, 4), np.float32))
    dy, pred = mods[2], mods[1]
    for c in range(n):
        B = int(rng.integers(1, 129)) if rng.random() < 0.8 else int(rng.integers(129, 201))
        S = int(rng.integers(1, 61))
        policy = "gumbel" if rng.random() < 0.3 else "muzero"
        max_depth = None if rng.random() < 0.6 else int(rng.integers(1, S + 1))
        emb = torch.rand(B, 6, 6, 64, generator=g).cuda() * float(rng.choice([0.1, 1.0, 5.0]))
        pl = torch.randn(B, A_, generator=g).cuda()
        v = torch.randn(B, generator=g).cuda()
        noise = torch.from_numpy(rng.dirichlet([0.3] * A_, B).astype(np.float32)).cuda()
        invalid = (rng.uniform(size=(B, A_)) < 0.15).astype(np.uint8)
        invalid[np.arange(B), rng.integers(0, A_, B)] = 0
        invalid = torch.from_numpy(invalid).cuda()

        def rec(action, flat):
            (r, disc, logits, val), ns = m._recurrent_inference(None, None, action, flat.reshape(B, 6, 6, 64))
            return r, disc, logits, val, ns.reshape(B, -1)

        outs = []
        for loop in (None, lambda h, b, e: dy.hip_search(pred, h, SUP, 0.99, b, e)):
            cfg = SearchConfig(A_, S, 2304, tiebreak=policy == "muzero", policy=policy, max_depth=max_depth)
            s = MuZeroSearch(B, cfg)
            kw = dict(dirichlet_noise=noise) if policy == "muzero" else {}
            o = s.search((pl, v, emb.reshape(B, -1)), rec, key=[c, seed], invalid_actions=invalid, with_tree=True, native_loop=loop, **kw)
            torch.cuda.synchronize()
            outs.append((o.action.clone(), o.action_weights.clone(), s.depth_sum.clone(),
                         {f: getattr(o.search_tree, f).clone() for f in o.search_tree._fields}))
            s.close()
        ok = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]) \
            and all(torch.equal(outs[0][3][f], outs[1][3][f]) for f in outs[0][3])
        if not ok or dy.pair_lost():
            bad += 1
            print(f"MISMATCH case {c}: B={B} S={S} policy={policy} md={max_depth} pair_lost={dy.pair_lost()}")
    print(f"search seed {seed}: {n} cases, {bad} mismatches")
