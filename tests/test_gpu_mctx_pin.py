"""The HIP path against golden captures of the REAL reference (tests/golden/mctx_*.npz, written by
tests/golden/capture_from_mctx.py where jax + mctx + haiku exist).  Absent captures -> loud skip; the harness is
exercised on a synthetic file in the same format either way (see tests/test_mctx_pin_cpu.py).  Through the product's
own entry points: MuZero.act() (everything from the key, as a caller of the reference gets it) and the fused
search with the captured root noise injected and the tree exported (isolates the search from the sampler's bits)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import mctx_fixture as fx  # noqa: E402

CAPTURES = fx.fixture_paths()
NO_CAPTURE = ("PARITY UNPINNED: no tests/golden/mctx_*.npz -- run tests/golden/capture_from_mctx.py on a machine with "
              "jax + mctx + dm-haiku and commit its outputs (INTEGRATION.md)")


def _model(case):
    import muax_amd as mx
    m = case["meta"]
    F = 2 * m["support_size"] + 1
    net = mx.nn.MZNetwork(mx.nn.Representation(m["E"]), mx.nn.Prediction(m["A"], F), mx.nn.Dynamic(m["E"], m["A"], F))
    model = mx.MuZero(net, policy=m["policy"], discount=m["discount"], support_size=m["support_size"],
                      recurrent_pred_on=m["recurrent_pred_on"])
    model.init(0, np.zeros((1, m["obs_dim"]), np.float32))
    with torch.no_grad():
        for k, p in mx.nn.mlp_trio_weights(model.network).items():
            p.copy_(torch.from_numpy(case["w"][k]))
    model.weights_changed()
    return model


def hip_run(case, dirichlet_from="key"):
    """-> {"action", "action_weights", "root_value", "tree"} from the HIP path on the capture's inputs."""
    m, key = case["meta"], case["key"]
    model = _model(case)
    kw = dict(num_simulations=m["num_simulations"], max_depth=m.get("max_depth"), with_tree=True)
    if m["policy"] == "muzero":
        kw.update(temperature=m["temperature"], dirichlet_fraction=m["dirichlet_fraction"],
                  dirichlet_alpha=m["dirichlet_alpha"], pb_c_init=m["pb_c_init"], pb_c_base=m["pb_c_base"])
        if dirichlet_from == "capture":
            kw["dirichlet_noise"] = torch.from_numpy(case["rng"]["dirichlet"]).cuda()
    else:
        kw.update(qtransform=m["qtransform"], max_num_considered_actions=m["max_num_considered_actions"],
                  gumbel_scale=m["gumbel_scale"])
    out, root_value = model._plan(model.params, key, torch.from_numpy(case["obs"]).cuda(), **kw)
    torch.cuda.synchronize()
    tree = {k: getattr(out.search_tree, k).cpu().numpy() for k in fx.TREE_INT + fx.TREE_FLOAT}
    got = {"action": out.action.cpu().numpy(), "action_weights": out.action_weights.cpu().numpy(),
           "root_value": root_value.cpu().numpy(), "tree": tree}
    # the reference's own call shape: NumPy in, NumPy out, everything from the key
    a, pi, v = model.act(key, case["obs"], with_pi=True, with_value=True, obs_from_batch=True,
                         **{k: v_ for k, v_ in kw.items() if k not in ("with_tree", "dirichlet_noise")})
    if dirichlet_from == "key":
        assert np.array_equal(a, got["action"]) and np.array_equal(pi, got["action_weights"]) and np.array_equal(v, got["root_value"])
    return got


@pytest.mark.parametrize("path", CAPTURES or [None], ids=[os.path.basename(p) for p in CAPTURES] or ["absent"])
def test_hip_path_matches_the_reference_capture(path):
    if path is None:
        pytest.skip(NO_CAPTURE)
    case = fx.load_case(path)
    msgs = fx.compare_outputs(case, hip_run(case, "capture")) if case["meta"]["policy"] == "muzero" else []
    msgs += [m + "  [everything from the key]" for m in fx.compare_outputs(case, hip_run(case, "key"))]
    assert not msgs, f"{os.path.basename(path)} ({case['meta']['versions']}):\n  " + "\n  ".join(msgs)


def _check_rollout(tr):
    case = {"meta": {"policy": "muzero", "A": 2, "E": 8, "obs_dim": 4, "support_size": 10, "discount": 0.99,
                     "recurrent_pred_on": tr["meta"].get("recurrent_pred_on", "child")}, "w": tr["w"]}
    model = _model(case)
    for t in range(len(tr["action"])):
        a, pi, v = model.act(tr["subkey"][t], tr["obs"][t], with_pi=True, with_value=True,
                             num_simulations=tr["meta"]["num_simulations"], temperature=1.0)
        assert isinstance(a, int) and a == int(tr["action"][t]), t
        assert pi.shape == (1, 2) and np.allclose(pi, tr["pi"][t].reshape(1, 2), atol=1e-5), t
        assert abs(v - float(tr["v"][t])) <= 1e-5 * max(1.0, abs(float(tr["v"][t]))), t


def test_hip_path_reproduces_the_reference_fit_loop_trace(oracle, tmp_path):
    """The reference's own acting loop (20 CartPole steps, one root, 10 simulations) through MuZero.act() unbatched --
    python int action, pi [1, A], python float value, as muax/model.py:160-179 returns them.  Always run on a
    synthetic trace in the same format (the oracle's output); on the real one when it exists."""
    _check_rollout(fx.load_rollout(fx.synthetic_rollout(oracle, str(tmp_path / "trace.npz"))))
    if not os.path.exists(fx.ROLLOUT_PATH):
        pytest.skip(NO_CAPTURE + " (fit-loop trace: mctx_rollout_cartpole_s10.npz)")
    _check_rollout(fx.load_rollout())


@pytest.mark.parametrize("policy,shape,S", [("muzero", (4, 8, 2), 50), ("muzero", (8, 32, 4), 50), ("gumbel", (4, 8, 2), 50),
                                            # round 6 -- the record kinds added since the kit was written, so that the
                                            # day a capture runs it pins them too: a LONG instance (root paths in HBM, 160
                                            # simulations; planned per policy) and a twelve-action one (all of a node's
                                            # scores in one lane), both built on demand through MuZero._plan
                                            ("muzero", (4, 8, 2), 160), ("muzero", (4, 8, 12), 50), ("gumbel", (4, 8, 12), 50)])
def test_harness_on_a_synthetic_file(oracle, tmp_path, policy, shape, S):
    """The same harness on a file in the capture format holding the ORACLE's output (not a pin): the HIP path must
    reproduce it exactly -- i.e. the route a real capture will take is known to work end to end."""
    od, E, A = shape
    path = fx.synthetic_case(oracle, str(tmp_path / "synthetic.npz"), policy=policy, obs_dim=od, E=E, A=A, S=S, seed=3)
    case = fx.load_case(path)
    for src in (("capture", "key") if policy == "muzero" else ("key",)):
        got = hip_run(case, src)
        assert not fx.compare_outputs(case, got), src
        for k in fx.TREE_INT + fx.TREE_FLOAT:  # against our own oracle the bar is equality, not 1e-5
            assert np.array_equal(case["tree"][k], got["tree"][k]), (src, k)
    bad = hip_run(case, "key")
    bad["tree"]["parents"] = bad["tree"]["parents"].copy()
    bad["tree"]["parents"][0, 5] ^= 1
    assert fx.compare_outputs(case, bad)


def _resnet_mods(mx, case, dev):
    m = case["meta"]
    g = torch.Generator().manual_seed(7)
    mods = (mx.nn.ResNetRepresentation(m["input_channels"], generator=g), mx.nn.ResNetPrediction(m["A"], m["F"], generator=g),
            mx.nn.ResNetDynamic(m["A"], m["F"], output_channels=m["dynamic_channels"], generator=g))
    with torch.no_grad():  # materialise the lazily shaped layers on the host, then move
        s = mods[0](torch.as_tensor(case["obs"][:1].astype(np.float32)))
        mods[1](s), mods[2](s, torch.zeros(1, dtype=torch.long))
    for mod in mods:
        mod.to(dev)
    fx.resnet_assign(mods, case["manifest"], case["seed"])
    return mods


def _check_resnet(mx, case):
    mods = _resnet_mods(mx, case, "cuda")
    msgs = []
    for hip in (False, True):  # the torch mirrors (MIOpen convolutions), then the one-launch HIP recurrent kernel
        root, rec = fx.resnet_mirror_outputs(mx, mods, case, device="cuda", hip=hip)
        msgs += [("hip kernel: " if hip else "torch mirror: ") + m for m in fx.compare_resnet(case, root, rec)]
    # the representation net stage by stage (HIP convolutions / residual blocks in inference): the first diverging stage
    import torch
    obs = torch.as_tensor(case["obs"].astype(np.float32), device="cuda")
    msgs += ["representation: " + m for m in fx.compare_resnet_stages(case, fx.resnet_mirror_stages(mx, mods[0], obs))]
    assert not msgs, "\n  ".join(msgs)


def test_resnet_nets_against_the_reference_capture(tmp_path):
    """The convolutional plugin nets (SAME geometry, LayerNorm axes, pooling, head order: muax/nn.py:118-148,313-395)
    and the one-launch recurrent kernel (mz_conv.cuh) against the reference's own root inference and recurrent_fn
    outputs on Atari-shaped frames.  Always on a synthetic file at the real shapes (the torch mirror's output: the
    manifest route and the HIP kernel are exercised; not a pin); on the real capture when it exists."""
    import muax_amd as mx
    case = fx.load_resnet(fx.synthetic_resnet(mx, str(tmp_path / "resnet.npz"), B=2, hw=84, A=18, c=32, dc=64, device="cuda"))
    assert case["root"]["embedding"].shape == (2, 6, 6, 64)
    _check_resnet(mx, case)
    if not os.path.exists(fx.RESNET_PATH):
        pytest.skip(NO_CAPTURE + " (ResNet nets: mctx_resnet_nets_seed0.npz)")
    _check_resnet(mx, fx.load_resnet())
