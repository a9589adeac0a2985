"""The oracle against golden captures of the REAL reference (jax + mctx + haiku + muax's glue), when they exist.

tests/golden/capture_from_mctx.py writes tests/golden/mctx_*.npz on a machine that has those packages (this container
and the GPU box do not: the captures are absent until someone runs it -- INTEGRATION.md, "Pinning the oracle").
With captures present these tests are the pin SURVEY.md 8(c) asks for: integers exact, floats 1e-5.  Without them
they SKIP, loudly, and the harness itself is still exercised on a synthetic file in the same format (the oracle's
own output: proves the reader / comparer / runner work and that a perturbed file is caught -- it pins nothing)."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import mctx_fixture as fx  # noqa: E402

CAPTURES = fx.fixture_paths()
NO_CAPTURE = ("PARITY UNPINNED: no tests/golden/mctx_*.npz -- run tests/golden/capture_from_mctx.py on a machine with "
              "jax + mctx + dm-haiku and commit its outputs (INTEGRATION.md)")


@pytest.mark.parametrize("path", CAPTURES or [None], ids=[os.path.basename(p) for p in CAPTURES] or ["absent"])
def test_oracle_matches_the_reference_capture(oracle, path):
    if path is None:
        pytest.skip(NO_CAPTURE)
    case = fx.load_case(path)
    msgs = fx.compare_rng(case, fx.oracle_rng(oracle, case))          # the PRNG walk on its own
    msgs += fx.compare_outputs(case, fx.oracle_run(oracle, case, dirichlet_from="capture"))  # the search on its own
    msgs += [m + "  [everything from the key]" for m in
             fx.compare_outputs(case, fx.oracle_run(oracle, case, dirichlet_from="oracle"))]
    assert not msgs, f"{os.path.basename(path)} ({case['meta']['versions']}):\n  " + "\n  ".join(msgs)


def test_oracle_reproduces_the_reference_fit_loop_trace(oracle):
    """20 CartPole steps of the reference's own acting loop (one root, 10 simulations): the oracle's act() for every
    recorded (sub-key, observation) -- action exact, pi and value to 1e-5."""
    if not os.path.exists(fx.ROLLOUT_PATH):
        pytest.skip(NO_CAPTURE + " (fit-loop trace: mctx_rollout_cartpole_s10.npz)")
    msgs = fx.oracle_rollout_mismatches(oracle, fx.load_rollout())
    assert not msgs, "\n  ".join(msgs)


def test_reference_checkpoint_reads_back(tmp_path):
    """A checkpoint written by the reference's own save (jnp.save of {'params', 'optimizer_state'},
    muax/model.py:203-212) through muax_amd.checkpoint, against the weights the capture script flattened."""
    ckpt = os.path.join(fx.HERE, "mctx_checkpoint.npy")
    want_path = os.path.join(fx.HERE, "mctx_checkpoint_expected.npz")
    if not (os.path.exists(ckpt) and os.path.exists(want_path)):
        pytest.skip("CHECKPOINT INTEROP UNVERIFIED: no tests/golden/mctx_checkpoint.npy -- capture_from_mctx.py writes it")
    import torch
    import muax_amd as mx
    want = np.load(want_path)
    E, A = want["repr_w"].shape[1], want["pp_w2"].shape[1]
    net = mx.nn.MZNetwork(mx.nn.Representation(E), mx.nn.Prediction(A, want["pv_w2"].shape[1]),
                          mx.nn.Dynamic(E, A, want["dr_w2"].shape[1]))
    m = mx.MuZero(net, device="cpu")
    m.init(0, np.zeros((1, want["repr_w"].shape[0]), np.float32))
    mx.checkpoint.load_reference_params(m, ckpt)
    got = {k: v.detach().numpy() for k, v in mx.nn.mlp_trio_weights(m.network).items()}
    for k in fx.WEIGHT_NAMES:
        assert np.array_equal(got[k], want[k]), k
    assert isinstance(torch.zeros(1), torch.Tensor)


# ------------------------------------------------------------------------------------------- harness self-test
@pytest.mark.parametrize("policy", ["muzero", "gumbel"])
def test_harness_on_a_synthetic_file(oracle, tmp_path, policy):
    """Reader / runner / comparer on a file in the capture format (NOT a pin, see the module docstring): passes on
    consistent data, and every kind of perturbation -- an index, a float beyond 1e-5, a tie-break uniform -- fails."""
    path = fx.synthetic_case(oracle, str(tmp_path / "synthetic.npz"), policy=policy, S=12)
    case = fx.load_case(path)
    assert case["meta"]["policy"] == policy and case["tree"]["children_index"].shape == (8, 13, 2)
    assert not fx.compare_rng(case, fx.oracle_rng(oracle, case))
    assert not fx.compare_outputs(case, fx.oracle_run(oracle, case))
    if policy == "muzero":
        assert not fx.compare_outputs(case, fx.oracle_run(oracle, case, dirichlet_from="capture"))
    got = fx.oracle_run(oracle, case)
    bad = dict(got, tree=dict(got["tree"]))
    bad["tree"]["children_visits"] = got["tree"]["children_visits"].copy()
    bad["tree"]["children_visits"][3, 0, 1] += 1
    assert any("children_visits" in m for m in fx.compare_outputs(case, bad))
    bad = dict(got, root_value=got["root_value"] * np.float32(1 + 3e-5))
    assert any("root_value" in m for m in fx.compare_outputs(case, bad))
    ok = dict(got, root_value=got["root_value"] * np.float32(1 + 2e-6))
    assert not fx.compare_outputs(case, ok)
    if policy == "muzero":
        rng = fx.oracle_rng(oracle, case)
        rng["tiebreak"] = rng["tiebreak"].copy()
        rng["tiebreak"][2, 1, 0, 0] = np.nextafter(rng["tiebreak"][2, 1, 0, 0], np.float32(2))
        assert any("tiebreak" in m for m in fx.compare_rng(case, rng))


def test_rollout_trace_harness_on_a_synthetic_file(oracle, tmp_path):
    """The fit-loop trace reader / checker on a file in the capture script's format (the oracle's own output: not a pin)."""
    tr = fx.load_rollout(fx.synthetic_rollout(oracle, str(tmp_path / "trace.npz")))
    assert tr["pi"].shape == (6, 1, 2) and not fx.oracle_rollout_mismatches(oracle, tr)
    tr["action"] = tr["action"].copy()
    tr["action"][2] ^= 1
    assert any("step 2 action" in m for m in fx.oracle_rollout_mismatches(oracle, tr))


def test_capture_script_flattens_haiku_params_in_the_checkpoint_readers_order():
    """capture_from_mctx.flatten_params on a haiku-shaped parameter tree (names as hk.transform makes them for
    muax/nn.py:59-115) gives the arrays muax_amd.checkpoint assigns from the same tree: one convention, two readers."""
    spec = importlib.util.spec_from_file_location("capture_from_mctx", os.path.join(fx.HERE, "capture_from_mctx.py"))
    cap = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cap)  # importing it needs no jax; running main() does
    rng = np.random.default_rng(0)
    E, A, F, od = 8, 2, 21, 4

    def lin(i, o):
        return {"w": rng.standard_normal((i, o)).astype(np.float32), "b": rng.standard_normal(o).astype(np.float32)}

    rep = {"representation/~/linear": lin(od, E)}
    pred = {"prediction/~/linear": lin(E, 16), "prediction/~/linear_1": lin(16, F),
            "prediction/~/linear_2": lin(E, 16), "prediction/~/linear_3": lin(16, A)}
    dyn = {"dynamic/~/linear": lin(E + A, 16), "dynamic/~/linear_1": lin(16, E),
           "dynamic/~/linear_2": lin(E + A, 16), "dynamic/~/linear_3": lin(16, F)}
    params = types.SimpleNamespace(representation=rep, prediction=pred, dynamic=dyn)
    w = cap.flatten_params(params, dict(obs_dim=od, E=E, A=A))
    assert set(w) == set(fx.WEIGHT_NAMES)
    assert w["pv_w2"] is not None and np.array_equal(w["pv_w2"], pred["prediction/~/linear_1"]["w"])
    assert np.array_equal(w["pp_w1"], pred["prediction/~/linear_2"]["w"])
    assert np.array_equal(w["dn_w2"], dyn["dynamic/~/linear_1"]["w"]) and np.array_equal(w["dr_b2"], dyn["dynamic/~/linear_3"]["b"])
    from muax_amd import checkpoint
    layers = checkpoint._linears(dyn)
    assert np.array_equal(layers[1]["w"], w["dn_w2"]) and np.array_equal(layers[2]["w"], w["dr_w1"])
    with pytest.raises(SystemExit):  # a swapped layer order is refused, not written
        swapped = dict(dyn)
        swapped["dynamic/~/linear_1"], swapped["dynamic/~/linear_3"] = dyn["dynamic/~/linear_3"], dyn["dynamic/~/linear_1"]
        cap.flatten_params(types.SimpleNamespace(representation=rep, prediction=pred, dynamic=swapped), dict(obs_dim=od, E=E, A=A))
