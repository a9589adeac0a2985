"""Shared input builders for the parity tests (seeded, no reference import)."""
import numpy as np

F32 = np.float32


def make_case(oracle, seed, B, obs_dim, E, A, S, support=10, bias_scale=0.1, invalid_frac=0.0):
    F = 2 * support + 1
    w = oracle.random_mlp_weights(seed, obs_dim, E, A, F, bias_scale=bias_scale)
    rng = np.random.default_rng(seed + 1000)
    obs = rng.uniform(-1, 1, (B, obs_dim)).astype(F32)
    noise = rng.dirichlet([0.3] * A, B).astype(F32)
    gum = rng.gumbel(size=(B, A)).astype(F32)
    invalid = None
    if invalid_frac > 0:
        invalid = (rng.uniform(size=(B, A)) < invalid_frac).astype(np.uint8)
        invalid[np.arange(B), rng.integers(0, A, B)] = 0  # keep one valid action per root
        invalid[0, :] = 1                                  # ... except one all-invalid root (mctx: argmax -> 0)
    return dict(w=w, obs=obs, noise=noise, gumbel=gum, invalid=invalid, B=B, obs_dim=obs_dim, E=E, A=A,
                F=F, S=S, support=support)


def assert_trees_equal(oracle_tree, gpu_tree, exact_floats=True):
    ref = oracle_tree.arrays()
    for name, a in ref.items():
        b = getattr(gpu_tree, name).cpu().numpy()
        assert a.shape == b.shape, name
        if a.dtype == np.int32 or exact_floats:
            bad = np.argwhere(a != b)
            assert bad.size == 0, f"{name}: {len(bad)} mismatches, first at {bad[0]}: {a[tuple(bad[0])]} vs {b[tuple(bad[0])]}"
        else:
            assert np.allclose(a, b, rtol=1e-5, atol=1e-5), name
