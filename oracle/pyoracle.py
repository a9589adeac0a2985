"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Loads ``oracle/_build/libmzoracle.so`` (built by ``oracle/Makefile``) and exposes
the restated reference path on NumPy arrays.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; ``muax_amd`` never does.  Parity is UNPINNED (see mz_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmzoracle.so")

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make decides from the time stamps whether anything is to do)."""
    if force or not os.path.exists(_SO) or os.path.exists(os.path.join(_HERE, "mz_oracle.c")) and \
            os.path.getmtime(os.path.join(_HERE, "mz_oracle.c")) > os.path.getmtime(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class _Tree(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("N", C.c_int32), ("A", C.c_int32), ("E", C.c_int32),
        ("node_visits", _i32p), ("raw_values", _f32p), ("node_values", _f32p),
        ("parents", _i32p), ("action_from_parent", _i32p),
        ("children_index", _i32p), ("children_prior_logits", _f32p),
        ("children_values", _f32p), ("children_visits", _i32p),
        ("children_rewards", _f32p), ("children_discounts", _f32p),
        ("embeddings", _f32p), ("root_invalid_actions", _u8p),
    ]


class _Cfg(C.Structure):
    _fields_ = [
        ("num_simulations", C.c_int32), ("max_depth", C.c_int32),
        ("pb_c_init", C.c_float), ("pb_c_base", C.c_float),
        ("tiebreak", C.c_int32),
        ("global_batch", C.c_int64), ("root_offset", C.c_int64),
    ]


_MLP_W = ["repr_w", "repr_b",
          "pv_w1", "pv_b1", "pv_w2", "pv_b2",
          "pp_w1", "pp_b1", "pp_w2", "pp_b2",
          "dr_w1", "dr_b1", "dr_w2", "dr_b2",
          "dn_w1", "dn_b1", "dn_w2", "dn_b2"]


class _Mlp(C.Structure):
    _fields_ = ([("obs_dim", C.c_int32), ("E", C.c_int32), ("A", C.c_int32),
                 ("F", C.c_int32), ("H", C.c_int32)]
                + [(n, _f32p) for n in _MLP_W]
                + [("discount", C.c_float), ("support_size", C.c_int32),
                   ("recurrent_pred_on", C.c_int32)])


def _p(a, t):
    return a.ctypes.data_as(t)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.mzo_exp.restype = C.c_float
        L.mzo_exp.argtypes = [C.c_float]
        L.mzo_expm1_neg.restype = C.c_float
        L.mzo_expm1_neg.argtypes = [C.c_float]
        L.mzo_elu.restype = C.c_float
        L.mzo_elu.argtypes = [C.c_float]
        L.mzo_log.restype = C.c_float
        L.mzo_log.argtypes = [C.c_float]
        L.mzo_sum16.restype = C.c_float
        L.mzo_sum16.argtypes = [_f32p, C.c_int]
        L.mzo_inv_scaling.restype = C.c_float
        L.mzo_inv_scaling.argtypes = [C.c_float]
        L.mzo_support_to_scalar.restype = C.c_float
        L.mzo_support_to_scalar.argtypes = [_f32p, C.c_int]
        L.mzo_uniform_from_bits.restype = C.c_float
        L.mzo_uniform_from_bits.argtypes = [C.c_uint32]
        L.mzo_gumbel_from_bits.restype = C.c_float
        L.mzo_gumbel_from_bits.argtypes = [C.c_uint32]
        L.mzo_log1p.restype = C.c_float
        L.mzo_log1p.argtypes = [C.c_float]
        L.mzo_erf_inv.restype = C.c_float
        L.mzo_erf_inv.argtypes = [C.c_float]
        L.mzo_loggamma_one.restype = C.c_float
        L.mzo_loggamma_one.argtypes = [_u32p, C.c_float]
        L.mzo_dirichlet.restype = None
        L.mzo_dirichlet.argtypes = [_u32p, C.c_float, C.c_int, C.c_int, C.c_int64, C.c_int64, _f32p]
        L.mzo_random_bits.restype = C.c_uint32
        L.mzo_random_bits.argtypes = [_u32p, C.c_int64, C.c_int64]
        L.mzo_select_action.restype = C.c_int
        L.mzo_div2eps_mismatches.restype = C.c_int64
        L.mzo_div2eps_mismatches.argtypes = [C.c_int, C.c_int]
        L.mzo_elu_clamped.restype = C.c_float
        L.mzo_elu_clamped.argtypes = [C.c_float]
        L.mzo_markstein_mismatches.restype = C.c_int64
        L.mzo_markstein_mismatches.argtypes = [C.c_int, C.c_int]
        for name in ("mzo_softmax", "mzo_min_max_normalize", "mzo_threefry2x32", "mzo_split",
                     "mzo_root_inference", "mzo_recurrent_inference", "mzo_root_prior",
                     "mzo_tree_init", "mzo_simulate", "mzo_expand", "mzo_backward",
                     "mzo_summary_sample", "mzo_step_select", "mzo_step_expand_backup",
                     "mzo_act_mlp", "mzo_qtransform", "mzo_considered_visits", "mzo_gumbel_step_select",
                     "mzo_gumbel_finish", "mzo_action_scores", "mzo_step_select_injected"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


# --------------------------------------------------------------------------
# scalar helpers (vectorised through python loops: test-sized inputs only)
# --------------------------------------------------------------------------

def exp(x):
    L = lib()
    return np.array([L.mzo_exp(float(v)) for v in np.ravel(x)], np.float32).reshape(np.shape(x))


def log(x):
    L = lib()
    return np.array([L.mzo_log(float(v)) for v in np.ravel(x)], np.float32).reshape(np.shape(x))


def elu(x):
    L = lib()
    return np.array([L.mzo_elu(float(v)) for v in np.ravel(x)], np.float32).reshape(np.shape(x))


def inv_scaling(x):
    L = lib()
    return np.array([L.mzo_inv_scaling(float(v)) for v in np.ravel(x)],
                    np.float32).reshape(np.shape(x))


def sum16(x):
    x = np.ascontiguousarray(x, np.float32)
    return float(lib().mzo_sum16(_p(x, _f32p), x.size))


def softmax(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().mzo_softmax(_p(x, _f32p), C.c_int(x.size), _p(out, _f32p))
    return out


def support_to_scalar(probs, support_size):
    probs = np.ascontiguousarray(probs, np.float32)
    return float(lib().mzo_support_to_scalar(_p(probs, _f32p), support_size))


def min_max_normalize(s):
    s = np.array(s, np.float32, copy=True)
    lib().mzo_min_max_normalize(_p(s, _f32p), C.c_int(s.size))
    return s


def markstein_mismatches(dmax=300, exponent=0):
    """Exhaustive check of the kernel's small-integer division identity (see mz_oracle.h)."""
    return int(lib().mzo_markstein_mismatches(dmax, exponent))


def threefry2x32(key, x0, x1):
    k = np.asarray(key, np.uint32)
    out = np.zeros(2, np.uint32)
    lib().mzo_threefry2x32(_p(k, _u32p), C.c_uint32(int(x0)), C.c_uint32(int(x1)), _p(out, _u32p))
    return out


def split(key, n=2):
    k = np.ascontiguousarray(key, np.uint32)
    out = np.zeros((n, 2), np.uint32)
    for r in range(n):
        lib().mzo_split(_p(k, _u32p), C.c_int64(n), C.c_int64(r), _p(out[r], _u32p))
    return out


def random_bits(key, size):
    k = np.ascontiguousarray(key, np.uint32)
    return np.array([lib().mzo_random_bits(_p(k, _u32p), size, i) for i in range(size)], np.uint32)


def uniform(key, size):
    return np.array([lib().mzo_uniform_from_bits(int(b)) for b in random_bits(key, size)],
                    np.float32)


def normal(key, size=None):
    """jax.random.normal(key, ()) as the gamma sampler of mz_oracle.c draws it (sqrt(2) * erf_inv(uniform(-1, 1))), or
    jax.random.normal(key, (size,))."""
    k = np.ascontiguousarray(key, np.uint32)
    if size is None:
        f = lib().mzo_normal
        f.restype = C.c_float
        return np.float32(f(_p(k, _u32p)))
    out = np.zeros(size, np.float32)
    lib().mzo_normal_vec(_p(k, _u32p), C.c_int64(size), _p(out, _f32p))
    return out


def gumbel(key, size):
    return np.array([lib().mzo_gumbel_from_bits(int(b)) for b in random_bits(key, size)],
                    np.float32)


def dirichlet(key, alpha, B, A, global_batch=None, root_offset=0):
    """Rows [root_offset, root_offset + B) of jax.random.dirichlet(key, full([A], alpha), (global_batch,))
    as restated in mz_oracle.c (spec-to-confirm)."""
    k = np.ascontiguousarray(key, np.uint32)
    out = np.zeros((B, A), np.float32)
    lib().mzo_dirichlet(_p(k, _u32p), C.c_float(alpha), B, A, global_batch or B, root_offset, _p(out, _f32p))
    return out


# --------------------------------------------------------------------------
# structures
# --------------------------------------------------------------------------

class Tree:
    """mctx.Tree in its own layout; owns NumPy storage."""

    FIELDS_I = ("node_visits", "parents", "action_from_parent")
    FIELDS_F = ("raw_values", "node_values")
    FIELDS_IA = ("children_index", "children_visits")
    FIELDS_FA = ("children_prior_logits", "children_values", "children_rewards",
                 "children_discounts")

    def __init__(self, B, N, A, E):
        self.B, self.N, self.A, self.E = B, N, A, E
        for f in self.FIELDS_I:
            setattr(self, f, np.zeros((B, N), np.int32))
        for f in self.FIELDS_F:
            setattr(self, f, np.zeros((B, N), np.float32))
        for f in self.FIELDS_IA:
            setattr(self, f, np.zeros((B, N, A), np.int32))
        for f in self.FIELDS_FA:
            setattr(self, f, np.zeros((B, N, A), np.float32))
        self.embeddings = np.zeros((B, N, E), np.float32)
        self.root_invalid_actions = np.zeros((B, A), np.uint8)

    def c(self):
        t = _Tree()
        t.B, t.N, t.A, t.E = self.B, self.N, self.A, self.E
        for f in self.FIELDS_I + self.FIELDS_IA:
            setattr(t, f, _p(getattr(self, f), _i32p))
        for f in self.FIELDS_F + self.FIELDS_FA + ("embeddings",):
            setattr(t, f, _p(getattr(self, f), _f32p))
        t.root_invalid_actions = _p(self.root_invalid_actions, _u8p)
        return t

    def arrays(self):
        names = (self.FIELDS_I + self.FIELDS_F + self.FIELDS_IA + self.FIELDS_FA
                 + ("embeddings",))
        return {n: getattr(self, n) for n in names}


@dataclass
class SearchCfg:
    num_simulations: int
    max_depth: int = 0
    pb_c_init: float = 1.25
    pb_c_base: float = 19652.0
    tiebreak: int = 0
    global_batch: int = 0
    root_offset: int = 0

    def c(self, B):
        return _Cfg(self.num_simulations, self.max_depth, self.pb_c_init, self.pb_c_base,
                    self.tiebreak, self.global_batch or B, self.root_offset)


class Mlp:
    """Default MLP trio weights (haiku layout, w[in][out])."""

    def __init__(self, weights: dict, obs_dim, E, A, F, H=16, discount=0.99, support_size=10,
                 recurrent_pred_on=0):
        assert F == 2 * support_size + 1
        self.w = {k: np.ascontiguousarray(weights[k], np.float32) for k in _MLP_W}
        self.obs_dim, self.E, self.A, self.F, self.H = obs_dim, E, A, F, H
        self.discount, self.support_size = discount, support_size
        self.recurrent_pred_on = recurrent_pred_on
        shapes = {"repr_w": (obs_dim, E), "repr_b": (E,),
                  "pv_w1": (E, H), "pv_b1": (H,), "pv_w2": (H, F), "pv_b2": (F,),
                  "pp_w1": (E, H), "pp_b1": (H,), "pp_w2": (H, A), "pp_b2": (A,),
                  "dr_w1": (E + A, H), "dr_b1": (H,), "dr_w2": (H, F), "dr_b2": (F,),
                  "dn_w1": (E + A, H), "dn_b1": (H,), "dn_w2": (H, E), "dn_b2": (E,)}
        for k, s in shapes.items():
            assert self.w[k].shape == s, (k, self.w[k].shape, s)

    def c(self):
        m = _Mlp()
        m.obs_dim, m.E, m.A, m.F, m.H = self.obs_dim, self.E, self.A, self.F, self.H
        for k in _MLP_W:
            setattr(m, k, _p(self.w[k], _f32p))
        m.discount, m.support_size = self.discount, self.support_size
        m.recurrent_pred_on = self.recurrent_pred_on
        return m


def random_mlp_weights(seed, obs_dim, E, A, F, H=16, bias_scale=0.0):
    """haiku-style init (TruncNormal(1/sqrt(fan_in)), b=0) from a NumPy seed."""
    rng = np.random.default_rng(seed)

    def tn(shape):
        fan_in = shape[0]
        x = rng.standard_normal(shape)
        while True:
            bad = np.abs(x) > 2.0
            if not bad.any():
                break
            x[bad] = rng.standard_normal(int(bad.sum()))
        return (x / np.sqrt(fan_in)).astype(np.float32)

    def bias(n):
        return (bias_scale * rng.standard_normal(n)).astype(np.float32)

    return {"repr_w": tn((obs_dim, E)), "repr_b": bias(E),
            "pv_w1": tn((E, H)), "pv_b1": bias(H), "pv_w2": tn((H, F)), "pv_b2": bias(F),
            "pp_w1": tn((E, H)), "pp_b1": bias(H), "pp_w2": tn((H, A)), "pp_b2": bias(A),
            "dr_w1": tn((E + A, H)), "dr_b1": bias(H), "dr_w2": tn((H, F)), "dr_b2": bias(F),
            "dn_w1": tn((E + A, H)), "dn_b1": bias(H), "dn_w2": tn((H, E)), "dn_b2": bias(E)}


# --------------------------------------------------------------------------
# nets and search
# --------------------------------------------------------------------------

def root_inference(mlp: Mlp, obs):
    obs = np.ascontiguousarray(obs, np.float32)
    B = obs.shape[0]
    emb = np.zeros((B, mlp.E), np.float32)
    pl = np.zeros((B, mlp.A), np.float32)
    v = np.zeros(B, np.float32)
    m = mlp.c()
    for b in range(B):
        vb = C.c_float()
        lib().mzo_root_inference(C.byref(m), _p(obs[b], _f32p), _p(emb[b], _f32p),
                                 _p(pl[b], _f32p), C.byref(vb))
        v[b] = vb.value
    return pl, v, emb


def recurrent_inference(mlp: Mlp, action, embedding):
    embedding = np.ascontiguousarray(embedding, np.float32)
    action = np.asarray(action, np.int32)
    B = embedding.shape[0]
    r = np.zeros(B, np.float32)
    d = np.zeros(B, np.float32)
    v = np.zeros(B, np.float32)
    pl = np.zeros((B, mlp.A), np.float32)
    ne = np.zeros((B, mlp.E), np.float32)
    m = mlp.c()
    for b in range(B):
        rb, db, vb = C.c_float(), C.c_float(), C.c_float()
        lib().mzo_recurrent_inference(C.byref(m), C.c_int(int(action[b])),
                                      _p(embedding[b], _f32p), C.byref(rb), C.byref(db),
                                      _p(pl[b], _f32p), C.byref(vb), _p(ne[b], _f32p))
        r[b], d[b], v[b] = rb.value, db.value, vb.value
    return r, d, pl, v, ne


def root_prior(prior_logits, dirichlet_noise, dirichlet_fraction, invalid=None):
    pl = np.ascontiguousarray(prior_logits, np.float32)
    B, A = pl.shape
    out = np.zeros_like(pl)
    for b in range(B):
        nz = None if dirichlet_noise is None else np.ascontiguousarray(dirichlet_noise[b], np.float32)
        iv = None if invalid is None else np.ascontiguousarray(invalid[b], np.uint8)
        lib().mzo_root_prior(_p(pl[b], _f32p), C.c_int(A),
                             _p(nz, _f32p) if nz is not None else None,
                             C.c_float(dirichlet_fraction),
                             _p(iv, _u8p) if iv is not None else None, _p(out[b], _f32p))
    return out


def tree_init(tree: Tree, prior_logits, value, embedding, invalid=None):
    pl = np.ascontiguousarray(prior_logits, np.float32)
    v = np.ascontiguousarray(value, np.float32)
    e = np.ascontiguousarray(embedding, np.float32)
    iv = None if invalid is None else np.ascontiguousarray(invalid, np.uint8)
    t = tree.c()
    lib().mzo_tree_init(C.byref(t), _p(pl, _f32p), _p(v, _f32p), _p(e, _f32p),
                        _p(iv, _u8p) if iv is not None else None)


def step_select(tree: Tree, cfg: SearchCfg, sim, sim_key=None):
    B = tree.B
    parent = np.zeros(B, np.int32)
    action = np.zeros(B, np.int32)
    depth = np.zeros(B, np.int32)
    k = np.ascontiguousarray(sim_key if sim_key is not None else [0, 0], np.uint32)
    t, c = tree.c(), cfg.c(B)
    lib().mzo_step_select(C.byref(t), C.byref(c), C.c_int(sim), _p(k, _u32p),
                          _p(parent, _i32p), _p(action, _i32p), _p(depth, _i32p))
    return parent, action, depth


def step_select_injected(tree: Tree, cfg: SearchCfg, sim, sim_key, uniforms):
    """step_select with this simulation's tie-break uniforms injected: `uniforms` [B, D, A] (a capture's
    rng_tiebreak[sim]); levels >= D draw from the key walk as usual."""
    B = tree.B
    parent, action, depth = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    k = np.ascontiguousarray(sim_key if sim_key is not None else [0, 0], np.uint32)
    u = np.ascontiguousarray(uniforms, np.float32)
    assert u.shape[0] == B and u.shape[2] == tree.A
    t, c = tree.c(), cfg.c(B)
    lib().mzo_step_select_injected(C.byref(t), C.byref(c), C.c_int(sim), _p(k, _u32p), _p(u, _f32p), C.c_int(u.shape[1]),
                                   _p(parent, _i32p), _p(action, _i32p), _p(depth, _i32p))
    return parent, action, depth


def action_scores(tree: Tree, cfg: SearchCfg, b, node):
    """(value_score [A], policy_score [A]) of muzero_action_selection at `node` of root b, the oracle's arithmetic."""
    vs, ps = np.zeros(tree.A, np.float32), np.zeros(tree.A, np.float32)
    t, c = tree.c(), cfg.c(tree.B)
    lib().mzo_action_scores(C.byref(t), C.c_int(b), C.c_int(node), C.byref(c), _p(vs, _f32p), _p(ps, _f32p))
    return vs, ps


def step_expand_backup(tree: Tree, sim, parent, action, reward, discount, prior_logits, value,
                       next_embedding):
    args = [np.ascontiguousarray(parent, np.int32), np.ascontiguousarray(action, np.int32),
            np.ascontiguousarray(reward, np.float32), np.ascontiguousarray(discount, np.float32),
            np.ascontiguousarray(prior_logits, np.float32), np.ascontiguousarray(value, np.float32),
            np.ascontiguousarray(next_embedding, np.float32)]
    t = tree.c()
    lib().mzo_step_expand_backup(C.byref(t), C.c_int(sim), _p(args[0], _i32p), _p(args[1], _i32p),
                                 _p(args[2], _f32p), _p(args[3], _f32p), _p(args[4], _f32p),
                                 _p(args[5], _f32p), _p(args[6], _f32p))


def summary_sample(tree: Tree, temperature, gumbel_noise):
    B, A = tree.B, tree.A
    g = np.ascontiguousarray(gumbel_noise, np.float32)
    action = np.zeros(B, np.int32)
    weights = np.zeros((B, A), np.float32)
    t = tree.c()
    for b in range(B):
        ab = C.c_int32()
        lib().mzo_summary_sample(C.byref(t), C.c_int(b), C.c_float(temperature), _p(g[b], _f32p),
                                 C.byref(ab), _p(weights[b], _f32p))
        action[b] = ab.value
    return action, weights


def sim_keys_from_act_key(key, S):
    """(k_sample, k_dirichlet, [simulate_key_s]) exactly as mctx consumes them."""
    ks = split(key, 3)
    k_sample, k_dir, rk = ks[0], ks[1], ks[2]
    sims = np.zeros((S, 2), np.uint32)
    for s in range(S):
        three = split(rk, 3)
        rk, sims[s] = three[0], three[1]
    return k_sample, k_dir, sims


def act_mlp(mlp: Mlp, cfg: SearchCfg, obs, key, dirichlet_noise=None, dirichlet_fraction=0.25,
            invalid_actions=None, temperature=1.0, gumbel_noise=None, nthreads=1, tree=None):
    """Whole MuZero.act() for the default MLP trio.  Returns a dict."""
    obs = np.ascontiguousarray(obs, np.float32)
    B = obs.shape[0]
    if tree is None:
        tree = Tree(B, cfg.num_simulations + 1, mlp.A, mlp.E)
    action = np.zeros(B, np.int32)
    weights = np.zeros((B, mlp.A), np.float32)
    root_value = np.zeros(B, np.float32)
    depth_sum = np.zeros(B, np.int64)
    k = np.ascontiguousarray(key, np.uint32)
    nz = None if dirichlet_noise is None else np.ascontiguousarray(dirichlet_noise, np.float32)
    iv = None if invalid_actions is None else np.ascontiguousarray(invalid_actions, np.uint8)
    g = None if gumbel_noise is None else np.ascontiguousarray(gumbel_noise, np.float32)
    if nz is None:
        dirichlet_fraction = 0.0
    t, c, m = tree.c(), cfg.c(B), mlp.c()
    lib().mzo_act_mlp(C.byref(m), C.byref(c), C.byref(t), _p(obs, _f32p), _p(k, _u32p),
                      _p(nz, _f32p) if nz is not None else None, C.c_float(dirichlet_fraction),
                      _p(iv, _u8p) if iv is not None else None, C.c_float(temperature),
                      _p(g, _f32p) if g is not None else None,
                      _p(action, _i32p), _p(weights, _f32p), _p(root_value, _f32p),
                      _p(depth_sum, _i64p), C.c_int(nthreads))
    return {"action": action, "action_weights": weights, "root_value": root_value,
            "depth_sum": depth_sum, "tree": tree}


# --------------------------------------------------------------------------
# Gumbel MuZero
# --------------------------------------------------------------------------

def considered_visits(m, num_simulations):
    seq = np.zeros(max(num_simulations, 1), np.int32)
    lib().mzo_considered_visits(C.c_int(m), C.c_int(num_simulations), _p(seq, _i32p))
    return seq[:num_simulations]


def qtransform(tree: Tree, node, kind):
    """kind 0: by_parent_and_siblings, 1: completed_by_mix_value; node [B] -> [B,A]."""
    out = np.zeros((tree.B, tree.A), np.float32)
    t = tree.c()
    for b in range(tree.B):
        lib().mzo_qtransform(C.byref(t), C.c_int(b), C.c_int(int(node[b])), C.c_int(kind), _p(out[b], _f32p))
    return out


def mask_root_logits(prior_logits, invalid=None):
    """mctx _mask_invalid_actions (the only root preprocessing of gumbel_muzero_policy)."""
    pl = np.array(prior_logits, np.float32, copy=True)
    if invalid is not None and np.asarray(invalid).any():
        inv = np.asarray(invalid).astype(bool)
        any_inv = inv.any(axis=1, keepdims=True)
        shifted = np.where(inv, np.finfo(np.float32).min, pl - pl.max(axis=1, keepdims=True)).astype(np.float32)
        pl = np.where(any_inv, shifted, pl)
    return pl


def gumbel_step_select(tree: Tree, cfg: SearchCfg, root_gumbel, qtransform_kind=1, max_considered=16):
    B = tree.B
    parent = np.zeros(B, np.int32)
    action = np.zeros(B, np.int32)
    depth = np.zeros(B, np.int32)
    g = np.ascontiguousarray(root_gumbel, np.float32)
    t, c = tree.c(), cfg.c(B)
    lib().mzo_gumbel_step_select(C.byref(t), C.byref(c), C.c_int(qtransform_kind), _p(g, _f32p),
                                 C.c_int(max_considered), _p(parent, _i32p), _p(action, _i32p), _p(depth, _i32p))
    return parent, action, depth


def gumbel_finish(tree: Tree, root_gumbel, qtransform_kind=1):
    B, A = tree.B, tree.A
    g = np.ascontiguousarray(root_gumbel, np.float32)
    action = np.zeros(B, np.int32)
    weights = np.zeros((B, A), np.float32)
    t = tree.c()
    for b in range(B):
        ab = C.c_int32()
        lib().mzo_gumbel_finish(C.byref(t), C.c_int(b), C.c_int(qtransform_kind), _p(g, _f32p), None,
                                C.byref(ab), _p(weights[b], _f32p))
        action[b] = ab.value
    return action, weights
