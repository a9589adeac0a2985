"""Host side of the HIP search: torch tensors in, C-ABI calls out.

Counterpart of what ``muax.MuZero._plan`` hands to ``mctx.muzero_policy``
(reference muax/model.py:222-243, muax/policy.py:13-30).  PyTorch is used for
device memory and streams only; all search arithmetic runs in
``libmzsearch.so`` (hand-written gfx950 kernels).  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import NamedTuple, Optional

import numpy as np
import torch

from . import _lib
from ._lib import (MLP_WEIGHT_NAMES, TREE_FIELDS, TREE_INT_FIELDS, MzsActArgs, MzsActHostArgs, MzsConfig,
                   MzsMlpWeights, MzsTreeView)


class SearchTree(NamedTuple):
    """mctx.Tree arrays ([B,N], [B,N,A], [B,N,E]) as torch tensors."""
    node_visits: torch.Tensor
    raw_values: torch.Tensor
    node_values: torch.Tensor
    parents: torch.Tensor
    action_from_parent: torch.Tensor
    children_index: torch.Tensor
    children_prior_logits: torch.Tensor
    children_values: torch.Tensor
    children_visits: torch.Tensor
    children_rewards: torch.Tensor
    children_discounts: torch.Tensor
    embeddings: torch.Tensor


class PolicyOutput(NamedTuple):
    """mctx.PolicyOutput (action, action_weights, search_tree)."""
    action: torch.Tensor
    action_weights: torch.Tensor
    search_tree: Optional[SearchTree]


@dataclass
class SearchConfig:
    """Keyword arguments of MuZero.act that reach mctx.muzero_policy (muax/model.py:86-95)."""
    num_actions: int
    num_simulations: int
    embed_dim: int
    max_depth: Optional[int] = None
    tiebreak: bool = True          # mctx adds 1e-7*uniform tie-break noise at every selection
    pb_c_init: float = 1.25
    pb_c_base: float = 19652.0
    global_batch: Optional[int] = None
    root_offset: int = 0
    policy: str = "muzero"         # "muzero" (muax/policy.py:13-30) or "gumbel" (muax/policy.py:33-47)
    qtransform: str = "qtransform_by_parent_and_siblings"   # or "qtransform_completed_by_mix_value" (gumbel)
    max_num_considered_actions: int = 16
    gumbel_scale: float = 1.0


_NO_ACT_CACHE = bool(__import__("os").environ.get("MUAX_AMD_NO_ACT_CACHE"))  # A/B switch of act_mlp's argument cache


def key_words(key) -> tuple:
    """Accept an int seed, a uint32[2] array/tensor (JAX PRNGKey data) or a (hi, lo) tuple."""
    if isinstance(key, (int, np.integer)):
        k = int(key)
        return ((k >> 32) & 0xFFFFFFFF, k & 0xFFFFFFFF)  # jax.random.PRNGKey(seed)
    if isinstance(key, torch.Tensor):
        key = key.detach().cpu().numpy()
    a = np.asarray(key).astype(np.uint64).ravel()
    if a.size != 2:
        raise ValueError("rng_key must be an int seed or two uint32 words")
    return (int(a[0]) & 0xFFFFFFFF, int(a[1]) & 0xFFFFFFFF)


def fn_identity(fn):
    """Hashable identity of a callable that survives bound-method re-creation."""
    return (getattr(fn, "__func__", fn), id(getattr(fn, "__self__", None)))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class MuZeroSearch:
    """One handle = one (device, batch shard, search configuration).

    Output aliasing: `action`, `action_weights`, `root_value`, `search_value`, `depth_sum` (and the tensors of an
    exported tree) are PERSISTENT buffers of the handle -- act_mlp() / finish() return views of them and the next
    call on the same handle overwrites them in stream order (select() also stages its actions in `action`).
    Clone what must outlive the next call (MuZero.act(device_outputs=True) does)."""

    def __init__(self, batch: int, cfg: SearchConfig, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("muax_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.batch, self.cfg = int(batch), cfg
        self._L = _lib.load()
        c = MzsConfig()
        c.struct_size = C.sizeof(MzsConfig)
        c.device = self._dev_index
        c.batch, c.num_actions = self.batch, cfg.num_actions
        c.num_simulations, c.embed_dim = cfg.num_simulations, cfg.embed_dim
        c.max_depth = cfg.max_depth or 0
        try:
            c.policy = {"muzero": 0, "gumbel": 1}[cfg.policy]
            c.qtransform = {"qtransform_by_parent_and_siblings": 0, "qtransform_completed_by_mix_value": 1}[
                getattr(cfg.qtransform, "__name__", cfg.qtransform)]
        except KeyError as e:
            raise ValueError(f"unknown policy / qtransform: {e}") from None
        c.tiebreak = int(bool(cfg.tiebreak))
        c.max_num_considered_actions, c.gumbel_scale = cfg.max_num_considered_actions, cfg.gumbel_scale
        c.pb_c_init, c.pb_c_base = cfg.pb_c_init, cfg.pb_c_base
        c.global_batch = cfg.global_batch or self.batch
        c.root_offset = cfg.root_offset
        h = C.c_void_p()
        _lib.check(self._L.mzs_create(C.byref(c), C.byref(h)))
        self._h = h
        self._weights = None
        B, A = self.batch, cfg.num_actions
        with torch.cuda.device(self.device):
            # action | action_weights | root_value in ONE allocation: a NumPy caller gets them with one copy
            self._out = torch.empty(B * (2 + A), dtype=torch.float32, device=self.device)
            self.action = self._out[:B].view(torch.int32)
            self.action_weights = self._out[B:B + B * A].view(B, A)
            self.root_value = self._out[B + B * A:]
            self.search_value = torch.empty(B, dtype=torch.float32, device=self.device)
            self.depth_sum = torch.empty(B, dtype=torch.int32, device=self.device)
        self._out_host = None   # pinned mirror of _out (outputs_to_host)
        self._obs_stage = None  # (pinned host [B, obs_dim], its NumPy view, device twin, copy-done event)
        self._tree = None
        self._parent_emb = None
        self._graphs = {}  # (recurrent_fn) -> captured simulation loop
        self._act_cache = None  # act_mlp: (argument signature, filled MzsActArgs, tree, kept tensors, noise given)

    def close(self):
        if getattr(self, "_h", None):
            self._L.mzs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        # torch's current stream of this device, raw (torch.cuda.current_stream() builds a Stream object: ~6 us per call)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return C.c_void_p(raw(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _f32(self, t, shape, name):
        if t is None:
            return None
        t = torch.as_tensor(t, device=self.device)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t.to(torch.float32).contiguous()

    def _u8(self, t, shape, name):
        if t is None:
            return None
        t = torch.as_tensor(t, device=self.device)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return (t != 0).to(torch.uint8).contiguous()

    def _alloc_tree(self) -> SearchTree:
        if self._tree is None:
            B, N = self.batch, self.cfg.num_simulations + 1
            A, E = self.cfg.num_actions, self.cfg.embed_dim
            shapes = {"embeddings": (B, N, E)}
            out = {}
            for f in TREE_FIELDS:
                shp = shapes.get(f, (B, N, A) if f.startswith("children_") else (B, N))
                dt = torch.int32 if f in TREE_INT_FIELDS else torch.float32
                out[f] = torch.empty(shp, dtype=dt, device=self.device)
            self._tree = SearchTree(**out)
        return self._tree

    def _tree_view(self, tree: SearchTree) -> MzsTreeView:
        v = MzsTreeView()
        for f in TREE_FIELDS:
            setattr(v, f, getattr(tree, f).data_ptr())
        return v

    def outputs_to_host(self):
        """(action int32 [B], action_weights f32 [B, A], root_value f32 [B]) as fresh NumPy arrays: ONE
        device-to-host copy into pinned memory and one stream synchronisation (the reference's np.asarray /
        .item() sync, muax/model.py:173-174)."""
        B, A = self.batch, self.cfg.num_actions
        if self._out_host is None:
            self._out_host = torch.empty(B * (2 + A), dtype=torch.float32).pin_memory()
            self._out_host_np = self._out_host.numpy()
        self._out_host.copy_(self._out, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        h = self._out_host_np
        return h[:B].view(np.int32).copy(), h[B:B + B * A].reshape(B, A).copy(), h[B + B * A:].copy()

    def outputs_clone(self):
        """(action, action_weights, root_value) of the last fused act() as views of ONE fresh device buffer: a single
        stream-ordered copy (the handle's own buffers are overwritten by the next act())."""
        B, A = self.batch, self.cfg.num_actions
        c = self._out.clone()
        return c[:B].view(torch.int32), c[B:B + B * A].view(B, A), c[B + B * A:]

    def _stage_obs(self, obs, obs_dim):
        """Host observations (NumPy / CPU tensor) -> device through a pinned staging buffer, asynchronously."""
        B = self.batch
        if isinstance(obs, torch.Tensor):
            if obs.is_cuda:
                return self._f32(obs, (B, obs_dim), "obs")
            obs = obs.detach().numpy()
        obs = np.asarray(obs)
        if obs.shape != (B, obs_dim):
            raise ValueError(f"obs: expected shape {(B, obs_dim)}, got {obs.shape}")
        if self._obs_stage is None or self._obs_stage[0].shape[1] != obs_dim:
            host = torch.empty(B, obs_dim, dtype=torch.float32).pin_memory()
            self._obs_stage = (host, host.numpy(), torch.empty(B, obs_dim, dtype=torch.float32, device=self.device),
                               torch.cuda.Event())
        host, host_np, dev, ev = self._obs_stage
        ev.synchronize()  # the previous upload has left the staging buffer (no-op when it has, or never ran)
        np.copyto(host_np, obs, casting="same_kind")
        with torch.cuda.device(self.device):
            dev.copy_(host, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.device))
        return dev

    # ------------------------------------------------------------------ fused path
    def set_mlp_weights(self, weights: dict, obs_dim: int, support_size: int = 10,
                        discount: float = 0.99, recurrent_pred_on: str = "child"):
        """Weights of the default MLP trio (muax/nn.py:59-115), haiku layout w[in][out]."""
        A, E = self.cfg.num_actions, self.cfg.embed_dim
        F, H = 2 * support_size + 1, 16
        shapes = {"repr_w": (obs_dim, E), "repr_b": (E,),
                  "pv_w1": (E, H), "pv_b1": (H,), "pv_w2": (H, F), "pv_b2": (F,),
                  "pp_w1": (E, H), "pp_b1": (H,), "pp_w2": (H, A), "pp_b2": (A,),
                  "dr_w1": (E + A, H), "dr_b1": (H,), "dr_w2": (H, F), "dr_b2": (F,),
                  "dn_w1": (E + A, H), "dn_b1": (H,), "dn_w2": (H, E), "dn_b2": (E,)}
        keep = {n: self._f32(weights[n], shapes[n], n) for n in MLP_WEIGHT_NAMES}
        w = MzsMlpWeights()
        w.struct_size = C.sizeof(MzsMlpWeights)
        w.obs_dim, w.support_size = obs_dim, support_size
        w.recurrent_pred_on = {"child": 0, "parent": 1}[recurrent_pred_on]
        w.discount = discount
        for n in MLP_WEIGHT_NAMES:
            setattr(w, n, keep[n].data_ptr())
        _lib.check(self._L.mzs_mlp_set_weights(self._h, C.byref(w)), self._h)
        self._weights = (keep, obs_dim)  # keep the device buffers alive

    def allow_generic(self, allow: bool = True):
        """Let act_mlp / act_mlp_host serve default-trio shapes without a fused-kernel instance through the library's
        generic one-launch search (mzs_mlp_allow_generic) instead of raising "no fused kernel instance"."""
        _lib.check(self._L.mzs_mlp_allow_generic(self._h, int(bool(allow))), self._h)

    def act_mlp(self, obs, key, dirichlet_noise=None, dirichlet_fraction: float = 0.25,
                invalid_actions=None, temperature: float = 1.0, gumbel=None,
                with_tree: bool = False) -> PolicyOutput:
        """Whole act(): root inference, S simulations, summary and sampling -- ONE kernel launch.
        Outputs live in self.action / action_weights / root_value / search_value / depth_sum."""
        if self._weights is None:
            raise ValueError("set_mlp_weights() first")
        B, A = self.batch, self.cfg.num_actions
        # An RL loop calls act() with the same device buffers step after step: when every tensor argument is the very
        # object (and storage) of the previous call, the checked / converted inputs and the filled argument block are
        # reused and only the key and the scalars are written (~8 us of host time per act otherwise, on the critical
        # path of a synchronised act: the GPU idles while the host prepares the launch).
        sig = (id(obs), id(dirichlet_noise), id(invalid_actions), id(gumbel), bool(with_tree),
               tuple(t.data_ptr() if isinstance(t, torch.Tensor) else None for t in (obs, dirichlet_noise, invalid_actions, gumbel)))
        fast = self._act_cache if (self._act_cache is not None and self._act_cache[0] == sig and not _NO_ACT_CACHE
                                   and isinstance(obs, torch.Tensor) and obs.is_cuda) else None
        if fast is not None:
            _, a, tree, keep, has_noise = fast
        else:
            obs_t = self._stage_obs(obs, self._weights[1])
            noise = self._f32(dirichlet_noise, (B, A), "dirichlet_noise")
            inv = self._u8(invalid_actions, (B, A), "invalid_actions")
            gum = self._f32(gumbel, (B, A), "gumbel")
            a = MzsActArgs()
            a.struct_size = C.sizeof(MzsActArgs)
            a.obs, a.dirichlet_noise = obs_t.data_ptr(), (noise.data_ptr() if noise is not None else None)
            a.invalid_actions = inv.data_ptr() if inv is not None else None
            a.gumbel = gum.data_ptr() if gum is not None else None
            a.action, a.action_weights = self.action.data_ptr(), self.action_weights.data_ptr()
            a.root_value, a.search_value = self.root_value.data_ptr(), self.search_value.data_ptr()
            a.depth_sum = self.depth_sum.data_ptr()
            tree = None
            if with_tree:
                tree = self._alloc_tree()
                view = self._tree_view(tree)
                a.tree = C.pointer(view)
            keep = (obs_t, noise, inv, gum, obs, dirichlet_noise, invalid_actions, gumbel, view if with_tree else None)  # (ids stay unique while held)
            has_noise = noise is not None
            # cached only when the kernel reads the caller's own storage (a converted copy would go stale under an
            # in-place update of the original)
            same = all(x is None or (isinstance(y, torch.Tensor) and x.data_ptr() == y.data_ptr())
                       for x, y in ((obs_t, obs), (noise, dirichlet_noise), (inv, invalid_actions), (gum, gumbel)))
            self._act_cache = (sig, a, tree, keep, has_noise) if same else None
        k = key_words(key)
        a.key[0], a.key[1] = k
        a.dirichlet_fraction = dirichlet_fraction if has_noise else 0.0
        a.temperature = temperature
        _lib.check(self._L.mzs_act_mlp(self._h, C.byref(a), self._stream()), self._h)
        self._keep = keep  # alive until the stream has consumed them
        return PolicyOutput(self.action, self.action_weights, tree)

    def act_mlp_host(self, obs: np.ndarray, key, dirichlet_noise=None, dirichlet_fraction: float = 0.25,
                     dirichlet_alpha: float = 0.3, invalid_actions=None, temperature: float = 1.0):
        """The reference's act() round trip in ONE C call (mzs_act_mlp_host): NumPy observations in, (action int32
        [B], action_weights f32 [B, A], root_value f32 [B]) NumPy out, synchronised.  Root noise: `dirichlet_noise`
        if given, else drawn on the device from split(key, 3)[1] (muzero policy, dirichlet_fraction != 0)."""
        if self._weights is None:
            raise ValueError("set_mlp_weights() first")
        B, A, od = self.batch, self.cfg.num_actions, self._weights[1]
        obs = np.ascontiguousarray(obs, np.float32)
        if obs.shape != (B, od):
            raise ValueError(f"obs: expected shape {(B, od)}, got {obs.shape}")
        a = MzsActHostArgs()
        a.struct_size = C.sizeof(MzsActHostArgs)
        a.obs = obs.ctypes.data
        keep = [obs]
        if dirichlet_noise is not None:
            nz = np.ascontiguousarray(dirichlet_noise, np.float32)
            if nz.shape != (B, A):
                raise ValueError(f"dirichlet_noise: expected shape {(B, A)}, got {nz.shape}")
            a.dirichlet_noise = nz.ctypes.data
            keep.append(nz)
        if invalid_actions is not None:
            iv = np.ascontiguousarray(np.asarray(invalid_actions) != 0, np.uint8)
            if iv.shape != (B, A):
                raise ValueError(f"invalid_actions: expected shape {(B, A)}, got {iv.shape}")
            a.invalid_actions = iv.ctypes.data
            keep.append(iv)
        a.draw_dirichlet = 1
        k = key_words(key)
        a.key[0], a.key[1] = k
        a.dirichlet_fraction, a.dirichlet_alpha, a.temperature = dirichlet_fraction, dirichlet_alpha, temperature
        action = np.empty(B, np.int32)
        weights = np.empty((B, A), np.float32)
        value = np.empty(B, np.float32)
        a.action, a.action_weights, a.root_value = action.ctypes.data, weights.ctypes.data, value.ctypes.data
        _lib.check(self._L.mzs_act_mlp_host(self._h, C.byref(a), self._stream()), self._h)
        return action, weights, value

    # ------------------------------------------------------------------ step-wise path
    def root(self, prior_logits, value, embedding, key=0, invalid_actions=None,
             dirichlet_noise=None, dirichlet_fraction: float = 0.25):
        """RootFnOutput -> tree (mctx muzero_policy prelude + instantiate_tree_from_root)."""
        B, A, E = self.batch, self.cfg.num_actions, self.cfg.embed_dim
        pl = self._f32(prior_logits, (B, A), "prior_logits")
        v = self._f32(value, (B,), "value")
        emb = self._f32(torch.as_tensor(embedding, device=self.device).reshape(B, -1), (B, E),
                        "embedding")
        inv = self._u8(invalid_actions, (B, A), "invalid_actions")
        noise = self._f32(dirichlet_noise, (B, A), "dirichlet_noise")
        kw = (C.c_uint32 * 2)(*key_words(key))
        _lib.check(self._L.mzs_root(self._h, _ptr(pl), _ptr(v), _ptr(emb), _ptr(inv), _ptr(noise),
                                    C.c_float(dirichlet_fraction if noise is not None else 0.0),
                                    C.byref(kw), self._stream()), self._h)
        self._keep = (pl, v, emb, inv, noise)
        if self._parent_emb is None:
            self._parent_emb = torch.empty(B, E, dtype=torch.float32, device=self.device)

    def root_gumbel(self, prior_logits, value, embedding, key=0, invalid_actions=None, gumbel=None):
        """Gumbel MuZero root (mctx gumbel_muzero_policy prelude): invalid-action mask only, root Gumbel
        noise drawn from the key (or injected)."""
        B, A, E = self.batch, self.cfg.num_actions, self.cfg.embed_dim
        pl = self._f32(prior_logits, (B, A), "prior_logits")
        v = self._f32(value, (B,), "value")
        emb = self._f32(torch.as_tensor(embedding, device=self.device).reshape(B, -1), (B, E), "embedding")
        inv = self._u8(invalid_actions, (B, A), "invalid_actions")
        g = self._f32(gumbel, (B, A), "gumbel")
        kw = (C.c_uint32 * 2)(*key_words(key))
        _lib.check(self._L.mzs_root_gumbel(self._h, _ptr(pl), _ptr(v), _ptr(emb), _ptr(inv), _ptr(g),
                                           C.byref(kw), self._stream()), self._h)
        self._keep = (pl, v, emb, inv, g)
        if self._parent_emb is None:
            self._parent_emb = torch.empty(B, E, dtype=torch.float32, device=self.device)

    def select(self, sim: int):
        """One mctx simulate(): returns (action [B] int32, parent embedding [B,E])."""
        if self._parent_emb is None:
            raise ValueError("root() first")
        _lib.check(self._L.mzs_select(self._h, sim, _ptr(self.action), _ptr(self._parent_emb),
                                      self._stream()), self._h)
        return self.action, self._parent_emb

    def expand_backup(self, sim: int, reward, discount, prior_logits, value, next_embedding):
        """RecurrentFnOutput + next embedding -> expand + backward."""
        B, A, E = self.batch, self.cfg.num_actions, self.cfg.embed_dim
        r = self._f32(reward, (B,), "reward")
        d = self._f32(discount, (B,), "discount")
        pl = self._f32(prior_logits, (B, A), "prior_logits")
        v = self._f32(value, (B,), "value")
        ne = self._f32(torch.as_tensor(next_embedding, device=self.device).reshape(B, -1), (B, E),
                       "next_embedding")
        _lib.check(self._L.mzs_expand_backup(self._h, sim, _ptr(r), _ptr(d), _ptr(pl), _ptr(v),
                                             _ptr(ne), self._stream()), self._h)
        self._keep = (r, d, pl, v, ne)

    def expand_backup_select(self, sim: int, reward, discount, prior_logits, value, next_embedding):
        """expand_backup(sim) and select(sim + 1) in ONE launch (mzs_expand_backup_select): returns the next
        simulation's (action, parent embedding) -- the same persistent buffers select() hands out -- or None after
        the last simulation."""
        if self._parent_emb is None:
            raise ValueError("root() first")
        B, A, E = self.batch, self.cfg.num_actions, self.cfg.embed_dim
        r = self._f32(reward, (B,), "reward")
        d = self._f32(discount, (B,), "discount")
        pl = self._f32(prior_logits, (B, A), "prior_logits")
        v = self._f32(value, (B,), "value")
        ne = self._f32(torch.as_tensor(next_embedding, device=self.device).reshape(B, -1), (B, E),
                       "next_embedding")
        pe = self._parent_emb
        if ne.untyped_storage().data_ptr() == pe.untyped_storage().data_ptr():
            # an identity (or slicing / offset-view) recurrent_fn: the launch's tail overwrites the buffer select()
            # handed out while the expansion still reads `ne` -- any view of that storage is copied first
            ne = ne.clone()
        _lib.check(self._L.mzs_expand_backup_select(self._h, sim, _ptr(r), _ptr(d), _ptr(pl), _ptr(v), _ptr(ne),
                                                    _ptr(self.action), _ptr(self._parent_emb), self._stream()),
                   self._h)
        self._keep = (r, d, pl, v, ne)
        return (self.action, self._parent_emb) if sim + 1 < self.cfg.num_simulations else None

    def finish(self, temperature: float = 1.0, gumbel=None, with_tree: bool = False) -> PolicyOutput:
        B, A = self.batch, self.cfg.num_actions
        gum = self._f32(gumbel, (B, A), "gumbel")
        _lib.check(self._L.mzs_finish(self._h, C.c_float(temperature), _ptr(gum), _ptr(self.action),
                                      _ptr(self.action_weights), _ptr(self.search_value),
                                      _ptr(self.depth_sum), self._stream()), self._h)
        self._keep = (gum,)
        tree = None
        if with_tree:
            tree = self._alloc_tree()
            view = self._tree_view(tree)
            _lib.check(self._L.mzs_tree_export(self._h, C.byref(view), self._stream()), self._h)
        return PolicyOutput(self.action, self.action_weights, tree)

    def search(self, root_fn_output, recurrent_fn, key=0, invalid_actions=None,
               dirichlet_noise=None, dirichlet_fraction=0.25, temperature=1.0, gumbel=None,
               with_tree=False, graph=False, graph_key=None, graph_version=0, native_loop=None) -> PolicyOutput:
        """mctx.muzero_policy with a caller-supplied recurrent_fn(action, embedding) ->
        (reward, discount, prior_logits, value, next_embedding) of torch tensors.

        graph=True captures the S x (select -> recurrent_fn -> expand_backup) loop into ONE hipGraph
        (torch.cuda.CUDAGraph; our kernels are launched on torch's current stream, so the capture sees
        them) and replays it on later calls: the loop is launch-bound -- two tree kernels plus the
        plugin's own small kernels per simulation -- and a graph launch removes the per-kernel host cost.
        recurrent_fn must then be capture-safe (no host synchronisation, static shapes); the per-call
        PRNG keys live in device memory (root) or in the un-captured root/finish calls.  `graph_version`
        is the caller's weights version: host-side work of recurrent_fn (e.g. re-packing convolution weights)
        is frozen into a capture, so a new version re-captures and replaces the old graph.

        `native_loop(handle, sim_begin, sim_end)`: nets whose recurrent_fn the library evaluates itself (the reference's
        ResNet nets: ResNetDynamic.hip_search -> mzs_resnet_search) run the WHOLE simulation loop as one launch -- every
        root advanced by its own workgroup(s), recurrent_fn and the tree update back to back, no per-simulation launch
        and no row copies; `recurrent_fn` is then only the fall-back (a tree without cached decisions)."""
        prior_logits, value, embedding = root_fn_output

        def do_root():
            if self.cfg.policy == "gumbel":
                self.root_gumbel(prior_logits, value, embedding, key, invalid_actions, gumbel)
            else:
                self.root(prior_logits, value, embedding, key, invalid_actions, dirichlet_noise,
                          dirichlet_fraction)

        def loop():
            # select(0), then S x (recurrent_fn -> expand + backward of sim AND the selection of sim + 1 in one launch)
            nxt = self.select(0)
            for sim in range(self.cfg.num_simulations):
                nxt = self.expand_backup_select(sim, *recurrent_fn(*nxt))

        do_root()
        if native_loop is not None and getattr(self, "_native_loop_unusable", False):
            native_loop = None  # (found out on an earlier call: this handle's tree has no cached decisions)
        if native_loop is not None:
            self.select(0)  # simulate() of simulation 0; every later one is the tail of its predecessor inside the launch
            ev = getattr(self, "time_native_loop", None)  # (start, end) events of a profiler (bench.py), else None
            try:
                if ev:
                    ev[0].record()
                native_loop(self, 0, self.cfg.num_simulations)
                if ev:
                    ev[1].record()
            except ValueError as e:
                if "cached decisions" not in str(e):
                    raise
                native_loop = None  # (trees beyond the cached-decision budget keep the per-simulation launches)
                self._native_loop_unusable = True
                do_root()  # select(0) above has already counted simulation 0's depth; the fall-back loop starts from the root again
        if native_loop is not None:
            pass
        elif not graph:
            loop()
        else:
            gkey = graph_key if graph_key is not None else fn_identity(recurrent_fn)
            entry = self._graphs.get(gkey)
            if entry is not None and entry[3] != graph_version:
                entry = None
            if entry is None:
                # warm-up run (lazy library initialisation must not happen under capture), then the tree
                # is rebuilt from the same root and the loop is captured (capture records, it does not run)
                with torch.no_grad():
                    side = torch.cuda.Stream(device=self.device)
                    side.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(side):
                        loop()
                    torch.cuda.current_stream(self.device).wait_stream(side)
                    do_root()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        loop()
                entry = self._graphs[gkey] = (g, recurrent_fn, self._keep, graph_version)
            entry[0].replay()
        return self.finish(temperature, None if self.cfg.policy == "gumbel" else gumbel, with_tree)
