"""MuZero model object with the reference's act() contract (muax/model.py:16-283; README-era
constructor muax/frameworks/coax/model.py:101-110), backed by the HIP search.

    model = MuZero(network)                       # muax/model.py:43-50
    model = MuZero(repr_fn, pred_fn, dy_fn)       # muax/frameworks/coax/model.py:101-110
    model.init(rng_key, sample_input)
    a, pi, v = model.act(rng_key, obs, with_pi=True, with_value=True, num_simulations=50)

What runs where: the default MLP trio of muax/nn.py is evaluated inside the fused gfx950 kernel (one
launch per act); any other torch plugin nets run as torch modules between the step-wise kernels, at
the points where mctx calls root_fn / recurrent_fn.  There is no CPU search path.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import nn as mz_nn
from . import prng
from . import utils as mx_utils
from .nn import MZNetwork, MZNetworkParams
from .optimizers import optimizer  # noqa: F401  (the README's `muax.model.optimizer(...)`)
from .policy import GumbelMuZeroPolicy, MuZeroPolicy, Policy
from .search import MuZeroSearch, PolicyOutput, SearchConfig  # noqa: F401  (PolicyOutput: re-exported type)


def _dirichlet(key_words, alpha: float, shape, device, global_batch=None, root_offset=0) -> torch.Tensor:
    """Root exploration noise of mctx.muzero_policy: rows [root_offset, root_offset + B) of
    jax.random.dirichlet(dirichlet_key, full([A], alpha), (global_batch,)) -- jax's sampler (loggamma by
    Marsaglia-Tsang rejection on the threefry key walk, softmax) restated on the device, ONE launch
    (mzs_dirichlet, muax_amd/csrc/mz_dirichlet.cuh).  The key walk is exact; the float bits are spec-to-confirm
    against a real jax (oracle/mz_oracle.c), so `dirichlet_noise=` stays the way to inject an exact array."""
    import ctypes as C

    from . import _lib
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("muax_amd draws the root noise on the GPU (there is no CPU search path)")
    B, A = shape
    kw = (C.c_uint32 * 2)(int(key_words[0]) & 0xFFFFFFFF, int(key_words[1]) & 0xFFFFFFFF)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    device = torch.device("cuda", idx)  # ONE ordinal for the buffer, the stream and the launch
    out = torch.empty(B, A, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.load().mzs_dirichlet(idx, C.byref(kw), float(alpha), B, A, global_batch or B,
                                             root_offset, out.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    return out


def _jit_tail():
    from . import _jit
    return _jit.build_log_tail(3)


class MuZero:
    r"""MuZero algorithm (muax/model.py:16-50).

    Parameters mirror the reference: `network` (MZNetwork) or the three plugin callables, `policy_class`
    / `policy`, `optimizer`, `loss_fn`, `discount`, `support_size`.  `recurrent_pred_on` selects which
    embedding the prediction net sees inside the search: "child" as muax/model.py:272, "parent" as the
    pip-release class muax/frameworks/coax/model.py:447-448.
    """

    def __init__(self, network=None, prediction_fn=None, dynamic_fn=None, policy_class=MuZeroPolicy,
                 policy: Optional[str] = None, optimizer=None, loss_fn=None, discount: float = 0.99,
                 support_size: int = 10, recurrent_pred_on: str = "child", device=None,
                 representation_fn=None, capture_graph: bool = False, root_graph_cache: int = 2):
        if isinstance(network, MZNetwork):
            self.network = network
        else:
            rep = representation_fn if representation_fn is not None else network
            if rep is None or prediction_fn is None or dynamic_fn is None:
                raise ValueError("give either an MZNetwork or representation_fn, prediction_fn and dynamic_fn")
            self.network = MZNetwork(rep, prediction_fn, dynamic_fn)
        if policy is not None:
            if policy not in ("muzero", "gumbel"):
                raise NotImplementedError(f"policy={policy!r}: 'muzero' and 'gumbel' are built (SURVEY.md 8(f))")
            policy_class = MuZeroPolicy if policy == "muzero" else GumbelMuZeroPolicy
        if not (isinstance(policy_class, type) and issubclass(policy_class, Policy)):
            raise TypeError("policy_class must be a subclass of muax_amd.policy.Policy")
        self.repr_func, self.pred_func, self.dy_func = self.network
        self._policy = policy_class()
        self._optimizer = optimizer
        self.loss_fn = loss_fn
        self._discount = float(discount)
        self._support_size = int(support_size)
        if recurrent_pred_on not in ("child", "parent"):
            raise ValueError("recurrent_pred_on must be 'child' or 'parent'")
        self._recurrent_pred_on = recurrent_pred_on
        self.device = torch.device(device) if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        # plugin (non-default) nets: run the S x (select, recurrent_fn, expand_backup) loop as one hipGraph
        self.capture_graph = bool(capture_graph)
        # captured root-inference graphs kept per weights version (one per observation shape; each pins its static
        # input and a private memory pool: hundreds of MB for Atari-shaped batches -- INTEGRATION.md)
        self.root_graph_cache = max(1, int(root_graph_cache))
        self._params = None
        self._opt_state_saved = None  # what init() / a checkpoint produced before an optimiser was bound
        self._loaded_opt_state = None
        self._param_list = None
        self._fused_train = None
        self._disc_const = None
        self._fused = {}
        self._root_graphs = {}
        self._root_graph_lru = []
        self._weights_version = 0

    # ------------------------------------------------------------------ init / params
    def init(self, rng_key, sample_input):
        """muax/model.py:62-80: materialise the three nets for `sample_input` ([B, ...])."""
        seed = int(prng.as_key(rng_key)[0]) << 32 | int(prng.as_key(rng_key)[1])
        torch.manual_seed(seed & 0x7FFFFFFFFFFFFFFF)
        x = torch.as_tensor(np.asarray(sample_input), dtype=torch.float32)
        for m in self.network:  # lazily built layers are created on the host, then everything moves
            if isinstance(m, torch.nn.Module):
                m.to("cpu")
        with torch.no_grad():
            s = self.repr_func(x)
            self.pred_func(s)
            self.dy_func(s, torch.zeros(s.shape[0], dtype=torch.long))
        for m in self.network:
            if isinstance(m, torch.nn.Module):
                m.to(self.device)
        self._params = MZNetworkParams(*[dict(m.named_parameters()) if isinstance(m, torch.nn.Module) else None
                                         for m in self.network])
        self._weights_version += 1
        self._fused_train = None
        return self._params

    @property
    def params(self):
        return self._params

    @property
    def _opt_state(self):
        # optimiser moments + schedule position, read when asked for (building the dict costs ~10 us: not per update())
        if self._optimizer is not None and self._optimizer.opt is not None:
            return self._optimizer.state_dict()
        return self._opt_state_saved

    @_opt_state.setter
    def _opt_state(self, value):
        self._opt_state_saved = value

    @property
    def optimizer_state(self):
        return self._opt_state

    def weights_changed(self):
        """Call after modifying parameters in place (the fused kernel reads the live tensors, but
        shape/dtype checks are cached per version)."""
        self._weights_version += 1

    # ------------------------------------------------------------------ sub-networks (coax API)
    def representation(self, obs):
        """muax/frameworks/coax/model.py:144-157."""
        with torch.no_grad():
            return self.repr_func(torch.as_tensor(obs, dtype=torch.float32, device=self.device))

    def prediction(self, s):
        """muax/frameworks/coax/model.py:159-173."""
        with torch.no_grad():
            return self.pred_func(torch.as_tensor(s, dtype=torch.float32, device=self.device))

    def dynamic(self, s, a):
        """muax/frameworks/coax/model.py:175-191."""
        with torch.no_grad():
            return self.dy_func(torch.as_tensor(s, dtype=torch.float32, device=self.device),
                                torch.as_tensor(a, device=self.device))

    # ------------------------------------------------------------------ inference glue
    def _root_inference(self, params, rng_key, obs):
        """muax/model.py:251-263 -> (prior_logits [B,A], value [B], embedding [B,...]).  With
        capture_graph=True the plugin nets' root inference (dozens of small torch kernels for a convolutional
        representation net: launch-bound, 8 of config 4's 39 ms per act were gaps between them) is captured once
        per (shape, weights version) into a hipGraph and replayed on a static input buffer; the returned tensors
        are the graph's static outputs (consumed by mzs_root in stream order before the next replay)."""
        if self.capture_graph and obs.is_cuda and not torch.cuda.is_current_stream_capturing():
            key = (tuple(obs.shape), obs.dtype, obs.device, self._weights_version)
            ent = self._root_graphs.get(key)
            if ent is None:
                # drop graphs of stale weights; keep the other shapes of THIS version (a B = 1 test rollout next to
                # a batched act() must not re-capture on every alternation), at most `root_graph_cache` of them;
                # _root_graph_lru lists the keys least recently used first
                for k in [k for k in self._root_graphs if k[3] != self._weights_version]:
                    del self._root_graphs[k]
                self._root_graph_lru = [k for k in self._root_graph_lru if k in self._root_graphs]
                while len(self._root_graphs) >= self.root_graph_cache:
                    del self._root_graphs[self._root_graph_lru.pop(0)]
                static_in = obs.clone()
                cur = torch.cuda.current_stream(obs.device)
                side = torch.cuda.Stream(device=obs.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):  # lazy layer creation and MIOpen's solver search stay out of the capture
                    for _ in range(2):
                        self._root_inference_eager(static_in)
                cur.wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = self._root_inference_eager(static_in)
                ent = self._root_graphs[key] = (g, static_in, out)
                self._root_graph_lru.append(key)
            else:
                self._root_graph_lru.remove(key)
                self._root_graph_lru.append(key)  # most recently used last
            ent[1].copy_(obs)
            ent[0].replay()
            return ent[2]
        return self._root_inference_eager(obs)

    def _root_inference_eager(self, obs):
        with torch.no_grad():
            hip_root = getattr(self.repr_func, "hip_root", None)
            if hip_root is not None:  # ResNet nets: last pool + min-max + prediction net + value decode in one launch
                out = hip_root(obs, self.pred_func, self._support_size)
                if out is not None:
                    return out[2], out[1], out[0]
            s = self.repr_func(obs)
            v, logits = self.pred_func(s)
            v = mx_utils.support_to_scalar(torch.softmax(v, dim=-1), self._support_size).flatten()
        return logits, v, s

    def _recurrent_inference(self, params, rng_key, action, embedding):
        """muax/model.py:265-282 -> ((reward, discount, prior_logits, value), next_embedding)."""
        if self._recurrent_pred_on == "child" and hasattr(self.dy_func, "hip_recurrent"):
            out = self.dy_func.hip_recurrent(self.pred_func, embedding, action, self._support_size)
            if out is not None:  # ResNet nets: the whole recurrent_fn is one HIP launch (mz_conv.cuh)
                r, v, logits, next_embedding = out
                if self._disc_const is None or self._disc_const.shape != r.shape or self._disc_const.device != r.device:
                    self._disc_const = torch.full_like(r, self._discount)  # one constant tensor, not a fill per simulation
                return (r, self._disc_const, logits, v), next_embedding
        with torch.no_grad():
            r, next_embedding = self.dy_func(embedding, action)
            v, logits = self.pred_func(embedding if self._recurrent_pred_on == "parent" else next_embedding)
            r = mx_utils.support_to_scalar(torch.softmax(r, dim=-1), self._support_size).flatten()
            v = mx_utils.support_to_scalar(torch.softmax(v, dim=-1), self._support_size).flatten()
            discount = torch.ones_like(r) * self._discount
        return (r, discount, logits, v), next_embedding

    def _with_jit(self, A, E, S, call, handle=None, gumbel=False):
        """call() -- a fused act(); when the library has no instance of the kernel for this shape, build one on demand
        (muax_amd/_jit.py: one translation unit, cached on disk, planned for the policy class `gumbel` names) and call
        again; a shape outside the kernel's limits (more than 16 actions, more than 255 simulations, embeddings wider than
        64) switches `handle` to the library's generic one-launch search (mzs_mlp_allow_generic).  Re-raises the
        ValueError when neither applies: the caller then runs the step-wise path with the torch modules."""
        try:
            return call()
        except ValueError as e:
            if "no fused kernel instance" not in str(e):
                raise
            from . import _jit
            if _jit.ensure_instance(A, E, 2 * self._support_size + 1, S, gumbel=gumbel):
                return call()
            tail = _jit.build_log_tail()
            if tail and _jit.last_build_log not in MuZero._warned_stepwise:  # a compiler was there and the build FAILED: say so
                MuZero._warned_stepwise.add(_jit.last_build_log)
                import warnings
                warnings.warn(f"muax_amd: the on-demand build of a fused act() instance for num_actions={A}, embedding_dim="
                              f"{E}, num_simulations={S} failed; compiler log {_jit.last_build_log}:\n{tail}",
                              RuntimeWarning, stacklevel=3)
            # outside the fused kernel's limits (or no compiler): the generic one-launch search of the library
            if os.environ.get("MUAX_AMD_GENERIC", "1") == "0" or handle is None:
                raise
            handle.allow_generic()
            return call()

    def _native_loop(self, root):
        """The one-launch simulation loop, when the nets are ones the library evaluates itself (the reference's ResNet
        nets with the prediction net on the child's embedding): a callable for MuZeroSearch.search, else None."""
        dy = self.dy_func
        if self._recurrent_pred_on != "child" or not hasattr(dy, "hip_search_ok"):
            return None
        if not dy.hip_search_ok(self.pred_func, tuple(root[2].shape[1:]), self._support_size):
            return None
        return lambda handle, b, e: dy.hip_search(self.pred_func, handle, self._support_size, self._discount, b, e)

    # ------------------------------------------------------------------ act
    def _fused_handle(self, B, A, E, obs_dim, S, max_depth, pb_c_init, pb_c_base, tiebreak, policy="muzero",
                      qtransform="qtransform_by_parent_and_siblings", max_considered=16, gumbel_scale=1.0,
                      global_batch=None, root_offset=0):
        """The MuZeroSearch handle of this act() configuration: created once (mzs_create + its device buffers);
        a new weights version only re-binds the weight pointers (mzs_mlp_set_weights), it does not re-create the
        handle -- no allocation on the update()-then-act() cycle of fit()."""
        key = (B, A, E, obs_dim, S, max_depth, pb_c_init, pb_c_base, tiebreak, policy, qtransform, max_considered,
               gumbel_scale, global_batch, root_offset)
        h = self._fused.get(key)
        if h is None:
            s = MuZeroSearch(B, SearchConfig(A, S, E, max_depth=max_depth, tiebreak=tiebreak,
                                             pb_c_init=pb_c_init, pb_c_base=float(pb_c_base), policy=policy,
                                             qtransform=qtransform, max_num_considered_actions=max_considered,
                                             gumbel_scale=float(gumbel_scale), global_batch=global_batch,
                                             root_offset=root_offset), self.device)
            h = self._fused[key] = [s, None]
        if h[1] != self._weights_version:
            w = {k: v.detach() for k, v in mz_nn.mlp_trio_weights(self.network).items()}
            h[0].set_mlp_weights(w, obs_dim, self._support_size, self._discount, self._recurrent_pred_on)
            h[1] = self._weights_version
        return h[0]

    def _plan(self, params, rng_key, obs, num_simulations=5, temperature=1., invalid_actions=None,
              max_depth=None, loop_fn=None, qtransform=None, dirichlet_fraction=0.25, dirichlet_alpha=0.3,
              pb_c_init=1.25, pb_c_base=19652, dirichlet_noise=None, gumbel=None, tiebreak=True,
              with_tree=False, max_num_considered_actions=16, gumbel_scale=1.0, global_batch=None, root_offset=0,
              host_io=False):
        """muax/model.py:222-243 -> (PolicyOutput, root value).  `host_io`: the caller gave host observations and
        wants host results; the fused path then makes the whole round trip in one C call (mzs_act_mlp_host) and the
        PolicyOutput holds NumPy arrays."""
        if self._params is None:
            raise ValueError("call init() first")
        gumbel_policy = isinstance(self._policy, GumbelMuZeroPolicy)
        if qtransform is None:
            qtransform = "qtransform_by_parent_and_siblings"  # muax/model.py:230-231, for EVERY policy
        qtransform = getattr(qtransform, "__name__", qtransform)
        if qtransform != "qtransform_by_parent_and_siblings" and not (
                gumbel_policy and qtransform == "qtransform_completed_by_mix_value"):
            raise ValueError(f"qtransform {qtransform!r} is not implemented for this policy")
        key = prng.as_key(rng_key)
        B = obs.shape[0]
        A = self.pred_func.num_actions if hasattr(self.pred_func, "num_actions") else None
        fused_ok = mz_nn.is_default_mlp_trio(self.network) and obs.ndim == 2
        self._last_fused = None  # the fused handle of this act(), if any (its outputs come back in one copy)
        host_io = (host_io and fused_ok and not with_tree and gumbel is None and not isinstance(obs, torch.Tensor)
                   and not isinstance(dirichlet_noise, torch.Tensor) and not isinstance(invalid_actions, torch.Tensor))
        if gumbel_policy and fused_ok and type(self._policy) is GumbelMuZeroPolicy:
            try:
                h = self._fused_handle(B, A, self.repr_func.embedding_dim, obs.shape[1], num_simulations, max_depth,
                                       1.25, 19652, False, "gumbel", qtransform, max_num_considered_actions,
                                       gumbel_scale, global_batch, root_offset)
                E_ = self.repr_func.embedding_dim
                if host_io:
                    a_, w_, v_ = self._with_jit(A, E_, num_simulations, lambda: h.act_mlp_host(
                        obs, key, dirichlet_fraction=0.0, invalid_actions=invalid_actions), h, gumbel=True)
                    return PolicyOutput(a_, w_, None), v_
                out = self._with_jit(A, E_, num_simulations, lambda: h.act_mlp(
                    obs, key, invalid_actions=invalid_actions, gumbel=gumbel, with_tree=with_tree), h, gumbel=True)
                self._last_fused = h
                return out, h.root_value
            except ValueError as e:
                if "no fused kernel instance" not in str(e):
                    raise
                self._warn_stepwise(A, self.repr_func.embedding_dim, num_simulations, e)
        def device_obs():  # plugin nets (and the step-wise fall-back) run on device tensors
            return obs if isinstance(obs, torch.Tensor) and obs.device == self.device \
                else torch.as_tensor(obs, dtype=torch.float32).to(self.device)

        if gumbel_policy:
            root = self._root_inference(params, key, device_obs())
            out = self._checked_search(lambda: self._policy(
                params, key, root, self._recurrent_inference, num_simulations=num_simulations,
                invalid_actions=invalid_actions, max_depth=max_depth, qtransform=qtransform,
                max_num_considered_actions=max_num_considered_actions, gumbel_scale=gumbel_scale,
                gumbel=gumbel, with_tree=with_tree, graph=self.capture_graph, graph_version=self._weights_version,
                global_batch=global_batch, root_offset=root_offset, native_loop=self._native_loop(root)))
            return out, root[1]
        if host_io and type(self._policy) is MuZeroPolicy:
            try:  # NumPy in, NumPy out: one C call (staging, root-noise draw from the key, search, one download, sync)
                h = self._fused_handle(B, A, self.repr_func.embedding_dim, obs.shape[1], num_simulations, max_depth,
                                       pb_c_init, pb_c_base, tiebreak, global_batch=global_batch, root_offset=root_offset)
                a_, w_, v_ = self._with_jit(A, self.repr_func.embedding_dim, num_simulations, lambda: h.act_mlp_host(
                    obs, key, dirichlet_noise=dirichlet_noise, dirichlet_fraction=dirichlet_fraction,
                    dirichlet_alpha=dirichlet_alpha, invalid_actions=invalid_actions, temperature=temperature), h)
                return PolicyOutput(a_, w_, None), v_
            except ValueError as e:
                if "no fused kernel instance" not in str(e):
                    raise
        if dirichlet_noise is None and dirichlet_fraction:
            k_dir = prng.split(key, 3)[1]  # mctx: rng_key, dirichlet_rng_key, search_rng_key = split(key, 3)
            if A is None:
                with torch.no_grad():
                    A = self.pred_func(self.repr_func(torch.as_tensor(obs[:1], dtype=torch.float32).to(self.device)))[1].shape[-1]
            # a shard draws exactly its rows of the whole batch's noise: results do not depend on the split
            dirichlet_noise = _dirichlet(k_dir, dirichlet_alpha, (B, A), self.device, global_batch, root_offset)
        if fused_ok and type(self._policy) is MuZeroPolicy:
            E = self.repr_func.embedding_dim
            try:
                h = self._fused_handle(B, A, E, obs.shape[1], num_simulations, max_depth, pb_c_init, pb_c_base,
                                       tiebreak, global_batch=global_batch, root_offset=root_offset)
                out = self._with_jit(A, E, num_simulations, lambda: h.act_mlp(
                    obs, key, dirichlet_noise=dirichlet_noise, dirichlet_fraction=dirichlet_fraction,
                    invalid_actions=invalid_actions, temperature=temperature, gumbel=gumbel, with_tree=with_tree), h)
                self._last_fused = h
                return out, h.root_value
            except ValueError as e:
                if "no fused kernel instance" not in str(e):
                    raise
                self._warn_stepwise(A, E, num_simulations, e)
        root = self._root_inference(params, key, device_obs())

        def run():
            return self._policy(params, key, root, self._recurrent_inference, num_simulations=num_simulations,
                                temperature=temperature, invalid_actions=invalid_actions, max_depth=max_depth,
                                dirichlet_fraction=dirichlet_fraction, dirichlet_noise=dirichlet_noise,
                                pb_c_init=pb_c_init, pb_c_base=pb_c_base, gumbel=gumbel, tiebreak=tiebreak,
                                with_tree=with_tree, graph=self.capture_graph, graph_version=self._weights_version,
                                global_batch=global_batch, root_offset=root_offset, native_loop=self._native_loop(root))

        return self._checked_search(run), root[1]

    _warned_stepwise = set()

    def _warn_stepwise(self, A, E, S, err):
        """Loud, once per shape: the default MLP trio normally runs inside the library (a fused instance, one built on
        demand, or the generic one-launch search); a shape none of them takes (support_size outside 8..31, more than 64
        actions, MUAX_AMD_JIT=0 and MUAX_AMD_GENERIC=0) drops to the step-wise kernels + torch modules."""
        key = (A, E, self._support_size, S)
        if key not in MuZero._warned_stepwise:
            MuZero._warned_stepwise.add(key)
            import warnings
            warnings.warn(f"muax_amd: no in-library act() route for num_actions={A}, embedding_dim={E}, support_size="
                          f"{self._support_size}, num_simulations={S} ({err}); falling back to the step-wise search "
                          f"with torch modules, which is far slower (about 70x a tuned instance at 4096 roots x 50 "
                          f"simulations; the generic one-launch route is about 16x)"
                          + (f"; last failed on-demand build: {_jit_tail()}" if _jit_tail() else ""),
                          RuntimeWarning, stacklevel=4)

    def _checked_search(self, run):
        """Run a step-wise search; if the ResNet recurrent kernel went through pair mode (two workgroups per
        root meeting in L2, mz_conv.cuh) and any rendezvous was lost -- in ANY launch of the search, graph
        replays included -- drop pair mode and repeat the search with one workgroup per root.  Both launch
        shapes produce the same bits, so the repeat is what the undisturbed search would have returned.
        Reading the status words synchronises the device: a pair-mode search (<= 128 roots of the ResNet nets) is
        not sync-free even with device_outputs=True."""
        out = run()
        dy = self.dy_func
        if getattr(dy, "_pair_scratch", None) and dy.pair_lost():
            import warnings
            warnings.warn("muax_amd: the pair-mode recurrent kernel lost a rendezvous between the two workgroups of a "
                          "root; pair mode is disabled for this process and the search was repeated with one "
                          "workgroup per root", RuntimeWarning)
            dy.disable_pair_mode()
            self._weights_version += 1  # captured graphs hold pair-mode launches: re-capture
            out = run()
            if getattr(dy, "_pair_scratch", None):  # the repeat must not have gone through pair mode again
                raise RuntimeError("muax_amd: pair mode is still active after it was disabled")
        return out

    def act(self, rng_key, obs, with_pi: bool = False, with_value: bool = False, obs_from_batch: bool = False,
            num_simulations: int = 5, temperature: float = 1., invalid_actions=None, max_depth: int = None,
            loop_fn=None, qtransform=None, dirichlet_fraction: float = 0.25, dirichlet_alpha: float = 0.3,
            pb_c_init: float = 1.25, pb_c_base: float = 19652, *, dirichlet_noise=None, gumbel=None,
            tiebreak: bool = True, device_outputs: bool = False, max_num_considered_actions: int = 16,
            gumbel_scale: float = 1.0, global_batch: Optional[int] = None, root_offset: int = 0):
        r"""Acts given environment observations (muax/model.py:82-179, same arguments and defaults).

        Returns `action[, action_weights][, root_value]` in the reference's order.  Unbatched: action is a
        python int, action_weights keeps its leading 1 ([1, A], as muax/model.py:176 leaves it), root_value
        a python float.  Batched (`obs_from_batch=True`): arrays of shape [B], [B, A], [B] -- NumPy by
        default (one host sync, like the reference's np.asarray), or device tensors with
        `device_outputs=True` (no sync).  `root_value` is the NETWORK value of the root, as the reference.
        Keyword-only extras: exact `dirichlet_noise` / `gumbel` arrays, `tiebreak=False` to drop mctx's
        1e-7 tie-break noise; for the Gumbel policy `max_num_considered_actions` and `gumbel_scale`, which
        the reference documents (muax/model.py:142-147) but never plumbs through.  Multi-GPU: a rank that holds
        rows [root_offset, root_offset + B) of a `global_batch`-root batch passes both, and gets exactly the rows
        the un-sharded call would have produced (per-root PRNG streams are indexed by the global root).
        """
        if isinstance(obs, torch.Tensor):
            obs = obs.to(torch.float32)
            if not obs_from_batch:
                obs = obs.unsqueeze(0)
        else:  # host observations stay NumPy: the search handle stages them through pinned memory
            obs = np.asarray(obs, dtype=np.float32)
            if not obs_from_batch:
                obs = obs[None]
        plan_output, root_value = self._plan(
            self.params, rng_key, obs, num_simulations=num_simulations, temperature=temperature,
            invalid_actions=invalid_actions, max_depth=max_depth, loop_fn=loop_fn, qtransform=qtransform,
            dirichlet_fraction=dirichlet_fraction, dirichlet_alpha=dirichlet_alpha, pb_c_init=pb_c_init,
            pb_c_base=pb_c_base, dirichlet_noise=dirichlet_noise, gumbel=gumbel, tiebreak=tiebreak,
            max_num_considered_actions=max_num_considered_actions, gumbel_scale=gumbel_scale,
            global_batch=global_batch, root_offset=root_offset, host_io=not (device_outputs and obs_from_batch))
        if isinstance(plan_output.action, np.ndarray):  # the fused path's host round trip: results are already NumPy
            action, weights = plan_output.action, plan_output.action_weights
            if not obs_from_batch:
                action, root_value = int(action[0]), float(root_value[0])
        elif device_outputs and obs_from_batch:
            # the search handle's output buffers are reused by the next act(): hand out copies (stream-ordered, no sync)
            if self._last_fused is not None and getattr(self._last_fused, "_out", None) is not None:
                action, weights, root_value = self._last_fused.outputs_clone()  # one copy kernel instead of three
            else:
                action, weights, root_value = plan_output.action.clone(), plan_output.action_weights.clone(), root_value.clone()
        else:
            if self._last_fused is not None:
                # fused path, device inputs: the three outputs share one allocation -> one device-to-host copy, one sync
                action, weights, root_value = self._last_fused.outputs_to_host()
            else:
                # one device-to-host copy instead of three (each costs a synchronisation): action, weights, value
                A = plan_output.action_weights.shape[1]
                B = plan_output.action.shape[0]
                host = torch.cat([plan_output.action.to(torch.float32), plan_output.action_weights.reshape(-1),
                                  root_value.reshape(-1).to(torch.float32)]).cpu().numpy()
                action = host[:B].astype(np.int32)
                weights = host[B:B + B * A].reshape(B, A).copy()
                root_value = host[B + B * A:].copy()
            if not obs_from_batch:
                action, root_value = int(action[0]), float(root_value[0])  # weights keep their leading 1 (muax/model.py:176)
        if with_pi and with_value:
            return action, weights, root_value
        elif not with_pi and with_value:
            return action, root_value
        elif with_pi and not with_value:
            return action, weights
        return action

    # ------------------------------------------------------------------ next tier
    def update(self, batch, *args, backend: str = "auto", dp_mean: bool = True, **kwargs):
        """muax/model.py:181-201: one gradient step on a batch of k-step trajectories; returns
        {'loss': float}.  With the default MLP trio and the default loss the loss and all gradients come from
        ONE fused forward+backward HIP kernel (mzs_mlp_loss_grad, muax_amd/csrc/mz_train.cuh) into a flat
        vector; the data-parallel gradient mean is then one all-reduce of that vector (RCCL on GPUs) and the
        optimiser (muax/optimizers.py mirror on torch.optim) consumes views of it.  Plugin nets or a custom
        loss_fn take the torch autograd route (backend="torch" forces it; "hip" refuses to fall back).
        `dp_mean=False` skips the gradient mean over the ranks of an initialised process group (a rank-local step:
        bench.py times update() with and without its collective)."""
        from . import loss as mz_loss
        from . import optimizers as mz_opt
        from .sharding import allreduce_mean_flat
        if self._params is None:
            raise ValueError("call init() first")
        def all_params():
            return [p for m in self.network if isinstance(m, torch.nn.Module) for p in m.parameters()]
        if self._optimizer is None:
            self._optimizer = mz_opt.create_optimizer()
        if self._optimizer.opt is None:
            loaded = self._loaded_opt_state
            self._opt_state = self._optimizer.init(all_params())
            if loaded is not None:
                # a checkpoint was loaded before the optimiser was bound to parameters: resume its moments,
                # step counts and schedule position (the reference restores opt_state, muax/model.py:210-212)
                self._optimizer.load_state_dict(loaded)
                self._loaded_opt_state = None
        fused = backend != "torch" and self.loss_fn is None and self.device.type == "cuda" \
            and mz_nn.is_default_mlp_trio(self.network) and (self.repr_func.obs_dim or 99) <= 16 \
            and not kwargs.get("pi_all_pairs", False)
        if backend == "hip" and not fused:
            raise ValueError("backend='hip' needs the default MLP trio on a GPU and the default loss")
        if fused:
            try:
                if self._fused_train is None:
                    self._fused_train = mz_loss.FusedLossGrad(self)
                try:
                    loss, flat = self._fused_train(batch, divide_by_length=kwargs.get("divide_by_length", False))
                except ValueError as e:
                    # a shape of the default trio the library lists no training instance for: build one on demand
                    # (muax_amd/_jit.py::ensure_train_instance, once per shape, cached on disk) and call again
                    from . import _jit
                    ft = self._fused_train
                    if "no kernel instance" not in str(e) or not _jit.ensure_train_instance(ft.A, ft.E, 2 * ft.S + 1):
                        raise
                    loss, flat = ft(batch, divide_by_length=kwargs.get("divide_by_length", False))
                if dp_mean:
                    allreduce_mean_flat([flat])
                for p, g in zip(self._fused_train.params, self._fused_train.views):
                    p.grad = g
            except ValueError as e:
                if backend == "hip" or "no kernel instance" not in str(e):
                    raise
                fused = False
        if not fused:
            loss_fn = self.loss_fn or mz_loss.default_loss_fn
            loss = loss_fn(self, batch, *args, **kwargs)
            loss.backward()
            if dp_mean:
                allreduce_mean_flat([p.grad for p in all_params()])
        self._optimizer.step()
        self._weights_version += 1
        return {"loss": float(loss.item())}

    def save_load(self, file, save=True):
        """muax/model.py:203-212, on torch state dicts (the reference pickles haiku params)."""
        mods = {n: m for n, m in zip(("representation", "prediction", "dynamic"), self.network)
                if isinstance(m, torch.nn.Module)}
        if save:
            torch.save({"params": {n: m.state_dict() for n, m in mods.items()}, "optimizer_state": self._opt_state},
                       file)
        elif str(file).endswith(".npy") or (not os.path.exists(file) and os.path.exists(f"{file}.npy")):
            # a checkpoint written by the reference (jnp.save of haiku params): best-effort reader, no jax needed
            from .checkpoint import load_reference_params
            load_reference_params(self, str(file))
        else:
            saved = torch.load(file, map_location=self.device)
            for n, m in mods.items():
                m.load_state_dict(saved["params"][n])
            st = saved.get("optimizer_state")
            self._opt_state = st
            if st is not None:
                if self._optimizer is not None and self._optimizer.opt is not None:
                    self._optimizer.load_state_dict(st)
                else:
                    self._loaded_opt_state = st  # applied when update() binds the optimiser
            self._weights_version += 1

    def save(self, file):
        """muax/model.py `save`: parameters + optimiser state to `file`."""
        self.save_load(file, save=True)

    def load(self, file):
        self.save_load(file, save=False)
