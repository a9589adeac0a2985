"""Policy adapters of the reference (muax/policy.py): a uniform __call__(params, rng_key, root,
recurrent_fn, **kwargs) over the search implementations -- here the HIP search instead of mctx."""
from __future__ import annotations

from abc import ABC, abstractmethod

from .search import MuZeroSearch, PolicyOutput, SearchConfig, fn_identity


class Policy(ABC):
    """muax/policy.py:7-10."""

    @abstractmethod
    def __call__(self, params, rng_key, root, recurrent_fn=None, decision_recurrent_fn=None,
                 chance_recurrent_fn=None, **kwargs):
        pass


class MuZeroPolicy(Policy):
    """muax/policy.py:13-30: mctx.muzero_policy with the same keyword defaults.

    `root` is (prior_logits [B,A], value [B], embedding [B,...]) as in mctx.RootFnOutput;
    `recurrent_fn(params, rng_key, action, embedding)` returns ((reward, discount, prior_logits, value),
    next_embedding) as in muax/model.py:265-282.  Runs the step-wise HIP search (any plugin nets)."""

    def __init__(self):
        self._handles = {}

    def _handle(self, batch, cfg_key, cfg, device):
        key = (batch, cfg_key, str(device))
        if key not in self._handles:
            self._handles[key] = MuZeroSearch(batch, cfg, device)
        return self._handles[key]

    def __call__(self, params, rng_key, root, recurrent_fn, **kwargs) -> PolicyOutput:
        prior_logits, value, embedding = root
        if kwargs.get("qtransform") not in (None, "qtransform_by_parent_and_siblings"):
            raise ValueError("only qtransform_by_parent_and_siblings is implemented")
        B, A = prior_logits.shape
        emb = embedding.reshape(B, -1)
        S = kwargs.get("num_simulations", 5)
        cfg_key = (A, S, emb.shape[1], kwargs.get("max_depth"), kwargs.get("tiebreak", True),
                   kwargs.get("pb_c_init", 1.25), kwargs.get("pb_c_base", 19652), kwargs.get("global_batch"),
                   kwargs.get("root_offset", 0))
        cfg = SearchConfig(A, S, emb.shape[1], max_depth=kwargs.get("max_depth"),
                           tiebreak=kwargs.get("tiebreak", True), pb_c_init=kwargs.get("pb_c_init", 1.25),
                           pb_c_base=float(kwargs.get("pb_c_base", 19652)), global_batch=kwargs.get("global_batch"),
                           root_offset=kwargs.get("root_offset", 0))
        h = self._handle(B, cfg_key, cfg, prior_logits.device)
        shape = tuple(embedding.shape[1:])

        def rec(action, flat_emb):
            (reward, discount, logits, v), nxt = recurrent_fn(params, None, action, flat_emb.reshape((B,) + shape))
            return reward, discount, logits, v, nxt.reshape(B, -1)

        return h.search((prior_logits, value, emb), rec, key=rng_key,
                        invalid_actions=kwargs.get("invalid_actions"),
                        dirichlet_noise=kwargs.get("dirichlet_noise"),
                        dirichlet_fraction=kwargs.get("dirichlet_fraction", 0.25),
                        temperature=kwargs.get("temperature", 1.0), gumbel=kwargs.get("gumbel"),
                        with_tree=kwargs.get("with_tree", False), graph=kwargs.get("graph", False),
                        graph_key=(fn_identity(recurrent_fn), id(params), shape),
                        graph_version=kwargs.get("graph_version", 0), native_loop=kwargs.get("native_loop"))


class GumbelMuZeroPolicy(Policy):
    """muax/policy.py:33-47: mctx.gumbel_muzero_policy with the same keyword defaults
    (max_num_considered_actions=16, gumbel_scale=1).  Runs on the step-wise HIP kernels.

    qtransform: mctx's own default for this policy is qtransform_completed_by_mix_value, which is what
    this adapter uses when called directly.  Note the reference quirk one level up: MuZero._plan always
    substitutes qtransform_by_parent_and_siblings when none is given (muax/model.py:230-231), so going
    through MuZero.act the Gumbel search runs with THAT transform unless one is passed explicitly --
    muax_amd.MuZero reproduces this."""

    def __init__(self):
        self._handles = {}

    def __call__(self, params, rng_key, root, recurrent_fn=None, decision_recurrent_fn=None,
                 chance_recurrent_fn=None, **kwargs) -> PolicyOutput:
        prior_logits, value, embedding = root
        B, A = prior_logits.shape
        emb = embedding.reshape(B, -1)
        S = kwargs.get("num_simulations", 5)
        qt = kwargs.get("qtransform") or "qtransform_completed_by_mix_value"
        qt = getattr(qt, "__name__", qt)
        key = (B, A, S, emb.shape[1], kwargs.get("max_depth"), qt, kwargs.get("max_num_considered_actions", 16),
               kwargs.get("gumbel_scale", 1), str(prior_logits.device), kwargs.get("global_batch"),
               kwargs.get("root_offset", 0))
        if key not in self._handles:
            cfg = SearchConfig(A, S, emb.shape[1], max_depth=kwargs.get("max_depth"), tiebreak=False,
                               policy="gumbel", qtransform=qt,
                               max_num_considered_actions=kwargs.get("max_num_considered_actions", 16),
                               gumbel_scale=float(kwargs.get("gumbel_scale", 1)), global_batch=kwargs.get("global_batch"),
                               root_offset=kwargs.get("root_offset", 0))
            self._handles[key] = MuZeroSearch(B, cfg, prior_logits.device)
        h = self._handles[key]
        shape = tuple(embedding.shape[1:])

        def rec(action, flat_emb):
            (reward, discount, logits, v), nxt = recurrent_fn(params, None, action, flat_emb.reshape((B,) + shape))
            return reward, discount, logits, v, nxt.reshape(B, -1)

        return h.search((prior_logits, value, emb), rec, key=rng_key,
                        invalid_actions=kwargs.get("invalid_actions"), gumbel=kwargs.get("gumbel"),
                        with_tree=kwargs.get("with_tree", False), graph=kwargs.get("graph", False),
                        graph_key=(fn_identity(recurrent_fn), id(params), shape),
                        graph_version=kwargs.get("graph_version", 0), native_loop=kwargs.get("native_loop"))


class StochasticMuZeroPolicy(Policy):
    """muax/policy.py:50-67: out of scope (chance nodes; nothing in the core reference supplies the
    decision/chance recurrent functions)."""

    def __call__(self, params, rng_key, root, recurrent_fn=None, decision_recurrent_fn=None,
                 chance_recurrent_fn=None, **kwargs):
        raise NotImplementedError("stochastic MuZero is out of scope (SURVEY.md section 2, row 2)")
