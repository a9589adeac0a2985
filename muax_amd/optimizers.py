"""Optimiser factories of the reference on torch.optim (muax/optimizers.py:5-87; README-era
muax/frameworks/coax/model.py:23-71).  Host-side plumbing, not a kernel."""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

import torch


class Optimizer:
    """A deferred torch optimiser: bound to parameters at MuZero.init() (optax's init/update split)."""

    def __init__(self, factory: Callable, lr_lambda: Optional[Callable] = None, clip_by_global_norm: float = 0.0):
        self._factory, self._lr_lambda, self.clip = factory, lr_lambda, clip_by_global_norm
        self.opt = self.sched = None

    def init(self, params):
        params = list(params)
        self.opt = self._factory(params)
        if isinstance(self.opt, (torch.optim.Adam, torch.optim.AdamW)) and params and all(p.is_cuda for p in params):
            try:  # one multi-tensor kernel for all 18 small arrays instead of a dozen launches
                fused = type(self.opt)(params, fused=True, **{k: v for k, v in self.opt.defaults.items()
                                                              if k in ("lr", "betas", "eps", "weight_decay", "amsgrad")})
                self.opt = fused
            except (RuntimeError, TypeError, ValueError):
                pass
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, self._lr_lambda) if self._lr_lambda else None
        return self.state_dict()

    def state_dict(self):
        """Optimiser moments AND the schedule position (optax keeps both in one opt_state; so does this)."""
        out = {"opt": self.opt.state_dict()}
        if self.sched is not None:
            out["sched"] = {"last_epoch": self.sched.last_epoch, "_step_count": getattr(self.sched, "_step_count", 0)}
        return out

    def load_state_dict(self, state):
        """Restore what state_dict() saved (after init()): moments, step counts, learning-rate schedule."""
        opt_state = state["opt"] if "opt" in state else state  # (older checkpoints stored the bare torch dict)
        self.opt.load_state_dict(opt_state)
        sched = state.get("sched") if isinstance(state, dict) else None
        if self.sched is not None and sched is not None:
            self.sched.last_epoch = int(sched["last_epoch"])
            self.sched._step_count = int(sched.get("_step_count", self.sched.last_epoch + 1))
            lrs = [base * lmbda(self.sched.last_epoch) for base, lmbda in zip(self.sched.base_lrs, self.sched.lr_lambdas)]
            for group, lr in zip(self.opt.param_groups, lrs):
                group["lr"] = lr
            self.sched._last_lr = lrs

    def step(self):
        if self.clip:
            torch.nn.utils.clip_grad_norm_([p for g in self.opt.param_groups for p in g["params"]], self.clip)
        self.opt.step()
        if self.sched:
            self.sched.step()
        self.opt.zero_grad(set_to_none=True)


def create_optimizer(optimizer_name: str = "adam", learning_rate: float = 1e-3, scheduler: Optional[str] = None,
                     scheduler_params: Optional[Dict[str, Any]] = None,
                     optimizer_params: Optional[Dict[str, Any]] = None) -> Optimizer:
    """muax/optimizers.py:5-36 (adam / adamw / sgd / rmsprop / adagrad; exponential_decay schedule)."""
    kw = dict(optimizer_params or {})
    table = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "sgd": torch.optim.SGD,
             "rmsprop": torch.optim.RMSprop, "adagrad": torch.optim.Adagrad}
    if optimizer_name not in table:
        raise ValueError(f"Unsupported optimizer: {optimizer_name}")
    if optimizer_name == "adam":
        kw.setdefault("eps", 1e-8)  # optax.adam default
    lr_lambda = None
    if scheduler == "exponential_decay":
        sp = scheduler_params or {}
        ts, dr = sp.get("transition_steps", 1), sp.get("decay_rate", 1.0)
        lr_lambda = lambda step: dr ** (step / ts)  # noqa: E731
    elif scheduler is not None:
        raise ValueError(f"Unsupported scheduler: {scheduler}")
    return Optimizer(lambda params: table[optimizer_name](params, lr=learning_rate, **kw), lr_lambda)


def optimizer(init_value=0, peak_value=2e-2, end_value=1e-3, warmup_steps=1000, transition_steps=10000,
              decay_rate=0.8, clip_by_global_norm=1.0) -> Optimizer:
    """muax/frameworks/coax/model.py:23-71: clip by global norm -> adam -> warmup + exponential decay."""

    def schedule(step):
        if step < warmup_steps:
            lr = init_value + (peak_value - init_value) * step / warmup_steps
        else:
            lr = peak_value * decay_rate ** ((step - warmup_steps) / transition_steps)
            lr = max(lr, end_value) if decay_rate < 1 else min(lr, end_value)
        return lr / peak_value

    return Optimizer(lambda params: torch.optim.Adam(params, lr=peak_value, eps=1e-8), schedule, clip_by_global_norm)
