"""k-step unrolled MuZero loss of the reference (muax/loss.py:10-88): SURVEY.md 8(f) n1.

Two routes to the same number.  `default_loss_fn` is the formula in plain PyTorch ops (autograd does the backward
pass): it serves plugin nets and custom losses, and it is the reference the fused kernel is tested against.
`FusedLossGrad` (below) is the product path for the default MLP trio: loss and every gradient from ONE hand-written
forward+backward HIP kernel (muax_amd/csrc/mz_train.cuh through mzs_mlp_loss_grad).  The reference's formula:

    loss = sum_{i<L} [ mean_B CE(r_logits_i, support(r_i)) + mean_B CE(v_logits_i, support(Rn_i))
                       + mean_B CE(pi_logits_i, pi_i) ]  +  1e-4 * 0.5 * sum ||param||^2

with the hidden state's gradient halved at the start of every dynamics step (Appendix G,
muax/loss.py:60-61).  Reference quirks handled explicitly:
  * muax/loss.py:83 reads `loss` before assignment; the only consistent reading is an initial 0;
  * the coax variant divides by L (muax/frameworks/coax/loss.py:70-71): `divide_by_length=True`;
  * muax stores pi per step as [1, A] (muax/model.py:176), so batch.pi[:, i] is [B, 1, A] against logits
    [B, A] and optax broadcasts to an all-pairs [B, B] cross entropy.  Here pi is squeezed to [B, L, A]
    (the intended per-sample target); `pi_all_pairs=True` reproduces the broadcast.
"""
from __future__ import annotations

import torch

from . import utils as mx_utils
from .episode_tracer import Transition  # noqa: F401  (muax.episode_tracer.Transition)


def softmax_cross_entropy(logits, labels):
    """optax.softmax_cross_entropy: -sum(labels * log_softmax(logits), -1)."""
    return -(labels * torch.log_softmax(logits, dim=-1)).sum(dim=-1)


def default_loss_fn(muzero_instance, batch: Transition, divide_by_length: bool = False,
                    pi_all_pairs: bool = False):
    """muax/loss.py:10-88.  `batch` fields are tensors/arrays of shape [B, L, ...]."""
    dev = muzero_instance.device
    t = lambda x, dt=torch.float32: torch.as_tensor(x, dtype=dt, device=dev)  # noqa: E731
    a = t(batch.a, torch.long)
    B, L = a.shape[:2]
    a = a.reshape(B, L)
    S = muzero_instance._support_size
    r_t = mx_utils.scalar_to_support(t(batch.r).reshape(B, L), S).detach()
    Rn_t = mx_utils.scalar_to_support(t(batch.Rn).reshape(B, L), S).detach()
    pi = t(batch.pi)
    pi = pi.reshape(B, L, 1, -1) if pi_all_pairs else pi.reshape(B, L, -1)
    obs = t(batch.obs)
    s = muzero_instance.repr_func(obs[:, 0])
    loss = torch.zeros((), device=dev)
    for i in range(L):
        v, logits = muzero_instance.pred_func(s)
        s = mx_utils.scale_gradient(s, 0.5)  # Appendix G
        r, ns = muzero_instance.dy_func(s, a[:, i])
        loss = loss + softmax_cross_entropy(r, r_t[:, i]).mean() \
            + softmax_cross_entropy(v, Rn_t[:, i]).mean() \
            + softmax_cross_entropy(logits, pi[:, i].detach()).mean()
        s = ns
    if divide_by_length:
        loss = loss / L
    l2 = 0.5 * sum((p ** 2).sum() for m in muzero_instance.network if isinstance(m, torch.nn.Module)
                   for p in m.parameters())
    return loss + 1e-4 * l2


# ---------------------------------------------------------------------------------------------------
# HIP path: loss and gradients of the default MLP trio in one fused forward+backward kernel
# (muax_amd/csrc/mz_train.cuh through the C-ABI entry mzs_mlp_loss_grad).
# ---------------------------------------------------------------------------------------------------
class FusedLossGrad:
    """value_and_grad of the default loss (muax/loss.py:10-88 via muax/model.py:245-249) for the default
    MLP trio.  The gradient comes back as ONE flat fp32 vector in the C-ABI's array order (that is also the
    buffer a data-parallel all-reduce works on); `views` maps it onto the 18 parameters."""

    def __init__(self, muzero_instance):
        import ctypes as C

        from . import _lib
        from . import nn as mz_nn
        if not mz_nn.is_default_mlp_trio(muzero_instance.network):
            raise ValueError("the fused training step is built for the default MLP trio")
        self.m, self._C, self._lib = muzero_instance, C, _lib
        self._L = _lib.load()
        params = mz_nn.mlp_trio_weights(muzero_instance.network)
        self.params = [params[n] for n in _lib.MLP_WEIGHT_NAMES]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("muax_amd needs a ROCm GPU (gfx950); there is no CPU fallback")
        r, p, _ = muzero_instance.network
        self.obs_dim, self.E, self.A = r.obs_dim, r.embedding_dim, p.num_actions
        self.S = muzero_instance._support_size
        n = int(self._L.mzs_mlp_num_params(self.obs_dim, self.E, self.A, self.S))
        assert n == sum(x.numel() for x in self.params)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for x in self.params:
            self.views.append(self.grads[off:off + x.numel()].view_as(x))
            off += x.numel()
        self._ws = None
        self._w = self._w_keep = self._w_ptrs = None

    def __call__(self, batch: Transition, divide_by_length: bool = False):
        C, _lib, dev = self._C, self._lib, self.grads.device

        def t(x, dt):  # (tensors that already are what the kernel reads pass through untouched)
            if isinstance(x, torch.Tensor) and x.dtype == dt and x.device == dev and x.is_contiguous():
                return x
            return torch.as_tensor(x, device=dev).to(dt).contiguous()
        a = t(batch.a, torch.int32)
        B, L = a.shape[:2]
        a = a.reshape(B, L)
        obs = t(batch.obs, torch.float32)[:, 0].reshape(B, -1).contiguous()
        r, Rn = t(batch.r, torch.float32).reshape(B, L), t(batch.Rn, torch.float32).reshape(B, L)
        pi = t(batch.pi, torch.float32).reshape(B, L, self.A)
        if obs.shape[1] != self.obs_dim:
            raise ValueError(f"batch.obs has {obs.shape[1]} features, the network takes {self.obs_dim}")
        need = int(self._L.mzs_mlp_train_workspace_bytes(B, self.obs_dim, self.E, self.A, self.S))
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
        # the weight struct is rebuilt only when a parameter tensor moved (optimisers update in place)
        ptrs = tuple(x.data_ptr() for x in self.params)
        if self._w is None or ptrs != self._w_ptrs or not all(x.is_contiguous() for x in self.params):
            w = _lib.MzsMlpWeights()
            w.struct_size = C.sizeof(_lib.MzsMlpWeights)
            w.obs_dim, w.support_size = self.obs_dim, self.S
            keep = [x.detach().contiguous() for x in self.params]
            for n, x in zip(_lib.MLP_WEIGHT_NAMES, keep):
                setattr(w, n, x.data_ptr())
            self._w, self._w_keep = w, keep
            self._w_ptrs = ptrs if all(k.data_ptr() == q for k, q in zip(keep, ptrs)) else None
        w, keep = self._w, self._w_keep
        w.discount = self.m._discount
        args = _lib.MzsTrainArgs()
        args.struct_size = C.sizeof(_lib.MzsTrainArgs)
        args.device = dev.index if dev.index is not None else torch.cuda.current_device()
        args.batch, args.unroll_steps, args.num_actions, args.embed_dim = B, L, self.A, self.E
        args.obs, args.actions, args.rewards = obs.data_ptr(), a.data_ptr(), r.data_ptr()
        args.returns, args.policy = Rn.data_ptr(), pi.data_ptr()
        args.loss_scale = 1.0 / (B * L) if divide_by_length else 1.0 / B
        args.l2_coeff = 1e-4
        args.loss, args.grads = self.loss.data_ptr(), self.grads.data_ptr()
        args.workspace, args.workspace_bytes = self._ws.data_ptr(), self._ws.numel() * 4
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        stream = C.c_void_p(raw(idx) if raw is not None else torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(self._L.mzs_mlp_loss_grad(C.byref(w), C.byref(args), stream))
        self._keep = (obs, a, r, Rn, pi, keep)
        return self.loss, self.grads
