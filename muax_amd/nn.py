"""Plugin surface of the reference (muax/nn.py) as torch modules.

`MZNetwork(representation_fn, prediction_fn, dynamic_fn)` keeps the reference's contract:
representation_fn(obs)->s, prediction_fn(s)->(v_logits, pi_logits), dynamic_fn(s, a)->(r_logits, ns)
(muax/model.py:52-54).  Parameters use haiku's Linear layout and default init (w[in][out] ~
TruncatedNormal(1/sqrt(fan_in)), b = 0) so reference checkpoints map one to one.  The default MLP trio
(muax/nn.py:59-115) is what the fused gfx950 kernel evaluates in place; any other torch module goes
through the step-wise search path.
"""
from __future__ import annotations

import math
import os
from typing import Callable, NamedTuple, Optional

import torch
from torch import nn

HIDDEN = 16  # hk.Linear(16) in every default head (muax/nn.py:77,81,97,101)


class MZNetworkParams(NamedTuple):
    representation: Optional[dict] = None
    prediction: Optional[dict] = None
    dynamic: Optional[dict] = None


class MZNetwork(NamedTuple):
    representation_fn: Callable
    prediction_fn: Callable
    dynamic_fn: Callable


def min_max_normalize(s: torch.Tensor) -> torch.Tensor:
    """muax/nn.py:37-44."""
    s_min = s.min(dim=1, keepdim=True).values
    s_max = s.max(dim=1, keepdim=True).values
    s_scale = s_max - s_min
    s_scale = torch.where(s_scale < 1e-5, s_scale + 1e-5, s_scale)
    return (s - s_min) / s_scale


class HkLinear(nn.Module):
    """hk.Linear: y = x @ w + b with w[in][out]; TruncatedNormal(1/sqrt(in)) / zeros init."""

    def __init__(self, in_features: int, out_features: int, generator: Optional[torch.Generator] = None):
        super().__init__()
        w = torch.empty(in_features, out_features)
        nn.init.trunc_normal_(w, 0.0, 1.0, -2.0, 2.0, generator=generator)
        self.w = nn.Parameter(w / math.sqrt(in_features))
        self.b = nn.Parameter(torch.zeros(out_features))

    def forward(self, x):
        return x @ self.w + self.b


def _mlp2(i, o, gen):
    return nn.ModuleList([HkLinear(i, HIDDEN, gen), HkLinear(HIDDEN, o, gen)])


def _apply_mlp2(m, x):
    return m[1](torch.nn.functional.elu(m[0](x)))


class Representation(nn.Module):
    """muax/nn.py:59-70: Linear(embedding_dim) + min_max_normalize."""

    def __init__(self, embedding_dim: int, obs_dim: Optional[int] = None, generator=None, name="representation"):
        super().__init__()
        self.embedding_dim, self.obs_dim = embedding_dim, obs_dim
        self.repr_func = HkLinear(obs_dim, embedding_dim, generator) if obs_dim else None
        self._gen = generator

    def materialize(self, obs_dim: int):
        if self.repr_func is None:
            self.obs_dim = obs_dim
            self.repr_func = HkLinear(obs_dim, self.embedding_dim, self._gen)

    def forward(self, obs):
        self.materialize(obs.shape[-1])
        return min_max_normalize(self.repr_func(obs))


class Prediction(nn.Module):
    """muax/nn.py:73-90: value head and policy head, two Linear(16)-elu-Linear stacks."""

    def __init__(self, num_actions: int, full_support_size: int, embedding_dim: Optional[int] = None,
                 generator=None, name="prediction"):
        super().__init__()
        self.num_actions, self.full_support_size, self.embedding_dim = num_actions, full_support_size, embedding_dim
        self._gen = generator
        self.v_func = self.pi_func = None
        if embedding_dim:
            self.materialize(embedding_dim)

    def materialize(self, embedding_dim: int):
        if self.v_func is None:
            self.embedding_dim = embedding_dim
            self.v_func = _mlp2(embedding_dim, self.full_support_size, self._gen)
            self.pi_func = _mlp2(embedding_dim, self.num_actions, self._gen)

    def forward(self, s):
        self.materialize(s.shape[-1])
        return _apply_mlp2(self.v_func, s), _apply_mlp2(self.pi_func, s)


class Dynamic(nn.Module):
    """muax/nn.py:93-115: reward head and next-state head on [s, onehot(a)]."""

    def __init__(self, embedding_dim: int, num_actions: int, full_support_size: int, generator=None,
                 name="dynamic"):
        super().__init__()
        self.embedding_dim, self.num_actions, self.full_support_size = embedding_dim, num_actions, full_support_size
        # creation order follows the reference: ns_func first, then r_func
        self.ns_func = _mlp2(embedding_dim + num_actions, embedding_dim, generator)
        self.r_func = _mlp2(embedding_dim + num_actions, full_support_size, generator)

    def forward(self, s, a):
        sa = torch.cat([s, torch.nn.functional.one_hot(a.long(), self.num_actions).to(s.dtype)], dim=1)
        return _apply_mlp2(self.r_func, sa), min_max_normalize(_apply_mlp2(self.ns_func, sa))


def _init_representation_func(representation_module, embedding_dim):
    """muax/nn.py:417-421: returns the representation_fn callable."""
    return representation_module(embedding_dim)


def _init_prediction_func(prediction_module, num_actions, full_support_size):
    """muax/nn.py:423-427."""
    return prediction_module(num_actions, full_support_size)


def _init_dynamic_func(dynamic_module, embedding_dim, num_actions, full_support_size):
    """muax/nn.py:429-433."""
    return dynamic_module(embedding_dim, num_actions, full_support_size)


def _init_resnet_representation_func(representation_module, input_channels):
    """muax/nn.py:435-439."""
    return representation_module(input_channels=input_channels)


def _init_resnet_prediction_func(prediction_module, num_actions, full_support_size, output_channels):
    """muax/nn.py:441-445."""
    return prediction_module(num_actions, full_support_size, output_channels)


def _init_resnet_dynamic_func(dynamic_module, num_actions, full_support_size, output_channels):
    """muax/nn.py:447-451."""
    return dynamic_module(num_actions, full_support_size, output_channels)


def _init_ez_representation_func(representation_module, embedding_dim):
    """muax/nn.py:398-402."""
    return representation_module(embedding_dim)


def _init_ez_prediction_func(prediction_module, num_actions, full_support_size, output_init_scale):
    """muax/nn.py:404-408."""
    return prediction_module(num_actions, full_support_size, output_init_scale)


def _init_ez_dynamic_func(dynamic_module, embedding_dim, num_actions, full_support_size, output_init_scale):
    """muax/nn.py:410-414."""
    return dynamic_module(embedding_dim, num_actions, full_support_size, output_init_scale)


def create_muzero_network(representation_module, prediction_module, dynamic_module, embedding_dim: int,
                          num_actions: int, full_support_size: int) -> MZNetwork:
    """muax/nn.py:23-34."""
    return MZNetwork(_init_representation_func(representation_module, embedding_dim),
                     _init_prediction_func(prediction_module, num_actions, full_support_size),
                     _init_dynamic_func(dynamic_module, embedding_dim, num_actions, full_support_size))


def is_default_mlp_trio(network: MZNetwork) -> bool:
    r, p, d = network
    return type(r) is Representation and type(p) is Prediction and type(d) is Dynamic


def mlp_trio_weights(network: MZNetwork) -> dict:
    """The 18 arrays of the default trio in the C-ABI's order/layout (include/mzsearch.h)."""
    r, p, d = network
    return {"repr_w": r.repr_func.w, "repr_b": r.repr_func.b,
            "pv_w1": p.v_func[0].w, "pv_b1": p.v_func[0].b, "pv_w2": p.v_func[1].w, "pv_b2": p.v_func[1].b,
            "pp_w1": p.pi_func[0].w, "pp_b1": p.pi_func[0].b, "pp_w2": p.pi_func[1].w, "pp_b2": p.pi_func[1].b,
            "dr_w1": d.r_func[0].w, "dr_b1": d.r_func[0].b, "dr_w2": d.r_func[1].w, "dr_b2": d.r_func[1].b,
            "dn_w1": d.ns_func[0].w, "dn_b1": d.ns_func[0].b, "dn_w2": d.ns_func[1].w, "dn_b2": d.ns_func[1].b}


# ---------------------------------------------------------------------------------------------------
# Convolutional plugin nets (muax/nn.py:47-56,118-395) as torch modules (MIOpen / hipBLASLt
# kernels underneath) that the step-wise search drives -- optionally as one hipGraph (MuZero(capture_graph=True)).
# Inside the search the ResNet nets' whole recurrent_fn (ResNetDynamic + ResNetPrediction + both decodes) runs as
# ONE hand-written fp32-MFMA launch instead (mzs_resnet_tower, muax_amd/csrc/mz_conv.cuh; ResNetDynamic._tower_hip
# below); the modules remain the definition it is tested against, the root's representation net
# and the training path.  Tensors keep the reference's NHWC layout at every module boundary (embedding [B, H, W, C]);
# inside a convolution the same memory is viewed as channels_last NCHW, so no transposes are made.
# ---------------------------------------------------------------------------------------------------
def min_max_normalize2d(s: torch.Tensor) -> torch.Tensor:
    """muax/nn.py:47-56: per (sample, channel) min/max over the H*W plane of an NHWC tensor."""
    s_min = s.amin(dim=(1, 2), keepdim=True)
    s_max = s.amax(dim=(1, 2), keepdim=True)
    s_scale = s_max - s_min
    s_scale = torch.where(s_scale < 1e-5, s_scale + 1e-5, s_scale)
    return (s - s_min) / s_scale


def _same_pad(size: int, k: int, stride: int):
    """TensorFlow/haiku 'SAME': out = ceil(size / stride), the odd pixel of padding goes after."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


class HkConv2D(nn.Module):
    """hk.Conv2D(channels, kernel_shape, stride, padding='SAME', with_bias=False) on NHWC tensors;
    weight stored HWIO like haiku (w[kh][kw][in][out]), TruncatedNormal(1/sqrt(fan_in)) init."""

    def __init__(self, out_channels: int, kernel_shape: int = 3, stride: int = 1, in_channels: Optional[int] = None,
                 generator=None):
        super().__init__()
        self.out_channels, self.k, self.stride, self._gen = out_channels, kernel_shape, stride, generator
        self.w = None
        if in_channels:
            self.materialize(in_channels)

    def materialize(self, cin: int):
        if self.w is None:
            w = torch.empty(self.k, self.k, cin, self.out_channels)
            nn.init.trunc_normal_(w, 0.0, 1.0, -2.0, 2.0, generator=self._gen)
            self.w = nn.Parameter(w / math.sqrt(self.k * self.k * cin))

    use_hip = True  # C -> C (16 / 32 / 64) 3x3 stride-1 convolutions of the representation nets on mzs_conv3x3_nhwc in inference

    def _hip_ok(self, x) -> bool:
        """A HIP convolution applies: inference on a dense fp32 NHWC map on the GPU, 3x3, and either stride 1 with C -> C
        channels, C = 16, 32 or 64 (mzs_conv3x3_nhwc: every layer inside the residual blocks of the representation nets,
        42 x 42 x 32, 21 x 21, 11 x 11, 6 x 6, and the EZ encoder's 16-channel first stage) or stride 2 with 4 -> 16,
        4 -> 32, 16 -> 32 or 32 -> 64 channels (mzs_conv3x3_stride2_nhwc: the stems and the EZ encoder's strided projection
        block); the rows a run of output pixels touches must fit a CU's LDS."""
        if not (self.use_hip and self.k == 3 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            return False
        c, co, (h, w) = x.shape[-1], self.out_channels, x.shape[1:3]
        if self.stride == 1 and c == co and c in (16, 32, 64):
            # (6 x 6 x 64 on 128 images is 13.6 us here against 10.7 for the library's kernel alone, but inside the EZ
            # encoder the library's side kernels make the same layers 25 us dearer each: every size stays here)
            tiles = (h * w + 15) // 16  # the library's choice of run length (mz_repr_host.h) -> its LDS bytes
            run = 16 * ((16 if c == 16 else 14) if tiles > 16 else (8 if tiles > 8 else 4))
            if ((run + w - 1) // w + 3) * (w + 2) * (c + 4) * 4 > 160 * 1024:
                return False
        elif self.stride == 2 and (c, co) in ((4, 16), (4, 32), (16, 32), (32, 64)):
            wo = -(-w // 2)
            if (2 * ((64 + wo - 1) // wo) + 3) * ((wo - 1) * 2 + 3) * (max(c, 16) + 4) * 4 > 160 * 1024:
                return False  # (even the shortest run, 4 tiles, does not fit)
        else:
            return False
        if torch.is_grad_enabled() and (x.requires_grad or self.w.requires_grad):
            return False
        return self.w.is_cuda and self.w.dtype == torch.float32 and self.w.device == x.device

    def _packed(self):
        """The HWIO kernel in the HIP kernels' order Wp[tap][c][g][co][i] = w[tap][16 c + 4 g + i][co] (fewer than 16 input
        channels: zero rows up to 16); rebuilt when the parameter changes."""
        sig = (self.w.data_ptr(), self.w._version, self.w.device)
        if getattr(self, "_pack_sig", None) != sig:
            co, cin = self.out_channels, self.w.shape[2]
            with torch.no_grad():
                w = self.w.detach().reshape(9, cin, co)
                if cin % 16:
                    w = torch.cat([w, w.new_zeros(9, 16 - cin % 16, co)], dim=1)
                self._pack = w.reshape(9, w.shape[1] // 16, 4, 4, co).permute(0, 1, 2, 4, 3).contiguous()
            self._pack_sig = sig
        return self._pack

    def _conv_hip(self, x, in_div=None, relu=False):
        import ctypes as C

        from . import _lib
        L = _lib.load()
        xc = x.contiguous()
        wp = self._packed()
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        if self.stride == 1:
            if in_div is not None:
                xc = xc / in_div
            y = torch.empty_like(xc)
            a = _lib.MzsConv3x3Args()
            a.struct_size = C.sizeof(_lib.MzsConv3x3Args)
            a.device = dev
            a.batch, a.height, a.width, a.channels, a.relu = xc.shape[0], xc.shape[1], xc.shape[2], xc.shape[3], int(relu)
            a.x, a.w_packed, a.y = xc.data_ptr(), wp.data_ptr(), y.data_ptr()
            with torch.cuda.device(x.device):
                _lib.check(L.mzs_conv3x3_nhwc(C.byref(a), stream))
            return y
        y = torch.empty(xc.shape[0], -(-xc.shape[1] // 2), -(-xc.shape[2] // 2), self.out_channels, dtype=torch.float32, device=x.device)
        a = _lib.MzsConv3x3sArgs()
        a.struct_size = C.sizeof(_lib.MzsConv3x3sArgs)
        a.device = dev
        a.batch, a.height, a.width, a.in_channels, a.out_channels = xc.shape[0], xc.shape[1], xc.shape[2], xc.shape[3], self.out_channels
        a.relu, a.in_div = int(relu), float(in_div) if in_div is not None else 0.0
        a.x, a.w_packed, a.y = xc.data_ptr(), wp.data_ptr(), y.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(L.mzs_conv3x3_stride2_nhwc(C.byref(a), stream))
        return y

    def scaled(self, x, in_div=None, relu=False):
        """[relu](conv(x [/ in_div])): the stems of the representation nets (muax/nn.py:189,299,303: observations / 255
        in front, relu behind), the division and the relu inside the HIP kernel when it applies."""
        self.materialize(x.shape[-1])
        if self._hip_ok(x):
            return self._conv_hip(x, in_div=in_div, relu=relu)
        y = self.forward(x if in_div is None else x / in_div)
        return torch.relu(y) if relu else y

    def forward(self, x):
        self.materialize(x.shape[-1])
        if self.k == 1 and self.stride == 1:  # a 1x1 convolution on NHWC is a matrix product over the channels
            return x @ self.w[0, 0]
        if self._hip_ok(x):
            return self._conv_hip(x)
        xc = x.permute(0, 3, 1, 2)
        (ht, hb), (wl, wr) = _same_pad(x.shape[1], self.k, self.stride), _same_pad(x.shape[2], self.k, self.stride)
        pad = (0, 0)
        if ht == hb and wl == wr:
            pad = (ht, wl)  # symmetric (every stride-1 3x3): the convolution pads, no copy of the padded map
        elif ht or hb or wl or wr:
            xc = torch.nn.functional.pad(xc, (wl, wr, ht, hb))
        w = self._oihw()
        if x.shape[-1] < 8 and not (torch.is_grad_enabled() and self.w.requires_grad):
            # few input channels (raw frames): MIOpen has no fast NHWC fp32 kernel and falls back to a naive
            # one (2.2 ms for 128 x 84 x 84 x 4); the plain NCHW problem gets a proper solver
            xc, w = xc.contiguous(), w.contiguous()
            y = torch.nn.functional.conv2d(xc, w, stride=self.stride, padding=pad).contiguous(memory_format=torch.channels_last)
            return y.permute(0, 2, 3, 1)
        if not xc.is_contiguous(memory_format=torch.channels_last):  # keep every conv on packed NHWC operands
            xc = xc.contiguous(memory_format=torch.channels_last)
        y = torch.nn.functional.conv2d(xc, w, stride=self.stride, padding=pad)
        return y.permute(0, 2, 3, 1)

    def _oihw(self):
        """The HWIO parameter as a packed channels_last OIHW tensor (a strided view sends MIOpen to its naive
        kernel); rebuilt when the parameter changes, differentiable view while training."""
        if torch.is_grad_enabled() and self.w.requires_grad:
            return self.w.permute(3, 2, 0, 1)
        sig = (self.w.data_ptr(), self.w._version, self.w.dtype, self.w.device)
        if getattr(self, "_oihw_sig", None) != sig:
            self._oihw_cache = self.w.detach().permute(3, 2, 0, 1).contiguous(memory_format=torch.channels_last)
            self._oihw_sig = sig
        return self._oihw_cache


class HkLayerNorm(nn.Module):
    """hk.LayerNorm(axis, create_scale=True, create_offset=True): statistics over `axis`, scale/offset
    over the last axis, eps 1e-5, biased variance."""

    def __init__(self, axis=(-3, -2, -1)):
        super().__init__()
        self.axis = tuple(axis) if isinstance(axis, (tuple, list)) else (axis,)
        self.scale = self.offset = None

    def materialize(self, x):
        if self.scale is None:
            self.scale = nn.Parameter(torch.ones(x.shape[-1], device=x.device))
            self.offset = nn.Parameter(torch.zeros(x.shape[-1], device=x.device))

    def forward(self, x):
        self.materialize(x)
        if self.fused_ok(x):
            return ln_act(x, self)
        return self.torch_forward(x)

    def torch_forward(self, x):
        self.materialize(x)
        mean = x.mean(dim=self.axis, keepdim=True)
        var = x.var(dim=self.axis, keepdim=True, unbiased=False)
        return (x - mean) * torch.rsqrt(var + 1e-5) * self.scale + self.offset

    use_hip = True  # the fused HIP kernels (mzs_layernorm_act) in inference on the GPU; False: the torch expression

    def fused_ok(self, x) -> bool:
        """Inference on a GPU tensor whose statistics run over the whole sample (every axis but the batch's):
        mzs_layernorm_act applies -- LayerNorm, the shortcut addition and the relu behind it in two launches."""
        if not (self.use_hip and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2):
            return False
        if torch.is_grad_enabled() and (x.requires_grad or (self.scale is not None and self.scale.requires_grad)):
            return False
        for t in (self.scale, self.offset):  # the kernel reads raw pointers: same device, fp32, dense
            if t is None or t.device != x.device or t.dtype != torch.float32 or not t.is_contiguous():
                return False
        if sorted(a % x.dim() for a in self.axis) != list(range(1, x.dim())):
            return False
        n = x[0].numel()
        return n % 4 == 0 and x.shape[-1] % 4 == 0 and n < 2 ** 31


def ln_act(x, ln: "HkLayerNorm", relu: bool = False, add_ln=None, residual=None):
    """[relu]( ln(x) [+ ln2(x2)] [+ residual] ) for NHWC tensors: `add_ln` = (x2, ln2), the projected shortcut of
    ResidualConvBlockV1 (muax/nn.py:118-148); `residual` its identity shortcut.  One fused HIP call
    (mzs_layernorm_act, muax_amd/csrc/mz_norm.cuh) in inference on the GPU, the torch expressions otherwise."""
    ln.materialize(x)
    ok = ln.fused_ok(x) and (residual is None or (
        residual.device == x.device and residual.dtype == torch.float32 and residual.shape == x.shape
        and not (torch.is_grad_enabled() and residual.requires_grad)))
    if add_ln is not None:
        add_ln[1].materialize(add_ln[0])
        ok = ok and add_ln[1].fused_ok(add_ln[0]) and add_ln[0].shape == x.shape
    if not ok:
        y = ln.torch_forward(x)
        if add_ln is not None:
            y = add_ln[1].torch_forward(add_ln[0]) + y
        if residual is not None:
            y = residual + y
        return torch.relu(y) if relu else y
    import ctypes as C

    from . import _lib
    L = _lib.load()
    B, n = x.shape[0], x[0].numel()
    xs = [x.contiguous()]
    a = _lib.MzsLayerNormArgs()
    a.struct_size = C.sizeof(_lib.MzsLayerNormArgs)
    a.device = x.device.index if x.device.index is not None else torch.cuda.current_device()
    a.batch, a.n, a.channels, a.relu, a.eps = B, n, x.shape[-1], int(relu), 1e-5
    a.x, a.scale, a.offset = xs[0].data_ptr(), ln.scale.data_ptr(), ln.offset.data_ptr()
    if add_ln is not None:
        xs.append(add_ln[0].contiguous())
        a.x2, a.scale2, a.offset2 = xs[1].data_ptr(), add_ln[1].scale.data_ptr(), add_ln[1].offset.data_ptr()
    if residual is not None:
        xs.append(residual.contiguous())
        a.residual = xs[-1].data_ptr()
    y = torch.empty_like(xs[0])
    ws = torch.empty(L.mzs_layernorm_workspace_bytes(B, n) // 8, dtype=torch.float64, device=x.device)
    a.y, a.workspace, a.workspace_bytes = y.data_ptr(), ws.data_ptr(), ws.numel() * 8
    with torch.cuda.device(x.device):
        _lib.check(L.mzs_layernorm_act(C.byref(a), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
    return y


class LazyHkLinear(nn.Module):
    """hk.Linear(out, with_bias) whose input width is known at the first call."""

    def __init__(self, out_features: int, with_bias: bool = True, generator=None):
        super().__init__()
        self.out_features, self.with_bias, self._gen = out_features, with_bias, generator
        self.w = self.b = None

    def forward(self, x):
        if self.w is None:
            w = torch.empty(x.shape[-1], self.out_features)
            nn.init.trunc_normal_(w, 0.0, 1.0, -2.0, 2.0, generator=self._gen)
            self.w = nn.Parameter((w / math.sqrt(x.shape[-1])).to(x.device))
            if self.with_bias:
                self.b = nn.Parameter(torch.zeros(self.out_features, device=x.device))
        y = x @ self.w
        return y + self.b if self.with_bias else y


def avg_pool_same(x: torch.Tensor, window: int = 3, stride: int = 2) -> torch.Tensor:
    """hk.AvgPool(window_shape=(3,3,1), strides=(2,2,1), padding='SAME') on NHWC: the mean of the VALID
    elements under each window."""
    xc = x.permute(0, 3, 1, 2)
    (ht, hb), (wl, wr) = _same_pad(x.shape[1], window, stride), _same_pad(x.shape[2], window, stride)
    s = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(xc, (wl, wr, ht, hb)), window, stride, divisor_override=1)
    key = (x.shape[1], x.shape[2], window, stride, x.dtype, x.device)
    cnt = _POOL_COUNTS.get(key)
    if cnt is None:
        # the number of valid elements under each window depends on the geometry only: computed once per shape and kept
        # (a tensor made while a hipGraph is being captured belongs to the capture: it is not kept)
        ones = torch.ones(1, 1, x.shape[1], x.shape[2], dtype=x.dtype, device=x.device)
        cnt = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(ones, (wl, wr, ht, hb)), window, stride, divisor_override=1)
        # not kept: a tensor made under a capture (it belongs to the capture) or under inference mode (an inference
        # tensor cannot be saved for a later backward pass)
        if not (x.is_cuda and torch.cuda.is_current_stream_capturing()) and not torch.is_inference_mode_enabled():
            if len(_POOL_COUNTS) >= 64:
                _POOL_COUNTS.clear()
            if x.is_cuda:
                torch.cuda.current_stream(x.device).synchronize()  # complete before any other stream may read it
            _POOL_COUNTS[key] = cnt
    return (s / cnt).permute(0, 2, 3, 1)


_POOL_COUNTS = {}


class ResidualConvBlockV1(nn.Module):
    """muax/nn.py:118-148: conv-LN-relu-conv-LN, (projected) shortcut, relu."""

    def __init__(self, channels: int, stride: int, use_projection: bool, generator=None):
        super().__init__()
        self.use_projection = use_projection
        if use_projection:
            self.proj_conv, self.proj_ln = HkConv2D(channels, 3, stride, generator=generator), HkLayerNorm()
        self.conv_0, self.ln_0 = HkConv2D(channels, 3, stride, generator=generator), HkLayerNorm()
        self.conv_1, self.ln_1 = HkConv2D(channels, 3, 1, generator=generator), HkLayerNorm()

    use_hip = True  # the whole block as one C call (mzs_resblock_v1: three launches) in GPU inference

    def _hip_ok(self, x) -> bool:
        """mzs_resblock_v1 applies: inference, every layer built, stride-1 C -> C convolutions the HIP kernel takes
        (HkConv2D._hip_ok), LayerNorms over the whole sample with dense fp32 parameters on x's device."""
        if not (self.use_hip and x.dim() == 4 and x.is_contiguous() and self.conv_0.w is not None and self.conv_1.w is not None):
            return False
        convs = [self.conv_0, self.conv_1] + ([self.proj_conv] if self.use_projection else [])
        lns = [self.ln_0, self.ln_1] + ([self.proj_ln] if self.use_projection else [])
        if any(c.w is None or not c._hip_ok(x) for c in convs):
            return False
        return all(ln.scale is not None and ln.fused_ok(x) for ln in lns)

    def _forward_hip(self, x):
        import ctypes as C

        from . import _lib
        L = _lib.load()
        B, H, W, Cc = x.shape
        a = _lib.MzsResblockArgs()
        a.struct_size = C.sizeof(_lib.MzsResblockArgs)
        a.device = x.device.index if x.device.index is not None else torch.cuda.current_device()
        a.batch, a.height, a.width, a.channels, a.eps = B, H, W, Cc, 1e-5
        keep = [self.conv_0._packed(), self.conv_1._packed()]
        a.x, a.w0, a.w1 = x.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr()
        if self.use_projection:
            keep.append(self.proj_conv._packed())
            a.w_proj, a.proj_scale, a.proj_offset = keep[2].data_ptr(), self.proj_ln.scale.data_ptr(), self.proj_ln.offset.data_ptr()
        a.ln0_scale, a.ln0_offset = self.ln_0.scale.data_ptr(), self.ln_0.offset.data_ptr()
        a.ln1_scale, a.ln1_offset = self.ln_1.scale.data_ptr(), self.ln_1.offset.data_ptr()
        y = torch.empty_like(x)
        ws = torch.empty(L.mzs_resblock_workspace_bytes(B, H, W, Cc) // 8, dtype=torch.float64, device=x.device)
        a.y, a.workspace, a.workspace_bytes = y.data_ptr(), ws.data_ptr(), ws.numel() * 8
        with torch.cuda.device(x.device):
            _lib.check(L.mzs_resblock_v1(C.byref(a), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return y

    def forward(self, x):
        # ln_act = the torch expressions in training / on the CPU, one fused HIP call per chain in GPU inference
        # (module calls in the reference's creation order -- projection, conv_0, conv_1: lazily built weights draw
        # from the generator in that order)
        if self._hip_ok(x):
            return self._forward_hip(x)
        cp = self.proj_conv(x) if self.use_projection else None
        out = self.conv_1(ln_act(self.conv_0(x), self.ln_0, relu=True))
        if self.use_projection:
            return ln_act(out, self.ln_1, relu=True, add_ln=(cp, self.proj_ln))
        return ln_act(out, self.ln_1, relu=True, residual=x)


class ResidualConvBlockV2(nn.Module):
    """muax/nn.py:151-178: pre-activation block (LN-relu first; the projection reads the activated input)."""

    def __init__(self, channels: int, stride: int, use_projection: bool, generator=None):
        super().__init__()
        self.use_projection = use_projection
        if use_projection:
            self.proj_conv = HkConv2D(channels, 3, stride, generator=generator)
        self.conv_0, self.ln_0 = HkConv2D(channels, 3, stride, generator=generator), HkLayerNorm()
        self.conv_1, self.ln_1 = HkConv2D(channels, 3, 1, generator=generator), HkLayerNorm()

    use_hip = True  # the identity-shortcut block as one C call (mzs_resblock_v2: three launches) in GPU inference

    def _hip_ok(self, x) -> bool:
        """mzs_resblock_v2 applies: inference, identity shortcut, every layer built, stride-1 C -> C convolutions the HIP
        kernel takes, LayerNorms over the whole sample with dense fp32 parameters on x's device."""
        if not (self.use_hip and not self.use_projection and x.dim() == 4 and x.is_contiguous()):
            return False
        if any(c.w is None or not c._hip_ok(x) for c in (self.conv_0, self.conv_1)):
            return False
        return all(ln.scale is not None and ln.fused_ok(x) for ln in (self.ln_0, self.ln_1))

    def _forward_hip(self, x):
        import ctypes as C

        from . import _lib
        L = _lib.load()
        B, H, W, Cc = x.shape
        a = _lib.MzsResblockArgs()
        a.struct_size = C.sizeof(_lib.MzsResblockArgs)
        a.device = x.device.index if x.device.index is not None else torch.cuda.current_device()
        a.batch, a.height, a.width, a.channels, a.eps = B, H, W, Cc, 1e-5
        keep = [self.conv_0._packed(), self.conv_1._packed()]
        a.x, a.w0, a.w1 = x.data_ptr(), keep[0].data_ptr(), keep[1].data_ptr()
        a.ln0_scale, a.ln0_offset = self.ln_0.scale.data_ptr(), self.ln_0.offset.data_ptr()
        a.ln1_scale, a.ln1_offset = self.ln_1.scale.data_ptr(), self.ln_1.offset.data_ptr()
        y = torch.empty_like(x)
        ws = torch.empty(L.mzs_resblock_v2_workspace_bytes(B, H, W, Cc) // 8, dtype=torch.float64, device=x.device)
        a.y, a.workspace, a.workspace_bytes = y.data_ptr(), ws.data_ptr(), ws.numel() * 8
        with torch.cuda.device(x.device):
            _lib.check(L.mzs_resblock_v2(C.byref(a), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return y

    def forward(self, x):
        if self._hip_ok(x):
            return self._forward_hip(x)
        out = ln_act(x, self.ln_0, relu=True)
        shortcut = self.proj_conv(out) if self.use_projection else x
        out = self.conv_1(ln_act(self.conv_0(out), self.ln_1, relu=True))
        return shortcut + out


class _Seq(nn.Sequential):
    pass


def _head(conv_channels, n_convs, hidden, out, gen):
    """1x1 conv (x n_convs) - relu - flatten - Linear(hidden) - relu - Linear(out) (muax/nn.py:317-337)."""
    layers = []
    for _ in range(n_convs):
        layers += [HkConv2D(conv_channels, 1, 1, generator=gen), nn.ReLU()]
    return _Seq(*layers, nn.Flatten(1), LazyHkLinear(hidden, generator=gen), nn.ReLU(), LazyHkLinear(out, generator=gen))


class _LNReluHead(nn.Module):
    """The head EZPrediction and EZDynamic share (muax/nn.py:232-243,246-257,277-288): LayerNorm - relu -
    conv1x1(16) - LayerNorm - relu - flatten - Linear(32, no bias) - LayerNorm over the vector - relu - Linear(out)
    with hk.initializers.VarianceScaling(output_init_scale) (fan_in, truncated normal) on the last layer."""

    def __init__(self, out_features: int, output_init_scale: float, generator=None):
        super().__init__()
        self.ln_in, self.conv, self.ln_mid = HkLayerNorm(), HkConv2D(16, 1, 1, generator=generator), HkLayerNorm()
        self.fc, self.ln_vec = LazyHkLinear(32, with_bias=False, generator=generator), HkLayerNorm(axis=(-1,))
        self.out = LazyHkLinear(out_features, generator=generator)
        self._scale = float(output_init_scale)

    def forward(self, x):
        h = ln_act(self.conv(ln_act(x, self.ln_in, relu=True)), self.ln_mid, relu=True)
        h = ln_act(self.fc(h.flatten(1)), self.ln_vec, relu=True)
        fresh = self.out.w is None
        y = self.out(h)
        if fresh:  # VarianceScaling(scale): stddev sqrt(scale / fan_in) of the truncated normal (haiku divides by .8796)
            with torch.no_grad():
                self.out.w.mul_(math.sqrt(self._scale) / 0.87962566103423978)
            y = self.out(h)
        return y


class EZStateEncoder(nn.Module):
    """muax/nn.py:180-207: EfficientZero encoder, 84x84 frames -> 6x6 x channels (strides 2, 2, pool, pool)."""

    def __init__(self, channels: int, use_v2: bool = True, generator=None):
        super().__init__()
        Block = ResidualConvBlockV2 if use_v2 else ResidualConvBlockV1
        g = generator
        self.use_v2 = use_v2
        self.stem = HkConv2D(channels // 2, 3, 2, generator=g)
        self.stem_ln = None if use_v2 else HkLayerNorm()
        self.block0 = Block(channels // 2, 1, False, g)
        self.block1 = Block(channels, 2, True, g)
        self.block2, self.block3, self.block4 = Block(channels, 1, False, g), Block(channels, 1, False, g), Block(channels, 1, False, g)

    def forward(self, observations):
        x = self.stem.scaled(observations.to(torch.float32), 255.)
        if not self.use_v2:
            x = ln_act(x, self.stem_ln, relu=True)
        x = self.block2(self.block1(self.block0(x)))
        x = self.block3(avg_pool_same(x))
        return self.block4(avg_pool_same(x))


class EZRepresentation(nn.Module):
    """muax/nn.py:210-218 (no min-max normalisation: the reference's EZ variant returns the encoder output)."""

    def __init__(self, embedding_dim: int, generator=None, name="representation"):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.repr_func = EZStateEncoder(embedding_dim, generator=generator)

    def forward(self, obs):
        return self.repr_func(obs)


class EZPrediction(nn.Module):
    """muax/nn.py:221-264: one residual block, then the value head and the policy head."""

    def __init__(self, num_actions: int, full_support_size: int, output_init_scale: float, use_v2: bool = True,
                 generator=None, name="prediction"):
        super().__init__()
        self.num_actions, self.full_support_size = num_actions, full_support_size
        self._block_cls, self._gen = (ResidualConvBlockV2 if use_v2 else ResidualConvBlockV1), generator
        self.block = None
        self.v_func = _LNReluHead(full_support_size, output_init_scale, generator)
        self.pi_func = _LNReluHead(num_actions, output_init_scale, generator)

    def forward(self, s):
        if self.block is None:  # channel count follows the embedding (muax/nn.py:260)
            self.block = self._block_cls(s.shape[-1], 1, False, self._gen).to(s.device)
        o = self.block(s)
        return self.v_func(o), self.pi_func(o)


class EZDynamic(nn.Module):
    """muax/nn.py:267-309: the RAW action index enters as one extra plane (no / num_actions here, unlike
    ResNetDynamic), one 3x3 convolution with a residual connection, one residual block, the reward head."""

    def __init__(self, embedding_dim: int, num_actions: int, full_support_size: int, output_init_scale: float,
                 use_v2: bool = True, generator=None, name="dynamic"):
        super().__init__()
        self.embedding_dim, self.num_actions, self.full_support_size = embedding_dim, num_actions, full_support_size
        self.use_v2, self._gen = use_v2, generator
        self._block_cls = ResidualConvBlockV2 if use_v2 else ResidualConvBlockV1
        self.ln_in = HkLayerNorm()
        self.conv = self.block = None
        self.ln_out = None if use_v2 else HkLayerNorm()
        self.r_func = _LNReluHead(full_support_size, output_init_scale, generator)

    def forward(self, s, a):
        n, h, w, c = s.shape
        if self.conv is None:
            self.conv = HkConv2D(c, 3, 1, generator=self._gen).to(s.device)
            self.block = self._block_cls(c, 1, False, self._gen).to(s.device)
        shortcut = s
        if self.use_v2:
            s = torch.relu(self.ln_in(s))
        sa = torch.cat([s, a.to(s.dtype).reshape(n, 1, 1, 1).expand(n, h, w, 1)], dim=-1)
        out = self.conv(sa)
        out = out + shortcut if self.use_v2 else torch.relu(self.ln_out(out) + shortcut)
        out = self.block(out)
        return self.r_func(out), out

    # ---- HIP path of the whole recurrent_fn (mzs_ez_recurrent, muax_amd/csrc/mz_ez.cuh) ----
    use_hip_recurrent = True

    @staticmethod
    def _pack_conv(w, cin_pad=None):
        """HWIO [3, 3, Cin, Cout] -> the kernels' Wp[tap][Cin / 16][g][Cout][i] = W[tap][16 c + 4 g + i][co]."""
        cin, cout = w.shape[2], w.shape[3]
        if cin_pad and cin_pad != cin:
            w = torch.cat([w, w.new_zeros(3, 3, cin_pad - cin, cout)], dim=2)
            cin = cin_pad
        return w.reshape(9, cin // 16, 4, 4, cout).permute(0, 1, 2, 4, 3).contiguous()

    def _ez_packed(self, pred):
        blocks, heads = (self.block, pred.block), (self.r_func, pred.v_func, pred.pi_func)
        ps = [self.ln_in.scale, self.ln_in.offset, self.conv.w] + [p for b in blocks for p in b.parameters()] \
            + [p for h in heads for p in h.parameters()]
        sig = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in ps)
        if getattr(self, "_ez_sig", None) != sig:
            C_ = self.conv.out_channels
            with torch.no_grad():
                ln = lambda m: torch.stack([m.scale, m.offset]).contiguous()  # noqa: E731
                pk = {"d_ln_in": ln(self.ln_in), "d_conv": self._pack_conv(self.conv.w, C_ + 16),
                      "d_ln0": ln(self.block.ln_0), "d_conv0": self._pack_conv(self.block.conv_0.w),
                      "d_ln1": ln(self.block.ln_1), "d_conv1": self._pack_conv(self.block.conv_1.w),
                      "p_ln0": ln(pred.block.ln_0), "p_conv0": self._pack_conv(pred.block.conv_0.w),
                      "p_ln1": ln(pred.block.ln_1), "p_conv1": self._pack_conv(pred.block.conv_1.w)}
                for name, h in zip("rvp", heads):
                    pk[name] = {"ln_in": ln(h.ln_in), "c1": h.conv.w.reshape(C_, 16).contiguous(), "ln_mid": ln(h.ln_mid),
                                "fc": h.fc.w.detach().contiguous(), "ln_vec": ln(h.ln_vec),
                                "out_w": h.out.w.detach().contiguous(), "out_b": h.out.b.detach().contiguous()}
            self._ez_pack, self._ez_sig = pk, sig
        return self._ez_pack

    def hip_recurrent(self, pred, s, a, support_size: int):
        """The whole recurrent_fn of muax/model.py:265-282 for the EZ nets in ONE HIP launch (mzs_ez_recurrent):
        returns (reward [B], value [B], prior_logits [B, A], next_state [B, 6, 6, C]), or None when the nets are not
        the shapes the kernel is built for (the caller then runs the torch modules)."""
        def blk_ok(b, c):
            return (isinstance(b, ResidualConvBlockV2) and not b.use_projection and b.conv_0.w is not None
                    and b.conv_1.w is not None and tuple(b.conv_0.w.shape) == (3, 3, c, c)
                    and tuple(b.conv_1.w.shape) == (3, 3, c, c) and b.ln_0.scale is not None and b.ln_1.scale is not None)

        def head_ok(h, c, n):
            return (isinstance(h, _LNReluHead) and h.conv.w is not None and tuple(h.conv.w.shape) == (1, 1, c, 16)
                    and h.fc.w is not None and tuple(h.fc.w.shape) == (576, 32) and h.out.w is not None
                    and tuple(h.out.w.shape) == (32, n) and h.out.b is not None and h.ln_in.scale is not None
                    and h.ln_mid.scale is not None and h.ln_vec.scale is not None)

        F = 2 * support_size + 1
        if not (self.use_hip_recurrent and self.use_v2 and isinstance(pred, EZPrediction) and s.is_cuda
                and s.dtype == torch.float32 and s.dim() == 4 and tuple(s.shape[1:3]) == (6, 6) and s.shape[3] in (32, 64)
                and self.conv is not None and self.conv.w is not None and self.ln_in.scale is not None
                and tuple(self.conv.w.shape) == (3, 3, s.shape[3] + 1, s.shape[3]) and F <= 64
                and pred.num_actions <= 64 and pred.num_actions == self.num_actions and pred.block is not None):
            return None  # (an inference-only entry point, like ResNetDynamic.hip_recurrent: no autograd through it)
        c = s.shape[3]
        if not (blk_ok(self.block, c) and blk_ok(pred.block, c) and head_ok(self.r_func, c, F)
                and head_ok(pred.v_func, c, F) and head_ok(pred.pi_func, c, pred.num_actions)):
            return None
        import ctypes as C

        from . import _lib
        L = _lib.load()
        pk = self._ez_packed(pred)
        B = s.shape[0]
        x = s.contiguous()
        act = a.to(torch.int32).contiguous()
        y = torch.empty_like(x)
        outs = (torch.empty(B, device=x.device), torch.empty(B, device=x.device),
                torch.empty(B, pred.num_actions, device=x.device))
        args = _lib.MzsEzArgs()
        args.struct_size = C.sizeof(_lib.MzsEzArgs)
        args.device = x.device.index if x.device.index is not None else torch.cuda.current_device()
        args.batch, args.channels, args.num_actions, args.support_size = B, c, pred.num_actions, support_size
        args.x, args.action, args.y = x.data_ptr(), act.data_ptr(), y.data_ptr()
        args.reward, args.value, args.prior_logits = (o.data_ptr() for o in outs)
        for name in _lib.MzsEzArgs.WEIGHTS:
            setattr(args, name, pk[name].data_ptr())
        for name in "rvp":
            hd = getattr(args, name)
            for f in _lib.MzsEzHead.FIELDS:
                setattr(hd, f, pk[name][f].data_ptr())
        with torch.cuda.device(x.device):
            _lib.check(L.mzs_ez_recurrent(C.byref(args), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        self._ez_keep = (x, act)  # alive until the stream has consumed them
        return outs[0], outs[1], outs[2], y


class ResNetRepresentation(nn.Module):
    """muax/nn.py:291-310: uint8-range NHWC frames -> [B, H/16, W/16, 2*input_channels], min-max
    normalised per channel (84x84x4 -> 6x6x64 at the default width)."""

    def __init__(self, input_channels: int = 32, generator=None, name="representation"):
        super().__init__()
        c, g = input_channels, generator
        self.embedding_dim = None
        self.stem0 = HkConv2D(c, 3, 2, generator=g)
        self.blocks0 = nn.ModuleList([ResidualConvBlockV1(c, 1, True, g) for _ in range(2)])
        self.stem1 = HkConv2D(2 * c, 3, 2, generator=g)
        self.blocks1 = nn.ModuleList([ResidualConvBlockV1(2 * c, 1, True, g) for _ in range(3)])
        self.blocks2 = nn.ModuleList([ResidualConvBlockV1(2 * c, 1, True, g) for _ in range(3)])

    def forward(self, obs, before_last_pool: bool = False):
        x = self.stem0.scaled(obs.to(torch.float32), 255., relu=True)
        for b in self.blocks0:
            x = b(x)
        x = self.stem1.scaled(x, relu=True)
        for b in self.blocks1:
            x = b(x)
        x = avg_pool_same(x)
        for b in self.blocks2:
            x = b(x)
        return x if before_last_pool else min_max_normalize2d(avg_pool_same(x))

    use_hip_root = True  # last pool + min-max + the prediction net + the value decode as one launch (mzs_resnet_root_tail)

    def hip_root(self, obs, pred, support_size: int):
        """Root inference of muax/model.py:251-263 with the tail in ONE HIP launch: returns (embedding [B, 6, 6, 64],
        value [B], prior_logits [B, A]) or None when the nets / the input are not what mzs_resnet_root_tail is built for
        (the caller then runs the modules).  Inference only."""
        def head_ok(seq, convs, hidden):
            ws = [m for m in seq if isinstance(m, HkConv2D)]
            ls = [m for m in seq if isinstance(m, LazyHkLinear)]
            return (len(ws) == len(convs) and len(ls) == 2 and all(m.w is not None and m.w.is_cuda for m in ws + ls)
                    and [tuple(m.w.shape[2:]) for m in ws] == convs and tuple(ls[0].w.shape) == (576, hidden)
                    and all(m.with_bias for m in ls))
        if not (self.use_hip_root and isinstance(pred, ResNetPrediction) and obs.is_cuda and obs.dim() == 4
                and not torch.is_grad_enabled() and 2 * support_size + 1 <= 64 and pred.num_actions <= 64):
            return None
        if not (head_ok(pred.v_func, [(64, 16), (16, 16)], 16) and head_ok(pred.pi_func, [(64, 16)], 16)
                and pred.v_func[7].w.shape[1] == 2 * support_size + 1 and pred.pi_func[5].w.shape[1] == pred.num_actions):
            return None
        # the map before the last pool, from the input's size alone (two stride-2 stems and one pool, SAME: ceil
        # division each) -- decided BEFORE any compute, so an unsupported size does not run the net twice
        def half(n):
            return -(-n // 2)
        hw = [half(half(half(n))) for n in obs.shape[1:3]]
        if not all(half(n) == 6 for n in hw):
            return None
        heads_t = [m.w for seq in (pred.v_func, pred.pi_func) for m in seq if isinstance(m, (HkConv2D, LazyHkLinear))]
        if not all(t.dtype == torch.float32 and t.device == obs.device for t in heads_t):
            return None
        x = self.forward(obs, before_last_pool=True)
        if not (x.dtype == torch.float32 and x.shape[3] == 64 and list(x.shape[1:3]) == hw):
            return None
        import ctypes as C

        from . import _lib
        L = _lib.load()
        x = x.contiguous()
        B = x.shape[0]
        vf, pf = pred.v_func, pred.pi_func
        heads = [vf[0].w, vf[2].w, vf[5].w, vf[5].b, vf[7].w, vf[7].b, pf[0].w, pf[3].w, pf[3].b, pf[5].w, pf[5].b]
        keep = [h.detach().contiguous() for h in heads]
        a = _lib.MzsRootTailArgs()
        a.struct_size = C.sizeof(_lib.MzsRootTailArgs)
        a.device = x.device.index if x.device.index is not None else torch.cuda.current_device()
        a.batch, a.height, a.width = B, x.shape[1], x.shape[2]
        a.num_actions, a.support_size, a.normalize = pred.num_actions, support_size, 1
        a.x = x.data_ptr()
        for name, t in zip(_lib.MzsRootTailArgs.HEAD_FIELDS, keep):
            setattr(a, name, t.data_ptr())
        emb = torch.empty(B, 6, 6, 64, device=x.device)
        value, logits = torch.empty(B, device=x.device), torch.empty(B, pred.num_actions, device=x.device)
        a.embedding, a.value, a.prior_logits = emb.data_ptr(), value.data_ptr(), logits.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(L.mzs_resnet_root_tail(C.byref(a), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        self._root_keep = keep  # alive until the stream has consumed them
        return emb, value, logits


class ResNetPrediction(nn.Module):
    """muax/nn.py:313-341."""

    def __init__(self, num_actions: int, full_support_size: int, output_channels: int = 16, generator=None,
                 name="prediction"):
        super().__init__()
        self.num_actions, self.full_support_size = num_actions, full_support_size
        # creation order follows the reference: pi_func, then v_func
        self.pi_func = _head(output_channels, 1, output_channels, num_actions, generator)
        self.v_func = _head(output_channels, 2, output_channels, full_support_size, generator)

    def forward(self, s):
        return self.v_func(s), self.pi_func(s)


class ResNetDynamic(nn.Module):
    """muax/nn.py:344-378: the action enters as one extra plane a / num_actions."""

    def __init__(self, embedding_dim, num_actions: int = None, full_support_size: int = None, output_channels: int = 64,
                 generator=None, name="dynamic"):
        super().__init__()
        self._pair_scratch = {}
        if full_support_size is None:  # the reference's own signature: (num_actions, full_support_size)
            embedding_dim, num_actions, full_support_size = None, embedding_dim, num_actions
        self.num_actions, self.full_support_size = num_actions, full_support_size
        g = generator
        self.r_func = _head(output_channels, 2, output_channels, full_support_size, g)
        self.ns_stem = HkConv2D(output_channels, 1, 1, generator=g)
        self.ns_blocks = nn.ModuleList([ResidualConvBlockV1(output_channels, 1, True, g) for _ in range(8)])

    def forward(self, s, a):
        n, h, w, _ = s.shape
        plane = (a.to(s.dtype) / self.num_actions).reshape(n, 1, 1, 1).expand(n, h, w, 1)
        sa = torch.cat([s, plane], dim=-1)
        if self._hip_tower_ok(s):
            ns = self._tower_hip(s, a)  # stem + residual tower + min_max_normalize2d in ONE HIP kernel
            return self.r_func(sa), ns
        ns = torch.relu(self.ns_stem(sa))
        for b in self.ns_blocks:
            ns = b(ns)
        return self.r_func(sa), min_max_normalize2d(ns)

    # ---- HIP path of the next-state tower (mzs_resnet_tower, muax_amd/csrc/mz_conv.cuh) ----
    use_hip_tower = True
    use_pair_tower = True

    def _hip_tower_ok(self, s, inference=False):
        return (self.use_hip_tower and s.is_cuda and s.dtype == torch.float32 and tuple(s.shape[1:]) == (6, 6, 64)
                and self.ns_stem.out_channels == 64 and self.ns_stem.w is not None
                and (inference or not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))))

    def _packed(self):
        ps = [self.ns_stem.w] + [p for b in self.ns_blocks for p in b.parameters()]
        sig = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in ps)
        if getattr(self, "_pack_sig", None) != sig:
            with torch.no_grad():
                # kernel layout of a 3x3 conv: Wp[tap][c][g][co][i] = W[tap][16 c + 4 g + i][co]  (mz_conv.cuh)
                pack = lambda w: w.reshape(9, 4, 4, 4, 64).permute(0, 1, 2, 4, 3)  # noqa: E731
                conv = torch.stack([torch.stack([pack(b.proj_conv.w), pack(b.conv_0.w), pack(b.conv_1.w)])
                                    for b in self.ns_blocks])
                ln = torch.stack([torch.stack([torch.stack([m.scale, m.offset]) for m in (b.proj_ln, b.ln_0, b.ln_1)])
                                  for b in self.ns_blocks])
                self._pack = (self.ns_stem.w.reshape(65, 64).contiguous(), conv.contiguous(), ln.contiguous())
            self._pack_sig = sig
        return self._pack

    def _tower_args(self, B, device, pred=None, support_size=None):
        """mzs_tower_args with the weights (and, with `pred`, the 17 head arrays + the three per-root outputs) filled in;
        returns (args, tensors to keep alive, (reward, value, prior_logits) or None, device ordinal)."""
        import ctypes as C

        from . import _lib
        stem, conv, ln = self._packed()
        args = _lib.MzsTowerArgs()
        args.struct_size = C.sizeof(_lib.MzsTowerArgs)
        dev_index = device.index if device.index is not None else torch.cuda.current_device()
        args.device = dev_index
        args.batch, args.blocks, args.normalize, args.num_actions = B, len(self.ns_blocks), 1, self.num_actions
        args.stem_w, args.conv_w, args.ln = stem.data_ptr(), conv.data_ptr(), ln.data_ptr()
        keep, outs = [stem, conv, ln], None
        if pred is not None:
            rf, vf, pf = self.r_func, pred.v_func, pred.pi_func
            heads = [rf[0].w, rf[2].w, rf[5].w, rf[5].b, rf[7].w, rf[7].b,
                     vf[0].w, vf[2].w, vf[5].w, vf[5].b, vf[7].w, vf[7].b,
                     pf[0].w, pf[3].w, pf[3].b, pf[5].w, pf[5].b]
            hk = [h.detach().contiguous() for h in heads]
            keep += hk
            for name, t in zip(_lib.MzsTowerArgs.HEAD_FIELDS, hk):
                setattr(args, name, t.data_ptr())
            outs = (torch.empty(B, device=device), torch.empty(B, device=device),
                    torch.empty(B, pred.num_actions, device=device))
            args.reward, args.value, args.prior_logits = (o.data_ptr() for o in outs)
            args.support_size = support_size
        return args, keep, outs, dev_index

    def _pair_scratch_for(self, L, args, B, device, dev_index, stream):
        """Attach the pair-mode scratch of (device, batch, stream) to `args` when pair mode applies (<= 128 roots: two
        workgroups per root so that the launch covers the chip, mz_conv.cuh); returns (key, first use) or (None, False)."""
        if self.use_pair_tower and os.environ.get("MZS_TOWER_PAIR", "1") != "0":
            nbytes = L.mzs_tower_pair_scratch_bytes(B)
            if nbytes:
                key = (dev_index, B, stream)
                first = key not in self._pair_scratch
                if first:
                    self._pair_scratch[key] = torch.zeros(nbytes // 4, dtype=torch.int32, device=device)
                args.pair_scratch, args.pair_scratch_bytes = self._pair_scratch[key].data_ptr(), nbytes
                return key, first
        return None, False

    def _tower_hip(self, s, a, pred=None, support_size=None):
        import ctypes as C

        from . import _lib
        L = _lib.load()
        x = s.contiguous()
        act = a.to(torch.int32).contiguous()
        y = torch.empty_like(x)
        args, keep, outs, dev_index = self._tower_args(x.shape[0], x.device, pred, support_size)
        args.x, args.action, args.y = x.data_ptr(), act.data_ptr(), y.data_ptr()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        key, first = self._pair_scratch_for(L, args, x.shape[0], x.device, dev_index, stream)
        with torch.cuda.device(x.device):
            _lib.check(L.mzs_resnet_tower(C.byref(args), C.c_void_p(stream)))
            if args.pair_scratch and first and not torch.cuda.is_current_stream_capturing():
                # first launch on this scratch: make sure the two halves of every root found each other (they
                # meet in one XCD's L2; a part or driver that places workgroups differently reports it here)
                if self.pair_status():
                    type(self).use_pair_tower = False
                    del self._pair_scratch[key]
                    args.pair_scratch, args.pair_scratch_bytes = None, 0
                    _lib.check(L.mzs_resnet_tower(C.byref(args), C.c_void_p(stream)))
        return y if outs is None else (outs[0], outs[1], outs[2], y)

    use_hip_search = True  # the whole simulation loop as one launch (mzs_resnet_search); MZS_RESNET_SEARCH=0 turns it off

    def hip_search_ok(self, pred, embedding_shape, support_size) -> bool:
        """Can MuZeroSearch run its simulation loop as ONE launch with these nets (mzs_resnet_search)?"""
        if not (self.use_hip_search and os.environ.get("MZS_RESNET_SEARCH", "1") != "0"):
            return False
        if tuple(embedding_shape) != (6, 6, 64) or self.ns_stem.w is None or not self.ns_stem.w.is_cuda:
            return False
        probe = torch.empty((1, 6, 6, 64), device=self.ns_stem.w.device)
        return self._heads_ok(pred, probe, support_size)

    def hip_search(self, pred, handle, support_size: int, discount: float, sim_begin: int, sim_end: int):
        """Simulations [sim_begin, sim_end) of the search on `handle` (a rooted MuZeroSearch whose simulate() of
        `sim_begin` has run) in ONE launch: recurrent_fn of muax/model.py:265-282 + mctx's expand / backward / next
        simulate per root, back to back on the workgroup(s) that own the root (muax_amd/csrc/mz_search_conv.hip).
        Pair mode (<= 128 roots): check pair_lost() afterwards, as for hip_recurrent."""
        import ctypes as C

        from . import _lib
        L = _lib.load()
        B, dev = handle.batch, handle.device
        args, keep, outs, dev_index = self._tower_args(B, dev, pred, support_size)
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._pair_scratch_for(L, args, B, dev, dev_index, stream)
        with torch.cuda.device(dev):
            _lib.check(L.mzs_resnet_search(handle._h, C.byref(args), C.c_float(discount), sim_begin, sim_end,
                                           C.c_void_p(stream)), handle._h)
        self._search_keep = (keep, outs)  # alive until the stream has consumed them

    def pair_status(self):
        """Roots whose two workgroups lost each other in any pair-mode launch so far (must be 0).  Reads device
        memory: synchronises."""
        return sum(int((t.view(-1)[-4 * k[1]:].view(-1, 4)[:, 3] != 0).sum()) for k, t in self._pair_scratch.items())

    def pair_lost(self) -> bool:
        """True if any pair-mode launch since the scratch was made lost a rendezvous (bounded spin ran out, or
        the halves of a root were placed on different XCDs): outputs of that launch are invalid.  MuZero checks
        this after EVERY search that went through pair mode and, when set, drops pair mode for good and repeats
        the search with one workgroup per root -- which gives the same bits as an undisturbed pair-mode run."""
        return bool(self._pair_scratch) and self.pair_status() != 0

    def disable_pair_mode(self):
        type(self).use_pair_tower = False
        self.__dict__.pop("use_pair_tower", None)  # an instance attribute (tests set one) would shadow the class's
        self._pair_scratch.clear()

    def _heads_ok(self, pred, s, support_size: int) -> bool:
        """The nets are the shapes the one-launch recurrent kernel is built for (muax/nn.py:313-378 at 64 channels)."""
        def head_ok(seq, convs, hidden, in_ch):
            ws = [m for m in seq if isinstance(m, HkConv2D)]
            ls = [m for m in seq if isinstance(m, LazyHkLinear)]
            return (len(ws) == len(convs) and len(ls) == 2 and all(m.w is not None for m in ws + ls)
                    and [tuple(m.w.shape[2:]) for m in ws] == convs and ls[0].w.shape[1] == hidden
                    and all(m.with_bias for m in ls))
        return (isinstance(pred, ResNetPrediction) and self._hip_tower_ok(s, inference=True) and 2 * support_size + 1 <= 64
                and pred.num_actions <= 64 and pred.num_actions == self.num_actions
                and head_ok(self.r_func, [(65, 64), (64, 64)], 64, 65)
                and head_ok(pred.v_func, [(64, 16), (16, 16)], 16, 64)
                and head_ok(pred.pi_func, [(64, 16)], 16, 64)
                and self.r_func[7].w.shape[1] == 2 * support_size + 1
                and pred.v_func[7].w.shape[1] == 2 * support_size + 1)

    def hip_recurrent(self, pred, s, a, support_size: int):
        """The whole recurrent_fn of muax/model.py:265-282 for the ResNet nets in ONE HIP launch: reward head,
        next-state tower, prediction heads on the next state, both support decodes.  Returns
        (reward [B], value [B], prior_logits [B, A], next_state [B, 6, 6, 64]) or None when the nets are not
        the shapes the kernel is built for (the caller then runs the torch modules)."""
        if not self._heads_ok(pred, s, support_size):
            return None
        return self._tower_hip(s, a, pred, support_size)
