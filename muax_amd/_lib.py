"""ctypes binding of include/mzsearch.h (the C-ABI of the HIP search library).

The product path has NO CPU fallback: if libmzsearch.so is missing, or there is
no gfx950 device, the calls below raise.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

_vp = C.c_void_p  # device pointers travel as integers

MZS_OK, MZS_E_INVALID, MZS_E_UNSUPPORTED, MZS_E_RUNTIME, MZS_E_NODEVICE = 0, -1, -2, -3, -4


class MzsConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32),
                ("num_actions", C.c_int32), ("num_simulations", C.c_int32), ("embed_dim", C.c_int32),
                ("max_depth", C.c_int32), ("qtransform", C.c_int32), ("tiebreak", C.c_int32),
                ("policy", C.c_int32), ("pb_c_init", C.c_float), ("pb_c_base", C.c_float),
                ("global_batch", C.c_int64), ("root_offset", C.c_int64),
                ("max_num_considered_actions", C.c_int32), ("gumbel_scale", C.c_float)]


MLP_WEIGHT_NAMES = ["repr_w", "repr_b", "pv_w1", "pv_b1", "pv_w2", "pv_b2", "pp_w1", "pp_b1", "pp_w2",
                    "pp_b2", "dr_w1", "dr_b1", "dr_w2", "dr_b2", "dn_w1", "dn_b1", "dn_w2", "dn_b2"]


class MzsMlpWeights(C.Structure):
    _fields_ = ([("struct_size", C.c_int32), ("obs_dim", C.c_int32), ("support_size", C.c_int32),
                 ("recurrent_pred_on", C.c_int32), ("discount", C.c_float), ("reserved0", C.c_float)]
                + [(n, _vp) for n in MLP_WEIGHT_NAMES])


TREE_FIELDS = ["node_visits", "raw_values", "node_values", "parents", "action_from_parent",
               "children_index", "children_prior_logits", "children_values", "children_visits",
               "children_rewards", "children_discounts", "embeddings"]
TREE_INT_FIELDS = {"node_visits", "parents", "action_from_parent", "children_index", "children_visits"}


class MzsTreeView(C.Structure):
    _fields_ = [(n, _vp) for n in TREE_FIELDS]


class MzsActArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("reserved0", C.c_int32), ("obs", _vp),
                ("dirichlet_noise", _vp), ("invalid_actions", _vp), ("gumbel", _vp),
                ("key", C.c_uint32 * 2), ("dirichlet_fraction", C.c_float), ("temperature", C.c_float),
                ("action", _vp), ("action_weights", _vp), ("root_value", _vp), ("search_value", _vp),
                ("depth_sum", _vp), ("tree", C.POINTER(MzsTreeView))]


class MzsActHostArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("draw_dirichlet", C.c_int32), ("obs", _vp), ("dirichlet_noise", _vp),
                ("invalid_actions", _vp), ("key", C.c_uint32 * 2), ("dirichlet_fraction", C.c_float),
                ("dirichlet_alpha", C.c_float), ("temperature", C.c_float), ("reserved0", C.c_float),
                ("action", _vp), ("action_weights", _vp), ("root_value", _vp)]


class MzsTrainArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32),
                ("unroll_steps", C.c_int32), ("num_actions", C.c_int32), ("embed_dim", C.c_int32),
                ("obs", _vp), ("actions", _vp), ("rewards", _vp), ("returns", _vp), ("policy", _vp),
                ("loss_scale", C.c_float), ("l2_coeff", C.c_float), ("loss", _vp), ("grads", _vp),
                ("workspace", _vp), ("workspace_bytes", C.c_int64)]


class MzsTowerArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("blocks", C.c_int32),
                ("normalize", C.c_int32), ("num_actions", C.c_int32), ("x", _vp), ("action", _vp),
                ("stem_w", _vp), ("conv_w", _vp), ("ln", _vp), ("y", _vp)]
    HEAD_FIELDS = ["r_c1", "r_c2", "r_l1", "r_b1", "r_l2", "r_b2", "v_c1", "v_c2", "v_l1", "v_b1", "v_l2", "v_b2",
                   "p_c1", "p_l1", "p_b1", "p_l2", "p_b2"]
    _fields_ += [(n, _vp) for n in HEAD_FIELDS] + [("reward", _vp), ("value", _vp), ("prior_logits", _vp),
                                                   ("support_size", C.c_int32), ("reserved0", C.c_int32),
                                                   ("pair_scratch", _vp), ("pair_scratch_bytes", C.c_int64)]


class MzsLayerNormArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("n", C.c_int32),
                ("channels", C.c_int32), ("relu", C.c_int32), ("eps", C.c_float),
                ("x", _vp), ("scale", _vp), ("offset", _vp), ("x2", _vp), ("scale2", _vp), ("offset2", _vp),
                ("residual", _vp), ("y", _vp), ("workspace", _vp), ("workspace_bytes", C.c_int64)]


class MzsEzHead(C.Structure):
    FIELDS = ["ln_in", "c1", "ln_mid", "fc", "ln_vec", "out_w", "out_b"]
    _fields_ = [(n, _vp) for n in FIELDS]


class MzsEzArgs(C.Structure):
    WEIGHTS = ["d_ln_in", "d_conv", "d_ln0", "d_conv0", "d_ln1", "d_conv1", "p_ln0", "p_conv0", "p_ln1", "p_conv1"]
    _fields_ = ([("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("channels", C.c_int32),
                 ("num_actions", C.c_int32), ("support_size", C.c_int32), ("x", _vp), ("action", _vp), ("y", _vp),
                 ("reward", _vp), ("value", _vp), ("prior_logits", _vp)] + [(n, _vp) for n in WEIGHTS]
                + [("r", MzsEzHead), ("v", MzsEzHead), ("p", MzsEzHead)])


class MzsConv3x3Args(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("channels", C.c_int32), ("relu", C.c_int32), ("reserved0", C.c_int32),
                ("x", _vp), ("w_packed", _vp), ("y", _vp)]


class MzsRootTailArgs(C.Structure):
    HEAD_FIELDS = ["v_c1", "v_c2", "v_l1", "v_b1", "v_l2", "v_b2", "p_c1", "p_l1", "p_b1", "p_l2", "p_b2"]
    _fields_ = ([("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32),
                 ("width", C.c_int32), ("num_actions", C.c_int32), ("support_size", C.c_int32), ("normalize", C.c_int32),
                 ("x", _vp)] + [(n, _vp) for n in HEAD_FIELDS] + [("embedding", _vp), ("value", _vp), ("prior_logits", _vp)])


class MzsConv3x3sArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("relu", C.c_int32),
                ("in_div", C.c_float), ("reserved0", C.c_int32), ("x", _vp), ("w_packed", _vp), ("y", _vp)]


class MzsResblockArgs(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("channels", C.c_int32), ("eps", C.c_float), ("reserved0", C.c_int32),
                ("x", _vp), ("w_proj", _vp), ("w0", _vp), ("w1", _vp), ("proj_scale", _vp), ("proj_offset", _vp),
                ("ln0_scale", _vp), ("ln0_offset", _vp), ("ln1_scale", _vp), ("ln1_offset", _vp), ("y", _vp),
                ("workspace", _vp), ("workspace_bytes", C.c_int64)]


EXPORTED_SYMBOLS = ["mzs_abi_version", "mzs_last_error", "mzs_create", "mzs_destroy",
                    "mzs_mlp_set_weights", "mzs_act_mlp", "mzs_root", "mzs_root_gumbel", "mzs_select",
                    "mzs_expand_backup", "mzs_expand_backup_select",
                    "mzs_finish", "mzs_tree_export", "mzs_mlp_loss_grad", "mzs_mlp_num_params",
                    "mzs_mlp_train_workspace_bytes", "mzs_resnet_tower", "mzs_tower_pair_scratch_bytes",
                    "mzs_dirichlet", "mzs_act_mlp_host", "mzs_selftest", "mzs_layernorm_act",
                    "mzs_layernorm_workspace_bytes", "mzs_ez_recurrent", "mzs_resnet_search",
                    "mzs_register_fused_dispatch", "mzs_register_fused_dispatch_muzero", "mzs_fused_jit_abi", "mzs_mlp_allow_generic", "mzs_conv3x3_nhwc",
                    "mzs_resblock_v1", "mzs_resblock_workspace_bytes", "mzs_conv3x3_stride2_nhwc", "mzs_resnet_root_tail",
                    "mzs_resblock_v2", "mzs_resblock_v2_workspace_bytes", "mzs_register_train_dispatch", "mzs_train_jit_abi"]

_lib = None


def load(build_if_missing: bool = True):
    """Load muax_amd/lib/libmzsearch.so; raises RuntimeError if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    # MUAX_AMD_LIB: tools only (A/B builds of the same ABI under tools/bin); the product always loads the in-tree library
    path = os.environ.get("MUAX_AMD_LIB") or _build.LIB_PATH
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError(f"{path} is missing: build it with `python -m muax_amd._build`")
        _build.build()
    L = C.CDLL(path)
    L.mzs_abi_version.restype = C.c_int
    L.mzs_last_error.restype = C.c_char_p
    L.mzs_last_error.argtypes = [_vp]
    L.mzs_create.argtypes = [C.POINTER(MzsConfig), C.POINTER(_vp)]
    L.mzs_destroy.argtypes = [_vp]
    L.mzs_mlp_set_weights.argtypes = [_vp, C.POINTER(MzsMlpWeights)]
    L.mzs_act_mlp.argtypes = [_vp, C.POINTER(MzsActArgs), _vp]
    L.mzs_act_mlp_host.argtypes = [_vp, C.POINTER(MzsActHostArgs), _vp]
    L.mzs_selftest.argtypes = [C.c_int32, C.POINTER(C.c_int64 * 4)]
    L.mzs_root.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.POINTER(C.c_uint32 * 2), _vp]
    L.mzs_root_gumbel.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint32 * 2), _vp]
    L.mzs_select.argtypes = [_vp, C.c_int32, _vp, _vp, _vp]
    L.mzs_expand_backup.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]
    L.mzs_expand_backup_select.argtypes = [_vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.mzs_finish.argtypes = [_vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]
    L.mzs_tree_export.argtypes = [_vp, C.POINTER(MzsTreeView), _vp]
    L.mzs_mlp_loss_grad.argtypes = [C.POINTER(MzsMlpWeights), C.POINTER(MzsTrainArgs), _vp]
    L.mzs_resnet_tower.argtypes = [C.POINTER(MzsTowerArgs), _vp]
    L.mzs_register_fused_dispatch.argtypes = [_vp, C.c_int32]
    L.mzs_register_fused_dispatch_muzero.argtypes = [_vp, C.c_int32]
    L.mzs_register_train_dispatch.argtypes = [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.mzs_mlp_allow_generic.argtypes = [_vp, C.c_int32]
    L.mzs_conv3x3_nhwc.argtypes = [C.POINTER(MzsConv3x3Args), _vp]
    L.mzs_resblock_v1.argtypes = [C.POINTER(MzsResblockArgs), _vp]
    L.mzs_conv3x3_stride2_nhwc.argtypes = [C.POINTER(MzsConv3x3sArgs), _vp]
    L.mzs_resnet_root_tail.argtypes = [C.POINTER(MzsRootTailArgs), _vp]
    L.mzs_resblock_workspace_bytes.argtypes = [C.c_int32] * 4
    L.mzs_resblock_v2.argtypes = [C.POINTER(MzsResblockArgs), _vp]
    L.mzs_resblock_v2_workspace_bytes.argtypes = [C.c_int32] * 4
    L.mzs_resnet_search.argtypes = [_vp, C.POINTER(MzsTowerArgs), C.c_float, C.c_int32, C.c_int32, _vp]
    L.mzs_mlp_num_params.argtypes = [C.c_int32] * 4
    L.mzs_mlp_train_workspace_bytes.argtypes = [C.c_int32] * 5
    for n in EXPORTED_SYMBOLS[2:]:
        getattr(L, n).restype = C.c_int
    L.mzs_mlp_num_params.restype = C.c_int64
    L.mzs_mlp_train_workspace_bytes.restype = C.c_int64
    L.mzs_dirichlet.argtypes = [C.c_int32, C.POINTER(C.c_uint32 * 2), C.c_float, C.c_int32, C.c_int32, C.c_int64,
                                C.c_int64, _vp, _vp]
    L.mzs_ez_recurrent.argtypes = [C.POINTER(MzsEzArgs), _vp]
    L.mzs_layernorm_act.argtypes = [C.POINTER(MzsLayerNormArgs), _vp]
    L.mzs_layernorm_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    L.mzs_layernorm_workspace_bytes.restype = C.c_int64
    L.mzs_resblock_workspace_bytes.restype = C.c_int64
    L.mzs_resblock_v2_workspace_bytes.restype = C.c_int64
    L.mzs_tower_pair_scratch_bytes.argtypes = [C.c_int32]
    L.mzs_tower_pair_scratch_bytes.restype = C.c_int64
    if L.mzs_abi_version() != 1:
        raise RuntimeError("libmzsearch.so ABI version mismatch")
    _lib = L
    return L


def check(code: int, handle=None):
    """Map C-ABI status codes onto Python exceptions (ValueError for bad arguments, as the
    reference raises for bad shapes/arguments; RuntimeError for device failures)."""
    if code == MZS_OK:
        return
    msg = load().mzs_last_error(handle)
    msg = msg.decode() if msg else f"mzsearch error {code}"
    if code in (MZS_E_INVALID, MZS_E_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)
