"""Reading the reference's checkpoints without jax -- BEST EFFORT, UNVERIFIED.

muax saves `{'params': MZNetworkParams(...), 'optimizer_state': ...}` with `jnp.save` (muax/model.py:203-212):
an object array in a .npy file, i.e. a pickle of haiku parameter dicts whose leaves are jax Arrays, plus optax
state tuples.  Unpickling that normally imports jax, haiku, optax and muax.  None of them exists here, so this
module reads the .npy header itself and runs the pickle through an Unpickler that
  * rebuilds jax Arrays as NumPy arrays (jax pickles an Array as
    `jax._src.array._reconstruct_array(fun, args, arr_state, aval_state)` around the NumPy reduce of its value),
  * maps `muax.nn.MZNetworkParams` onto ours and turns every other foreign class (optax states, haiku
    mappings, jax avals) into an inert stand-in,
and then assigns the arrays to the default MLP trio in haiku's creation order (hk.Linear: w[in][out], b[out] --
the layout the kernels already use): representation: linear; prediction: v_func (linear, linear_1), pi_func
(linear_2, linear_3); dynamic: ns_func (linear, linear_1), r_func (linear_2, linear_3) (muax/nn.py:59-115).

No reference checkpoint and no jax are available in this build environment: the pickle layout above is the
published one as remembered, exercised only by a synthetic file of that layout (tests/test_host_cpu.py).  Shapes
are checked on assignment; anything unexpected raises ValueError with what was found."""
from __future__ import annotations

import pickle
from typing import Dict, List

import numpy as np

from .nn import MZNetworkParams


class _Stub:
    """Inert stand-in for a class that cannot be imported (optax state, jax aval, ...)."""

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __setstate__(self, state):
        self.state = state

    def __call__(self, *args, **kwargs):
        return _Stub(*args, **kwargs)


def _reconstruct_array(fun, args, arr_state, aval_state=None):
    value = fun(*args)
    value.__setstate__(arr_state)
    return np.asarray(value)


def _flat_mapping(*args, **kwargs):
    return dict(args[0]) if args and isinstance(args[0], dict) else dict(**kwargs)


# the only globals a checkpoint of arrays needs; everything else becomes an inert _Stub (builtins.eval,
# numpy helpers with side effects, ... never resolve)
_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("collections", "OrderedDict"), ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "list"),
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "int"), ("builtins", "float"),
    ("builtins", "bool"), ("builtins", "complex"), ("builtins", "slice"), ("_codecs", "encode"),
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module.split(".")[0] == "numpy" and module.endswith("dtypes") and name.endswith("DType"):
            return super().find_class(module, name)  # numpy >= 1.25 pickles dtypes through numpy.dtypes.*DType
        if name == "_reconstruct_array":
            return _reconstruct_array
        if name == "MZNetworkParams":
            return MZNetworkParams
        if name == "FlatMapping":
            return _flat_mapping
        return type(name, (_Stub,), {"__module__": module})


def read_reference_checkpoint(path: str) -> dict:
    """The saved dict with NumPy leaves: {'params': MZNetworkParams(representation, prediction, dynamic), ...}."""
    if not path.endswith(".npy"):
        path = f"{path}.npy"
    with open(path, "rb") as f:
        version = np.lib.format.read_magic(f)
        shape, _, dtype = (np.lib.format.read_array_header_1_0 if version == (1, 0)
                           else np.lib.format.read_array_header_2_0)(f)
        if dtype != np.dtype(object) or shape != ():
            raise ValueError(f"{path}: not a pickled-object .npy (dtype {dtype}, shape {shape})")
        obj = _Unpickler(f).load()
    obj = obj.item() if isinstance(obj, np.ndarray) else obj
    if not isinstance(obj, dict) or "params" not in obj:
        raise ValueError(f"{path}: expected a dict with 'params', found {type(obj).__name__}")
    return obj


def _linears(tree) -> List[Dict[str, np.ndarray]]:
    """{'module/~/linear': {'w', 'b'}, ...} -> [{'w', 'b'}, ...] in haiku's creation order (linear, linear_1, ...)."""
    if not isinstance(tree, dict):
        raise ValueError(f"expected a haiku parameter dict, found {type(tree).__name__}")

    def order(name):
        tail = name.rsplit("/", 1)[-1]
        return int(tail.rsplit("_", 1)[1]) if "_" in tail and tail.rsplit("_", 1)[1].isdigit() else 0

    out = []
    for name in sorted(tree, key=order):
        leaf = tree[name]
        if not (isinstance(leaf, dict) and "w" in leaf and "b" in leaf):
            raise ValueError(f"module {name!r}: expected {{'w', 'b'}}, found {sorted(leaf) if isinstance(leaf, dict) else leaf}")
        out.append({"w": np.asarray(leaf["w"], np.float32), "b": np.asarray(leaf["b"], np.float32)})
    return out


def load_reference_params(model, path: str) -> None:
    """Copy a reference checkpoint's parameters into `model` (a MuZero on the default MLP trio, init() done)."""
    import torch

    from . import nn as mz_nn
    if not mz_nn.is_default_mlp_trio(model.network):
        raise ValueError("reference checkpoints can be mapped onto the default MLP trio only")
    params = read_reference_checkpoint(path)["params"]
    rep, pred, dyn = (params.representation, params.prediction, params.dynamic) if hasattr(params, "representation") \
        else (params[0], params[1], params[2])
    r, p, d = model.network
    targets = [(r.repr_func, _linears(rep), "representation"),
               (list(p.v_func) + list(p.pi_func), _linears(pred), "prediction"),
               (list(d.ns_func) + list(d.r_func), _linears(dyn), "dynamic")]
    with torch.no_grad():
        for mods, arrays, name in targets:
            mods = mods if isinstance(mods, list) else [mods]
            if len(mods) != len(arrays):
                raise ValueError(f"{name}: {len(arrays)} linear layers in the checkpoint, {len(mods)} in the model")
            for i, (m, a) in enumerate(zip(mods, arrays)):
                if tuple(m.w.shape) != a["w"].shape or tuple(m.b.shape) != a["b"].shape:
                    raise ValueError(f"{name} layer {i}: checkpoint w{a['w'].shape} b{a['b'].shape}, "
                                     f"model w{tuple(m.w.shape)} b{tuple(m.b.shape)}")
                m.w.copy_(torch.from_numpy(a["w"]))
                m.b.copy_(torch.from_numpy(a["b"]))
    model.weights_changed()
