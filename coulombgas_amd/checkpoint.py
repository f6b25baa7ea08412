"""Mirror of src/checkpoint.py (ckpt_filename, load_data, save_data) that does not need JAX.

The reference pickles pytrees of jax arrays (main.py:374-381: keys, x, params_van, params_flow, opt_state).  Unpickling
those normally imports jax / optax; `load_data` maps the array reconstructor to numpy and every other jax / optax /
haiku class to an inert stub, so the shipped `epoch_*.pkl` files load as nested dicts of numpy arrays.  `save_data`
writes nested dicts of numpy arrays, which the reference's own `load_data` (plain pickle) reads back unchanged."""
import os
import pickle
import numpy as np


def pretrained_model_filename(freefermion_path):
    return os.path.join(freefermion_path, "params_van.pkl")


def ckpt_filename(epoch, path):
    return os.path.join(path, "epoch_%06d.pkl" % epoch)


class _Stub:
    """Placeholder for jax / optax / haiku objects (optimizer state tuples, ...).  Records its constructor arguments in
    __new__ as well: namedtuples are unpickled with NEWOBJ, which never calls __init__ -- the mu / nu / count of an optax
    Adam state would otherwise be dropped."""
    def __new__(cls, *a, **k):
        o = object.__new__(cls)
        o.args, o.kwargs = a, k
        return o

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.state = state


def _reconstruct_array(fun, args, arr_state, aval_state=None):
    """jax._src.array._reconstruct_array: rebuild the numpy array the jax array was pickled from."""
    a = fun(*args)
    a.__setstate__(arr_state)
    return a


class _Unpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        root = mod.split(".")[0]
        if root in ("jax", "jaxlib", "optax", "haiku", "flax", "chex"):
            return _reconstruct_array if name == "_reconstruct_array" else _Stub
        return super().find_class(mod, name)


def _to_numpy(obj):
    if isinstance(obj, dict):
        return {k: _to_numpy(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_numpy(v) for v in obj) if not hasattr(obj, "_fields") else obj
    if isinstance(obj, np.ndarray) or np.isscalar(obj) or obj is None or isinstance(obj, _Stub):
        return obj
    try:
        return np.asarray(obj)
    except Exception:
        return obj


def load_data(filename):
    with open(filename, "rb") as f:
        return _to_numpy(_Unpickler(f).load())


def save_data(data, filename):
    with open(filename, "wb") as f:
        pickle.dump(data, f)


def adam_state_from_ckpt(opt_state):
    """Adam moments of a checkpoint as this package's adam() state {"count", "mu", "nu"}: passes such a dict through, and
    converts an optax state pickled by the reference (a tuple holding ScaleByAdamState(count, mu, nu), loaded as _Stub)."""
    if isinstance(opt_state, dict) and {"count", "mu", "nu"} <= set(opt_state):
        return opt_state
    stack = [opt_state]
    while stack:
        o = stack.pop()
        if isinstance(o, _Stub):
            if len(o.args) == 3:
                return {"count": int(np.asarray(o.args[0])), "mu": _to_numpy(o.args[1]), "nu": _to_numpy(o.args[2])}
            stack.extend(o.args)
        elif isinstance(o, (list, tuple)):
            stack.extend(o)
    return None


def load_log(filename):
    """data.txt of main.py:367-372 as a dict of columns (energies in Ry / rs^2)."""
    a = np.atleast_2d(np.loadtxt(filename))
    names = ["epoch", "F", "F_std", "E", "E_std", "K", "K_std", "V", "V_std", "S", "S_std", "accept_rate"]
    return {k: a[:, i] for i, k in enumerate(names)}
