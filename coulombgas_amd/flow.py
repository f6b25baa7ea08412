"""Host-side mirror of the reference's FermiNet (src/flow.py:5-55) and of the hk.transform
interface main.py uses (`flow.init(key, x)`, `flow.apply(params, None, x)`, main.py:152-157).
The arithmetic runs in libcoulombgas_hip.so (cg_flow_forward)."""
import numpy as np
from .engine import Engine

_ENGINES = {}


def get_engine(n, dim, depth, spsize, tpsize, L, sp_indices=None, device=None):
    from . import utils
    device = utils.current_device() if device is None else device
    key = (n, dim, depth, spsize, tpsize, float(L), None if sp_indices is None else np.asarray(sp_indices, dtype=np.float64).tobytes(), device)
    eng = _ENGINES.get(key)
    if eng is None or eng._ctx is None:
        eng = _ENGINES[key] = Engine(n, dim, depth, spsize, tpsize, L, sp_indices, device)
    return eng


def close_engines():
    for e in list(_ENGINES.values()):
        e.close()
    _ENGINES.clear()


def layer_names(depth):
    """Haiku module names (src/flow.py:11-14,54)."""
    nm = lambda i: "fermi_net/~/linear" + ("" if i == 0 else "_%d" % i)
    return [nm(i) for i in range(depth)], [nm(depth + i) for i in range(depth - 1)], "fermi_net/linear"


def param_shapes(depth, spsize, tpsize, dim):
    sp, tp, fin = layer_names(depth)
    shapes = {sp[0]: (4 * dim + 1, spsize)}
    for i in range(1, depth):
        shapes[sp[i]] = (2 * spsize + tpsize, spsize)
    shapes[tp[0]] = (2 * dim + 1, tpsize)
    for i in range(1, depth - 1):
        shapes[tp[i]] = (tpsize, tpsize)
    shapes[fin] = (spsize, dim)
    return shapes


def ravel_order(depth, spsize, tpsize, dim):
    """jax.flatten_util.ravel_pytree order: sorted module names, 'b' before 'w' (main.py:159)."""
    out = []
    shapes = param_shapes(depth, spsize, tpsize, dim)
    for name in sorted(shapes):
        fin, fout = shapes[name]
        out.append((name, "b", (fout,)))
        out.append((name, "w", (fin, fout)))
    return out


class FermiNet:
    def __init__(self, depth, spsize, tpsize, L, init_stddev=0.01):
        if depth < 2:
            raise ValueError("FermiNet needs depth >= 2 (src/flow.py:52 is ill-formed for depth 1)")
        self.depth, self.spsize, self.tpsize, self.L, self.init_stddev = depth, spsize, tpsize, float(L), init_stddev

    # -- parameter pytree -------------------------------------------------------------
    def init(self, rng, x):
        """N(0, init_stddev^2) weights, zero biases (src/flow.py:6-14,54).  rng: numpy Generator or seed
        (jax.random streams are not reproduced)."""
        rng = np.random.default_rng(rng) if not isinstance(rng, np.random.Generator) else rng
        dim = np.asarray(x).shape[-1]
        params = {}
        for name, leaf, shp in ravel_order(self.depth, self.spsize, self.tpsize, dim):
            params.setdefault(name, {})[leaf] = np.zeros(shp) if leaf == "b" else self.init_stddev * rng.standard_normal(shp)
        return params

    def ravel(self, params, dim):
        if isinstance(params, np.ndarray):
            return np.ascontiguousarray(params, dtype=np.float64).ravel()
        return np.concatenate([np.asarray(params[n][l], dtype=np.float64).ravel()
                               for n, l, _ in ravel_order(self.depth, self.spsize, self.tpsize, dim)])

    def unravel(self, theta, dim):
        theta = np.asarray(theta, dtype=np.float64)
        params, off = {}, 0
        for name, leaf, shp in ravel_order(self.depth, self.spsize, self.tpsize, dim):
            sz = int(np.prod(shp))
            params.setdefault(name, {})[leaf] = theta[off:off + sz].reshape(shp).copy()
            off += sz
        if off != theta.size:
            raise ValueError("theta has %d entries, expected %d" % (theta.size, off))
        return params

    # -- forward ------------------------------------------------------------------------
    def engine(self, n, dim, sp_indices=None):
        return get_engine(n, dim, self.depth, self.spsize, self.tpsize, self.L, sp_indices)

    def apply(self, params, rng, x):
        x = np.asarray(x, dtype=np.float64)
        n, dim = x.shape[-2:]
        eng = self.engine(n, dim)
        eng.set_params(self.ravel(params, dim))
        return eng.flow_forward(x)
