"""Host-side mirror of src/logpsi.py: the factories keep the reference's names and signatures,
the closures take / return numpy arrays of the reference's shapes; the arithmetic is
libcoulombgas_hip.so (cg_logpsi, cg_logphi_logjacdet, cg_logp, cg_grad_laplacian, cg_quantum_score).

The reference's closures are un-batched and get vmapped (src/logpsi.py:58,63,176,185); here every
closure accepts x of shape (..., n, dim) with matching state_idx (..., n)."""
import numpy as np
from . import _lib


class _WaveFn:
    """What the reference's closures capture: (flow, sp_indices, L)."""

    def __init__(self, flow, sp_indices, L):
        self.flow, self.sp_indices, self.L = flow, np.ascontiguousarray(sp_indices, dtype=np.float64), float(L)

    def engine(self, x, params):
        n, dim = np.shape(x)[-2:]
        eng = self.flow.engine(n, dim, self.sp_indices)
        eng.set_params(self.flow.ravel(params, dim))
        return eng


def make_logpsi(flow, sp_indices, L):
    """src/logpsi.py:7-33"""
    wf = _WaveFn(flow, sp_indices, L)

    def logpsi(x, params, state_idx):
        return wf.engine(x, params).logpsi(x, state_idx)

    logpsi.wf = wf
    return logpsi


def make_logphi_logjacdet(flow, sp_indices, L):
    """src/logpsi.py:35-53"""
    wf = _WaveFn(flow, sp_indices, L)

    def logphi(x, params, state_idx):
        return wf.engine(x, params).logphi_logjacdet(x, state_idx)[0]

    def logjacdet(x, params):
        n = np.shape(x)[-2]
        lead = np.shape(x)[:-2]
        sidx = np.broadcast_to(np.arange(n, dtype=np.int32), lead + (n,))
        return wf.engine(x, params).logphi_logjacdet(x, sidx)[1]

    logphi.wf = logjacdet.wf = wf
    return logphi, logjacdet


def make_logp(logpsi):
    """src/logpsi.py:174-181.  The returned closure has `.bind(params, state_indices)` giving the
    object `mcmc` needs to run the whole chain on the GPU."""
    wf = logpsi.wf

    def logp(x, params, state_idx):
        return wf.engine(x, params).logp(x, state_idx)

    class BoundLogp:
        def __init__(self, params, state_indices):
            self.wf, self.params = wf, params
            self.state_indices = state_indices if hasattr(state_indices, "ptr") else np.ascontiguousarray(state_indices, dtype=np.int32)

        def __call__(self, x):
            return logp(x, self.params, self.state_indices)

    logp.wf = wf
    logp.bind = BoundLogp
    return logp


def _draw_v(key, shape):
    """Hutchinson probe (src/logpsi.py:110).  `key`: ndarray of x.shape (explicit probe, parity
    mode), numpy Generator, or an int / SeedSequence seed."""
    if isinstance(key, np.ndarray) and key.shape == tuple(shape):
        return np.ascontiguousarray(key, dtype=np.float64)
    rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
    return rng.standard_normal(shape)


def _is_device(a):
    return not isinstance(a, np.ndarray) and hasattr(a, "ptr")


def make_logpsi_grad_laplacian(logpsi, forloop=True, hutchinson=False, logphi=None, logjacdet=None):
    """src/logpsi.py:55-172.  `forloop` selects between two algebraically identical reference
    variants (:86-100) and has no effect here.  With DeviceArray arguments the results are DeviceArrays too and the
    probe is drawn on the device (cg_randn) unless `key` is an explicit probe array."""
    wf = logpsi.wf

    def logpsi_vmapped(x, params, state_idx):
        eng = wf.engine(x, params)
        if _is_device(x):
            return eng.logpsi_d(x, state_idx if _is_device(state_idx) else eng.asdevice(state_idx, "sidx", np.int32))
        out = eng.logpsi(x, state_idx)
        return out[..., 0] + 1j * out[..., 1]

    if not hutchinson:
        mode = _lib.CG_LAP_EXACT
    elif logphi is None and logjacdet is None:
        mode = _lib.CG_LAP_HUTCHINSON
    else:
        mode = _lib.CG_LAP_HUTCHINSON_SPLIT

    def logpsi_grad_laplacian(x, params, state_indices, key, with_scores=False):
        """with_scores (device arrays only; not in the reference's signature): also leave the per-sample scores d log Psi / d theta of
        the same walkers resident -- make_loss asks for it, because the jacrev of main.py:278 follows on the same x"""
        eng = wf.engine(x, params)
        if _is_device(x):
            v_d = None
            if hutchinson:
                if isinstance(key, np.ndarray) and key.shape == tuple(x.shape):
                    v_d = eng.asdevice(key, "probe")
                else:
                    from .mcmc import _seed_of
                    v_d = eng.randn_d("probe", x.shape, _seed_of(key))
            s_d = state_indices if _is_device(state_indices) else eng.asdevice(state_indices, "sidx", np.int32)
            return eng.grad_laplacian_d(x, s_d, mode, v_d, with_scores=with_scores)
        v = _draw_v(key, np.shape(x)) if hutchinson else None
        return eng.grad_laplacian(x, state_indices, mode, v)

    logpsi_vmapped.wf = logpsi_grad_laplacian.wf = wf
    logpsi_grad_laplacian.mode = mode
    logpsi_grad_laplacian.takes_with_scores = True
    return logpsi_vmapped, logpsi_grad_laplacian


def make_quantum_score(logpsi):
    """src/logpsi.py:183-203: per-sample d log Psi / d theta, complex, as a params-shaped pytree with a
    leading batch axis."""
    wf = logpsi.wf

    def quantum_score_fn(x, params, state_idx):
        eng = wf.engine(x, params)
        sc = eng.quantum_score(x, state_idx)              # (..., P) complex
        dim = np.shape(x)[-1]
        from .flow import ravel_order
        out, off = {}, 0
        for name, leaf, shp in ravel_order(wf.flow.depth, wf.flow.spsize, wf.flow.tpsize, dim):
            sz = int(np.prod(shp))
            out.setdefault(name, {})[leaf] = sc[..., off:off + sz].reshape(sc.shape[:-1] + shp)
            off += sz
        return out

    quantum_score_fn.wf = wf
    return quantum_score_fn
