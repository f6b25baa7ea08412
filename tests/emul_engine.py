"""tests/emul_engine.py -- TEST INFRASTRUCTURE ONLY.
An object with the interface of coulombgas_amd.engine.Engine backed by tests/host_emul/libcg_emul.so (the
device code compiled for the host with a 1-thread workgroup shim).  It lets the GPU-less CPU suite exercise
(a) the kernel arithmetic against the oracle and the golden vectors, (b) the host logic of make_loss /
sample_stateindices_and_x / the comm layer, including world_size-2 gloo runs.  The product never imports it."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        from coulombgas_amd.build import build_emul
        _LIB = C.CDLL(build_emul())
        _LIB.emu_mcmc.restype = C.c_long
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class EmulEngine:
    def __init__(self, n, dim, depth, spsize, tpsize, L, sp_indices=None, device=0):
        assert depth == 2
        self.n, self.dim, self.hs, self.ht, self.L = n, dim, spsize, tpsize, float(L)
        self.sp = np.ascontiguousarray(sp_indices if sp_indices is not None else np.zeros((n, dim)), dtype=np.float64)
        from coulombgas_amd.flow import ravel_order
        self.P = sum(int(np.prod(s)) for _, _, s in ravel_order(2, spsize, tpsize, dim))
        self.theta = None
        self.ew = None
        self.lds_budget = 10080        # doubles; which blocks of cg_lap.hpp's layout count as LDS-resident

    def set_params(self, theta):
        self.theta = np.ascontiguousarray(theta, dtype=np.float64).ravel().copy()
        assert self.theta.size == self.P

    def set_ewald(self, kappa, G, rs):
        self.ew = (float(kappa), np.ascontiguousarray(G, dtype=np.int64), float(rs))

    def _xb(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return x.reshape(-1, self.n, self.dim), x.shape[:-2]

    def _args(self):
        return (self.n, self.dim, self.hs, self.ht, C.c_double(self.L), _p(self.theta), _p(self.sp), self.sp.shape[0])

    def _raw(self, x, sidx, want_zJ=False):
        xb, lead = self._xb(x)
        B = xb.shape[0]
        s = np.ascontiguousarray(sidx, dtype=np.int32).reshape(B, self.n)
        N = self.n * self.dim
        lp, h = np.empty((B, 2)), np.empty(B)
        z = np.empty((B, self.n, self.dim)); J = np.empty((B, N, N)) if want_zJ else None
        assert lib().emu_logpsi(*self._args(), _p(s), _p(xb), B, _p(lp), _p(h), _p(z), _p(J)) == 0
        return lp, h, z, J, lead

    def flow_forward(self, x):
        lp, h, z, J, lead = self._raw(x, np.zeros(np.shape(x)[:-2] + (self.n,), dtype=np.int32) + np.arange(self.n, dtype=np.int32))
        return z.reshape(lead + (self.n, self.dim))

    def flow_jacobian(self, x):
        lp, h, z, J, lead = self._raw(x, np.zeros(np.shape(x)[:-2] + (self.n,), dtype=np.int32) + np.arange(self.n, dtype=np.int32), True)
        return J.reshape(lead + J.shape[1:])

    def logphi_logjacdet(self, x, sidx):
        lp, h, z, J, lead = self._raw(x, sidx)
        return lp.reshape(lead + (2,)), h.reshape(lead)

    def logpsi(self, x, sidx):
        lp, h, z, J, lead = self._raw(x, sidx)
        out = lp.copy(); out[:, 0] += h
        return out.reshape(lead + (2,))

    def logp(self, x, sidx):
        return 2 * self.logpsi(x, sidx)[..., 0]

    def mcmc(self, x, sidx, mc_steps, mc_stddev, seed=0, walker_offset=0, noise=None, unif=None):
        xb, lead = self._xb(x)
        xb = xb.copy(); B = xb.shape[0]
        s = np.ascontiguousarray(sidx, dtype=np.int32).reshape(B, self.n)
        if noise is None:      # the Philox stream lives in the GPU kernel only; emulate with a seeded numpy stream
            rng = np.random.default_rng([int(seed), int(walker_offset)])
            noise = rng.standard_normal((mc_steps, B, self.n, self.dim)); unif = rng.uniform(size=(mc_steps, B))
        noise = np.ascontiguousarray(noise, dtype=np.float64); unif = np.ascontiguousarray(unif, dtype=np.float64)
        logp = np.empty(B)
        nacc = lib().emu_mcmc(*self._args(), _p(s), _p(xb), B, int(mc_steps), C.c_double(mc_stddev), _p(noise), _p(unif), _p(logp))
        return xb.reshape(lead + (self.n, self.dim)), logp.reshape(lead), int(nacc)

    def wrap(self, x):
        x = np.asarray(x, dtype=np.float64)
        return x - self.L * np.floor(x / self.L)

    def ewald(self, x):
        xb, lead = self._xb(x)
        kappa, G, rs = self.ew
        V = np.empty(xb.shape[0])
        assert lib().emu_ewald(self.n, self.dim, C.c_double(self.L), C.c_double(kappa), C.c_double(rs), _p(G), G.shape[0], _p(xb), xb.shape[0], _p(V)) == 0
        return V.reshape(lead)

    def grad_laplacian(self, x, sidx, mode=0, v=None):
        xb, lead = self._xb(x)
        B = xb.shape[0]
        s = np.ascontiguousarray(sidx, dtype=np.int32).reshape(B, self.n)
        if v is not None:
            v = np.ascontiguousarray(v, dtype=np.float64).reshape(B, self.n, self.dim)
        g = np.empty((B, self.n, self.dim, 2)); l = np.empty((B, 2))
        assert lib().emu_grad_laplacian(*self._args(), _p(s), _p(xb), B, int(mode), _p(v), _p(g), _p(l), C.c_long(self.lds_budget)) == 0
        return (g[..., 0] + 1j * g[..., 1]).reshape(lead + (self.n, self.dim)), (l[:, 0] + 1j * l[:, 1]).reshape(lead)

    def param_vjp(self, x, sidx, w_re, w_im):
        xb, _ = self._xb(x)
        B = xb.shape[0]
        s = np.ascontiguousarray(sidx, dtype=np.int32).reshape(B, self.n)
        w_re = np.ascontiguousarray(w_re, dtype=np.float64).reshape(B); w_im = np.ascontiguousarray(w_im, dtype=np.float64).reshape(B)
        g = np.zeros(self.P)
        assert lib().emu_param_vjp(*self._args(), _p(s), _p(xb), B, _p(w_re), _p(w_im), _p(g), None) == 0
        return g

    def quantum_score(self, x, sidx):
        xb, lead = self._xb(x)
        B = xb.shape[0]
        s = np.ascontiguousarray(sidx, dtype=np.int32).reshape(B, self.n)
        sc = np.empty((B, self.P, 2))
        assert lib().emu_param_vjp(*self._args(), _p(s), _p(xb), B, None, None, None, _p(sc)) == 0
        return (sc[..., 0] + 1j * sc[..., 1]).reshape(lead + (self.P,))

    def quantum_fisher(self, x, sidx):
        sc = self.quantum_score(x, sidx).reshape(-1, self.P)
        return (sc.conj().T @ sc).real / sc.shape[0], sc.mean(axis=0)

    def close(self):
        pass


def install(monkeypatch):
    """Route coulombgas_amd's engine factory to the host emulation (CPU tests only)."""
    import coulombgas_amd.flow as fl
    cache = {}

    def get_engine(n, dim, depth, spsize, tpsize, L, sp_indices=None, device=None):
        key = (n, dim, depth, spsize, tpsize, float(L), None if sp_indices is None else np.asarray(sp_indices, dtype=np.float64).tobytes())
        if key not in cache:
            cache[key] = EmulEngine(n, dim, depth, spsize, tpsize, L, sp_indices)
        return cache[key]
    monkeypatch.setattr(fl, "get_engine", get_engine)
