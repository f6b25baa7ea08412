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

    def quantum_score2(self, x, sidx):
        """second-generation score code (csrc/cg_score.hpp: the small-n kernel of the GPU library)"""
        xb, lead = self._xb(x)
        B = xb.shape[0]
        s = np.ascontiguousarray(sidx, dtype=np.int32).reshape(B, self.n)
        sc = np.empty((B, self.P, 2))
        assert lib().emu_scores2(*self._args(), _p(s), _p(xb), B, _p(sc)) == 0
        return (sc[..., 0] + 1j * sc[..., 1]).reshape(lead + (self.P,))

    def quantum_fisher(self, x, sidx):
        sc = self.quantum_score(x, sidx).reshape(-1, self.P)
        return (sc.conj().T @ sc).real / sc.shape[0], sc.mean(axis=0)

    # ---- the device-resident API of coulombgas_amd.engine.Engine with numpy arrays as "device" handles (CPU tests of the
    # host logic in vmc.py / driver.py / sr.py; the arithmetic below restates the small kernels k_local_energy,
    # k_abs_dev, k_clip_weights for that purpose -- the GPU tests compare the kernels themselves with the oracle) ----
    def asdevice(self, a, tag, dtype=np.float64):
        return np.array(a, dtype=dtype, copy=True, order="C")

    def to_host(self, a):
        return np.asarray(a)

    def scratch(self, tag, shape, dtype=np.float64, complex_pairs=False):
        pool = self.__dict__.setdefault("_scratch", {})
        key = (tag, tuple(shape), bool(complex_pairs))
        if key not in pool:
            pool[key] = np.zeros(shape, dtype=np.complex128 if complex_pairs else dtype)
        return pool[key]

    def mcmc_d(self, x, s, mc_steps, mc_stddev, seed=0, walker_offset=0, noise=None, unif=None):
        xn, _, nacc = self.mcmc(x, s, mc_steps, mc_stddev, seed, walker_offset, noise, unif)
        x[...] = xn
        return nacc

    def wrap_d(self, x):
        x -= self.L * np.floor(x / self.L)
        return x

    def randn_d(self, tag, shape, seed, offset=0):
        return np.random.default_rng([int(seed) % (2 ** 63), int(offset)]).standard_normal(shape)

    def logpsi_d(self, x, s):
        out = self.logpsi(x, s)
        return out[..., 0] + 1j * out[..., 1]

    def grad_laplacian_d(self, x, s, mode, v=None, with_scores=False):
        if with_scores:                                    # (the device engine's fused call: the scores of the same walkers stay resident)
            self.scores_compute_d(x, s)
        return self.grad_laplacian(x, s, mode, v)

    def ewald_d(self, x):
        return self.ewald(x)

    def local_energy_d(self, grad, lap, V, logp_states, Vconst, beta):
        kin = -lap - (grad ** 2).sum(axis=(-2, -1))
        pot = V + Vconst
        eloc = kin + pot
        lps = np.zeros(V.shape) if logp_states is None else logp_states
        floc = lps / beta + eloc.real
        mom = np.array([kin.real.mean(), (kin.real ** 2).mean(), pot.mean(), (pot ** 2).mean(), eloc.real.mean(), (eloc.real ** 2).mean(),
                        floc.mean(), (floc ** 2).mean(), -lps.mean(), (lps ** 2).mean()])
        return eloc, floc, mom

    def abs_dev_d(self, e, center, tag="tv"):
        return np.array([np.abs(e - center[0][center[1]]).mean()])

    def clip_weights_d(self, e, center, tv, scale, tag="w"):
        c, t = float(center[0][center[1]]), float(tv[0])
        lo, hi = c - 5 * t, c + 5 * t
        if not np.iscomplexobj(e):
            return scale * np.clip(e, lo, hi), None
        re, im = e.real.copy(), e.imag.copy()
        m = (re < lo) | ((re == lo) & (im < 0)); re[m] = lo; im[m] = 0.0
        m = (hi < re) | ((hi == re) & (0 < im)); re[m] = hi; im[m] = 0.0
        return scale * re, scale * im

    def scores_compute_d(self, x, s):
        self._S = self.quantum_score(x, s).reshape(-1, self.P)

    def scores_vjp_d(self, w_re, w_im, out, out_index=0):
        out[out_index:out_index + self.P] = (w_re[:, None] * self._S.real + w_im[:, None] * self._S.imag).sum(axis=0)

    def scores_mean_d(self, out, out_index=0):
        m = self._S.mean(axis=0)
        out[out_index:out_index + 2 * self.P:2] = m.real; out[out_index + 1:out_index + 2 * self.P:2] = m.imag

    def scores_fisher_d(self, out, fisher_index, mean_index):
        out[fisher_index:fisher_index + self.P ** 2] = ((self._S.conj().T @ self._S).real / self._S.shape[0]).ravel()
        self.scores_mean_d(out, mean_index)

    def axpby_d(self, a, x, b, y, count=None, x_index=0, y_index=0):
        xf, yf = x.reshape(-1), y.reshape(-1)
        n = yf.size - y_index if count is None else count
        yf[y_index:y_index + n] = a * xf[x_index:x_index + n] + (b * yf[y_index:y_index + n] if b != 0 else 0.0)
        return y

    def scale_d(self, y, s, count=None, index=0):
        yf = y.reshape(-1)
        n = yf.size - index if count is None else count
        yf[index:index + n] *= s
        return y

    def fisher_real_d(self, score, tag="classical_fisher"):
        S = np.asarray(score, dtype=np.float64)
        return S.T @ S / S.shape[0]

    def view(self, base, index, shape):
        return base.reshape(-1)[index:index + int(np.prod(shape))].reshape(shape)

    def to_host_slice(self, a, start, count):
        return np.array(a.reshape(-1)[start:start + count])

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
