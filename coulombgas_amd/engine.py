"""Engine: one cg_ctx (one GPU, one (n, dim, flow architecture, orbital table)) + numpy marshalling.

Host-side mirror of what the reference's closures capture (main.py:152-171): the flow
architecture, the orbital table `sp_indices_twist`, the box L and the Ewald constants.
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import lib, check


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """fp64/int32 array resident in HBM, owned by an Engine (freed with it or explicitly)."""

    def __init__(self, eng, shape, dtype=np.float64):
        self.eng, self.shape, self.dtype = eng, tuple(int(s) for s in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().cg_dev_alloc(eng._ctx, self.nbytes, C.byref(p)), eng._ctx)
        self.ptr = p
        eng._buffers.append(self)

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        check(lib().cg_memcpy_h2d(self.eng._ctx, self.ptr, _p(a), self.nbytes), self.eng._ctx)
        return self

    def download(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().cg_memcpy_d2h(self.eng._ctx, _p(out), self.ptr, self.nbytes), self.eng._ctx)
        return out

    def free(self):
        if self.ptr is not None and self.eng._ctx is not None:
            lib().cg_dev_free(self.eng._ctx, self.ptr)
            self.ptr = None


class Engine:
    def __init__(self, n, dim, depth, spsize, tpsize, L, sp_indices=None, device=0):
        self.n, self.dim, self.depth, self.spsize, self.tpsize, self.L = int(n), int(dim), int(depth), int(spsize), int(tpsize), float(L)
        if sp_indices is None:           # flow-only use: a dummy table (never touched by cg_flow_forward)
            sp_indices = np.zeros((self.n, self.dim))
        self.sp_indices = _f64(sp_indices)
        assert self.sp_indices.ndim == 2 and self.sp_indices.shape[1] == self.dim
        self._ctx = None
        self._buffers = []
        ctx = C.c_void_p()
        rc = lib().cg_create(C.byref(ctx), int(device), self.n, self.dim, self.depth, self.spsize, self.tpsize,
                             self.L, _p(self.sp_indices), self.sp_indices.shape[0])
        check(rc, None)
        self._ctx = ctx
        self.device = int(device)
        self.P = lib().cg_num_params(ctx)
        self._theta = None
        self._ewald = None
        self._mode = _lib.CG_PTR_HOST

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if self._ctx is not None:
            for b in self._buffers:
                b.free()
            lib().cg_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ------------------------------------------------------------------------
    def set_params(self, theta):
        theta = _f64(theta).ravel()
        if theta.shape[0] != self.P:
            raise ValueError("expected %d flow parameters, got %d" % (self.P, theta.shape[0]))
        if self._theta is None or not np.array_equal(theta, self._theta):
            check(lib().cg_set_flow_params(self._ctx, _p(theta)), self._ctx)
            self._theta = theta.copy()

    def set_ewald(self, kappa, G, rs):
        G = np.ascontiguousarray(G, dtype=np.int64)
        key = (float(kappa), G.tobytes(), float(rs))
        if self._ewald != key:
            check(lib().cg_set_ewald(self._ctx, float(kappa), _p(G), G.shape[0], float(rs)), self._ctx)
            self._ewald = key

    def device_mode(self, on=True):
        self._mode = _lib.CG_PTR_DEVICE if on else _lib.CG_PTR_HOST
        check(lib().cg_set_pointer_mode(self._ctx, self._mode), self._ctx)

    def sync(self):
        check(lib().cg_sync(self._ctx), self._ctx)

    def alloc(self, shape, dtype=np.float64):
        return DeviceBuffer(self, shape, dtype)

    def launch_info(self):
        info = (C.c_int64 * 8)()
        check(lib().cg_get_launch_info(self._ctx, info), self._ctx)
        return {"threads": info[0], "lds_bytes": info[1], "cu_count": info[2], "P": info[3], "fast": info[4]}

    def set_block_threads(self, t):
        check(lib().cg_set_block_threads(self._ctx, int(t)), self._ctx)

    def timer_start(self):
        check(lib().cg_timer_start(self._ctx), self._ctx)

    def timer_stop(self):
        ms = C.c_float()
        check(lib().cg_timer_stop(self._ctx, C.byref(ms)), self._ctx)
        return float(ms.value)

    def microbench_fp64(self, which=0):
        t = C.c_double()
        check(lib().cg_microbench_fp64(self._ctx, int(which), C.byref(t)), self._ctx)
        return float(t.value)

    # -- helpers ----------------------------------------------------------------------
    def _xb(self, x):
        x = _f64(x)
        if x.shape[-2:] != (self.n, self.dim):
            raise ValueError("x must have trailing shape (%d,%d), got %s" % (self.n, self.dim, x.shape))
        return x.reshape(-1, self.n, self.dim), x.shape[:-2]

    def _sb(self, s, B):
        s = _i32(s).reshape(-1, self.n)
        if s.shape[0] != B:
            raise ValueError("state_idx batch %d != x batch %d" % (s.shape[0], B))
        if s.size and (s.min() < 0 or s.max() >= self.sp_indices.shape[0]):
            raise IndexError("state_idx out of range [0,%d)" % self.sp_indices.shape[0])
        return s

    # -- host-pointer API (numpy in, numpy out) ---------------------------------------
    def flow_forward(self, x):
        xb, lead = self._xb(x)
        z = np.empty_like(xb)
        check(lib().cg_flow_forward(self._ctx, _p(xb), xb.shape[0], _p(z)), self._ctx)
        return z.reshape(lead + (self.n, self.dim))

    def flow_jacobian(self, x):
        xb, lead = self._xb(x)
        N = self.n * self.dim
        J = np.empty((xb.shape[0], N, N))
        check(lib().cg_flow_jacobian(self._ctx, _p(xb), xb.shape[0], _p(J)), self._ctx)
        return J.reshape(lead + (N, N))

    def logpsi(self, x, state_idx):
        xb, lead = self._xb(x)
        s = self._sb(state_idx, xb.shape[0])
        out = np.empty((xb.shape[0], 2))
        check(lib().cg_logpsi(self._ctx, _p(xb), _p(s), xb.shape[0], _p(out)), self._ctx)
        return out.reshape(lead + (2,))

    def logphi_logjacdet(self, x, state_idx):
        xb, lead = self._xb(x)
        s = self._sb(state_idx, xb.shape[0])
        lp, h = np.empty((xb.shape[0], 2)), np.empty(xb.shape[0])
        check(lib().cg_logphi_logjacdet(self._ctx, _p(xb), _p(s), xb.shape[0], _p(lp), _p(h)), self._ctx)
        return lp.reshape(lead + (2,)), h.reshape(lead)

    def logp(self, x, state_idx):
        xb, lead = self._xb(x)
        s = self._sb(state_idx, xb.shape[0])
        out = np.empty(xb.shape[0])
        check(lib().cg_logp(self._ctx, _p(xb), _p(s), xb.shape[0], _p(out)), self._ctx)
        return out.reshape(lead)

    def mcmc(self, x, state_idx, mc_steps, mc_stddev, seed=0, walker_offset=0, noise=None, unif=None):
        """-> x_new (B,n,dim), logp (B,), n_accept (int).  noise (steps,B,n,dim), unif (steps,B) or both None."""
        xb, lead = self._xb(x)
        xb = xb.copy()
        B = xb.shape[0]
        s = self._sb(state_idx, B)
        if noise is not None:
            noise = _f64(noise).reshape(mc_steps, B, self.n, self.dim)
            unif = _f64(unif).reshape(mc_steps, B)
        logp = np.empty(B)
        nacc = C.c_int64(0)
        check(lib().cg_mcmc(self._ctx, _p(xb), _p(s), B, int(mc_steps), float(mc_stddev), int(seed), int(walker_offset),
                            _p(noise), _p(unif), _p(logp), C.byref(nacc)), self._ctx)
        return xb.reshape(lead + (self.n, self.dim)), logp.reshape(lead), int(nacc.value)

    def wrap(self, x):
        xb, lead = self._xb(x)
        xb = xb.copy()
        check(lib().cg_wrap(self._ctx, _p(xb), xb.shape[0]), self._ctx)
        return xb.reshape(lead + (self.n, self.dim))

    def ewald(self, x):
        xb, lead = self._xb(x)
        V = np.empty(xb.shape[0])
        check(lib().cg_ewald(self._ctx, _p(xb), xb.shape[0], _p(V)), self._ctx)
        return V.reshape(lead)

    def grad_laplacian(self, x, state_idx, mode=_lib.CG_LAP_EXACT, v=None):
        xb, lead = self._xb(x)
        B = xb.shape[0]
        s = self._sb(state_idx, B)
        if v is not None:
            v = _f64(v).reshape(B, self.n, self.dim)
        g = np.empty((B, self.n, self.dim, 2))
        l = np.empty((B, 2))
        check(lib().cg_grad_laplacian(self._ctx, _p(xb), _p(s), B, int(mode), _p(v), _p(g), _p(l)), self._ctx)
        grad = (g[..., 0] + 1j * g[..., 1]).reshape(lead + (self.n, self.dim))
        lap = (l[:, 0] + 1j * l[:, 1]).reshape(lead)
        return grad, lap

    # Per-sample scores resident on the device (cg_scores_*): one score computation (two reverse sweeps) serves the two
    # theta-VJPs of jax.jacrev(quantum_lossfn) (main.py:278) and the Fisher matrix of the SR optimizer, instead of one
    # sweep (and one set-up) each.  Valid while (x, state_idx, theta) are unchanged.
    SCORE_CACHE_MAX_BYTES = 4 << 30

    def _scores_ready(self, xb, s):
        B = xb.shape[0]
        if B == 0 or B * self.P * 16 > self.SCORE_CACHE_MAX_BYTES:
            return False
        key = getattr(self, "_score_key", None)
        if key is not None and key[2] is self._theta and key[0].shape == xb.shape and np.array_equal(key[0], xb) and np.array_equal(key[1], s):
            return True
        check(lib().cg_scores_compute(self._ctx, _p(xb), _p(s), B), self._ctx)
        self._score_key = (xb.copy(), s.copy(), self._theta)
        return True

    def param_vjp(self, x, state_idx, w_re, w_im, use_scores=True):
        xb, _ = self._xb(x)
        B = xb.shape[0]
        s = self._sb(state_idx, B)
        w_re, w_im = _f64(w_re).reshape(B), _f64(w_im).reshape(B)
        g = np.empty(self.P)
        if use_scores and self._mode == _lib.CG_PTR_HOST and self._scores_ready(xb, s):
            check(lib().cg_scores_vjp(self._ctx, _p(w_re), _p(w_im), _p(g)), self._ctx)
        else:
            check(lib().cg_param_vjp(self._ctx, _p(xb), _p(s), B, _p(w_re), _p(w_im), _p(g)), self._ctx)
        return g

    def quantum_score(self, x, state_idx):
        xb, lead = self._xb(x)
        B = xb.shape[0]
        s = self._sb(state_idx, B)
        sc = np.empty((B, self.P, 2))
        check(lib().cg_quantum_score(self._ctx, _p(xb), _p(s), B, _p(sc)), self._ctx)
        return (sc[..., 0] + 1j * sc[..., 1]).reshape(lead + (self.P,))

    def quantum_fisher(self, x, state_idx, reuse_out=False):
        """fishers_fn of src/sr.py:62-80 for this device: (Re(S^H S)/B (P,P), mean_b S (P,) complex).
        reuse_out: return the engine's own (P,P) buffer, overwritten by the next such call (an optimisation loop consumes
        the matrix at once; a fresh 9 MB array per epoch costs its page faults again every time)."""
        xb, _ = self._xb(x)
        B = xb.shape[0]
        s = self._sb(state_idx, B)
        if reuse_out:
            if getattr(self, "_fisher_out", None) is None:
                self._fisher_out = np.empty((self.P, self.P))
            F = self._fisher_out
        else:
            F = np.empty((self.P, self.P))
        sm = np.empty((self.P, 2))
        if self._mode == _lib.CG_PTR_HOST and self._scores_ready(xb, s):
            check(lib().cg_scores_fisher(self._ctx, _p(F), _p(sm)), self._ctx)
        else:
            self._score_key = None      # cg_quantum_fisher overwrites the resident scores: the cached key no longer describes them
            check(lib().cg_quantum_fisher(self._ctx, _p(xb), _p(s), B, _p(F), _p(sm)), self._ctx)
        return F, sm[:, 0] + 1j * sm[:, 1]

    def fisher_real(self, score):
        """S^T S / B of a real (B, P) score matrix on the device (classical Fisher matrix, src/sr.py:36,74)."""
        S = _f64(score)
        B, P = S.shape
        F = np.empty((P, P))
        check(lib().cg_fisher_real(self._ctx, _p(S), B, P, _p(F)), self._ctx)
        return F

    def cholesky(self, A):
        """Lower Cholesky factor of a symmetric positive definite matrix on the device (blocked, f64 MFMA); returns L
        with zeros above the diagonal."""
        L = np.array(A, dtype=np.float64, order="C")
        check(lib().cg_cholesky(self._ctx, _p(L), L.shape[0]), self._ctx)
        return np.tril(L)

    def spd_solve(self, A, b, damping=0.0, center=None):
        """(A - Re(conj(m) m^T) + damping I)^-1 b, entirely on the device (optional centring with the complex vector
        m = center, shift, blocked Cholesky, both triangular solves): the damped solve of src/sr.py:38-41 / 88, 102-112.
        Raises if the matrix is not positive definite."""
        A = _f64(A); b = _f64(b).reshape(-1)
        assert A.shape == (b.size, b.size)
        x = np.empty(b.size)
        cre = cim = None
        if center is not None:
            center = np.asarray(center).reshape(-1)
            assert center.size == b.size
            cre, cim = _f64(center.real), _f64(center.imag)
        check(lib().cg_spd_solve(self._ctx, _p(A), b.size, float(damping), _p(cre), _p(cim), _p(b), _p(x)), self._ctx)
        return x

    # -- device-pointer API (DeviceBuffer in / out, asynchronous) ---------------------
    def mcmc_dev(self, x_buf, sidx_buf, B, mc_steps, mc_stddev, seed=0, walker_offset=0, logp_buf=None):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_mcmc(self._ctx, x_buf.ptr, sidx_buf.ptr, int(B), int(mc_steps), float(mc_stddev), int(seed),
                            int(walker_offset), None, None, logp_buf.ptr if logp_buf is not None else None, None), self._ctx)

    def mcmc_accepts(self):
        v = C.c_int64(0)
        check(lib().cg_mcmc_accepts(self._ctx, C.byref(v)), self._ctx)
        return int(v.value)

    def logp_dev(self, x_buf, sidx_buf, B, out_buf):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_logp(self._ctx, x_buf.ptr, sidx_buf.ptr, int(B), out_buf.ptr), self._ctx)

    def ewald_dev(self, x_buf, B, out_buf):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_ewald(self._ctx, x_buf.ptr, int(B), out_buf.ptr), self._ctx)

    def wrap_dev(self, x_buf, B):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_wrap(self._ctx, x_buf.ptr, int(B)), self._ctx)
