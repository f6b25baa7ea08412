"""Engine: one cg_ctx (one GPU, one (n, dim, flow architecture, orbital table)) + numpy marshalling.

Host-side mirror of what the reference's closures capture (main.py:152-171): the flow
architecture, the orbital table `sp_indices_twist`, the box L and the Ewald constants.
"""
import ctypes as C
import itertools
import numpy as np
from . import _lib
from ._lib import lib, check

_TOKENS = itertools.count(1)      # identity of device arrays / parameter sets in cache keys: never reused (id() is, after a free)


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


class DeviceArray:
    """An array that lives in HBM -- what a jax device array is to the reference's closures.  The closures of this package
    accept either numpy arrays (staged through the device per call) or DeviceArrays (no copies: walkers, local energies,
    scores and Fisher matrices then stay on the GPU for the whole optimisation step).  `np.asarray(a)` downloads.
    complex_pairs: the buffer holds (..., 2) real pairs that read back as complex128 (the C-ABI's complex layout)."""

    def __init__(self, eng, shape, dtype=np.float64, complex_pairs=False):
        self.eng, self.shape, self.dtype = eng, tuple(int(v) for v in shape), np.dtype(dtype)
        self.complex_pairs = bool(complex_pairs)
        self.buf = DeviceBuffer(eng, self.shape + ((2,) if complex_pairs else ()), self.dtype)
        self.version = 0          # bumped by every call that writes the buffer (cache keys)
        self.token = next(_TOKENS)  # which array this is, for cache keys (monotonic: a recycled address is not a match)
        self.base, self.index = self, 0

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape, dtype=np.int64)))
    ptr = property(lambda self: self.buf.ptr)

    def ptr_at(self, index):
        """device pointer to element `index` of the underlying real buffer"""
        return C.c_void_p(self.buf.ptr.value + int(index) * self.dtype.itemsize)

    @staticmethod
    def from_numpy(eng, a, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        d = DeviceArray(eng, a.shape, dtype)
        d.buf.upload(a)
        return d

    def upload(self, a):
        self.buf.upload(np.ascontiguousarray(a, dtype=self.dtype).reshape(self.buf.shape))
        self.version += 1
        return self

    def numpy(self, start=0, count=None):
        """download; start / count (elements of the real buffer) select a flat slice"""
        if start == 0 and count is None:
            a = self.buf.download()
            return (a[..., 0] + 1j * a[..., 1]) if self.complex_pairs else a
        out = np.empty(int(count), dtype=self.dtype)
        check(lib().cg_memcpy_d2h(self.eng._ctx, _p(out), self.ptr_at(start), out.nbytes), self.eng._ctx)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def free(self):
        self.buf.free()


class DeviceView:
    """(P, Q) window into a DeviceArray's buffer (no copy): e.g. the Fisher matrix inside the packed [F | mean score] buffer
    of hybrid_fisher_sr.  Behaves like a DeviceArray for the accumulators (axpby_d / scale_d) and the solver."""

    def __init__(self, base, index, shape):
        self.base, self.index, self.shape = base, int(index), tuple(int(v) for v in shape)
        self.eng, self.dtype, self.complex_pairs = base.eng, base.dtype, False

    size = property(lambda self: int(np.prod(self.shape, dtype=np.int64)))
    ptr = property(lambda self: self.base.ptr_at(self.index))
    version = property(lambda self: self.base.version, lambda self, v: setattr(self.base, "version", v))
    token = property(lambda self: (self.base.token, self.index, self.shape))

    def ptr_at(self, i):
        return self.base.ptr_at(self.index + int(i))

    def numpy(self):
        return self.base.numpy(self.index, self.size).reshape(self.shape)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)


def is_device(a):
    return isinstance(a, DeviceArray)


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
        self._theta_version = 0       # bumped whenever new flow parameters reach the device (cache keys)
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
            self._theta_version += 1

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
        if key is not None and key[2] == self._theta_version and key[0].shape == xb.shape and np.array_equal(key[0], xb) and np.array_equal(key[1], s):
            return True
        self._score_key = self._score_key_d = None      # the resident scores are about to be overwritten: neither key describes them
        check(lib().cg_scores_compute(self._ctx, _p(xb), _p(s), B), self._ctx)
        self._score_key = (xb.copy(), s.copy(), self._theta_version)
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
            self._score_key = self._score_key_d = None      # cg_quantum_fisher overwrites the resident scores: no cached key describes them
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

    # -- device-resident API: DeviceArray in / out, no host copies, asynchronous -----------
    # These are what make_loss / make_update / hybrid_fisher_sr run on: walkers, gradients, local energies, weights, scores
    # and Fisher matrices stay in HBM between the sampling call and the parameter update (src/VMC.py:33-76, main.py:270-307).
    def _dev_call(self, fn, *args):
        """one C-ABI call in device-pointer mode (restores the mode the engine was in)"""
        prev = self._mode
        if prev != _lib.CG_PTR_DEVICE:
            check(lib().cg_set_pointer_mode(self._ctx, _lib.CG_PTR_DEVICE), self._ctx)
        try:
            check(fn(self._ctx, *args), self._ctx)
        finally:
            if prev != _lib.CG_PTR_DEVICE:
                check(lib().cg_set_pointer_mode(self._ctx, prev), self._ctx)

    def scratch(self, tag, shape, dtype=np.float64, complex_pairs=False):
        """persistent named DeviceArray of this engine (re-allocated only when the shape changes)"""
        pool = self.__dict__.setdefault("_scratch", {})
        a = pool.get(tag)
        shape = tuple(int(v) for v in shape)
        if a is None or a.shape != shape or a.dtype != np.dtype(dtype) or a.complex_pairs != bool(complex_pairs):
            if a is not None:
                a.free()
            a = pool[tag] = DeviceArray(self, shape, dtype, complex_pairs)
        return a

    def asdevice(self, a, tag, dtype=np.float64):
        """DeviceArray as is; a numpy array is uploaded into the scratch array `tag` -- skipped when the same values are
        already there (state indices of a zero-temperature sampler, an unchanged walker batch)."""
        if isinstance(a, DeviceArray):
            return a
        a = np.ascontiguousarray(a, dtype=dtype)
        d = self.scratch(tag, a.shape, dtype)
        cache = self.__dict__.setdefault("_upload_cache", {})
        old = cache.get(tag)
        if old is None or old[0] is not d or old[2] != d.version or old[1].shape != a.shape or not np.array_equal(old[1], a):
            d.upload(a)
            cache[tag] = (d, a.copy(), d.version)
        return d

    def to_host(self, a):
        return np.asarray(a)

    def view(self, base, index, shape):
        return DeviceView(base, index, shape)

    def to_host_slice(self, a, start, count):
        return a.numpy(start, count)

    def mcmc_d(self, x_d, sidx_d, mc_steps, mc_stddev, seed=0, walker_offset=0, noise=None, unif=None, count=True):
        """the chain of src/MCMC.py:22-39 in place on x_d; returns the number of accepted moves (8-byte read-back; count=False: None,
        the caller takes the rate from mcmc_accept_rate instead)"""
        B = x_d.shape[0]
        nz = un = None
        if noise is not None:
            nz = self.asdevice(_f64(noise).reshape(mc_steps, B, self.n, self.dim), "mc_noise")
            un = self.asdevice(_f64(unif).reshape(mc_steps, B), "mc_unif")
        self._dev_call(lib().cg_mcmc, x_d.ptr, sidx_d.ptr, int(B), int(mc_steps), float(mc_stddev), int(seed), int(walker_offset),
                       nz.ptr if nz is not None else None, un.ptr if un is not None else None, None, None)
        x_d.version += 1
        return self.mcmc_accepts() if count else None

    def wrap_d(self, x_d):
        self._dev_call(lib().cg_wrap, x_d.ptr, int(x_d.shape[0]))
        x_d.version += 1
        return x_d

    def randn_d(self, tag, shape, seed, offset=0):
        out = self.scratch(tag, shape)
        self._dev_call(lib().cg_randn, out.ptr, out.size, int(seed) & (2 ** 64 - 1), int(offset))
        out.version += 1
        return out

    def logpsi_d(self, x_d, sidx_d):
        out = self.scratch("logpsi", (x_d.shape[0],), complex_pairs=True)
        self._dev_call(lib().cg_logpsi, x_d.ptr, sidx_d.ptr, int(x_d.shape[0]), out.ptr)
        out.version += 1
        return out

    def grad_laplacian_d(self, x_d, sidx_d, mode, v_d=None, with_scores=False):
        """with_scores: the per-sample scores of the same walkers are left resident as well (cg_grad_laplacian_scores: one fused kernel where
        the library has one, else the two calls) -- the scores_compute_d that follows in an optimisation step then finds them there"""
        B = x_d.shape[0]
        g = self.scratch("grad", (B, self.n, self.dim), complex_pairs=True)
        l = self.scratch("lap", (B,), complex_pairs=True)
        key = (x_d.token, x_d.version, sidx_d.token, sidx_d.version, self._theta_version)
        if with_scores and v_d is not None and getattr(self, "_score_key_d", None) != key:
            self._score_key = self._score_key_d = None          # (also if the call below fails half-way)
            self._dev_call(lib().cg_grad_laplacian_scores, x_d.ptr, sidx_d.ptr, int(B), int(mode), v_d.ptr, g.ptr, l.ptr)
            self._score_key_d = key
        else:
            self._dev_call(lib().cg_grad_laplacian, x_d.ptr, sidx_d.ptr, int(B), int(mode), v_d.ptr if v_d is not None else None, g.ptr, l.ptr)
        g.version += 1; l.version += 1
        return g, l

    def ewald_d(self, x_d):
        V = self.scratch("V", (x_d.shape[0],))
        self._dev_call(lib().cg_ewald, x_d.ptr, int(x_d.shape[0]), V.ptr)
        V.version += 1
        return V

    def local_energy_d(self, grad_d, lap_d, V_d, logp_states_d, Vconst, beta):
        """K8: (E_loc (B) complex, F_loc (B), moments (10)) of src/VMC.py:39-58 before the pmean"""
        B = V_d.shape[0]
        eloc = self.scratch("eloc", (B,), complex_pairs=True)
        floc = self.scratch("floc", (B,))
        mom = self.scratch("moments", (10,))
        self._dev_call(lib().cg_local_energy, grad_d.ptr, lap_d.ptr, V_d.ptr, logp_states_d.ptr if logp_states_d is not None else None,
                       int(B), float(Vconst), float(beta), eloc.ptr, floc.ptr, mom.ptr)
        for a in (eloc, floc, mom):
            a.version += 1
        return eloc, floc, mom

    def abs_dev_d(self, e_d, center, tag="tv"):
        """mean_b |e_b - center| with center = (DeviceArray, index) read on the device (src/VMC.py:63,72 before the pmean)"""
        out = self.scratch(tag, (1,))
        self._dev_call(lib().cg_abs_dev, e_d.ptr, int(e_d.shape[0]), 1 if e_d.complex_pairs else 0, center[0].ptr_at(center[1]), out.ptr)
        out.version += 1
        return out

    def clip_weights_d(self, e_d, center, tv_d, scale, tag="w"):
        B = e_d.shape[0]
        w_re = self.scratch(tag + "_re", (B,))
        w_im = self.scratch(tag + "_im", (B,)) if e_d.complex_pairs else None
        self._dev_call(lib().cg_clip_weights, e_d.ptr, int(B), 1 if e_d.complex_pairs else 0, center[0].ptr_at(center[1]), tv_d.ptr,
                       float(scale), w_re.ptr, w_im.ptr if w_im is not None else None)
        w_re.version += 1
        if w_im is not None:
            w_im.version += 1
        return w_re, w_im

    def scores_compute_d(self, x_d, sidx_d):
        """per-sample scores S = d log Psi / d theta resident on the device (src/logpsi.py:183-203); recomputed only when the
        walkers, the state indices or theta changed"""
        key = (x_d.token, x_d.version, sidx_d.token, sidx_d.version, self._theta_version)
        if getattr(self, "_score_key_d", None) != key:
            self._score_key = self._score_key_d = None          # (also if the call below fails half-way)
            self._dev_call(lib().cg_scores_compute, x_d.ptr, sidx_d.ptr, int(x_d.shape[0]))
            self._score_key_d = key

    def scores_vjp_d(self, w_re_d, w_im_d, out, out_index=0):
        self._dev_call(lib().cg_scores_vjp, w_re_d.ptr, w_im_d.ptr, out.ptr_at(out_index))
        out.version += 1

    def scores_mean_d(self, out, out_index=0):
        self._dev_call(lib().cg_scores_mean, out.ptr_at(out_index))
        out.version += 1

    def scores_fisher_d(self, out, fisher_index, mean_index):
        self._dev_call(lib().cg_scores_fisher, out.ptr_at(fisher_index), out.ptr_at(mean_index))
        out.version += 1

    def axpby_d(self, a, x_d, b, y_d, count=None, x_index=0, y_index=0):
        """y = a x + b y on the device (accumulators of main.py:281-289)"""
        n = int(count if count is not None else y_d.size * (2 if y_d.complex_pairs else 1))
        check(lib().cg_axpby(self._ctx, float(a), x_d.ptr_at(x_index), float(b), y_d.ptr_at(y_index), n), self._ctx)
        y_d.version += 1
        return y_d

    def scale_d(self, y_d, s, count=None, index=0):
        n = int(count if count is not None else y_d.size * (2 if y_d.complex_pairs else 1))
        check(lib().cg_scale_dev(self._ctx, y_d.ptr_at(index), n, float(s)), self._ctx)
        y_d.version += 1
        return y_d

    def fisher_real_d(self, score, tag="classical_fisher"):
        """classical Fisher matrix S^T S / B formed on the device and LEFT there (src/sr.py:74)"""
        S = self.asdevice(_f64(score), tag + "_scores")
        B, P = S.shape
        F = self.scratch(tag, (P, P))
        self._dev_call(lib().cg_fisher_real, S.ptr, int(B), int(P), F.ptr)
        F.version += 1
        return F

    def spd_solve_d(self, A_d, b, damping=0.0, center=None, index=0, P=None):
        """(A - Re(conj(m) m^T) + damping I)^-1 b with A (P,P) at A_d[index:] on the device (factored in a device-side
        copy); b, m, x are O(P) host vectors.  src/sr.py:88, 102-112."""
        b = _f64(b).reshape(-1)
        P = int(P if P is not None else b.size)
        vec = self.scratch("solve_vec_%d" % P, (4 * P,))          # (per size: the two solves of a hybrid step alternate, and a
                                                                     # re-allocation is a hipFree -- a device-wide synchronisation)
        host = np.zeros(4 * P)
        host[:P] = b
        if center is not None:
            center = np.asarray(center).reshape(-1)
            host[P:2 * P] = center.real; host[2 * P:3 * P] = center.imag
        vec.upload(host)
        work = self.scratch("solve_matrix_%d" % P, (P * P,))    # factor a copy: the caller's matrix stays intact
        self.axpby_d(1.0, A_d, 0.0, work, count=P * P, x_index=index)
        self._dev_call(lib().cg_spd_solve, work.ptr, P, float(damping),
                       vec.ptr_at(P) if center is not None else None, vec.ptr_at(2 * P) if center is not None else None,
                       vec.ptr_at(0), vec.ptr_at(3 * P))
        return vec.numpy(3 * P, P)

    # -- density-matrix Transformer on the device (cg_van_*) -------------------------------
    def van_set_params(self, cfg, sp_indices, flat, owner=None):
        """cfg = (M, num_layers, model_size, num_heads, hidden_size); flat: parameters in the order of include/coulombgas.h.
        Uploads only when something changed.  owner: whoever set them last (make_autoregressive_sampler uses it to skip the
        flattening when it is handed the same parameter arrays again and nobody else has been here in between)."""
        flat = _f64(flat).ravel(); sp = _f64(sp_indices)
        key = getattr(self, "_van_key", None)
        if key is not None and key[0] == tuple(cfg) and np.array_equal(key[1], flat) and np.array_equal(key[2], sp):
            self._van_key = key[:3] + (owner,)
            return
        M, nl, ms, nh, hs = (int(v) for v in cfg)
        need = lib().cg_van_num_params(M, nl, ms, nh, hs, self.dim)
        if need != flat.size:
            raise ValueError("Transformer parameter count %d, expected %d" % (flat.size, need))
        check(lib().cg_van_set_params(self._ctx, M, nl, ms, nh, hs, _p(sp), _p(flat)), self._ctx)
        self._van_key = (tuple(cfg), flat.copy(), sp.copy(), owner)
        self._van_version = getattr(self, "_van_version", 0) + 1

    def van_sample_d(self, B, seed, offset=0, unif=None):
        """sampler of src/sampler.py:30-38 on the device -> (state_idx (B,n) int32, log p (B)) DeviceArrays"""
        sidx = self.scratch("van_sidx", (B, self.n), np.int32)
        logp = self.scratch("van_logp", (B,))
        u = self.asdevice(unif, "van_unif") if unif is not None else None
        self._dev_call(lib().cg_van_sample, int(B), int(seed) & (2 ** 64 - 1), int(offset), u.ptr if u is not None else None, sidx.ptr, logp.ptr)
        sidx.version += 1; logp.version += 1
        logp.tag = (sidx.token, sidx.version, self._van_version)      # these log-probabilities belong to these samples + parameters
        return sidx, logp

    def van_log_prob_d(self, sidx_d):
        cached = self.__dict__.get("_scratch", {}).get("van_logp")
        if cached is not None and getattr(cached, "tag", None) == (sidx_d.token, sidx_d.version, self._van_version):
            return cached                                                # computed by the sampling pass itself
        logp = self.scratch("van_logp2", (sidx_d.shape[0],))
        self._dev_call(lib().cg_van_log_prob, sidx_d.ptr, int(sidx_d.shape[0]), logp.ptr)
        logp.version += 1
        return logp

    def van_scores_compute_d(self, sidx_d):
        """per-sample classical scores d log p_b / d params (flat order) for these samples; resident in the context until the
        samples or the parameters change (cg_van_scores_compute)"""
        key = (sidx_d.token, sidx_d.version, self._van_version)
        if getattr(self, "_van_scores_key", None) != key:
            self._van_scores_key = None
            self._dev_call(lib().cg_van_scores_compute, sidx_d.ptr, int(sidx_d.shape[0]))
            self._van_scores_key = key
            self._van_scores_B = int(sidx_d.shape[0])
        return self._van_scores_B

    def van_scores_vjp_d(self, w_d, out_d):
        """out (count) = sum_b w[b] S_b"""
        self._dev_call(lib().cg_van_scores_vjp, w_d.ptr, out_d.ptr)
        out_d.version += 1
        return out_d

    def van_scores_fisher_d(self, perm=None):
        """(count, count) S^T S / B, rows / columns in the order perm (ravel index -> flat index) when given"""
        n = lib().cg_van_num_params(*self._van_key[0], self.dim)
        F = self.scratch("van_fisher", (n, n))
        p = self.asdevice(np.ascontiguousarray(perm, dtype=np.int32), "van_perm", np.int32) if perm is not None else None
        self._dev_call(lib().cg_van_scores_fisher, p.ptr if p is not None else None, F.ptr)
        F.version += 1
        return F

    def van_scores_get(self):
        n = lib().cg_van_num_params(*self._van_key[0], self.dim)
        out = self.scratch("van_scores_copy", (self._van_scores_B, n))
        self._dev_call(lib().cg_van_scores_get, out.ptr)
        return self.to_host(out)

    def van_log_prob(self, state_idx):
        s = _i32(state_idx).reshape(-1, self.n)
        out = np.empty(s.shape[0])
        check(lib().cg_van_log_prob(self._ctx, _p(s), s.shape[0], _p(out)), self._ctx)
        return out.reshape(np.shape(state_idx)[:-1])

    def van_sample(self, B, seed, offset=0, unif=None):
        s = np.empty((B, self.n), dtype=np.int32); lp = np.empty(B)
        if unif is not None:
            unif = _f64(unif).reshape(B, self.n, -1)
        check(lib().cg_van_sample(self._ctx, int(B), int(seed) & (2 ** 64 - 1), int(offset), _p(unif), _p(s), _p(lp)), self._ctx)
        return s, lp

    # -- device-pointer API (DeviceBuffer in / out, asynchronous) ---------------------
    def mcmc_dev(self, x_buf, sidx_buf, B, mc_steps, mc_stddev, seed=0, walker_offset=0, logp_buf=None):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_mcmc(self._ctx, x_buf.ptr, sidx_buf.ptr, int(B), int(mc_steps), float(mc_stddev), int(seed),
                            int(walker_offset), None, None, logp_buf.ptr if logp_buf is not None else None, None), self._ctx)

    def mcmc_accepts(self):
        v = C.c_int64(0)
        check(lib().cg_mcmc_accepts(self._ctx, C.byref(v)), self._ctx)
        return int(v.value)

    def mcmc_accept_rate(self, denom, comm_handle=None):
        """accepted moves of the last chain / denom, averaged over the ranks of an RCCL communicator on the device (src/MCMC.py:39)"""
        v = C.c_double(0.0)
        check(lib().cg_mcmc_accept_rate(self._ctx, comm_handle, float(denom), C.byref(v)), self._ctx)
        return float(v.value)

    def logp_dev(self, x_buf, sidx_buf, B, out_buf):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_logp(self._ctx, x_buf.ptr, sidx_buf.ptr, int(B), out_buf.ptr), self._ctx)

    def ewald_dev(self, x_buf, B, out_buf):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_ewald(self._ctx, x_buf.ptr, int(B), out_buf.ptr), self._ctx)

    def wrap_dev(self, x_buf, B):
        assert self._mode == _lib.CG_PTR_DEVICE
        check(lib().cg_wrap(self._ctx, x_buf.ptr, int(B)), self._ctx)
