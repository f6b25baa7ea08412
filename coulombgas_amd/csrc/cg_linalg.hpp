// cg_linalg.hpp -- small dense fp64 linear algebra on one workgroup, operands in LDS.
//
// Replaces the reference's jnp.linalg.slogdet / jnp.linalg.inv (LAPACK getrf/getri) call
// sites: src/logpsi.py:29,50 (real nd x nd Jacobian), src/slater.py:18,38,42 (complex n x n).
// Partial (row) pivoting like getrf; rows are addressed through a permutation vector so no
// row is physically swapped.  The pivot search is done redundantly by every thread
// (broadcast LDS reads) so that no reduction / extra barrier is needed.
#pragma once
#include "cg_common.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ unsigned cg_wave_max_u32(unsigned v);
#endif
// Workgroup-wide argmax of a non-negative key (ties -> smallest index).  `scratch` (>= 3*16 doubles worth, LDS):
// per-wave partial results.  Every thread returns the winning index.
CG_DEVI int cg_block_argmax(const CgBlk& b, double v, int idx, double* scratch) {
#if defined(__HIP_DEVICE_COMPILE__)
    {   // wave level: DPP max on the high word of the key (a near-maximal pivot is as good as the maximal one)
        const unsigned key = v < 0.0 ? 0u : (unsigned)(__double_as_longlong(v) >> 32) + 1u;
        const unsigned mx = cg_wave_max_u32(key);
        const unsigned long long mask = __ballot(key == mx);
        const int src = (int)__builtin_ctzll(mask);
        idx = __builtin_amdgcn_readlane(idx, src);
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)__double_as_longlong(v), src);
        const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(__double_as_longlong(v) >> 32), src);
        v = __longlong_as_double(((long long)hi << 32) | lo);
    }
    const int nw = b.nthr >> 6;
    if (nw == 1) return idx;
    int* si = (int*)(scratch + 16);
    if ((b.tid & 63) == 0) { scratch[b.tid >> 6] = v; si[b.tid >> 6] = idx; }
    b.sync();
    double bv = scratch[0]; int bi = si[0];
    for (int w = 1; w < nw; ++w) {
        const double ov = scratch[w]; const int oi = si[w];
        const bool take = (ov > bv) || (ov == bv && oi < bi);
        bv = take ? ov : bv; bi = take ? oi : bi;
    }
    b.sync();                      // scratch may be rewritten by the next call
    return bi;
#else
    (void)b; (void)v; (void)scratch;
    return idx;                    // host shim: one thread has already scanned every candidate
#endif
}

// log|det A| of a real N x N matrix (row-major, leading dimension lda) held in LDS (or any memory the workgroup
// shares).  A is destroyed.  scratch: >= 40 doubles.  getrf-style partial pivoting with physical row swaps; the
// pivot search is one candidate per thread + a shuffle/LDS argmax (a few hundred cycles per column instead of a
// serial scan).  Returns the value in every thread.
CG_DEVI double cg_lu_logabsdet(const CgBlk& b, double* A, int N, int lda, int* scratch_i, int* sign_out = nullptr, bool ool = false) {
    double* scratch = (double*)scratch_i;
    const int TX = b.nthr < 16 ? b.nthr : 16;
    const int tx = b.tid % TX, ty = b.tid / TX, TY = b.nthr / TX;
    CgScaledProd prod; prod.init();
    int sgn = 1;
    for (int k = 0; k < N; ++k) {
        double best = -1.0; int bi = 0x7fffffff;
        for (int i = k + b.tid; i < N; i += b.nthr) {
            const double v = fabs(A[i * lda + k]);
            if (v > best) { best = v; bi = i; }
        }
        int p = cg_block_argmax(b, best, bi, scratch);
        if (p < k || p >= N) p = k;                 // all-NaN column: keep the diagonal (NaN propagates)
        if (p != k) {
            for (int j = k + b.tid; j < N; j += b.nthr) { const double t = A[k * lda + j]; A[k * lda + j] = A[p * lda + j]; A[p * lda + j] = t; }
            sgn = -sgn;
        }
        b.sync();
        const double piv = A[k * lda + k];
        if (piv < 0) sgn = -sgn;
        prod.mul(piv);
        const double rinv = 1.0 / piv;
        for (int i = k + 1 + ty; i < N; i += TY) {
            const double l = A[i * lda + k] * rinv;
            for (int j = k + 1 + tx; j < N; j += TX) A[i * lda + j] -= l * A[k * lda + j];
        }
        b.sync();
    }
    if (sign_out) *sign_out = sgn;
    return prod.logabs(ool);
}

// Complex N x N (interleaved re,im; row-major, lda in complex elements) log det:
// returns log|det| and arg(det) in (-pi, pi]  (jnp.linalg.slogdet + jnp.log(phase), src/slater.py:18-19)
CG_DEVI void cg_lu_logdet_complex(const CgBlk& b, double* A, int N, int lda, int* scratch_i,
                                  double& logabs, double& arg, bool ool = false) {
    double* scratch = (double*)scratch_i;
    const int TX = b.nthr < 16 ? b.nthr : 16;
    const int tx = b.tid % TX, ty = b.tid / TX, TY = b.nthr / TX;
    CgCplx pm = {1.0, 0.0}; int pe = 0;             // running product, exponent apart
    for (int k = 0; k < N; ++k) {
        double best = -1.0; int bi = 0x7fffffff;
        for (int i = k + b.tid; i < N; i += b.nthr) {
            const double* a = A + 2 * (i * lda + k);
            const double v = a[0] * a[0] + a[1] * a[1];
            if (v > best) { best = v; bi = i; }
        }
        int p = cg_block_argmax(b, best, bi, scratch);
        if (p < k || p >= N) p = k;
        if (p != k) {
            for (int j = k + b.tid; j < N; j += b.nthr) {
                double* x = A + 2 * (k * lda + j); double* y = A + 2 * (p * lda + j);
                const double t0 = x[0], t1 = x[1]; x[0] = y[0]; x[1] = y[1]; y[0] = t0; y[1] = t1;
            }
        }
        b.sync();
        const CgCplx piv = {A[2 * (k * lda + k)], A[2 * (k * lda + k) + 1]};
        pm = cmul(pm, piv);
        if (p != k) { pm.re = -pm.re; pm.im = -pm.im; }
        {   // renormalise
            int ex; const double mx = fmax(fabs(pm.re), fabs(pm.im));
            (void)frexp(mx, &ex);
            pm.re = ldexp(pm.re, -ex); pm.im = ldexp(pm.im, -ex); pe += ex;
        }
        const CgCplx rinv = cinv(piv);
        for (int i = k + 1 + ty; i < N; i += TY) {
            const CgCplx aik = {A[2 * (i * lda + k)], A[2 * (i * lda + k) + 1]};
            const CgCplx l = cmul(aik, rinv);
            for (int j = k + 1 + tx; j < N; j += TX) {
                double* aij = A + 2 * (i * lda + j);
                const double* akj = A + 2 * (k * lda + j);
                aij[0] -= l.re * akj[0] - l.im * akj[1];
                aij[1] -= l.re * akj[1] + l.im * akj[0];
            }
        }
        b.sync();
    }
    const double m2 = pm.re * pm.re + pm.im * pm.im;
    logabs = 0.5 * (ool ? cg_log_ool(m2) : log(m2)) + (double)pe * 0.693147180559945309417232121458;
    arg = ool ? cg_atan2_ool(pm.im, pm.re) : atan2(pm.im, pm.re);
}

// In-place inverse by Gauss-Jordan with partial pivoting on [A | I] -> [I | A^-1].
// A: N x N real (lda), Ainv: N x N (ldi) output; A destroyed.  Also returns log|det A|.
CG_DEVI double cg_inverse_real(const CgBlk& b, double* A, int N, int lda, double* Ainv, int ldi, int* perm) {
    for (int e = b.tid; e < N * N; e += b.nthr) { int i = e / N, j = e - i * N; Ainv[i * ldi + j] = (i == j) ? 1.0 : 0.0; }
    for (int i = b.tid; i < N; i += b.nthr) perm[i] = i;
    b.sync();
    const int TX = b.nthr < 16 ? b.nthr : 16;
    const int tx = b.tid % TX, ty = b.tid / TX, TY = b.nthr / TX;
    CgScaledProd prod; prod.init();
    for (int k = 0; k < N; ++k) {
        int p = k; double best = -1.0;
        for (int i = k; i < N; ++i) {
            double v = fabs(A[perm[i] * lda + k]);
            if (v > best) { best = v; p = i; }
        }
        const int rk = perm[p], rk_old = perm[k];
        b.sync();
        if (b.tid == 0) { perm[p] = rk_old; perm[k] = rk; }
        const double piv = A[rk * lda + k];
        prod.mul(piv);
        const double rinv = 1.0 / piv;
        b.sync();
        // eliminate column k from every other physical row (Jordan step); pivot row scaled afterwards
        for (int r = ty; r < N; r += TY) {
            if (r == rk) continue;
            const double l = A[r * lda + k] * rinv;
            for (int j = k + 1 + tx; j < N; j += TX) A[r * lda + j] -= l * A[rk * lda + j];
            for (int j = tx; j < N; j += TX) Ainv[r * ldi + j] -= l * Ainv[rk * ldi + j];
        }
        b.sync();
        for (int j = b.tid; j < N; j += b.nthr) {
            if (j > k) A[rk * lda + j] *= rinv;
            Ainv[rk * ldi + j] *= rinv;
        }
        b.sync();
    }
    // physical row perm[k] now holds row k of the inverse; un-permute through A as scratch
    for (int e = b.tid; e < N * N; e += b.nthr) { int k = e / N, j = e - k * N; A[k * lda + j] = Ainv[perm[k] * ldi + j]; }
    b.sync();
    for (int e = b.tid; e < N * N; e += b.nthr) { int k = e / N, j = e - k * N; Ainv[k * ldi + j] = A[k * lda + j]; }
    b.sync();
    return prod.logabs();
}

// Complex inverse + log det, same scheme.  A, Ainv interleaved complex.
CG_DEVI void cg_inverse_complex(const CgBlk& b, double* A, int N, int lda, double* Ainv, int ldi, int* perm,
                                double& logabs, double& arg) {
    for (int e = b.tid; e < N * N; e += b.nthr) {
        int i = e / N, j = e - i * N;
        Ainv[2 * (i * ldi + j)] = (i == j) ? 1.0 : 0.0; Ainv[2 * (i * ldi + j) + 1] = 0.0;
    }
    for (int i = b.tid; i < N; i += b.nthr) perm[i] = i;
    b.sync();
    const int TX = b.nthr < 16 ? b.nthr : 16;
    const int tx = b.tid % TX, ty = b.tid / TX, TY = b.nthr / TX;
    CgCplx pm = {1.0, 0.0}; int pe = 0;
    for (int k = 0; k < N; ++k) {
        int p = k; double best = -1.0;
        for (int i = k; i < N; ++i) {
            const double* a = A + 2 * (perm[i] * lda + k);
            double v = a[0] * a[0] + a[1] * a[1];
            if (v > best) { best = v; p = i; }
        }
        const int rk = perm[p], rk_old = perm[k];
        b.sync();
        if (b.tid == 0) { perm[p] = rk_old; perm[k] = rk; }
        CgCplx piv = {A[2 * (rk * lda + k)], A[2 * (rk * lda + k) + 1]};
        pm = cmul(pm, piv);
        if (p != k) { pm.re = -pm.re; pm.im = -pm.im; }
        { int ex; double mx = fmax(fabs(pm.re), fabs(pm.im)); (void)frexp(mx, &ex);
          pm.re = ldexp(pm.re, -ex); pm.im = ldexp(pm.im, -ex); pe += ex; }
        const CgCplx rinv = cinv(piv);
        b.sync();
        for (int r = ty; r < N; r += TY) {
            if (r == rk) continue;
            CgCplx ark = {A[2 * (r * lda + k)], A[2 * (r * lda + k) + 1]};
            CgCplx l = cmul(ark, rinv);
            for (int j = k + 1 + tx; j < N; j += TX) {
                double* x = A + 2 * (r * lda + j); const double* y = A + 2 * (rk * lda + j);
                x[0] -= l.re * y[0] - l.im * y[1]; x[1] -= l.re * y[1] + l.im * y[0];
            }
            for (int j = tx; j < N; j += TX) {
                double* x = Ainv + 2 * (r * ldi + j); const double* y = Ainv + 2 * (rk * ldi + j);
                x[0] -= l.re * y[0] - l.im * y[1]; x[1] -= l.re * y[1] + l.im * y[0];
            }
        }
        b.sync();
        for (int j = b.tid; j < N; j += b.nthr) {
            if (j > k) { double* x = A + 2 * (rk * lda + j); CgCplx v = cmul({x[0], x[1]}, rinv); x[0] = v.re; x[1] = v.im; }
            double* y = Ainv + 2 * (rk * ldi + j); CgCplx w = cmul({y[0], y[1]}, rinv); y[0] = w.re; y[1] = w.im;
        }
        b.sync();
    }
    for (int e = b.tid; e < N * N; e += b.nthr) {
        int k = e / N, j = e - k * N;
        A[2 * (k * lda + j)] = Ainv[2 * (perm[k] * ldi + j)]; A[2 * (k * lda + j) + 1] = Ainv[2 * (perm[k] * ldi + j) + 1];
    }
    b.sync();
    for (int e = b.tid; e < N * N; e += b.nthr) {
        int k = e / N, j = e - k * N;
        Ainv[2 * (k * ldi + j)] = A[2 * (k * lda + j)]; Ainv[2 * (k * ldi + j) + 1] = A[2 * (k * lda + j) + 1];
    }
    b.sync();
    logabs = 0.5 * log(pm.re * pm.re + pm.im * pm.im) + (double)pe * 0.693147180559945309417232121458;
    arg = atan2(pm.im, pm.re);
}

// In-place Gauss-Jordan inverse with partial pivoting, the matrix resident in LDS (the derivative kernels at N > 32).
// A (N x N, row-major, lda) is overwritten by (P A)^-1; rowsrc[l] = original row now at position l, so that
//     A^-1[i][rowsrc[l]] = A_out[i][l]            (apply with cg_inverse_scatter_* while copying the result out).
// vec: 2 N doubles of scratch (pivot column, scaled pivot row); sc: >= 48 doubles (argmax scratch); rowsrc: N ints.
// Per column: parallel pivot search (one candidate per thread + wave/LDS argmax), physical row swap, then ONE sweep over the whole
// matrix with the pivot row and column staged apart -- 5 barriers per column, every operand in LDS.  (The [A | I] version above
// scans the pivot column serially in every thread and runs out of the HBM workspace at large N: 4-5 ms per matrix at N = 114.)
CG_DEVI void cg_inverse_inplace_real(const CgBlk& b, double* A, int N, int lda, double* vec, double* sc, int* rowsrc, unsigned mN /* cg_div_magic(N) */) {
    double* col = vec; double* row = vec + N;
    for (int i = b.tid; i < N; i += b.nthr) rowsrc[i] = i;
    b.sync();
    for (int k = 0; k < N; ++k) {
        double best = -1.0; int bi = 0x7fffffff;
        for (int i = k + b.tid; i < N; i += b.nthr) {
            const double v = fabs(A[i * lda + k]);
            if (v > best) { best = v; bi = i; }
        }
#if !defined(__HIP_DEVICE_COMPILE__)
        (void)sc;
#endif
        int p = cg_block_argmax(b, best, bi, sc);
        if (p < k || p >= N) p = k;                 // all-NaN column: keep the diagonal (NaN propagates)
        if (p != k) {
            for (int j = b.tid; j < N; j += b.nthr) { const double t = A[k * lda + j]; A[k * lda + j] = A[p * lda + j]; A[p * lda + j] = t; }
            if (b.tid == 0) { const int t = rowsrc[k]; rowsrc[k] = rowsrc[p]; rowsrc[p] = t; }
        }
        b.sync();
        const double rinv = 1.0 / A[k * lda + k];
        for (int j = b.tid; j < N; j += b.nthr) {   // stage the pivot column and the scaled pivot row
            col[j] = A[j * lda + k];
            row[j] = j == k ? rinv : A[k * lda + j] * rinv;
        }
        b.sync();
        for (int e = b.tid; e < N * N; e += b.nthr) {
            const int i = cg_udiv(e, mN), j = e - i * N;
            double v;
            if (i == k) v = row[j];
            else if (j == k) v = -col[i] * rinv;
            else v = A[i * lda + j] - col[i] * row[j];
            A[i * lda + j] = v;
        }
        b.sync();
    }
}
CG_DEVI void cg_inverse_scatter_real(const CgBlk& b, const double* A, int N, int lda, const int* rowsrc, double* Ainv, int ldi, unsigned mN) {
    for (int e = b.tid; e < N * N; e += b.nthr) { const int i = cg_udiv(e, mN), l = e - i * N; Ainv[i * ldi + rowsrc[l]] = A[i * lda + l]; }
}
// complex version (interleaved re, im; lda, ldi in complex elements).  vec: 4 N doubles.
CG_DEVI void cg_inverse_inplace_complex(const CgBlk& b, double* A, int N, int lda, double* vec, double* sc, int* rowsrc, unsigned mN) {
    double* col = vec; double* row = vec + 2 * N;
    for (int i = b.tid; i < N; i += b.nthr) rowsrc[i] = i;
    b.sync();
    for (int k = 0; k < N; ++k) {
        double best = -1.0; int bi = 0x7fffffff;
        for (int i = k + b.tid; i < N; i += b.nthr) {
            const double* a = A + 2 * (i * lda + k);
            const double v = a[0] * a[0] + a[1] * a[1];
            if (v > best) { best = v; bi = i; }
        }
        int p = cg_block_argmax(b, best, bi, sc);
        if (p < k || p >= N) p = k;
        if (p != k) {
            for (int j = b.tid; j < 2 * N; j += b.nthr) { const double t = A[2 * k * lda + j]; A[2 * k * lda + j] = A[2 * p * lda + j]; A[2 * p * lda + j] = t; }
            if (b.tid == 0) { const int t = rowsrc[k]; rowsrc[k] = rowsrc[p]; rowsrc[p] = t; }
        }
        b.sync();
        const CgCplx rinv = cinv({A[2 * (k * lda + k)], A[2 * (k * lda + k) + 1]});
        for (int j = b.tid; j < N; j += b.nthr) {
            col[2 * j] = A[2 * (j * lda + k)]; col[2 * j + 1] = A[2 * (j * lda + k) + 1];
            const CgCplx r = j == k ? rinv : cmul({A[2 * (k * lda + j)], A[2 * (k * lda + j) + 1]}, rinv);
            row[2 * j] = r.re; row[2 * j + 1] = r.im;
        }
        b.sync();
        for (int e = b.tid; e < N * N; e += b.nthr) {
            const int i = cg_udiv(e, mN), j = e - i * N;
            CgCplx v;
            if (i == k) v = {row[2 * j], row[2 * j + 1]};
            else if (j == k) { const CgCplx t = cmul({col[2 * i], col[2 * i + 1]}, rinv); v = {-t.re, -t.im}; }
            else { const CgCplx t = cmul({col[2 * i], col[2 * i + 1]}, {row[2 * j], row[2 * j + 1]}); v = {A[2 * (i * lda + j)] - t.re, A[2 * (i * lda + j) + 1] - t.im}; }
            A[2 * (i * lda + j)] = v.re; A[2 * (i * lda + j) + 1] = v.im;
        }
        b.sync();
    }
}
CG_DEVI void cg_inverse_scatter_complex(const CgBlk& b, const double* A, int N, int lda, const int* rowsrc, double* Ainv, int ldi, unsigned mN) {
    for (int e = b.tid; e < N * N; e += b.nthr) {
        const int i = cg_udiv(e, mN), l = e - i * N;
        Ainv[2 * (i * ldi + rowsrc[l])] = A[2 * (i * lda + l)]; Ainv[2 * (i * ldi + rowsrc[l]) + 1] = A[2 * (i * lda + l) + 1];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Wave-level helpers (gfx950 only) shared by the register / LDS LUs below.
// ------------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double cg_readlane_f64(double v, int lane) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
// wave-uniform maximum of an unsigned key (DPP row reduction + 4 readlanes)
__device__ __forceinline__ unsigned cg_wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8
    const unsigned a = __builtin_amdgcn_readlane((int)v, 15), b = __builtin_amdgcn_readlane((int)v, 31),
                   c = __builtin_amdgcn_readlane((int)v, 47), d = __builtin_amdgcn_readlane((int)v, 63);
    return max(max(a, b), max(c, d));
}

#endif

// ------------------------------------------------------------------------------------------------------------
// Register-tiled, panel-blocked Gauss-Jordan inverse (gfx950; the derivative kernels at 32 < N <= 128).  Every thread keeps a TR x TC
// tile of the matrix in registers for the whole elimination; rows never move (the pivot of column k is the unused row p_k with the
// largest high word), no matrix traffic at all.  The threads that own tile column kp -- LR adjacent lanes of ONE wave (thread = tile
// column * LR + tile row) -- run the in-place Gauss-Jordan of their N x TC panel among themselves (pivot search: lane-local + a DPP row
// reduction; pivot row by readlane: no LDS, no barrier), publish the panel G (= -A[:, panel] P^-1 on the other rows, P^-1 on the pivot
// rows) minus the identity on the pivot rows, and everybody else applies
//     A[:, c] += (G - I_piv) A[pivot rows, c]                    (rank TC, the pivot rows as they stood before the panel)
// to its tile: two barriers per TC columns, and the panel of kp + 1 starts as soon as its owners have updated.  In-place bookkeeping:
// slot k of the tile matrix ends as column p_k of the inverse of the row-permuted system,  A^-1[k][p_j] = S[p_k][j]  (scattered straight
// to the destination).  A, Ainv: any memory.
// History (tools/lu_bench/tile_inv_bench.hip, cycles per real + complex pair at n = 57 / 29): column-at-a-time with the candidates read
// back one by one 1.10 M; the same with 16-byte reads and padded buffers 644 k / 298 k; panels 410 k / 154 k.  What is left is the
// panel owners' dependent chain (~1.4 k cycles per column, latency not arithmetic) and the LDS -> register bandwidth of the rank-TC
// update (96 doubles per 256 operations and thread).
// Shapes (cg_inv_shape_*): N <= 64: <2,4,32> with ceil(N/4) * 32 threads, or <4,4,16> with ceil(N/4) * 16; N <= 128: <4,8,32> with ceil(N/8) * 32.
// sc: LDS, 16-byte aligned, cg_inv_panel_scratch() doubles.
// ------------------------------------------------------------------------------------------------------------
#ifndef CG_INV_T
#define CG_INV_T(i)
#define CG_INV_T_DECL
#endif
struct CgInvShape { int TR, TC, LR; };
CG_HD CgInvShape cg_inv_shape_real(int N, int nthr) {
    if (N <= 64 && ((N + 3) / 4) * 32 <= nthr) return {2, 4, 32};
    if (N <= 64 && ((N + 3) / 4) * 16 <= nthr) return {4, 4, 16};
    if (N <= 128 && ((N + 7) / 8) * 32 <= nthr) return {4, 8, 32};
    return {0, 0, 0};
}
CG_HD CgInvShape cg_inv_shape_complex(int n, int nthr) {
    if (n <= 64 && ((n + 3) / 4) * 32 <= nthr) return {2, 4, 32};
    if (n <= 64 && ((n + 3) / 4) * 16 <= nthr) return {4, 4, 16};
    return {0, 0, 0};
}
// LDS doubles of the inverses of an N x N real and an n x n complex matrix (one after the other in the same scratch); 0: not served
CG_HD size_t cg_inv_panel_scratch(int N, int n, int nthr) {
    const CgInvShape r = cg_inv_shape_real(N, nthr), c = cg_inv_shape_complex(n, nthr);
    if (!r.TR || !c.TR) return 0;
    const size_t NRr = (size_t)((N + r.TR - 1) / r.TR) * r.TR, NCr = (size_t)((N + r.TC - 1) / r.TC) * r.TC;
    const size_t NRc = (size_t)((n + c.TR - 1) / c.TR) * c.TR, NCc = (size_t)((n + c.TC - 1) / c.TC) * c.TC;
    const size_t sr = 2 * r.TC * NRr + r.TC * NCr + 8 + (size_t)N, scx = 2 * (2 * c.TC * NRc + c.TC * NCc) + 8 + (size_t)n;
    return ((sr > scx ? sr : scx) + 1) & ~(size_t)1;
}
#if defined(__HIP_DEVICE_COMPILE__)
template <int LR>
__device__ __forceinline__ unsigned cg_group_max_u32(unsigned v, int g0) {      // maximum over the LR (16 / 32) lanes from g0 on; all of them active
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8
    unsigned m = (unsigned)__builtin_amdgcn_readlane((int)v, g0 + 15);
    if (LR == 32) m = max(m, (unsigned)__builtin_amdgcn_readlane((int)v, g0 + 31));
    return m;
}
template <int TR, int TC, int LR>
__device__ __forceinline__ void cg_inverse_panel_real(const CgBlk& b, const double* A, int N, int lda, double* Ainv, int ldi, double* sc, bool tr_out = false) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef int i4_t __attribute__((ext_vector_type(4)));
    static_assert((TR == 2 || TR == 4 || TR == 8) && TC % 2 == 0 && TC <= 8 && (LR == 16 || LR == 32), "tile shape");
    constexpr int LTR = TR == 2 ? 1 : TR == 4 ? 2 : 3, LLR = LR == 16 ? 4 : 5;
    const int tcn = (N + TC - 1) / TC, trn = (N + TR - 1) / TR;
    const int tc = b.tid >> LLR, tr = b.tid & (LR - 1), lane = b.tid & 63;
    const bool act = tc < tcn && tr < trn;
    const int i0 = tr * TR, j0 = tc * TC;
    const int NR = trn * TR, NC = tcn * TC;
    double* Et = sc; double* R = sc + 2 * TC * NR; int* prb = (int*)(R + TC * NC); int* piv = prb + 16; int* kinv = piv + N;
    double a[TR][TC];
#pragma unroll
    for (int ii = 0; ii < TR; ++ii)
#pragma unroll
        for (int jj = 0; jj < TC; ++jj) {          // (clamped, unpredicated loads)
            const double v = A[(size_t)(i0 + ii < N ? i0 + ii : N - 1) * lda + (j0 + jj < N ? j0 + jj : N - 1)];
            a[ii][jj] = (act && i0 + ii < N && j0 + jj < N) ? v : 0.0;
        }
    CG_INV_T_DECL
    unsigned used = 0;
#pragma unroll
    for (int ii = 0; ii < TR; ++ii) if (!act || i0 + ii >= N) used |= 1u << ii;  // rows beyond the matrix never pivot
    for (int kp = 0; kp < tcn; ++kp) {
        const int kc = kp * TC, kb = N - kc < TC ? N - kc : TC, buf = kp & 1;
        CG_INV_T(0)
        if (tc == kp) {
            // ---- the panel, in place, among its owners
            const int g0 = (kp << LLR) & 63;
            int pr[TC];
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                pr[j] = -1;
                if (j < kb) {
                    unsigned best = 0; int bi = 0;
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii) {
                        const unsigned k = ((used >> ii) & 1u) ? 0u : ((unsigned)((unsigned long long)__double_as_longlong(a[ii][j]) >> 32) & 0x7fffffffu) + 1u;
                        if (k > best) { best = k; bi = ii; }
                    }
                    const unsigned mx = cg_group_max_u32<LR>(best, g0);
                    const int pl = (int)__builtin_ctzll(__ballot(best == mx));           // (a free row always exists: mx >= 1)
                    const int ip = __builtin_amdgcn_readlane(bi, pl);
                    pr[j] = ((pl & (LR - 1)) << LTR) + ip;
                    double rr[TC];
#pragma unroll
                    for (int jj = 0; jj < TC; ++jj) rr[jj] = 0.0;
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
                        if (ip == ii) {                                               // (scalar branch)
#pragma unroll
                            for (int jj = 0; jj < TC; ++jj) rr[jj] = cg_readlane_f64(a[ii][jj], pl);
                        }
                    const double pv = rr[j], r0 = __builtin_amdgcn_rcp(pv);
                    double r1 = fma(r0, fma(-pv, r0, 1.0), r0);
                    r1 = fma(r1, fma(-pv, r1, 1.0), r1);
                    const double rinv = fabs(pv) > 0.0 ? r1 : r0;
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii) {
                        const double fr = a[ii][j] * rinv;
#pragma unroll
                        for (int jj = 0; jj < TC; ++jj) if (jj != j) a[ii][jj] = fma(-fr, rr[jj], a[ii][jj]);
                        a[ii][j] = -fr;
                    }
                    const bool me = lane == pl;
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
                        if (ip == ii) {
#pragma unroll
                            for (int jj = 0; jj < TC; ++jj) a[ii][jj] = me ? (jj == j ? rinv : rr[jj] * rinv) : a[ii][jj];
                        }
                    if (me) used |= 1u << ip;
                }
            }
            if (tr < trn) {
#pragma unroll
                for (int j = 0; j < TC; ++j)
#pragma unroll
                    for (int ii = 0; ii < TR; ii += 2)
                        *(d2_t*)(Et + (buf * TC + j) * NR + i0 + ii) = d2_t{a[ii][j] - (i0 + ii == pr[j] ? 1.0 : 0.0), a[ii + 1][j] - (i0 + ii + 1 == pr[j] ? 1.0 : 0.0)};
            }
            if (tr == 0) {
#pragma unroll
                for (int j = 0; j < TC; ++j) { prb[buf * 8 + j] = pr[j]; if (j < kb) piv[kc + j] = pr[j]; }
            }
        }
        CG_INV_T(1)
        b.sync();
        CG_INV_T(2)
        const bool upd = act && tc != kp;
        if (upd) {                                    // the pivot rows as they stand, from the tiles that hold them
            int pq[8];
            { const i4_t q0 = *(const i4_t*)(prb + buf * 8), q1 = *(const i4_t*)(prb + buf * 8 + 4);
              pq[0] = q0[0]; pq[1] = q0[1]; pq[2] = q0[2]; pq[3] = q0[3]; pq[4] = q1[0]; pq[5] = q1[1]; pq[6] = q1[2]; pq[7] = q1[3]; }
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                const int p = __builtin_amdgcn_readfirstlane(pq[j]);
                if (j < kb && tr == (p >> LTR)) {
                    const int ip = p & (TR - 1);
                    used |= 1u << ip;                 // (every tile of the row learns that it has served)
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
                        if (ip == ii) {
#pragma unroll
                            for (int jj = 0; jj < TC; jj += 2) *(d2_t*)(R + j * NC + j0 + jj) = d2_t{a[ii][jj], a[ii][jj + 1]};
                        }
                }
            }
        }
        CG_INV_T(3)
        b.sync();
        CG_INV_T(4)
        if (upd) {
#pragma unroll
            for (int j = 0; j < TC; ++j)
                if (j < kb) {
                    double e[TR], r[TC];
#pragma unroll
                    for (int ii = 0; ii < TR; ii += 2) { const d2_t t = *(const d2_t*)(Et + (buf * TC + j) * NR + i0 + ii); e[ii] = t[0]; e[ii + 1] = t[1]; }
#pragma unroll
                    for (int jj = 0; jj < TC; jj += 2) { const d2_t t = *(const d2_t*)(R + j * NC + j0 + jj); r[jj] = t[0]; r[jj + 1] = t[1]; }
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
#pragma unroll
                        for (int jj = 0; jj < TC; ++jj) a[ii][jj] = fma(e[ii], r[jj], a[ii][jj]);
                }
        }
        CG_INV_T(5)
    }
    b.sync();
    for (int t = b.tid; t < N; t += b.nthr) kinv[piv[t]] = t;
    b.sync();
    if (act) {
#pragma unroll
        for (int ii = 0; ii < TR; ++ii)
#pragma unroll
            for (int jj = 0; jj < TC; ++jj)
                if (i0 + ii < N && j0 + jj < N) {              // tr_out: the transposed inverse (what the reverse sweeps read row-wise)
                    if (tr_out) Ainv[(size_t)piv[j0 + jj] * ldi + kinv[i0 + ii]] = a[ii][jj];
                    else Ainv[(size_t)kinv[i0 + ii] * ldi + piv[j0 + jj]] = a[ii][jj];
                }
    }
    b.sync();
}
// complex version: interleaved (re, im), lda / ldi in complex elements
template <int TR, int TC, int LR>
__device__ __forceinline__ void cg_inverse_panel_complex(const CgBlk& b, const double* A, int N, int lda, double* Ainv, int ldi, double* sc) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef int i4_t __attribute__((ext_vector_type(4)));
    static_assert((TR == 2 || TR == 4) && TC <= 8 && (LR == 16 || LR == 32), "tile shape");
    constexpr int LTR = TR == 2 ? 1 : 2, LLR = LR == 16 ? 4 : 5;
    const int tcn = (N + TC - 1) / TC, trn = (N + TR - 1) / TR;
    const int tc = b.tid >> LLR, tr = b.tid & (LR - 1), lane = b.tid & 63;
    const bool act = tc < tcn && tr < trn;
    const int i0 = tr * TR, j0 = tc * TC;
    const int NR = trn * TR, NC = tcn * TC;
    double* Et = sc; double* R = sc + 4 * TC * NR; int* prb = (int*)(R + 2 * TC * NC); int* piv = prb + 16; int* kinv = piv + N;
    double ar[TR][TC], ai[TR][TC];
#pragma unroll
    for (int ii = 0; ii < TR; ++ii)
#pragma unroll
        for (int jj = 0; jj < TC; ++jj) {
            const bool ok = act && i0 + ii < N && j0 + jj < N;
            const double* q = A + 2 * ((size_t)(i0 + ii < N ? i0 + ii : N - 1) * lda + (j0 + jj < N ? j0 + jj : N - 1));     // (clamped, unpredicated loads)
            const double vr = q[0], vi = q[1];
            ar[ii][jj] = ok ? vr : 0.0; ai[ii][jj] = ok ? vi : 0.0;
        }
    unsigned used = 0;
#pragma unroll
    for (int ii = 0; ii < TR; ++ii) if (!act || i0 + ii >= N) used |= 1u << ii;
    for (int kp = 0; kp < tcn; ++kp) {
        const int kc = kp * TC, kb = N - kc < TC ? N - kc : TC, buf = kp & 1;
        if (tc == kp) {
            const int g0 = (kp << LLR) & 63;
            int pr[TC];
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                pr[j] = -1;
                if (j < kb) {
                    unsigned best = 0; int bi = 0;
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii) {
                        const double m2 = ar[ii][j] * ar[ii][j] + ai[ii][j] * ai[ii][j];
                        const unsigned k = ((used >> ii) & 1u) ? 0u : ((unsigned)((unsigned long long)__double_as_longlong(m2) >> 32) & 0x7fffffffu) + 1u;
                        if (k > best) { best = k; bi = ii; }
                    }
                    const unsigned mx = cg_group_max_u32<LR>(best, g0);
                    const int pl = (int)__builtin_ctzll(__ballot(best == mx));
                    const int ip = __builtin_amdgcn_readlane(bi, pl);
                    pr[j] = ((pl & (LR - 1)) << LTR) + ip;
                    CgCplx rr[TC];
#pragma unroll
                    for (int jj = 0; jj < TC; ++jj) rr[jj] = CgCplx{0.0, 0.0};
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
                        if (ip == ii) {
#pragma unroll
                            for (int jj = 0; jj < TC; ++jj) rr[jj] = CgCplx{cg_readlane_f64(ar[ii][jj], pl), cg_readlane_f64(ai[ii][jj], pl)};
                        }
                    const CgCplx rinv = cinv(rr[j]);
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii) {
                        const CgCplx fr = cmul({ar[ii][j], ai[ii][j]}, rinv);
#pragma unroll
                        for (int jj = 0; jj < TC; ++jj)
                            if (jj != j) {
                                ar[ii][jj] = fma(fr.im, rr[jj].im, fma(-fr.re, rr[jj].re, ar[ii][jj]));
                                ai[ii][jj] = fma(-fr.im, rr[jj].re, fma(-fr.re, rr[jj].im, ai[ii][jj]));
                            }
                        ar[ii][j] = -fr.re; ai[ii][j] = -fr.im;
                    }
                    const bool me = lane == pl;
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
                        if (ip == ii) {
#pragma unroll
                            for (int jj = 0; jj < TC; ++jj) {
                                const CgCplx t = jj == j ? rinv : cmul(rr[jj], rinv);
                                ar[ii][jj] = me ? t.re : ar[ii][jj]; ai[ii][jj] = me ? t.im : ai[ii][jj];
                            }
                        }
                    if (me) used |= 1u << ip;
                }
            }
            if (tr < trn) {
#pragma unroll
                for (int j = 0; j < TC; ++j)
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
                        *(d2_t*)(Et + 2 * ((buf * TC + j) * NR + i0 + ii)) = d2_t{ar[ii][j] - (i0 + ii == pr[j] ? 1.0 : 0.0), ai[ii][j]};
            }
            if (tr == 0) {
#pragma unroll
                for (int j = 0; j < TC; ++j) { prb[buf * 8 + j] = pr[j]; if (j < kb) piv[kc + j] = pr[j]; }
            }
        }
        b.sync();
        const bool upd = act && tc != kp;
        if (upd) {
            int pq[8];
            { const i4_t q0 = *(const i4_t*)(prb + buf * 8), q1 = *(const i4_t*)(prb + buf * 8 + 4);
              pq[0] = q0[0]; pq[1] = q0[1]; pq[2] = q0[2]; pq[3] = q0[3]; pq[4] = q1[0]; pq[5] = q1[1]; pq[6] = q1[2]; pq[7] = q1[3]; }
#pragma unroll
            for (int j = 0; j < TC; ++j) {
                const int p = __builtin_amdgcn_readfirstlane(pq[j]);
                if (j < kb && tr == (p >> LTR)) {
                    const int ip = p & (TR - 1);
                    used |= 1u << ip;
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
                        if (ip == ii) {
#pragma unroll
                            for (int jj = 0; jj < TC; ++jj) *(d2_t*)(R + 2 * (j * NC + j0 + jj)) = d2_t{ar[ii][jj], ai[ii][jj]};
                        }
                }
            }
        }
        b.sync();
        if (upd) {
#pragma unroll
            for (int j = 0; j < TC; ++j)
                if (j < kb) {
                    d2_t e[TR], r[TC];
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii) e[ii] = *(const d2_t*)(Et + 2 * ((buf * TC + j) * NR + i0 + ii));
#pragma unroll
                    for (int jj = 0; jj < TC; ++jj) r[jj] = *(const d2_t*)(R + 2 * (j * NC + j0 + jj));
#pragma unroll
                    for (int ii = 0; ii < TR; ++ii)
#pragma unroll
                        for (int jj = 0; jj < TC; ++jj) {
                            ar[ii][jj] = fma(-e[ii][1], r[jj][1], fma(e[ii][0], r[jj][0], ar[ii][jj]));
                            ai[ii][jj] = fma(e[ii][1], r[jj][0], fma(e[ii][0], r[jj][1], ai[ii][jj]));
                        }
                }
        }
    }
    b.sync();
    for (int t = b.tid; t < N; t += b.nthr) kinv[piv[t]] = t;
    b.sync();
    if (act) {
#pragma unroll
        for (int ii = 0; ii < TR; ++ii)
#pragma unroll
            for (int jj = 0; jj < TC; ++jj)
                if (i0 + ii < N && j0 + jj < N) {
                    double* o = Ainv + 2 * ((size_t)kinv[i0 + ii] * ldi + piv[j0 + jj]);
                    o[0] = ar[ii][jj]; o[1] = ai[ii][jj];
                }
    }
    b.sync();
}
// dispatch on the shape (the caller has checked cg_inv_panel_scratch(N, n, nthr) != 0)
__device__ __forceinline__ void cg_inverse_panel_real(const CgBlk& b, const double* A, int N, int lda, double* Ainv, int ldi, double* sc, bool tr_out = false) {
    const CgInvShape s = cg_inv_shape_real(N, b.nthr);
    if (s.TC == 8) cg_inverse_panel_real<4, 8, 32>(b, A, N, lda, Ainv, ldi, sc, tr_out);
    else if (s.LR == 32) cg_inverse_panel_real<2, 4, 32>(b, A, N, lda, Ainv, ldi, sc, tr_out);
    else cg_inverse_panel_real<4, 4, 16>(b, A, N, lda, Ainv, ldi, sc, tr_out);
}
__device__ __forceinline__ void cg_inverse_panel_complex(const CgBlk& b, const double* A, int n, int lda, double* Ainv, int ldi, double* sc) {
    const CgInvShape s = cg_inv_shape_complex(n, b.nthr);
    if (s.LR == 32) cg_inverse_panel_complex<2, 4, 32>(b, A, n, lda, Ainv, ldi, sc);
    else cg_inverse_panel_complex<4, 4, 16>(b, A, n, lda, Ainv, ldi, sc);
}
#endif

// ------------------------------------------------------------------------------------------------------------
// Workgroup GEMM on the f64 matrix cores for the dense contractions of the derivative kernels (C = J J^T, M = J^-1 J', the
// N x N x 16 adjoint products): C(row, col) = sum_k fa(row, k) fb(k, col), rows < M, cols < Nc, k < K.  One wave per 16 x 16 tile,
// tiles dealt round-robin to the waves; the operands come straight from wherever the functors read them (LDS, or the L2-resident
// workspace), one double per lane and MFMA -- fine for these sizes (K <= 128), with the k loop unrolled for loads in flight.
// fc(row, col, value) stores.  v_mfma_f64_16x16x4: A lane l -> (row l & 15, k l >> 4), B lane l -> (k l >> 4, col l & 15),
// C/D lane l -> rows (l >> 4) + 4 r, col l & 15.  Host shim: the plain triple loop.
// ------------------------------------------------------------------------------------------------------------
template <class FA, class FB, class FC>
CG_DEVI void cg_gemm_wg(const CgBlk& b, int M, int Nc, int K, FA fa, FB fb, FC fc) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const int lane = b.tid & 63, col = lane & 15, kq = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
    const int tm = (M + 15) >> 4, tn = (Nc + 15) >> 4;
    for (int t = wave; t < tm * tn; t += nw) {
        const int ti = t / tn, tj = t - ti * tn;
        const int ra = 16 * ti + col, cb = 16 * tj + col;
        const bool aok = ra < M, bok = cb < Nc;
        d4_t c = {0, 0, 0, 0};
#pragma unroll 4
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + kq;
            const bool kok = k < K;
            const double av = (aok && kok) ? fa(ra, k) : 0.0;
            const double bv = (bok && kok) ? fb(k, cb) : 0.0;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 16 * ti + kq + 4 * r;
            if (rr < M && bok) fc(rr, cb, c[r]);
        }
    }
#else
    for (int e = b.tid; e < M * Nc; e += b.nthr) {
        const int r = e / Nc, c = e - r * Nc;
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += fa(r, k) * fb(k, c);
        fc(r, c, acc);
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Wave-level LU, 2-D lane layout (gfx950): the sampler's determinants at N = n*d <= 32, n <= 16.
//   real    : lane = 2 r + c holds row r, columns j = 2 m + c  (m < NMAX/2)   -> 2 N of 64 lanes busy, N/2 doubles each
//   complex : lane = 4 r + c holds row r, columns j = 4 m + c  (m < 4)        -> 4 n of 64 lanes busy
// Per column: the candidate column entry is shared inside the lane pair / quad by DPP quad_perm, the pivot is
// found by a DPP max on the high word of |a_rk| + ballot (a near-maximal pivot is as good as the maximal one), the
// pivot row's tail goes through a small LDS scratch (written by the lanes that own it, read back by every lane with
// ds_read2_b64: LDS executes one wave's instructions in order, so no barrier) instead of two v_readlane per element,
// and reciprocals are v_rcp_f64 + two Newton steps instead of IEEE divisions.  Rows are never swapped: a row that
// has served as pivot stops updating.  Must be called by all 64 lanes of ONE wave.
// ------------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
// 1/x to ~1 ulp (x = 0 -> +-inf like the division it replaces)
__device__ __forceinline__ double cg_fast_rcp(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r0, 1.0);
    double r = fma(r0, e, r0);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return fabs(x) > 0.0 ? r : r0;
}
// 1/x with ONE Newton step (~1e-14 relative): enough for LU multipliers (the determinant uses the pivots themselves)
__device__ __forceinline__ double cg_fast_rcp1(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double r = fma(r0, fma(-x, r0, 1.0), r0);
    return fabs(x) > 0.0 ? r : r0;
}
// sqrt(x) and 1/sqrt(x) for x > 0 (normal range) from v_rsq_f64 + two Newton steps
__device__ __forceinline__ void cg_fast_sqrt_rsqrt(double x, double& sq, double& rs) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    const double g = x * y;                           // sqrt(x) with one correction step: g + (x - g^2) y / 2
    const double q = fma(fma(-g, g, x), 0.5 * y, g);
    const bool pos = x > 0.0;                         // x = 0: sqrt 0, 1/sqrt inf (as sqrt() and the division would give)
    sq = pos ? q : x;
    rs = pos ? y : __builtin_amdgcn_rsq(x);
}
template <int CTRL>
__device__ __forceinline__ double cg_dpp_f64(double v) {
    const long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)u, CTRL, 0xf, 0xf, true);      // (quad_perm controls: every lane has a source;
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xf, 0xf, true); //  bound_ctrl spares the zero-initialised destination)
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// log|det A|, A: N x N real in LDS (row-major, lda), N <= NMAX <= 32 (NMAX even).  scr: 32 doubles of LDS.
template <int NMAX>
__device__ __forceinline__ double cg_wave_lu2_logabsdet(const double* A, int N, int lda, double* scr) {
    constexpr int MH = NMAX / 2, MHP = (MH + 1) & ~1;
    const int lane = threadIdx.x & 63, r = lane >> 1, c = lane & 1;
    double a[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) { const int j = 2 * m + c; a[m] = (r < N && j < N) ? A[r * lda + j] : 0.0; }
    bool done = r >= N;
    unsigned long long donemask = N >= 32 ? 0ull : ~0ull << (2 * N);          // wave-uniform copy of `done`
    double* mine = scr + c * MHP;
    CgScaledProd prod; prod.init();
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        if (k < N) {
            const int ck = k & 1, mk = k >> 1;
            const double ak = ck ? cg_dpp_f64<0xF5>(a[mk]) : cg_dpp_f64<0xA0>(a[mk]);      // column k of this lane's row
            // threshold pivoting: the first unfinished row serves unless some candidate is more than 4x larger
            // (growth bounded by 4 per step); only then the full search runs.  The common case (J = I + small) never
            // leaves the short path, which takes ~30 dependent instructions out of every step.
            int p = (int)__builtin_ctzll(~donemask);
            double piv = cg_readlane_f64(ak, p);
            if (__ballot(!done && fabs(ak) * 0.25 > fabs(piv))) {
                const unsigned key = done ? 0u : (unsigned)(__double_as_longlong(fabs(ak)) >> 32) + 1u;
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx && !done);
                p = mask ? (int)__builtin_ctzll(mask) : p;
                piv = cg_readlane_f64(ak, p);
            }
            prod.mul(piv);
            const double rinv = cg_fast_rcp1(piv);
            const bool isp = r == (p >> 1);
            const double l = (done || isp) ? 0.0 : ak * rinv;
            const int m0 = ck ? mk + 1 : mk;          // first local column that still lies right of column k (or is k itself)
            if (isp) {
#pragma unroll
                for (int m = m0; m < MH; ++m) mine[m] = a[m];
            }
            asm volatile("" ::: "memory");            // cross-lane hand-off: the reads below must stay behind the stores
#pragma unroll
            for (int m = m0; m < MH; ++m) a[m] = fma(-l, mine[m], a[m]);
            asm volatile("" ::: "memory");
            done = done || isp;
            donemask |= 3ull << (p & ~1);
        }
    }
    return prod.logabs(true);
}

// complex version: A interleaved (re,im) N x N in LDS, N <= NMAX <= 16; scr: 32 doubles of LDS.
// Returns log|det| and arg(det) including the sign of the equivalent row permutation.
template <int NMAX>
__device__ __forceinline__ void cg_wave_lu2_logdet_complex(const double* A, int N, int lda, double* scr, double& logabs, double& arg) {
    constexpr int MQ = (NMAX + 3) / 4;
    const int lane = threadIdx.x & 63, r = lane >> 2, c = lane & 3;
    double ar[MQ], ai[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        const int j = 4 * m + c;
        const bool ok = r < N && j < N;
        ar[m] = ok ? A[2 * (r * lda + j)] : 0.0;
        ai[m] = ok ? A[2 * (r * lda + j) + 1] : 0.0;
    }
    bool done = r >= N;
    unsigned long long donemask = N >= 16 ? 0ull : ~0ull << (4 * N);          // wave-uniform copy of `done`
    int mypos = r;                    // position of this row under the equivalent sequence of row swaps
    int parity = 0;
    CgCplx pm = {1.0, 0.0}; int pe = 0;
    double* mine = scr + 2 * c;       // complex element j = 4 m + c of the published row at scr[2 j]
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        if (k < N) {
            const int ck = k & 3, mk = k >> 2;
            double akr, aki;
            if (ck == 0) { akr = cg_dpp_f64<0x00>(ar[mk]); aki = cg_dpp_f64<0x00>(ai[mk]); }
            else if (ck == 1) { akr = cg_dpp_f64<0x55>(ar[mk]); aki = cg_dpp_f64<0x55>(ai[mk]); }
            else if (ck == 2) { akr = cg_dpp_f64<0xAA>(ar[mk]); aki = cg_dpp_f64<0xAA>(ai[mk]); }
            else { akr = cg_dpp_f64<0xFF>(ar[mk]); aki = cg_dpp_f64<0xFF>(ai[mk]); }
            const double m2 = akr * akr + aki * aki;
            // threshold pivoting (see the real version): first unfinished row unless a candidate is > 4x larger in modulus
            int p = (int)__builtin_ctzll(~donemask);
            if (__ballot(!done && m2 * 0.0625 > cg_readlane_f64(m2, p))) {
                const unsigned key = done ? 0u : (unsigned)(__double_as_longlong(m2) >> 32) + 1u;
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx && !done);
                p = mask ? (int)__builtin_ctzll(mask) : p;
            }
            const bool isp = r == (p >> 2);
            const int posp = __builtin_amdgcn_readlane(mypos, p);
            if (posp != k) {          // swap positions k <-> posp
                parity ^= 1;
                if (mypos == k) mypos = posp;
                if (isp) mypos = k;
            }
            const CgCplx piv = {cg_readlane_f64(akr, p), cg_readlane_f64(aki, p)};
            pm = cmul(pm, piv);
            if ((k & 3) == 3) {       // |piv| = O(1..n): four factors cannot overflow between renormalisations
                int ex; const double mxv = fmax(fabs(pm.re), fabs(pm.im)); (void)frexp(mxv, &ex);
                pm.re = ldexp(pm.re, -ex); pm.im = ldexp(pm.im, -ex); pe += ex;
            }
            const double rd = cg_fast_rcp1(piv.re * piv.re + piv.im * piv.im);
            const CgCplx rinv = {piv.re * rd, -piv.im * rd};
            CgCplx l = cmul({akr, aki}, rinv);
            if (done || isp) { l.re = 0.0; l.im = 0.0; }
            const int m0 = ck == 3 ? mk + 1 : mk;
            if (isp) {
#pragma unroll
                for (int m = m0; m < MQ; ++m) { mine[8 * m] = ar[m]; mine[8 * m + 1] = ai[m]; }
            }
            asm volatile("" ::: "memory");            // cross-lane hand-off through LDS (see the real version)
#pragma unroll
            for (int m = m0; m < MQ; ++m) {
                const double pr = mine[8 * m], pi = mine[8 * m + 1];
                ar[m] = fma(-l.re, pr, fma(l.im, pi, ar[m]));
                ai[m] = fma(-l.re, pi, fma(-l.im, pr, ai[m]));
            }
            asm volatile("" ::: "memory");
            done = done || isp;
            donemask |= 15ull << (p & ~3);
        }
    }
    if (parity) { pm.re = -pm.re; pm.im = -pm.im; }
    logabs = 0.5 * cg_log_pos(pm.re * pm.re + pm.im * pm.im) + (double)pe * 0.693147180559945309417232121458;
    arg = cg_atan2_ool(pm.im, pm.re);
}
// Both determinants of one walker in ONE instruction stream (single-wave workgroups, n <= 16): the real step k and -- on every
// other k -- a complex step are issued together.  Each factorisation alone is a chain of dependent f64 operations (~12 cycles
// per instruction at two waves per SIMD); interleaved, one chain's latencies are filled by the other's instructions.  The two
// rarely-taken pivot searches share one branch, so that the common path of a step is a single basic block the scheduler can
// interleave.  Arithmetic per matrix identical to cg_wave_lu2_logabsdet / cg_wave_lu2_logdet_complex (same results).
// NR = 2 NC.  scr_r, scr_c: 32 doubles of LDS each.
template <int NR, int NC>
__device__ __forceinline__ void cg_wave_lu2_both(const double* A, int N, int lda, double* scr_r, const double* Cm, int n, int ldc, double* scr_c,
                                                 double& logabs_r, double& logabs_c, double& arg_c) {
    constexpr int MH = NR / 2, MHP = (MH + 1) & ~1, MQ = (NC + 3) / 4;
    const int lane = threadIdx.x & 63;
    const int rr = lane >> 1, rc_ = lane & 1;            // real: row, column parity
    const int cr = lane >> 2, cc = lane & 3;             // complex: row, column mod 4
    double a[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) { const int j = 2 * m + rc_; a[m] = (rr < N && j < N) ? A[rr * lda + j] : 0.0; }
    double ar[MQ], ai[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        const int j = 4 * m + cc; const bool ok = cr < n && j < n;
        ar[m] = ok ? Cm[2 * (cr * ldc + j)] : 0.0; ai[m] = ok ? Cm[2 * (cr * ldc + j) + 1] : 0.0;
    }
    bool rdone = rr >= N, cdone = cr >= n;
    unsigned long long rmask = N >= 32 ? 0ull : ~0ull << (2 * N), cmask = n >= 16 ? 0ull : ~0ull << (4 * n);
    int mypos = cr, parity = 0;
    double* rmine = scr_r + rc_ * MHP;
    double* cmine = scr_c + 2 * cc;
    CgScaledProd prod; prod.init();
    CgCplx pm = {1.0, 0.0}; int pe = 0;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const bool rs = k < N;                           // (wave-uniform) real step k
        const int kc = k >> 1;
        const bool cs = (k & 1) == 0 && kc < NC && kc < n;   // complex step kc rides on the even real steps
        if (!rs && !cs) continue;
        // ---- candidates
        const int ck = k & 1, mk = k >> 1;
        const double ak = ck ? cg_dpp_f64<0xF5>(a[mk]) : cg_dpp_f64<0xA0>(a[mk]);
        int p = (int)__builtin_ctzll(~rmask);
        double piv = cg_readlane_f64(ak, p);
        const int qk = kc & 3, qm = kc >> 2;
        double akr = 0.0, aki = 0.0;
        if (cs) {
            if (qk == 0) { akr = cg_dpp_f64<0x00>(ar[qm]); aki = cg_dpp_f64<0x00>(ai[qm]); }
            else if (qk == 1) { akr = cg_dpp_f64<0x55>(ar[qm]); aki = cg_dpp_f64<0x55>(ai[qm]); }
            else if (qk == 2) { akr = cg_dpp_f64<0xAA>(ar[qm]); aki = cg_dpp_f64<0xAA>(ai[qm]); }
            else { akr = cg_dpp_f64<0xFF>(ar[qm]); aki = cg_dpp_f64<0xFF>(ai[qm]); }
        }
        const double m2 = akr * akr + aki * aki;
        int pc = (int)__builtin_ctzll(~cmask);
        const unsigned long long tr = rs ? __ballot(!rdone && fabs(ak) * 0.25 > fabs(piv)) : 0ull;
        const unsigned long long tc = cs ? __ballot(!cdone && m2 * 0.0625 > cg_readlane_f64(m2, pc)) : 0ull;
        if (tr | tc) {                                   // a candidate more than 4x larger than the first unfinished row: full search
            if (tr) {
                const unsigned key = rdone ? 0u : (unsigned)(__double_as_longlong(fabs(ak)) >> 32) + 1u;
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx && !rdone);
                p = mask ? (int)__builtin_ctzll(mask) : p;
                piv = cg_readlane_f64(ak, p);
            }
            if (tc) {
                const unsigned key = cdone ? 0u : (unsigned)(__double_as_longlong(m2) >> 32) + 1u;
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx && !cdone);
                pc = mask ? (int)__builtin_ctzll(mask) : pc;
            }
        }
        // ---- multipliers
        const bool risp = rr == (p >> 1);
        double l = 0.0;
        if (rs) {
            prod.mul(piv);
            const double rinv = cg_fast_rcp1(piv);
            l = (rdone || risp) ? 0.0 : ak * rinv;
        }
        const bool cisp = cr == (pc >> 2);
        CgCplx lc = {0.0, 0.0};
        if (cs) {
            const int posp = __builtin_amdgcn_readlane(mypos, pc);
            parity ^= (posp != kc) ? 1 : 0;
            if (mypos == kc) mypos = posp;
            if (cisp) mypos = kc;
            const CgCplx cpiv = {cg_readlane_f64(akr, pc), cg_readlane_f64(aki, pc)};
            pm = cmul(pm, cpiv);
            if ((kc & 3) == 3) {
                int ex; const double mxv = fmax(fabs(pm.re), fabs(pm.im)); (void)frexp(mxv, &ex);
                pm.re = ldexp(pm.re, -ex); pm.im = ldexp(pm.im, -ex); pe += ex;
            }
            const double rd = cg_fast_rcp1(cpiv.re * cpiv.re + cpiv.im * cpiv.im);
            const CgCplx rinv = {cpiv.re * rd, -cpiv.im * rd};
            lc = cmul({akr, aki}, rinv);
            if (cdone || cisp) { lc.re = 0.0; lc.im = 0.0; }
        }
        // ---- pivot rows through LDS (written by their owners, read back by every lane; in-order per wave), updates
        const int m0 = ck ? mk + 1 : mk;
        const int q0 = qk == 3 ? qm + 1 : qm;
        if (rs && risp) {
#pragma unroll
            for (int m = m0; m < MH; ++m) rmine[m] = a[m];
        }
        if (cs && cisp) {
#pragma unroll
            for (int m = q0; m < MQ; ++m) { cmine[8 * m] = ar[m]; cmine[8 * m + 1] = ai[m]; }
        }
        asm volatile("" ::: "memory");
        if (rs) {
#pragma unroll
            for (int m = m0; m < MH; ++m) a[m] = fma(-l, rmine[m], a[m]);
        }
        if (cs) {
#pragma unroll
            for (int m = q0; m < MQ; ++m) {
                const double pr = cmine[8 * m], pi = cmine[8 * m + 1];
                ar[m] = fma(-lc.re, pr, fma(lc.im, pi, ar[m]));
                ai[m] = fma(-lc.re, pi, fma(-lc.im, pr, ai[m]));
            }
        }
        asm volatile("" ::: "memory");
        if (rs) { rdone = rdone || risp; rmask |= 3ull << (p & ~1); }
        if (cs) { cdone = cdone || cisp; cmask |= 15ull << (pc & ~3); }
    }
    if (parity) { pm.re = -pm.re; pm.im = -pm.im; }
    logabs_r = prod.logabs(true);
    logabs_c = 0.5 * cg_log_pos(pm.re * pm.re + pm.im * pm.im) + (double)pe * 0.693147180559945309417232121458;
    arg_c = cg_atan2_ool(pm.im, pm.re);
}
#endif

// ------------------------------------------------------------------------------------------------------------
// Wave-level INVERSES (gfx950), same lane layouts as the wave-level LUs above: Gauss-Jordan on [A | I] held in registers,
// one wave, no barriers.  Rows never move: the row that serves as the pivot of step k ends up holding row k of A^-1 and
// stores it there.  The pivot row goes through an LDS scratch (A part right of the pivot column + the whole I part),
// every lane reads it back (LDS executes one wave's accesses in order).  A is left intact.  Used by the set-up of the
// derivative kernels (J^-1 and D^-1 of one walker concurrently on two waves) instead of the workgroup-wide
// Gauss-Jordan with four barriers per column.
//   real    : N <= NMAX <= 32, lane = 2 r + c, columns j = 2 m + c;   scr: 4 * (NMAX/2 + 1) doubles of LDS
//   complex : N <= NMAX <= 16, lane = 4 r + c, columns j = 4 m + c;   scr: 4 * NMAX + 16 doubles of LDS
// ------------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
template <int NMAX>
__device__ __forceinline__ void cg_wave_inverse_real(const double* A, int N, int lda, double* Ainv, int ldi, double* scr) {
    // IN-PLACE Gauss-Jordan (round 3): the slot of column k receives the inverse's column as soon as column k is eliminated, so a
    // lane carries NMAX / 2 doubles instead of NMAX ([A | I]) and a step is NMAX / 2 multiply-adds and LDS words per lane instead of
    // ~NMAX.  Rows never move (a finished pivot row stops being a candidate); with p_k the pivot row of column k the slots end as
    //     A^-1[k][p_j] = S[p_k][j],   resolved while storing.
    constexpr int MH = NMAX / 2, MHP = (MH + 1) & ~1;
    const int lane = threadIdx.x & 63, r = lane >> 1, c = lane & 1;
    double a[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) {
        const int j = 2 * m + c;
        a[m] = (r < N && j < N) ? A[r * lda + j] : 0.0;
    }
    bool done = r >= N;
    unsigned long long donemask = N >= 32 ? 0ull : ~0ull << (2 * N);
    int myk = -1;
    int pv[MH];                                          // pivot row of the lane's own columns j = 2 m + c (registers: static indices only)
    double* mine = scr + c * MHP;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        if ((k & 1) == 0) pv[k >> 1] = 0;
        if (k < N) {
            const int ck = k & 1, mk = k >> 1;
            const double ak = ck ? cg_dpp_f64<0xF5>(a[mk]) : cg_dpp_f64<0xA0>(a[mk]);
            int p = (int)__builtin_ctzll(~donemask);
            double piv = cg_readlane_f64(ak, p);
            if (__ballot(!done && fabs(ak) * 0.25 > fabs(piv))) {      // threshold pivoting (see cg_wave_lu2_logabsdet)
                const unsigned key = done ? 0u : (unsigned)(__double_as_longlong(fabs(ak)) >> 32) + 1u;
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx && !done);
                p = mask ? (int)__builtin_ctzll(mask) : p;
                piv = cg_readlane_f64(ak, p);
            }
            if (c == ck) pv[mk] = p >> 1;
            const double rinv = cg_fast_rcp(piv);
            const bool isp = r == (p >> 1);
            const double l = isp ? 0.0 : ak * rinv;          // Jordan step: every other row, finished ones included
            if (isp) {
#pragma unroll
                for (int m = 0; m < MH; ++m) mine[m] = a[m];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int m = 0; m < MH; ++m) a[m] = fma(-l, mine[m], a[m]);
            asm volatile("" ::: "memory");
            if (isp) {                                       // the pivot row itself: scaled
#pragma unroll
                for (int m = 0; m < MH; ++m) a[m] *= rinv;
                myk = k;
            }
            if (c == ck) a[mk] = isp ? rinv : -l;            // the slot of column k now holds the inverse's column
            done = done || isp;
            donemask |= 3ull << (p & ~1);
        }
    }
    if (myk >= 0) {
#pragma unroll
        for (int m = 0; m < MH; ++m) {
            const int j = 2 * m + c;
            if (j < N) Ainv[myk * ldi + pv[m]] = a[m];
        }
    }
}

template <int NMAX>
__device__ __forceinline__ void cg_wave_inverse_complex(const double* A, int N, int lda, double* Ainv, int ldi, double* scr) {
    // in-place Gauss-Jordan, as the real version: lane = 4 row + c holds the columns 4 m + c of its row
    constexpr int MQ = (NMAX + 3) / 4;
    const int lane = threadIdx.x & 63, r = lane >> 2, c = lane & 3;
    double ar[MQ], ai[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) {
        const int j = 4 * m + c;
        const bool ok = r < N && j < N;
        ar[m] = ok ? A[2 * (r * lda + j)] : 0.0;
        ai[m] = ok ? A[2 * (r * lda + j) + 1] : 0.0;
    }
    bool done = r >= N;
    unsigned long long donemask = N >= 16 ? 0ull : ~0ull << (4 * N);
    int myk = -1;
    int pv[MQ];                                         // pivot row of the lane's own columns j = 4 m + c
    double* mine = scr + 2 * c;                         // complex element j = 4 m + c at scr[2 j]
#pragma unroll
    for (int k = 0; k < 4 * MQ; ++k) {
        if ((k & 3) == 0) pv[k >> 2] = 0;
        if (k < N) {
            const int ck = k & 3, mk = k >> 2;
            double akr, aki;
            if (ck == 0) { akr = cg_dpp_f64<0x00>(ar[mk]); aki = cg_dpp_f64<0x00>(ai[mk]); }
            else if (ck == 1) { akr = cg_dpp_f64<0x55>(ar[mk]); aki = cg_dpp_f64<0x55>(ai[mk]); }
            else if (ck == 2) { akr = cg_dpp_f64<0xAA>(ar[mk]); aki = cg_dpp_f64<0xAA>(ai[mk]); }
            else { akr = cg_dpp_f64<0xFF>(ar[mk]); aki = cg_dpp_f64<0xFF>(ai[mk]); }
            const double m2 = akr * akr + aki * aki;
            int p = (int)__builtin_ctzll(~donemask);
            if (__ballot(!done && m2 * 0.0625 > cg_readlane_f64(m2, p))) {
                const unsigned key = done ? 0u : (unsigned)(__double_as_longlong(m2) >> 32) + 1u;
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx && !done);
                p = mask ? (int)__builtin_ctzll(mask) : p;
            }
            if (c == ck) pv[mk] = p >> 2;
            const bool isp = r == (p >> 2);
            const CgCplx piv = {cg_readlane_f64(akr, p), cg_readlane_f64(aki, p)};
            const double rd = cg_fast_rcp(piv.re * piv.re + piv.im * piv.im);
            const CgCplx rinv = {piv.re * rd, -piv.im * rd};
            CgCplx l = cmul({akr, aki}, rinv);
            if (isp) { l.re = 0.0; l.im = 0.0; }
            if (isp) {
#pragma unroll
                for (int m = 0; m < MQ; ++m) { mine[8 * m] = ar[m]; mine[8 * m + 1] = ai[m]; }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int m = 0; m < MQ; ++m) {
                const double pr = mine[8 * m], pi = mine[8 * m + 1];
                ar[m] = fma(-l.re, pr, fma(l.im, pi, ar[m]));
                ai[m] = fma(-l.re, pi, fma(-l.im, pr, ai[m]));
            }
            asm volatile("" ::: "memory");
            if (isp) {
#pragma unroll
                for (int m = 0; m < MQ; ++m) { const CgCplx v = cmul({ar[m], ai[m]}, rinv); ar[m] = v.re; ai[m] = v.im; }
                myk = k;
            }
            if (c == ck) { ar[mk] = isp ? rinv.re : -l.re; ai[mk] = isp ? rinv.im : -l.im; }
            done = done || isp;
            donemask |= 15ull << (p & ~3);
        }
    }
    if (myk >= 0) {
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            const int j = 4 * m + c;
            if (j < N) { Ainv[2 * (myk * ldi + pv[m])] = ar[m]; Ainv[2 * (myk * ldi + pv[m]) + 1] = ai[m]; }
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------------------------
// Workgroup-wide BLOCKED LU on a matrix in LDS (gfx950): sizes beyond the register LUs (N > 32; n = 29, 57, ...).
// Right-looking, panel width 4 = the K of v_mfma_f64_16x16x4_f64:
//   panel   (wave 0): partial pivoting on the panel's 4 columns (row swaps applied to whole rows), L21 scaled in place,
//                     U12 = L11^-1 A12 by forward substitution (one lane per column);
//   trailing (all waves): A22 -= L21 U12 as 16 x 16 MFMA tiles with K = 4 (one MFMA per tile, four for complex),
//                     tiles dealt round-robin to the waves; two workgroup barriers per panel.
// The single-wave LDS LU this replaces spent ~7 k cycles per column on LDS round trips of one wave while the other
// waves of the workgroup idled (68 % of a log Psi evaluation at n = 57).
// ------------------------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
typedef double cg_d4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void cg_wave_lds_fence_b() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// returns log|det A| in every thread; A (N x N, row-major, lda) is destroyed.  res: >= 2 doubles of LDS scratch.
__device__ __forceinline__ double cg_blocked_lu_logabsdet_lds(const CgBlk& b, double* A, int N, int lda, double* res) {
    const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
    const int col = lane & 15, kq = lane >> 4;
    CgScaledProd prod; prod.init();                       // meaningful in wave 0
    for (int k0 = 0; k0 < N; k0 += 4) {
        const int kb = N - k0 < 4 ? N - k0 : 4;
        if (wave == 0) {
            for (int j = 0; j < kb; ++j) {
                const int k = k0 + j;
                unsigned key = 0; int bi = k;
                for (int i = k + lane; i < N; i += 64) {
                    const unsigned kk = (unsigned)(__double_as_longlong(fabs(A[i * lda + k])) >> 32) + 1u;
                    if (kk > key) { key = kk; bi = i; }
                }
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx);
                const int p = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(mask));
                if (p != k)
                    for (int c = lane; c < N; c += 64) { const double t = A[k * lda + c]; A[k * lda + c] = A[p * lda + c]; A[p * lda + c] = t; }
                cg_wave_lds_fence_b();
                const double piv = A[k * lda + k];
                prod.mul(piv);
                const double rinv = cg_fast_rcp(piv);
                for (int i = k + 1 + lane; i < N; i += 64) {
                    const double l = A[i * lda + k] * rinv;
                    A[i * lda + k] = l;
                    for (int jj = j + 1; jj < kb; ++jj) A[i * lda + k0 + jj] = fma(-l, A[k * lda + k0 + jj], A[i * lda + k0 + jj]);
                }
                cg_wave_lds_fence_b();
            }
            for (int c = k0 + kb + lane; c < N; c += 64) {    // U12 = L11^-1 A12 (unit lower triangle)
                double u[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = r < kb ? A[(k0 + r) * lda + c] : 0.0;
#pragma unroll
                for (int r = 1; r < 4; ++r)
                    if (r < kb) {
#pragma unroll
                        for (int q = 0; q < r; ++q) u[r] = fma(-A[(k0 + r) * lda + k0 + q], u[q], u[r]);
                        A[(k0 + r) * lda + c] = u[r];
                    }
            }
        }
        b.sync();
        const int m0 = k0 + kb, M = N - m0;
        if (M > 0) {
            const int tiles = (M + 15) >> 4;
            for (int tt = wave; tt < tiles * tiles; tt += nw) {
                const int ti = tt / tiles, tj = tt - ti * tiles;
                const int r0 = m0 + 16 * ti, c0 = m0 + 16 * tj;
                const int ar = r0 + col, bc = c0 + col;
                const double av = (ar < N && kq < kb) ? -A[ar * lda + k0 + kq] : 0.0;
                const double bv = (bc < N && kq < kb) ? A[(k0 + kq) * lda + bc] : 0.0;
                cg_d4_t c;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int rr = r0 + kq + 4 * r; c[r] = (rr < N && bc < N) ? A[rr * lda + bc] : 0.0; }
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int rr = r0 + kq + 4 * r; if (rr < N && bc < N) A[rr * lda + bc] = c[r]; }
            }
        }
        b.sync();
    }
    if (b.tid == 0) res[0] = prod.logabs(true);
    b.sync();
    const double v = res[0];
    b.sync();
    return v;
}

// complex version: A interleaved (re,im), lda in complex elements.  Returns log|det| and arg(det) (with the sign of
// the row permutation) in every thread.  res: >= 2 doubles of LDS scratch.
__device__ __forceinline__ void cg_blocked_lu_logdet_complex_lds(const CgBlk& b, double* A, int N, int lda, double* res,
                                                                double& logabs, double& arg) {
    const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
    const int col = lane & 15, kq = lane >> 4;
    CgCplx pm = {1.0, 0.0}; int pe = 0;                   // meaningful in wave 0
    for (int k0 = 0; k0 < N; k0 += 4) {
        const int kb = N - k0 < 4 ? N - k0 : 4;
        if (wave == 0) {
            for (int j = 0; j < kb; ++j) {
                const int k = k0 + j;
                unsigned key = 0; int bi = k;
                for (int i = k + lane; i < N; i += 64) {
                    const double* a = A + 2 * (i * lda + k);
                    const unsigned kk = (unsigned)(__double_as_longlong(a[0] * a[0] + a[1] * a[1]) >> 32) + 1u;
                    if (kk > key) { key = kk; bi = i; }
                }
                const unsigned mx = cg_wave_max_u32(key);
                const unsigned long long mask = __ballot(key == mx);
                const int p = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(mask));
                if (p != k)
                    for (int c = lane; c < N; c += 64) {
                        double* x = A + 2 * (k * lda + c); double* y = A + 2 * (p * lda + c);
                        const double t0 = x[0], t1 = x[1]; x[0] = y[0]; x[1] = y[1]; y[0] = t0; y[1] = t1;
                    }
                cg_wave_lds_fence_b();
                const CgCplx piv = {A[2 * (k * lda + k)], A[2 * (k * lda + k) + 1]};
                pm = cmul(pm, piv);
                if (p != k) { pm.re = -pm.re; pm.im = -pm.im; }
                { int ex; const double mxv = fmax(fabs(pm.re), fabs(pm.im)); (void)frexp(mxv, &ex);
                  pm.re = ldexp(pm.re, -ex); pm.im = ldexp(pm.im, -ex); pe += ex; }
                const double rd = cg_fast_rcp(piv.re * piv.re + piv.im * piv.im);
                const CgCplx rinv = {piv.re * rd, -piv.im * rd};
                for (int i = k + 1 + lane; i < N; i += 64) {
                    double* aik = A + 2 * (i * lda + k);
                    const CgCplx l = cmul({aik[0], aik[1]}, rinv);
                    aik[0] = l.re; aik[1] = l.im;
                    for (int jj = j + 1; jj < kb; ++jj) {
                        double* x = A + 2 * (i * lda + k0 + jj); const double* y = A + 2 * (k * lda + k0 + jj);
                        const double re = x[0] - (l.re * y[0] - l.im * y[1]), im = x[1] - (l.re * y[1] + l.im * y[0]);
                        x[0] = re; x[1] = im;
                    }
                }
                cg_wave_lds_fence_b();
            }
            for (int c = k0 + kb + lane; c < N; c += 64) {    // U12 = L11^-1 A12
                CgCplx u[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) u[r] = r < kb ? CgCplx{A[2 * ((k0 + r) * lda + c)], A[2 * ((k0 + r) * lda + c) + 1]} : CgCplx{0.0, 0.0};
#pragma unroll
                for (int r = 1; r < 4; ++r)
                    if (r < kb) {
#pragma unroll
                        for (int q = 0; q < r; ++q) {
                            const CgCplx l = {A[2 * ((k0 + r) * lda + k0 + q)], A[2 * ((k0 + r) * lda + k0 + q) + 1]};
                            u[r] = csub(u[r], cmul(l, u[q]));
                        }
                        A[2 * ((k0 + r) * lda + c)] = u[r].re; A[2 * ((k0 + r) * lda + c) + 1] = u[r].im;
                    }
            }
        }
        b.sync();
        const int m0 = k0 + kb, M = N - m0;
        if (M > 0) {
            const int tiles = (M + 15) >> 4;
            for (int tt = wave; tt < tiles * tiles; tt += nw) {
                const int ti = tt / tiles, tj = tt - ti * tiles;
                const int r0 = m0 + 16 * ti, c0 = m0 + 16 * tj;
                const int ar = r0 + col, bc = c0 + col;
                const bool aok = ar < N && kq < kb, bok = bc < N && kq < kb;
                const double a_re = aok ? A[2 * (ar * lda + k0 + kq)] : 0.0, a_im = aok ? A[2 * (ar * lda + k0 + kq) + 1] : 0.0;
                const double b_re = bok ? A[2 * ((k0 + kq) * lda + bc)] : 0.0, b_im = bok ? A[2 * ((k0 + kq) * lda + bc) + 1] : 0.0;
                cg_d4_t cr, ci;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + kq + 4 * r; const bool ok = rr < N && bc < N;
                    cr[r] = ok ? A[2 * (rr * lda + bc)] : 0.0; ci[r] = ok ? A[2 * (rr * lda + bc) + 1] : 0.0;
                }
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_re, b_re, cr, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a_im, b_im, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_re, b_im, ci, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_im, b_re, ci, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + kq + 4 * r;
                    if (rr < N && bc < N) { A[2 * (rr * lda + bc)] = cr[r]; A[2 * (rr * lda + bc) + 1] = ci[r]; }
                }
            }
        }
        b.sync();
    }
    if (b.tid == 0) {
        res[0] = 0.5 * cg_log_ool(pm.re * pm.re + pm.im * pm.im) + (double)pe * 0.693147180559945309417232121458;
        res[1] = cg_atan2_ool(pm.im, pm.re);
    }
    b.sync();
    logabs = res[0]; arg = res[1];
    b.sync();
}

// ---- register-resident panels + look-ahead --------------------------------------------------------------------------
// Blocked right-looking LU with PW = 8 columns per panel.
//   panel   : factored by wave 0 IN REGISTERS (lane = row, up to two rows per lane for the real matrix: N <= 128; one
//             complex row per lane: N <= 64): pivot search by DPP max, the pivot row's entries broadcast by readlane, the row
//             exchange inside the panel done on registers; LDS is touched twice (load, store) instead of ~15 dependent
//             round trips per column.
//   deferred exchanges + U12 : the exchange of the REST of the two rows is deferred: the pivots go to LDS and the owner of a
//             column applies the exchanges to it and forward-substitutes its U12 entries (column-local); columns left of
//             the panel are never needed again (only the determinant is wanted).
//   trailing : A22 -= L21 U12 on 16 x 16 MFMA tiles with K = PW, two tiles per trip.
//   look-ahead: the sequential pivot chain of the panels is the critical path (one wave, ~1 k cycles per column).  While
//             the other waves apply panel k to the columns right of panel k+1 (each wave owns whole 16-column blocks:
//             exchanges + U12 for its columns, then their row tiles -- no synchronisation between the waves), wave 0
//             applies panel k to the columns of panel k+1 only and factors panel k+1.  ONE workgroup barrier per panel
//             (the LDS version: two per 4 columns with all other waves idle during the panel).
// res: >= 12 doubles of LDS (result + double-buffered pivots).
#define CG_LU_PW 8
struct CgLuPanelReal {
    // factor panel starting at column k0 (kb columns) -> L in place, pivots to pv; returns through `prod`
    // sk0 >= 0: the rows of this panel's columns have not yet received panel (sk0, skb) -- its U12 is in place (column()) --
    // and get A22 -= L21 U12 here, in the registers they are factored in (L21: 8 contiguous doubles of the lane's row, U12: 64
    // broadcast reads); the MFMA tile route cost an LDS round trip of the strip plus the latency of seven dependent trips.
    // FULL: kb == PW known at compile time (every panel but the last): no per-column width tests in the unrolled chain
    static __device__ __forceinline__ void factor(double* A, int N, int lda, int k0, int kb, int lane, int* pv, CgScaledProd& prod,
                                                  int sk0 = -1, int skb = 0) {
        double lrow[CG_LU_PW]; int piv[CG_LU_PW];
        factor_keep(A, N, lda, k0, kb, lane, pv, prod, sk0, skb, lrow, piv);
    }
    // lrow / piv: on return lane j < kb holds row k0 + j of the factored panel (L11 below its diagonal) and piv[j] the pivot
    // rows as wave-uniform values: column_reg() takes both from registers instead of 36 LDS reads on the pivot chain
    static __device__ __forceinline__ void factor_keep(double* A, int N, int lda, int k0, int kb, int lane, int* pv, CgScaledProd& prod,
                                                       int sk0, int skb, double (&lrow)[CG_LU_PW], int (&piv)[CG_LU_PW]) {
        if (kb == CG_LU_PW) factor_t<true>(A, N, lda, k0, CG_LU_PW, lane, pv, prod, sk0, skb, lrow, piv);
        else factor_t<false>(A, N, lda, k0, kb, lane, pv, prod, sk0, skb, lrow, piv);
    }
    template <bool FULL>
    static __device__ __forceinline__ void factor_t(double* A, int N, int lda, int k0, int kb, int lane, int* pv, CgScaledProd& prod,
                                                    int sk0, int skb, double (&lrow)[CG_LU_PW], int (&piv)[CG_LU_PW]) {
        constexpr int PW = CG_LU_PW;
        const int r0 = k0 + lane, r1 = r0 + 64;
        double a0[PW], a1[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            a0[j] = (r0 < N && (FULL || j < kb)) ? A[r0 * lda + k0 + j] : 0.0;
            a1[j] = (r1 < N && (FULL || j < kb)) ? A[r1 * lda + k0 + j] : 0.0;
        }
        CG_STAMP(21)
        if (sk0 >= 0) {
            const bool two = N > k0 + 64;                                 // any second row of a lane (wave-uniform)
            if (FULL && skb == PW && (lda & 1) == 0) {                // full panels: unconditional, vector LDS reads
                typedef double d2_t __attribute__((ext_vector_type(2)));
                double l0[PW], l1[PW];
                const d2_t* p0 = (const d2_t*)(A + (r0 < N ? r0 : N - 1) * lda + sk0);
                const d2_t* p1 = (const d2_t*)(A + (r1 < N ? r1 : N - 1) * lda + sk0);
#pragma unroll
                for (int q = 0; q < PW / 2; ++q) {
                    const d2_t v0 = p0[q]; l0[2 * q] = v0[0]; l0[2 * q + 1] = v0[1];
                    if (two) { const d2_t v1 = p1[q]; l1[2 * q] = v1[0]; l1[2 * q + 1] = v1[1]; } else { l1[2 * q] = 0.0; l1[2 * q + 1] = 0.0; }
                }
#pragma unroll
                for (int q = 0; q < PW; ++q) {
                    const d2_t* pu = (const d2_t*)(A + (sk0 + q) * lda + k0);   // wave-uniform address: LDS broadcast
#pragma unroll
                    for (int j = 0; j < PW / 2; ++j) {
                        const d2_t u = pu[j];
                        a0[2 * j] = fma(-l0[q], u[0], a0[2 * j]); a0[2 * j + 1] = fma(-l0[q], u[1], a0[2 * j + 1]);
                        if (two) { a1[2 * j] = fma(-l1[q], u[0], a1[2 * j]); a1[2 * j + 1] = fma(-l1[q], u[1], a1[2 * j + 1]); }
                    }
                }
            } else {
                double l0[PW], l1[PW];
#pragma unroll
                for (int q = 0; q < PW; ++q) {
                    l0[q] = (r0 < N && q < skb) ? A[r0 * lda + sk0 + q] : 0.0;
                    l1[q] = (r1 < N && q < skb) ? A[r1 * lda + sk0 + q] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < PW; ++q) {
                    if (q < skb) {
#pragma unroll
                        for (int j = 0; j < PW; ++j) {
                            const double u = j < kb ? A[(sk0 + q) * lda + k0 + j] : 0.0;
                            a0[j] = fma(-l0[q], u, a0[j]);
                            a1[j] = fma(-l1[q], u, a1[j]);
                        }
                    }
                }
            }
        }
        CG_STAMP(22)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (FULL || j < kb) {
                const int k = k0 + j;
                // threshold pivoting (as in the wave-level LUs): row k serves unless some candidate is more than 4x larger
                // (growth bounded by 5 per step); only then the full search and the row exchange run.  For J = I + small
                // the short path is the only one taken; it removes the DPP max chain from the sequential pivot chain.
                const double c0v = (r0 >= k && r0 < N) ? fabs(a0[j]) : 0.0, c1v = (r1 < N) ? fabs(a1[j]) : 0.0;
                const double akk = cg_readlane_f64(c0v, j);
                int p = k;
                if (__ballot(fmax(c0v, c1v) * 0.25 > akk)) {
                    const unsigned key0 = (r0 >= k && r0 < N) ? (unsigned)(__double_as_longlong(c0v) >> 32) + 1u : 0u;
                    const unsigned key1 = (r1 < N) ? (unsigned)(__double_as_longlong(c1v) >> 32) + 1u : 0u;
                    const unsigned key = key1 > key0 ? key1 : key0;
                    const int bi = key1 > key0 ? r1 : r0;
                    const unsigned mx = cg_wave_max_u32(key);
                    const unsigned long long mask = __ballot(key == mx);
                    p = __builtin_amdgcn_readlane(bi, (int)__builtin_ctzll(mask));
                }
                const int lp = (p - k0) & 63, sp = (p - k0) >> 6;         // wave-uniform
                double rp[PW];
                if (p != k) {                                             // exchange rows k and p inside the panel
#pragma unroll
                    for (int jj = 0; jj < PW; ++jj) {
                        const double rk = cg_readlane_f64(a0[jj], j), rq = cg_readlane_f64(sp ? a1[jj] : a0[jj], lp);
                        if (lane == j) a0[jj] = rq;
                        if (lane == lp) { if (sp) a1[jj] = rk; else a0[jj] = rk; }
                    }
                }
#pragma unroll
                for (int jj = j; jj < PW; ++jj) rp[jj] = cg_readlane_f64(a0[jj], j);   // the pivot row now sits in lane j
                if (lane == 0) pv[j] = p;
                piv[j] = p;
                const double pvt = rp[j];
                prod.mul(pvt);
                const double rinv = cg_fast_rcp1(pvt);                     // (the determinant uses the pivots themselves)
                if (r0 > k && r0 < N) {
                    const double l = a0[j] * rinv; a0[j] = l;
#pragma unroll
                    for (int jj = j + 1; jj < PW; ++jj) a0[jj] = fma(-l, rp[jj], a0[jj]);
                }
                if (r1 < N) {
                    const double l = a1[j] * rinv; a1[j] = l;
#pragma unroll
                    for (int jj = j + 1; jj < PW; ++jj) a1[jj] = fma(-l, rp[jj], a1[jj]);
                }
            }
        }
        CG_STAMP(23)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (FULL || j < kb) {
                if (r0 < N) A[r0 * lda + k0 + j] = a0[j];
                if (r1 < N) A[r1 * lda + k0 + j] = a1[j];
            } else piv[j] = k0 + j;
            lrow[j] = a0[j];
        }
        CG_STAMP_END(24)
    }
    // column() for the pivot chain: L11 and the pivots of the (full) panel come from the registers factor_keep() left them in
    static __device__ __forceinline__ void column_reg(double* A, int lda, int k0, const double (&lrow)[CG_LU_PW], const int (&piv)[CG_LU_PW],
                                                      int c, bool on) {
        constexpr int PW = CG_LU_PW;
        double u[PW];
#pragma unroll
        for (int r = 0; r < PW; ++r) u[r] = on ? A[(k0 + r) * lda + c] : 0.0;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int p = piv[j];                                          // wave-uniform: scalar branch, almost never taken
            if (p != k0 + j) {
                if (p < k0 + PW) {
#pragma unroll
                    for (int q = j + 1; q < PW; ++q) if (p - k0 == q) { const double t = u[j]; u[j] = u[q]; u[q] = t; }
                } else if (on) {
                    const double t = A[p * lda + c]; A[p * lda + c] = u[j]; u[j] = t;
                }
            }
        }
#pragma unroll
        for (int r = 1; r < PW; ++r)
#pragma unroll
            for (int q = 0; q < r; ++q) u[r] = fma(-cg_readlane_f64(lrow[q], r), u[q], u[r]);
        if (on) {
#pragma unroll
            for (int r = 0; r < PW; ++r) A[(k0 + r) * lda + c] = u[r];
        }
    }
    // deferred row exchanges of panel (k0, kb, pv) applied to column c, then U12[:, c] = L11^-1 A12[:, c]
    static __device__ __forceinline__ void column(double* A, int lda, int k0, int kb, const int* pv, int c) {
        constexpr int PW = CG_LU_PW;
        double u[PW];
#pragma unroll
        for (int r = 0; r < PW; ++r) u[r] = r < kb ? A[(k0 + r) * lda + c] : 0.0;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (j < kb) {
                const int p = pv[j];
                if (p != k0 + j) {
                    if (p < k0 + PW) {
#pragma unroll
                        for (int q = j + 1; q < PW; ++q) if (p - k0 == q) { const double t = u[j]; u[j] = u[q]; u[q] = t; }
                    } else {
                        const double t = A[p * lda + c]; A[p * lda + c] = u[j]; u[j] = t;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 1; r < PW; ++r)
            if (r < kb) {
#pragma unroll
                for (int q = 0; q < r; ++q) u[r] = fma(-A[(k0 + r) * lda + k0 + q], u[q], u[r]);
            }
#pragma unroll
        for (int r = 0; r < PW; ++r) if (r < kb) A[(k0 + r) * lda + c] = u[r];
    }
    // row tiles ti = 0.. of the column block [c0, cend) (cend - c0 <= 16): A22 -= L21 U12 for rows >= m0, two tiles per trip
    static __device__ __forceinline__ void tiles(double* A, int N, int lda, int k0, int kb, int m0, int c0, int cend, int lane) {
        constexpr int PW = CG_LU_PW;
        const int col = lane & 15, kq = lane >> 4;
        const int rtiles = (N - m0 + 15) >> 4;
        const int bc = c0 + col;
        const bool cok = bc < cend;
        double bv[PW / 4];
#pragma unroll
        for (int ks = 0; ks < PW / 4; ++ks) { const int kk = 4 * ks + kq; bv[ks] = (cok && kk < kb) ? A[(k0 + kk) * lda + bc] : 0.0; }
        for (int t0 = 0; t0 < rtiles; t0 += 2) {
            cg_d4_t c[2]; double av[2][PW / 4]; int rr0[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int r0 = m0 + 16 * (t0 + h);
                const int ar = r0 + col;
                rr0[h] = r0 + kq;
#pragma unroll
                for (int ks = 0; ks < PW / 4; ++ks) { const int kk = 4 * ks + kq; av[h][ks] = (ar < N && kk < kb) ? -A[ar * lda + k0 + kk] : 0.0; }
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int rr = rr0[h] + 4 * r; c[h][r] = (rr < N && cok) ? A[rr * lda + bc] : 0.0; }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < PW / 4; ++ks) c[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][ks], bv[ks], c[h], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int rr = rr0[h] + 4 * r; if (rr < N && cok) A[rr * lda + bc] = c[h][r]; }
        }
    }
};

__device__ __forceinline__ double cg_blocked_lu_logabsdet(const CgBlk& b, double* A, int N, int lda, double* res) {
    const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
    if (N > 128 || nw < 2) return cg_blocked_lu_logabsdet_lds(b, A, N, lda, res);
    constexpr int PW = CG_LU_PW;
    int* pivs = (int*)(res + 2);                          // [2][PW]
    const int npan = (N + PW - 1) / PW;
    CgScaledProd prod; prod.init();                       // meaningful in wave 0
    if (wave == 0) CgLuPanelReal::factor(A, N, lda, 0, N < PW ? N : PW, lane, pivs, prod);
    b.sync();
    for (int k = 0; k < npan; ++k) {
        const int k0 = k * PW, kb = N - k0 < PW ? N - k0 : PW, m0 = k0 + kb;
        const int nb = N - m0 < PW ? N - m0 : PW;          // width of panel k + 1 (0: none)
        const int* pv = pivs + (k & 1) * PW;
        if (wave == 0) {
            if (nb > 0) {                                  // panel k applied to the columns of panel k + 1, then panel k + 1
                CG_STAMP_START(16)
                if (lane < nb) CgLuPanelReal::column(A, lda, k0, kb, pv, m0 + lane);
                asm volatile("" ::: "memory");             // (LDS executes one wave's accesses in order)
                CG_STAMP(16)
                CG_STAMP(17)
                CgLuPanelReal::factor(A, N, lda, m0, nb, lane, pivs + ((k + 1) & 1) * PW, prod, k0, kb);   // strip update in registers
                CG_STAMP_END(18)
            }
        } else {
            const int cb0 = m0 + nb;                       // the other waves: whole 16-column blocks right of panel k + 1
            const int ctiles = (N - cb0 + 15) >> 4;
            CG_STAMP_START(19)
            for (int tj = wave - 1; tj < ctiles; tj += nw - 1) {
                const int c0 = cb0 + 16 * tj, cend = c0 + 16 < N ? c0 + 16 : N;
                if (lane < 16 && c0 + lane < N) CgLuPanelReal::column(A, lda, k0, kb, pv, c0 + lane);
                asm volatile("" ::: "memory");
                CgLuPanelReal::tiles(A, N, lda, k0, kb, m0, c0, cend, lane);
            }
            CG_STAMP_END(19)
        }
        b.sync();
    }
    if (b.tid == 0) res[0] = prod.logabs(true);
    b.sync();
    const double v = res[0];
    b.sync();
    return v;
}

struct CgLuPanelCplx {
    // sk0 >= 0: the rows of this panel's columns get A22 -= L21 U12 of panel (sk0, PW) here, in registers (see the real version)
    static __device__ __forceinline__ void factor(double* A, int N, int lda, int k0, int kb, int lane, int* pv, CgCplx& pm, int& pe, int sk0 = -1) {
        if (kb == CG_LU_PW) factor_t<true>(A, N, lda, k0, CG_LU_PW, lane, pv, pm, pe, sk0);
        else factor_t<false>(A, N, lda, k0, kb, lane, pv, pm, pe, sk0);
    }
    template <bool FULL>
    static __device__ __forceinline__ void factor_t(double* A, int N, int lda, int k0, int kb, int lane, int* pv, CgCplx& pm, int& pe, int sk0) {
        constexpr int PW = CG_LU_PW;
        const int r0 = k0 + lane;
        double ar_[PW], ai_[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const bool ok = r0 < N && (FULL || j < kb);
            ar_[j] = ok ? A[2 * (r0 * lda + k0 + j)] : 0.0;
            ai_[j] = ok ? A[2 * (r0 * lda + k0 + j) + 1] : 0.0;
        }
        CG_STAMP(26)
        if (sk0 >= 0) {                                                   // strip update by the full panel (sk0, PW)
            typedef double d2_t __attribute__((ext_vector_type(2)));
            const d2_t* pl = (const d2_t*)(A + 2 * ((r0 < N ? r0 : N - 1) * lda + sk0));
            d2_t l[PW];
#pragma unroll
            for (int q = 0; q < PW; ++q) l[q] = pl[q];
#pragma unroll
            for (int q = 0; q < PW; ++q) {
                const d2_t* pu = (const d2_t*)(A + 2 * ((sk0 + q) * lda + k0));   // wave-uniform address: LDS broadcast
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    if (FULL || j < kb) {
                        const d2_t u = pu[j];
                        ar_[j] = ar_[j] - (l[q][0] * u[0] - l[q][1] * u[1]);
                        ai_[j] = ai_[j] - (l[q][0] * u[1] + l[q][1] * u[0]);
                    }
                }
            }
        }
        CG_STAMP(27)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (FULL || j < kb) {
                const int k = k0 + j;
                // threshold pivoting: row k serves unless some candidate is more than 4x larger in modulus
                const double m2 = (r0 >= k && r0 < N) ? ar_[j] * ar_[j] + ai_[j] * ai_[j] : 0.0;
                const double mkk = cg_readlane_f64(m2, j);
                int lp = j;                                               // lane of the pivot row (one row per lane)
                if (__ballot(m2 * 0.0625 > mkk)) {
                    const unsigned key = (r0 >= k && r0 < N) ? (unsigned)(__double_as_longlong(m2) >> 32) + 1u : 0u;
                    const unsigned mx = cg_wave_max_u32(key);
                    const unsigned long long mask = __ballot(key == mx);
                    lp = (int)__builtin_ctzll(mask);
                }
                const int p = k0 + lp;
                double rpr[PW], rpi[PW];
                if (p != k) {
#pragma unroll
                    for (int jj = 0; jj < PW; ++jj) {
                        const double rkr = cg_readlane_f64(ar_[jj], j), rki = cg_readlane_f64(ai_[jj], j);
                        const double rqr = cg_readlane_f64(ar_[jj], lp), rqi = cg_readlane_f64(ai_[jj], lp);
                        if (lane == j) { ar_[jj] = rqr; ai_[jj] = rqi; }
                        if (lane == lp) { ar_[jj] = rkr; ai_[jj] = rki; }
                    }
                }
#pragma unroll
                for (int jj = j; jj < PW; ++jj) { rpr[jj] = cg_readlane_f64(ar_[jj], j); rpi[jj] = cg_readlane_f64(ai_[jj], j); }   // pivot row: lane j
                if (lane == 0) pv[j] = p;
                const CgCplx piv = {rpr[j], rpi[j]};
                pm = cmul(pm, piv);
                if (p != k) { pm.re = -pm.re; pm.im = -pm.im; }
                { int ex; const double mxv = fmax(fabs(pm.re), fabs(pm.im)); (void)frexp(mxv, &ex);
                  pm.re = ldexp(pm.re, -ex); pm.im = ldexp(pm.im, -ex); pe += ex; }
                const double rd = cg_fast_rcp1(piv.re * piv.re + piv.im * piv.im);
                const CgCplx rinv = {piv.re * rd, -piv.im * rd};
                if (r0 > k && r0 < N) {
                    const CgCplx l = cmul({ar_[j], ai_[j]}, rinv);
                    ar_[j] = l.re; ai_[j] = l.im;
#pragma unroll
                    for (int jj = j + 1; jj < PW; ++jj) {
                        const double re = ar_[jj] - (l.re * rpr[jj] - l.im * rpi[jj]), im = ai_[jj] - (l.re * rpi[jj] + l.im * rpr[jj]);
                        ar_[jj] = re; ai_[jj] = im;
                    }
                }
            }
        }
        CG_STAMP(28)
#pragma unroll
        for (int j = 0; j < PW; ++j)
            if ((FULL || j < kb) && r0 < N) { A[2 * (r0 * lda + k0 + j)] = ar_[j]; A[2 * (r0 * lda + k0 + j) + 1] = ai_[j]; }
        CG_STAMP_END(29)
    }
    static __device__ __forceinline__ void column(double* A, int lda, int k0, int kb, const int* pv, int c) {
        constexpr int PW = CG_LU_PW;
        CgCplx u[PW];
#pragma unroll
        for (int r = 0; r < PW; ++r) u[r] = r < kb ? CgCplx{A[2 * ((k0 + r) * lda + c)], A[2 * ((k0 + r) * lda + c) + 1]} : CgCplx{0.0, 0.0};
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (j < kb) {
                const int p = pv[j];
                if (p != k0 + j) {
                    if (p < k0 + PW) {
#pragma unroll
                        for (int q = j + 1; q < PW; ++q) if (p - k0 == q) { const CgCplx t = u[j]; u[j] = u[q]; u[q] = t; }
                    } else {
                        double* y = A + 2 * (p * lda + c);
                        const CgCplx t = {y[0], y[1]}; y[0] = u[j].re; y[1] = u[j].im; u[j] = t;
                    }
                }
            }
        }
#pragma unroll
        for (int r = 1; r < PW; ++r)
            if (r < kb) {
#pragma unroll
                for (int q = 0; q < r; ++q) {
                    const CgCplx l = {A[2 * ((k0 + r) * lda + k0 + q)], A[2 * ((k0 + r) * lda + k0 + q) + 1]};
                    u[r] = csub(u[r], cmul(l, u[q]));
                }
            }
#pragma unroll
        for (int r = 0; r < PW; ++r) if (r < kb) { A[2 * ((k0 + r) * lda + c)] = u[r].re; A[2 * ((k0 + r) * lda + c) + 1] = u[r].im; }
    }
    static __device__ __forceinline__ void tiles(double* A, int N, int lda, int k0, int kb, int m0, int c0, int cend, int lane) {
        constexpr int PW = CG_LU_PW;
        const int col = lane & 15, kq = lane >> 4;
        const int rtiles = (N - m0 + 15) >> 4;
        const int bc = c0 + col;
        const bool cok = bc < cend;
        double b_re[PW / 4], b_im[PW / 4];
#pragma unroll
        for (int ks = 0; ks < PW / 4; ++ks) {
            const int kk = 4 * ks + kq; const bool bok = cok && kk < kb;
            b_re[ks] = bok ? A[2 * ((k0 + kk) * lda + bc)] : 0.0; b_im[ks] = bok ? A[2 * ((k0 + kk) * lda + bc) + 1] : 0.0;
        }
        for (int ti = 0; ti < rtiles; ++ti) {
            const int r0 = m0 + 16 * ti, ar = r0 + col;
            double a_re[PW / 4], a_im[PW / 4];
#pragma unroll
            for (int ks = 0; ks < PW / 4; ++ks) {
                const int kk = 4 * ks + kq; const bool aok = ar < N && kk < kb;
                a_re[ks] = aok ? A[2 * (ar * lda + k0 + kk)] : 0.0; a_im[ks] = aok ? A[2 * (ar * lda + k0 + kk) + 1] : 0.0;
            }
            cg_d4_t cr, ci;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + kq + 4 * r; const bool ok = rr < N && cok;
                cr[r] = ok ? A[2 * (rr * lda + bc)] : 0.0; ci[r] = ok ? A[2 * (rr * lda + bc) + 1] : 0.0;
            }
#pragma unroll
            for (int ks = 0; ks < PW / 4; ++ks) {
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_re[ks], b_re[ks], cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_re[ks], b_im[ks], ci, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a_im[ks], b_im[ks], cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_im[ks], b_re[ks], ci, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = r0 + kq + 4 * r;
                if (rr < N && cok) { A[2 * (rr * lda + bc)] = cr[r]; A[2 * (rr * lda + bc) + 1] = ci[r]; }
            }
        }
    }
};

__device__ __forceinline__ void cg_blocked_lu_logdet_complex(const CgBlk& b, double* A, int N, int lda, double* res,
                                                            double& logabs, double& arg) {
    const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
    if (N > 64 || nw < 2) { cg_blocked_lu_logdet_complex_lds(b, A, N, lda, res, logabs, arg); return; }
    constexpr int PW = CG_LU_PW;
    int* pivs = (int*)(res + 2);                          // [2][PW]
    const int npan = (N + PW - 1) / PW;
    CgCplx pm = {1.0, 0.0}; int pe = 0;                   // meaningful in wave 0
    if (wave == 0) CgLuPanelCplx::factor(A, N, lda, 0, N < PW ? N : PW, lane, pivs, pm, pe);
    b.sync();
    for (int k = 0; k < npan; ++k) {
        const int k0 = k * PW, kb = N - k0 < PW ? N - k0 : PW, m0 = k0 + kb;
        const int nb = N - m0 < PW ? N - m0 : PW;
        const int* pv = pivs + (k & 1) * PW;
        if (wave == 0) {
            if (nb > 0) {
                if (lane < nb) CgLuPanelCplx::column(A, lda, k0, kb, pv, m0 + lane);
                asm volatile("" ::: "memory");
                CgLuPanelCplx::tiles(A, N, lda, k0, kb, m0, m0, m0 + nb, lane);
                asm volatile("" ::: "memory");
                CgLuPanelCplx::factor(A, N, lda, m0, nb, lane, pivs + ((k + 1) & 1) * PW, pm, pe);
            }
        } else {
            const int cb0 = m0 + nb;
            const int ctiles = (N - cb0 + 15) >> 4;
            for (int tj = wave - 1; tj < ctiles; tj += nw - 1) {
                const int c0 = cb0 + 16 * tj, cend = c0 + 16 < N ? c0 + 16 : N;
                if (lane < 16 && c0 + lane < N) CgLuPanelCplx::column(A, lda, k0, kb, pv, c0 + lane);
                asm volatile("" ::: "memory");
                CgLuPanelCplx::tiles(A, N, lda, k0, kb, m0, c0, cend, lane);
            }
        }
        b.sync();
    }
    if (b.tid == 0) {
        res[0] = 0.5 * cg_log_ool(pm.re * pm.re + pm.im * pm.im) + (double)pe * 0.693147180559945309417232121458;
        res[1] = cg_atan2_ool(pm.im, pm.re);
    }
    b.sync();
    logabs = res[0]; arg = res[1];
    b.sync();
}
// ---- both determinants of log Psi at once (large n) --------------------------------------------------------------------
// log|det J| (real N x N) and log det D (complex n x n) are independent, and each blocked LU is bound by the sequential pivot
// chain of its panel wave while the other waves wait.  With both matrices in LDS (CgFastLds::dual) the two chains run
// CONCURRENTLY and DECOUPLED from the trailing updates:
//   wave 0 : the real panels.    Step k: panel k applied to the columns of panel k + 1 (U12 + the strip in registers), panel
//            k + 1 factored, its L and pivots published (flag pub_r = k + 2).
//   wave 1 : the complex panels, likewise (pub_c).
//   waves 2.. : helpers.  A task = one published panel applied to one absolute 16-column block of one matrix (deferred
//            exchanges + U12 per column, then the MFMA row tiles); per block the panels go in order (claim[j] = next panel,
//            taken by compare-and-swap; app[j] = panels applied).  A chain wave waits only for the block that holds its next
//            panel (app[(k + 1) >> 1] >= k); a free helper takes that urgent task first, whoever applied the previous panel.
//   No workgroup barrier inside: flags in LDS with release / acquire at workgroup scope.  Every spin is bounded (a broken
//   dependency would give wrong numbers, never a hung GPU).  The arithmetic on each matrix is that of the sequential drivers
//   in the same order: bitwise identical results.
// res: CG_LU_DUAL_DOUBLES doubles of LDS.
#define CG_LU_DUAL_DOUBLES 164
__device__ __forceinline__ int cg_flag_load(const int* f) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void cg_flag_store(int* f, int v, int lane) {
    if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// publication by the wave that wrote the data: LDS executes one wave's accesses in order, so a plain store issued after the data
// stores becomes visible after them -- no s_waitcnt before it (the release store above drains the wave's LDS queue first)
__device__ __forceinline__ void cg_flag_post(int* f, int v, int lane) {
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void cg_flag_wait(const int* f, int need) {
    for (int spin = 0; spin < (1 << 22) && cg_flag_load(f) < need; ++spin) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void cg_blocked_lu_dual(const CgBlk& b, double* A, int N, int lda, double* C, int n, int ldc, double* res,
                                                   double& logabs_real, double& logabs_c, double& arg_c) {
    const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
    constexpr int PW = CG_LU_PW;
    int* fl = (int*)(res + 4);
    int* pub_r = fl;       int* pub_c = fl + 1;           // panels published
    int* claim_r = fl + 2;                                // [8] real + [4] complex: next panel to apply to absolute block j
    int* app_r = fl + 14;  int* app_c = fl + 22;          // [8], [4]: panels applied to absolute block j
    int* pivr = fl + 26;   int* pivc = fl + 26 + 128;     // pivots of every panel (helpers may lag several panels behind)
    const int npr = (N + PW - 1) / PW, npc = (n + PW - 1) / PW;
    for (int e = b.tid; e < 26; e += b.nthr) fl[e] = 0;
    b.sync();
    if (wave == 0) {
        CG_STAMP_START(16)
        CgScaledProd prod; prod.init();
        double lrow[PW]; int piv[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) { lrow[j] = 0.0; piv[j] = 0; }
        for (int k = -1; k + 1 < npr; ++k) {               // k = -1: panel 0 itself (one call site of the panel code)
            const int k0 = k * PW, m0 = k0 + PW, nb = N - m0 < PW ? N - m0 : PW;
            if (k > 0) {
                CG_STAMP_START(18)
                cg_flag_wait(app_r + ((k + 1) >> 1), k);
                CG_STAMP_END(18)
            }
            CG_STAMP_START(20)
            if (k >= 0) CgLuPanelReal::column_reg(A, lda, k0, lrow, piv, m0 + lane, lane < nb);
            asm volatile("" ::: "memory");                 // (LDS executes one wave's accesses in order)
            CG_STAMP(20)
            CgLuPanelReal::factor_keep(A, N, lda, m0, nb, lane, pivr + m0, prod, k >= 0 ? k0 : -1, PW, lrow, piv);
            cg_flag_store(pub_r, k + 2, lane);
        }
        if (lane == 0) res[0] = prod.logabs(true);
        CG_STAMP_END(16)
    } else if (wave == 1) {
        CG_STAMP_START(17)
        CgCplx pm = {1.0, 0.0}; int pe = 0;
        for (int k = -1; k + 1 < npc; ++k) {
            const int k0 = k * PW, m0 = k0 + PW, nb = n - m0 < PW ? n - m0 : PW;
            CG_STAMP_START(30)
            if (k > 0) cg_flag_wait(app_c + ((k + 1) >> 1), k);
            CG_STAMP_END(30)
            CG_STAMP_START(25)
            if (k >= 0 && lane < nb) CgLuPanelCplx::column(C, ldc, k0, PW, pivc + k0, m0 + lane);
            asm volatile("" ::: "memory");
            CG_STAMP(25)
            CgLuPanelCplx::factor(C, n, ldc, m0, nb, lane, pivc + m0, pm, pe, k >= 0 ? k0 : -1);
            cg_flag_store(pub_c, k + 2, lane);
        }
        if (lane == 0) {
            res[1] = 0.5 * cg_log_ool(pm.re * pm.re + pm.im * pm.im) + (double)pe * 0.693147180559945309417232121458;
            res[2] = cg_atan2_ool(pm.im, pm.re);
        }
        CG_STAMP_END(17)
    } else {
        // helpers: lane l < 8 watches real block l, lanes 8..11 complex block l - 8; one ballot picks the task
        const int nbr = (N + 15) >> 4, nbc = (n + 15) >> 4;
        const bool isr = lane < 8, mine = isr ? lane < nbr : (lane < 12 && lane - 8 < nbc);
        const int j = isr ? lane : lane - 8;
        const int limit = !mine ? 0 : (isr ? (2 * j < npr - 1 ? 2 * j : npr - 1) : (2 * j < npc - 1 ? 2 * j : npc - 1));   // panels this block receives
        int idle = 0;
        while (idle < (1 << 22)) {
            const int pr = cg_flag_load(pub_r), pc = cg_flag_load(pub_c);
            int q = 0, ap = 0;
            if (lane < 12) {
                q = __hip_atomic_load(claim_r + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);       // claim_c follows claim_r
                ap = __hip_atomic_load(app_r + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);        // app_c follows app_r
            }
            const bool pending = mine && q < limit;
            if (!__ballot(pending)) break;                               // every task of both matrices has been claimed
            const bool ready = pending && (isr ? pr : pc) > q && ap >= q;
            const unsigned long long mu = __ballot(ready && j == ((q + 2) >> 1)), me = __ballot(ready);
            if (!me) { ++idle; __builtin_amdgcn_s_sleep(1); continue; }
            const int pick = (int)__builtin_ctzll(mu ? mu : me);          // urgent (the block a chain waits for) first; real before complex
            const int tq = __builtin_amdgcn_readlane(q, pick);
            int got = 0;
            if (lane == 0) got = atomicCAS(claim_r + pick, tq, tq + 1) == tq ? 1 : 0;
            if (!__builtin_amdgcn_readfirstlane(got)) continue;          // another helper took it
            CG_STAMP_START(19)
            const int k0 = tq * PW, m0 = k0 + PW;
            if (pick < 8) {
                const int c0 = 16 * pick > m0 + PW ? 16 * pick : m0 + PW, cend = 16 * pick + 16 < N ? 16 * pick + 16 : N;
                if (c0 < cend) {
                    if (lane < cend - c0) CgLuPanelReal::column(A, lda, k0, PW, pivr + k0, c0 + lane);
                    asm volatile("" ::: "memory");
                    CgLuPanelReal::tiles(A, N, lda, k0, PW, m0, c0, cend, lane);
                }
            } else {
                const int jb = pick - 8;
                const int c0 = 16 * jb > m0 + PW ? 16 * jb : m0 + PW, cend = 16 * jb + 16 < n ? 16 * jb + 16 : n;
                if (c0 < cend) {
                    if (lane < cend - c0) CgLuPanelCplx::column(C, ldc, k0, PW, pivc + k0, c0 + lane);
                    asm volatile("" ::: "memory");
                    CgLuPanelCplx::tiles(C, n, ldc, k0, PW, m0, c0, cend, lane);
                }
            }
            cg_flag_store(app_r + pick, tq + 1, lane);
            CG_STAMP_END(19)
        }
    }
    b.sync();
    logabs_real = res[0]; logabs_c = res[1]; arg_c = res[2];
    b.sync();
}

// ---- second generation of the concurrent pair (round 4): rows never move, helpers are pure GEMMs -----------------------------
// What bounded cg_blocked_lu_dual was its two chain waves (profiles/r04a_lu_bench_*): per 8-column panel the real chain spent
// 3.2 k cycles in the column steps and 7.6 k around them (U12 of its next columns by a readlane triangular solve, the strip
// update out of LDS, the L panel stored and re-read in the shifted lane <-> row map, waiting for a helper), the helpers spent
// half of every task in the 28-term triangular solve of their U12 block on 16 lanes.  Here
//   * lane <-> row is FIXED (lane l owns rows l and l + 64) and rows are never exchanged: the pivot of a column is the natural
//     row (row k if unused, else the first unused one) unless some unused row is more than 4x larger, then the largest (the
//     threshold rule of the first generation); `used` rows keep multiplier 0 and drop out by themselves.  No exchange code,
//     no deferred exchanges in the helpers, the multipliers stay in the registers they were formed in.
//   * while a panel is factored the chain also carries E = -L21 L11^-1 (the column steps replayed on the identity: 28 terms,
//     independent of the pivot chain, they fill its latency slots).  With E the Schur update needs no U12 at all:
//         A22 - L21 (L11^-1 A12) = A22 + E A12[pivot rows, raw]
//     so (1) the chain updates its NEXT eight columns in registers as a plain rank-8 update with broadcast pivot-row entries
//     (64 independent terms per row instead of a triangular solve + strip), (2) E is what is published, and a helper task is
//     a pure MFMA GEMM whose B operand is gathered by pivot index -- no triangular solve, no exchanges.
//   * a row tile whose 16 rows have all served as pivots is skipped by the helpers (`live` mask per panel).
// Flags / task scheme as in the first generation.  res: CG_LU_DUAL_DOUBLES doubles of LDS, 16-byte aligned.
#if !defined(CG_LU_TRACE)
#define CG_LU_TRACE(chain, k)          /* tools/lu_bench: per-panel time stamps of the chain waves */
#endif
struct CgLu2 {
    static constexpr int PW = CG_LU_PW;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef int i4_t __attribute__((ext_vector_type(4)));
    // int offsets inside the flag area
    static constexpr int O_PUB_R = 0, O_PUB_C = 1, O_CLAIM = 2, O_APP = 14;
    static constexpr int REC = 12, O_REC_R = 32, O_REC_C = 32 + 16 * REC;   // per panel {pivot rows[8], live row tiles, -, -, -}: 16 real + 8 complex panels

    // per-lane select by a wave-uniform lane mask (the mask IS the condition register of the select)
    static __device__ __forceinline__ int seli(unsigned long long mask, int v) {
        int r; asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(mask)); return r;
    }
    static __device__ __forceinline__ double sel(unsigned long long mask, double v) {
        const unsigned long long u = __double_as_longlong(v);
        const int lo = seli(mask, (int)(unsigned)u), hi = seli(mask, (int)(unsigned)(u >> 32));
        return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
    static __device__ __forceinline__ int hi_abs(double v) { return (int)((unsigned)((unsigned long long)__double_as_longlong(v) >> 32) & 0x7fffffffu); }
    // row tiles (16 rows) that still hold a live row: bit t of the result (l0: rows 0..63, l1: rows 64..127)
    static __device__ __forceinline__ int live_tiles(unsigned long long l0, unsigned long long l1) {
        auto four = [](unsigned long long l) {
            l |= l >> 8; l |= l >> 4; l |= l >> 2; l |= l >> 1;             // bit 0 of every 16-bit group = OR of the group
            l &= 0x0001000100010001ull;
            return (int)(((l * 0x0001000200040008ull) >> 48) & 0xfull);     // bits 0, 16, 32, 48 gathered into bits 0..3 (group g lands on bit 48 + g)
        };
        return four(l0) | (four(l1) << 4);
    }

    // ---------------------------------------------------------------- real chain (one wave) ----------------------------------
    // S = 1: N <= 64 (one row per lane), S = 2: N <= 128 (register slot s of lane l holds row l + 64 (s ^ swapped_l): a lane's
    // two rows trade slots when the pivot turns up in the slot the panel code does not read -- rare, and lane-local).
    // Rows that have served as pivot are not masked in the arithmetic: they compute garbage that nothing reads (the magnitude
    // keys, E at publication and the pivot broadcasts only ever look at live rows).
    template <int S>
    struct Real {
        double p[S][PW], e[S][PW];
        unsigned long long live[2], swm;                                   // wave-uniform: live register positions per slot; lanes whose rows are swapped
        int prow[PW], pl[PW];                                              // pivot rows / lanes of the current panel (wave-uniform)
        double pm;
    };
    // column steps of one panel: pivots in slot SP (BOTH: the other slot takes part as well)
    template <int S, int SP, bool BOTH, bool FULL>
    static __device__ __forceinline__ void steps_real(Real<S>& st, int k0, int kb, int lane, CgScaledProd& prod) {
        constexpr int S0 = BOTH ? 0 : SP, S1 = BOTH ? S : SP + 1;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (FULL || j < kb) {
                // natural row of the column in its natural place, alive, and no live entry more than ~4x larger (high words)?
                int cl = (k0 + j) & 63;
                int key[S];
#pragma unroll
                for (int s = S0; s < S1; ++s) key[s] = seli(st.live[s], hi_abs(st.p[s][j]));
                const int kmx = (BOTH && S == 2) ? max(key[0], key[S - 1]) : key[SP];
                const bool nat = ((st.live[SP] & ~st.swm) >> cl) & 1ull;
                const int akk = __builtin_amdgcn_readlane(key[SP], cl);
                if (!nat || __ballot((unsigned)kmx > (unsigned)akk + 0x00200000u)) {
                    // rare: the largest live entry; a pivot in the other slot is brought over by a lane-local exchange
                    unsigned kk = ((st.live[SP] >> lane) & 1ull) ? (unsigned)hi_abs(st.p[SP][j]) + 1u : 0u;
                    int sl = SP;
                    if (BOTH && S == 2) {
                        const unsigned ko = ((st.live[1 - SP] >> lane) & 1ull) ? (unsigned)hi_abs(st.p[1 - SP][j]) + 1u : 0u;
                        if (ko > kk) { kk = ko; sl = 1 - SP; }
                    }
                    const unsigned mx = cg_wave_max_u32(kk);
                    cl = (int)__builtin_ctzll(__ballot(kk == mx));
                    if (BOTH && S == 2) {
                        if (__builtin_amdgcn_readlane(sl, cl) != SP) {
                            const bool me = lane == cl;
#pragma unroll
                            for (int c = 0; c < PW; ++c) {
                                const double t0 = st.p[0][c], t1 = st.p[S - 1][c], u0 = st.e[0][c], u1 = st.e[S - 1][c];
                                st.p[0][c] = me ? t1 : t0; st.p[S - 1][c] = me ? t0 : t1;
                                st.e[0][c] = me ? u1 : u0; st.e[S - 1][c] = me ? u0 : u1;
                            }
                            const unsigned long long bit = 1ull << cl, b0 = st.live[0] & bit, b1 = st.live[1] & bit;
                            st.live[0] = (st.live[0] & ~bit) | b1; st.live[1] = (st.live[1] & ~bit) | b0;
                            st.swm ^= bit;
                        }
                    }
                }
                st.pl[j] = cl;
                st.prow[j] = S == 2 ? cl + 64 * (SP ^ (int)((st.swm >> cl) & 1ull)) : cl;
                st.live[SP] &= ~(1ull << cl);
                double rp[PW];
#pragma unroll
                for (int jj = j; jj < PW; ++jj) rp[jj] = cg_readlane_f64(st.p[SP][jj], cl);
                st.pm *= rp[j];
                if ((j & 3) == 3 || (!FULL && j + 1 == kb)) { prod.mul(st.pm); st.pm = 1.0; }
                const double rinv = cg_fast_rcp1(rp[j]);
                double u[PW];
#pragma unroll
                for (int c = 0; c < j; ++c) u[c] = cg_readlane_f64(st.e[SP][c], cl);
#pragma unroll
                for (int s = S0; s < S1; ++s) {
                    const double l = st.p[s][j] * rinv;
                    st.p[s][j] = l;
#pragma unroll
                    for (int jj = j + 1; jj < PW; ++jj) st.p[s][jj] = fma(-l, rp[jj], st.p[s][jj]);
                    // E = -L21 L11^-1: the same row operations on the identity columns (off the pivot chain)
#pragma unroll
                    for (int c = 0; c < j; ++c) st.e[s][c] = fma(-l, u[c], st.e[s][c]);
                    st.e[s][j] = -l;
                }
            } else {
                st.pl[j] = 0; st.prow[j] = 0;
#pragma unroll
                for (int s = S0; s < S1; ++s) st.e[s][j] = 0.0;
            }
        }
    }
    // E published, the next eight columns fetched (once every older panel has reached their block) and updated in registers
    template <int S, int SP, bool BOTH>
    static __device__ __forceinline__ void next_real(Real<S>& st, double* A, int N, int lda, int k, int lane, int* fl) {
        constexpr int S0 = BOTH ? 0 : SP, S1 = BOTH ? S : SP + 1;
        const int k0 = k * PW, m0 = k0 + PW;
        int row[S];
#pragma unroll
        for (int s = 0; s < S; ++s) row[s] = S == 2 ? lane + 64 * (s ^ (int)((st.swm >> lane) & 1ull)) : lane;
#pragma unroll
        for (int s = S0; s < S1; ++s) {
#pragma unroll
            for (int c = 0; c < PW; ++c) st.e[s][c] = sel(st.live[s], st.e[s][c]);
            if (row[s] < N) {
                d2_t* q = (d2_t*)(A + row[s] * lda + k0);
#pragma unroll
                for (int c = 0; c < PW / 2; ++c) q[c] = d2_t{st.e[s][2 * c], st.e[s][2 * c + 1]};
            }
        }
        if (!BOTH && S == 2) {                                             // the slot that takes no part holds used rows only: E = 0 there
            if (row[1 - SP] < N) {
                d2_t* q = (d2_t*)(A + row[1 - SP] * lda + k0);
#pragma unroll
                for (int c = 0; c < PW / 2; ++c) q[c] = d2_t{0.0, 0.0};
            }
        }
        if (lane == 0) {
            i4_t* q = (i4_t*)(fl + O_REC_R + REC * k);
            const unsigned long long l0 = (st.live[0] & ~st.swm) | (st.live[1] & st.swm), l1 = (st.live[1] & ~st.swm) | (st.live[0] & st.swm);
            q[0] = i4_t{st.prow[0], st.prow[1], st.prow[2], st.prow[3]};
            q[1] = i4_t{st.prow[4], st.prow[5], st.prow[6], st.prow[7]};
            q[2] = i4_t{live_tiles(l0, l1), 0, 0, 0};
        }
        cg_flag_post(fl + O_PUB_R, k + 1, lane);
        CG_STAMP(21)
        if (k > 0) cg_flag_wait(fl + O_APP + ((k + 1) >> 1), k);
        CG_STAMP(22)
        double b[S][PW];
#pragma unroll
        for (int s = S0; s < S1; ++s) {
            const d2_t* q = (const d2_t*)(A + (row[s] < N ? row[s] : N - 1) * lda + m0);
#pragma unroll
            for (int c = 0; c < PW / 2; ++c) { const d2_t v = q[c]; b[s][2 * c] = v[0]; b[s][2 * c + 1] = v[1]; }   // (columns >= N of the last panel: never used)
        }
        // p = b + E A12, A12 = the pivot rows' entries of the new columns as they stand in LDS (wave-uniform addresses: broadcast reads);
        // the reads of pivot row j + 1 are issued before the products of row j
        d2_t u[2][PW / 2];
        {
            const d2_t* q = (const d2_t*)(A + st.prow[0] * lda + m0);
#pragma unroll
            for (int c = 0; c < PW / 2; ++c) u[0][c] = q[c];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (j + 1 < PW) {
                const d2_t* q = (const d2_t*)(A + st.prow[j + 1] * lda + m0);
#pragma unroll
                for (int c = 0; c < PW / 2; ++c) u[(j + 1) & 1][c] = q[c];
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int s = S0; s < S1; ++s)
#pragma unroll
                for (int c = 0; c < PW; ++c) st.p[s][c] = fma(st.e[s][j], u[j & 1][c >> 1][c & 1], j == 0 ? b[s][c] : st.p[s][c]);
        }
    }
    template <int S, int SP, bool BOTH>
    static __device__ __forceinline__ bool panel_real(Real<S>& st, double* A, int N, int lda, int k, int npan, int lane, int* fl, CgScaledProd& prod) {
        const int k0 = k * PW;
        CG_STAMP_START(20)
        if (k + 1 == npan) {
            steps_real<S, SP, BOTH, false>(st, k0, N - k0, lane, prod);
            CG_STAMP_END(20)
            return true;
        }
        steps_real<S, SP, BOTH, true>(st, k0, PW, lane, prod);
        CG_STAMP(20)
        next_real<S, SP, BOTH>(st, A, N, lda, k, lane, fl);
        CG_STAMP_END(23)
        return false;
    }
    template <int S>
    static __device__ __forceinline__ void chain_real(double* A, int N, int lda, int lane, int* fl, double* res) {
        const int npan = (N + PW - 1) / PW;
        Real<S> st;
        st.live[0] = N >= 64 ? ~0ull : ((1ull << N) - 1ull);
        st.live[1] = (S == 2 && N > 64) ? (N >= 128 ? ~0ull : ((1ull << (N - 64)) - 1ull)) : 0ull;
        st.swm = 0ull; st.pm = 1.0;
        CgScaledProd prod; prod.init();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int row = lane + 64 * s;
            const d2_t* q = (const d2_t*)(A + (row < N ? row : N - 1) * lda);
#pragma unroll
            for (int c = 0; c < PW / 2; ++c) { const d2_t v = q[c]; st.p[s][2 * c] = v[0]; st.p[s][2 * c + 1] = v[1]; }
#pragma unroll
            for (int c = 0; c < PW; ++c) st.e[s][c] = 0.0;
        }
        for (int k = 0; k < npan; ++k) {
            bool done;
            CG_LU_TRACE(0, k)
            if (S == 1) done = panel_real<S, 0, false>(st, A, N, lda, k, npan, lane, fl, prod);
            else {
                int sp = (k * PW) >> 6;
                if (!(sp ? st.live[S - 1] : st.live[0])) sp ^= 1;
                const bool both = (sp ? st.live[0] : st.live[S - 1]) != 0ull;
                if (sp == 0) done = both ? panel_real<S, 0, true>(st, A, N, lda, k, npan, lane, fl, prod) : panel_real<S, 0, false>(st, A, N, lda, k, npan, lane, fl, prod);
                else done = both ? panel_real<S, S - 1, true>(st, A, N, lda, k, npan, lane, fl, prod) : panel_real<S, S - 1, false>(st, A, N, lda, k, npan, lane, fl, prod);
            }
            if (done) break;
        }
        if (lane == 0) res[0] = prod.logabs(true);
    }

    // ---------------------------------------------------------------- complex chain (one wave, n <= 64) ----------------------
    struct Cplx {
        double pr[PW], pi[PW], er[PW], ei[PW];
        unsigned long long live;
        int pl[PW];
        CgCplx pm; int pe, parity;
    };
    template <bool FULL>
    static __device__ __forceinline__ void steps_cplx(Cplx& st, int k0, int kb, int lane) {
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (FULL || j < kb) {
                int cl = k0 + j;
                const int key = seli(st.live, hi_abs(st.pr[j] * st.pr[j] + st.pi[j] * st.pi[j]));
                const int mkk = __builtin_amdgcn_readlane(key, cl);
                // (squared moduli: 16x = 4x in modulus = four binades of the high word)
                if (!((st.live >> cl) & 1ull) || __ballot((unsigned)key > (unsigned)mkk + 0x00400000u)) {
                    const unsigned kk = ((st.live >> lane) & 1ull) ? (unsigned)hi_abs(st.pr[j] * st.pr[j] + st.pi[j] * st.pi[j]) + 1u : 0u;
                    const unsigned mx = cg_wave_max_u32(kk);
                    cl = (int)__builtin_ctzll(__ballot(kk == mx));
                }
                st.pl[j] = cl;
                st.parity += __builtin_popcountll(st.live & ((1ull << cl) - 1ull));   // position of the pivot among the live rows
                st.live &= ~(1ull << cl);
                double rpr[PW], rpi[PW];
#pragma unroll
                for (int jj = j; jj < PW; ++jj) { rpr[jj] = cg_readlane_f64(st.pr[jj], cl); rpi[jj] = cg_readlane_f64(st.pi[jj], cl); }
                st.pm = cmul(st.pm, CgCplx{rpr[j], rpi[j]});
                if ((j & 3) == 3 || (!FULL && j + 1 == kb)) {
                    int ex; const double mxv = fmax(fabs(st.pm.re), fabs(st.pm.im)); (void)frexp(mxv, &ex);
                    st.pm.re = ldexp(st.pm.re, -ex); st.pm.im = ldexp(st.pm.im, -ex); st.pe += ex;
                }
                const double rd = cg_fast_rcp1(rpr[j] * rpr[j] + rpi[j] * rpi[j]);
                const double qr = rpr[j] * rd, qi = -rpi[j] * rd;      // 1 / pivot
                const double lr = st.pr[j] * qr - st.pi[j] * qi, li = st.pr[j] * qi + st.pi[j] * qr;
                st.pr[j] = lr; st.pi[j] = li;
#pragma unroll
                for (int jj = j + 1; jj < PW; ++jj) {
                    st.pr[jj] = fma(li, rpi[jj], fma(-lr, rpr[jj], st.pr[jj]));
                    st.pi[jj] = fma(-li, rpr[jj], fma(-lr, rpi[jj], st.pi[jj]));
                }
                // E = -L21 L11^-1 (the row operations on the identity columns)
#pragma unroll
                for (int c = 0; c < j; ++c) {
                    const double ur = cg_readlane_f64(st.er[c], cl), ui = cg_readlane_f64(st.ei[c], cl);
                    st.er[c] = fma(li, ui, fma(-lr, ur, st.er[c]));
                    st.ei[c] = fma(-li, ur, fma(-lr, ui, st.ei[c]));
                }
                st.er[j] = -lr; st.ei[j] = -li;
            } else { st.pl[j] = 0; st.er[j] = 0.0; st.ei[j] = 0.0; }
        }
    }
    static __device__ __forceinline__ void chain_cplx(double* C, int n, int ldc, int lane, int* fl, double* res) {
        const int npan = (n + PW - 1) / PW;
        Cplx st;
        st.live = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        st.pm = CgCplx{1.0, 0.0}; st.pe = 0; st.parity = 0;
        const int rowc = lane < n ? lane : n - 1;                          // (clamped: lanes beyond the matrix read a valid row and are never live)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const d2_t v = *(const d2_t*)(C + 2 * (rowc * ldc + j));     // (n >= 8 on this path)
            st.pr[j] = v[0]; st.pi[j] = v[1]; st.er[j] = 0.0; st.ei[j] = 0.0;
        }
        for (int k = 0; k < npan; ++k) {
            const int k0 = k * PW;
            CG_LU_TRACE(1, k)
            CG_STAMP_START(25)
            if (k + 1 == npan) { steps_cplx<false>(st, k0, n - k0, lane); CG_STAMP_END(25) break; }
            steps_cplx<true>(st, k0, PW, lane);
            CG_STAMP(25)
#pragma unroll
            for (int c = 0; c < PW; ++c) { st.er[c] = sel(st.live, st.er[c]); st.ei[c] = sel(st.live, st.ei[c]); }
            if (lane < n) {
#pragma unroll
                for (int c = 0; c < PW; ++c) *(d2_t*)(C + 2 * (lane * ldc + k0 + c)) = d2_t{st.er[c], st.ei[c]};
            }
            if (lane == 0) {
                i4_t* q = (i4_t*)(fl + O_REC_C + REC * k);
                q[0] = i4_t{st.pl[0], st.pl[1], st.pl[2], st.pl[3]};
                q[1] = i4_t{st.pl[4], st.pl[5], st.pl[6], st.pl[7]};
                q[2] = i4_t{live_tiles(st.live, 0ull), 0, 0, 0};
            }
            cg_flag_post(fl + O_PUB_C, k + 1, lane);
            CG_STAMP(26)
            const int m0 = k0 + PW;
            if (k > 0) cg_flag_wait(fl + O_APP + 8 + ((k + 1) >> 1), k);
            CG_STAMP(27)
            // p = b + E A12 with the pivot rows' entries read where they stand (wave-uniform addresses: broadcast reads).  The reads of
            // pivot row j + 1 are issued before the products of row j (two rows of eight 16-byte reads in flight: the LDS latency is
            // paid once, not per pair of reads)
            if (m0 + PW <= n) {
                const d2_t* qb = (const d2_t*)(C + 2 * (rowc * ldc + m0));
                d2_t bv[PW], u[2][PW];
#pragma unroll
                for (int c = 0; c < PW; ++c) bv[c] = qb[c];
                {
                    const d2_t* q = (const d2_t*)(C + 2 * (st.pl[0] * ldc + m0));
#pragma unroll
                    for (int c = 0; c < PW; ++c) u[0][c] = q[c];
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    if (j + 1 < PW) {
                        const d2_t* q = (const d2_t*)(C + 2 * (st.pl[j + 1] * ldc + m0));
#pragma unroll
                        for (int c = 0; c < PW; ++c) u[(j + 1) & 1][c] = q[c];
                        asm volatile("" ::: "memory");
                    }
#pragma unroll
                    for (int c = 0; c < PW; ++c) {
                        const d2_t uu = u[j & 1][c];
                        st.pr[c] = fma(-st.ei[j], uu[1], fma(st.er[j], uu[0], j == 0 ? bv[c][0] : st.pr[c]));
                        st.pi[c] = fma(st.ei[j], uu[0], fma(st.er[j], uu[1], j == 0 ? bv[c][1] : st.pi[c]));
                    }
                }
            } else {                                                       // the last, narrower panel (once per factorisation): clamped columns
                double br[PW], bi[PW];
#pragma unroll
                for (int c = 0; c < PW; ++c) {
                    const int cc = m0 + c < n ? m0 + c : n - 1;
                    const d2_t v = *(const d2_t*)(C + 2 * (rowc * ldc + cc));
                    br[c] = v[0]; bi[c] = v[1];
                }
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    const d2_t* q = (const d2_t*)(C + 2 * (st.pl[j] * ldc));
#pragma unroll
                    for (int c = 0; c < PW; ++c) {
                        const int cc = m0 + c < n ? m0 + c : n - 1;
                        const d2_t u = q[cc];
                        st.pr[c] = fma(-st.ei[j], u[1], fma(st.er[j], u[0], j == 0 ? br[c] : st.pr[c]));
                        st.pi[c] = fma(st.ei[j], u[0], fma(st.er[j], u[1], j == 0 ? bi[c] : st.pi[c]));
                    }
                }
            }
            CG_STAMP_END(28)
        }
        if (lane == 0) {
            if (st.parity & 1) { st.pm.re = -st.pm.re; st.pm.im = -st.pm.im; }
            res[1] = 0.5 * cg_log_ool(st.pm.re * st.pm.re + st.pm.im * st.pm.im) + (double)st.pe * 0.693147180559945309417232121458;
            res[2] = cg_atan2_ool(st.pm.im, st.pm.re);
        }
    }

    // ---------------------------------------------------------------- helper tasks: pure GEMMs on MFMA ------------------------
    // block columns [c0, cend) (<= 16) += E[:, k0 .. k0+8) * A[pivot rows of the panel, block], over the live row tiles.  Loads are never
    // predicated (row / column indices beyond the matrix are clamped: what they produce lands in accumulator rows / columns that are
    // not stored), four tiles per trip so that the LDS latency is paid once per 8 MFMAs.
    static __device__ __forceinline__ void task_real(double* A, int N, int lda, int k0, int c0, int cend, const int* rec, int lane) {
        const int col = lane & 15, kq = lane >> 4, bc = c0 + col;
        const bool cok = bc < cend;
        const int bcc = cok ? bc : c0;
        const int p0 = rec[kq], p1 = rec[4 + kq];
        int tiles = __builtin_amdgcn_readfirstlane(rec[8]);
        const double bv0 = A[p0 * lda + bcc], bv1 = A[p1 * lda + bcc];
        constexpr int T = 4;
        while (tiles) {
            int t[T];
#pragma unroll
            for (int h = 0; h < T; ++h) { t[h] = tiles ? __builtin_ctz(tiles) : -1; tiles &= tiles - 1; }   // (0 & -1 = 0)
            cg_d4_t c[T]; double av[T][2];
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                const int r0 = 16 * t[h], ar = r0 + col < N ? r0 + col : N - 1;
                const double* pa = A + ar * lda + k0 + kq;
                av[h][0] = pa[0]; av[h][1] = pa[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int rr = r0 + kq + 4 * r; c[h][r] = A[(rr < N ? rr : N - 1) * lda + bcc]; }
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                c[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][0], bv0, c[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                c[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[h][1], bv1, c[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                const int r0 = 16 * t[h];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int rr = r0 + kq + 4 * r; if (rr < N && cok) A[rr * lda + bc] = c[h][r]; }
            }
        }
    }
    static __device__ __forceinline__ void task_cplx(double* C, int n, int ldc, int k0, int c0, int cend, const int* rec, int lane) {
        const int col = lane & 15, kq = lane >> 4, bc = c0 + col;
        const bool cok = bc < cend;
        const int bcc = cok ? bc : c0;
        const int p0 = rec[kq], p1 = rec[4 + kq];
        int tiles = __builtin_amdgcn_readfirstlane(rec[8]);
        const d2_t b0 = *(const d2_t*)(C + 2 * (p0 * ldc + bcc)), b1 = *(const d2_t*)(C + 2 * (p1 * ldc + bcc));
        constexpr int T = 2;
        while (tiles) {
            int t[T];
#pragma unroll
            for (int h = 0; h < T; ++h) { t[h] = tiles ? __builtin_ctz(tiles) : -1; tiles &= tiles - 1; }
            cg_d4_t cr[T], ci[T]; d2_t a0[T], a1[T];
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                const int r0 = 16 * t[h], ar = r0 + col < n ? r0 + col : n - 1;
                const d2_t* pa = (const d2_t*)(C + 2 * (ar * ldc + k0 + kq));
                a0[h] = pa[0]; a1[h] = pa[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + kq + 4 * r;
                    const d2_t v = *(const d2_t*)(C + 2 * ((rr < n ? rr : n - 1) * ldc + bcc));
                    cr[h][r] = v[0]; ci[h][r] = v[1];
                }
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                cr[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h][0], b0[0], cr[h], 0, 0, 0);
                ci[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h][0], b0[1], ci[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                cr[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[h][1], b0[1], cr[h], 0, 0, 0);
                ci[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h][1], b0[0], ci[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                cr[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h][0], b1[0], cr[h], 0, 0, 0);
                ci[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h][0], b1[1], ci[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                cr[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[h][1], b1[1], cr[h], 0, 0, 0);
                ci[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h][1], b1[0], ci[h], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < T; ++h) {
                if (t[h] < 0) continue;
                const int r0 = 16 * t[h];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = r0 + kq + 4 * r;
                    if (rr < n && cok) *(d2_t*)(C + 2 * (rr * ldc + bc)) = d2_t{cr[h][r], ci[h][r]};
                }
            }
        }
    }
};

__device__ __forceinline__ void cg_blocked_lu_dual2(const CgBlk& b, double* A, int N, int lda, double* C, int n, int ldc, double* res,
                                                    double& logabs_real, double& logabs_c, double& arg_c) {
    const int lane = b.tid & 63, wave = b.tid >> 6;
    constexpr int PW = CG_LU_PW;
    int* fl = (int*)(res + 4);
    const int npr = (N + PW - 1) / PW, npc = (n + PW - 1) / PW;
    for (int e = b.tid; e < 32; e += b.nthr) fl[e] = 0;
    b.sync();
    // the complex chain rides on wave 5 when there is one (SIMD 1 either way -- waves go to the SIMDs round-robin -- but as the younger
    // of the two waves there: 108.4 k -> 107.0 k cycles per pair at n = 57, 84.9 k -> 82.5 k at n = 49, tools/lu_bench; wave 4 = the real
    // chain's SIMD costs 3 %)
    const int cw = b.nthr >= 384 ? 5 : 1;
    if (wave == 0) {
        CG_STAMP_START(16)
        if (N <= 64) CgLu2::chain_real<1>(A, N, lda, lane, fl, res); else CgLu2::chain_real<2>(A, N, lda, lane, fl, res);
        CG_STAMP_END(16)
    } else if (wave == cw) {
        CG_STAMP_START(17)
        CgLu2::chain_cplx(C, n, ldc, lane, fl, res);
        CG_STAMP_END(17)
    } else {
        // helpers: lane l < 8 watches real block l, lanes 8..11 complex block l - 8; one ballot picks the task
        int* claim = fl + CgLu2::O_CLAIM; int* app = fl + CgLu2::O_APP;
        const int nbr = (N + 15) >> 4, nbc = (n + 15) >> 4;
        const bool isr = lane < 8, mine = isr ? lane < nbr : (lane < 12 && lane - 8 < nbc);
        const int j = isr ? lane : lane - 8;
        const int limit = !mine ? 0 : (isr ? (2 * j < npr - 1 ? 2 * j : npr - 1) : (2 * j < npc - 1 ? 2 * j : npc - 1));   // panels this block receives
        int idle = 0;
        while (idle < (1 << 22)) {
            // every flag in ONE LDS round trip (plain loads: LDS is coherent inside the workgroup, and what a flag guards is read only
            // after the decision that depends on its value)
            const int lc = lane < 12 ? lane : 0;
            const int q = *(volatile const int*)(claim + lc), ap = *(volatile const int*)(app + lc), pub = *(volatile const int*)(fl + (isr ? 0 : 1));
            const bool pending = mine && q < limit;
            if (!__ballot(pending)) break;                               // every task of both matrices has been claimed
            const bool ready = pending && pub > q && ap >= q;
            const unsigned long long me = __ballot(ready);
            if (!me) { ++idle; __builtin_amdgcn_s_sleep(1); continue; }
            // the block a chain needs next (j = (q + 2) / 2) first, then the rest; real before complex.  (Measured and dropped, tools/lu_bench:
            // earliest-deadline order, serving the block a chain is stalled on first, helpers with a preferred matrix -- 108 k cycles
            // per pair at n = 57 with this order against 111 - 120 k.)
            const unsigned long long mu = __ballot(ready && j == ((q + 2) >> 1));
            const int pick = (int)__builtin_ctzll(mu ? mu : me);
            const int tq = __builtin_amdgcn_readlane(q, pick);
            int got = 0;
            if (lane == 0) got = atomicCAS(claim + pick, tq, tq + 1) == tq ? 1 : 0;
            if (!__builtin_amdgcn_readfirstlane(got)) continue;          // another helper took it
            // acquire side of the chains' plain-store publication: the record, E and pivot-row reads of the task must not be hoisted or
            // merged above the flag loads / the claim by the compiler (the hardware executes a wave's LDS accesses in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            CG_STAMP_START(19)
            const int k0 = tq * PW;
            if (pick < 8) {
                const int c0 = 16 * pick > k0 + 2 * PW ? 16 * pick : k0 + 2 * PW, cend = 16 * pick + 16 < N ? 16 * pick + 16 : N;
                if (c0 < cend) CgLu2::task_real(A, N, lda, k0, c0, cend, fl + CgLu2::O_REC_R + CgLu2::REC * tq, lane);
            } else {
                const int jb = pick - 8;
                const int c0 = 16 * jb > k0 + 2 * PW ? 16 * jb : k0 + 2 * PW, cend = 16 * jb + 16 < n ? 16 * jb + 16 : n;
                if (c0 < cend) CgLu2::task_cplx(C, n, ldc, k0, c0, cend, fl + CgLu2::O_REC_C + CgLu2::REC * tq, lane);
            }
            cg_flag_post(app + pick, tq + 1, lane);
            CG_STAMP_END(19)
        }
    }
    b.sync();
    logabs_real = res[0]; logabs_c = res[1]; arg_c = res[2];
    b.sync();
}
#endif
