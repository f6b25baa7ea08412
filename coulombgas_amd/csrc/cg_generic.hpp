// cg_generic.hpp -- log Psi, Metropolis chain and x-derivatives on top of the general-depth flow
// (cg_flow_generic.hpp).  Same mathematics as the fast path (cg_flow_fast.hpp / cg_derivs.hpp), runtime sizes,
// every array in a per-workgroup HBM workspace.  theta-gradients are not provided for this path (cg_param_vjp and
// cg_quantum_score return CG_ERR_UNSUPPORTED for networks the depth-2 fast path does not cover).
#pragma once
#include "cg_flow_generic.hpp"
#include "cg_linalg.hpp"

struct CgGenWs {      // offsets in doubles
    size_t da, ja, Dm, Dc, Dinv, Jc, Jinv, M, Ta, Kd, gz, xj, xc, xp, perm, total;
};
static inline CgGenWs cg_gen_ws(const CgGenModel& m) {
    CgGenWs w; size_t t = 0;
    const size_t n = m.n, N = (size_t)m.n * m.dim, d = m.dim;
    auto take = [&](size_t c) { size_t r = t; t += (c + 1) & ~(size_t)1; return r; };
    w.da = take(m.total); w.ja = take(3 * m.total);
    w.Dm = take(2 * n * n); w.Dc = take(2 * n * n); w.Dinv = take(2 * n * n);
    w.Jc = take(N * N); w.Jinv = take(N * N); w.M = take(N * N);
    w.Ta = take(2 * d * n * n); w.Kd = take(2 * d * d * n); w.gz = take(2 * N);
    w.xj = take(3 * N); w.xc = take(N); w.xp = take(N); w.perm = take(N + 42);
    w.total = t;
    return w;
}

struct CgGenK {
    // Slater matrix D_ij = exp(i k_j . z_i)  (src/slater.py:14-17, the L^{-d/2} factor is added analytically)
    static CG_DEVI void slater(const CgBlk& b, const double* z, const double* __restrict__ spk, const int* __restrict__ sidx,
                               int n, int d, double* Dm) {
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = e / n, j = e - i * n;
            double ph = 0.0;
            for (int a = 0; a < d; ++a) ph += spk[(size_t)sidx[j] * d + a] * z[i * d + a];
            double s, c; sincos(ph, &s, &c);
            Dm[2 * e] = c; Dm[2 * e + 1] = s;
        }
        b.sync();
    }
    // [Re log phi, Im log phi, 1/2 log|det J|]; x may be any double pointer
    static CG_DEVI void logpsi(const CgBlk& b, const CgGenModel& m, const CgGenWs& w, const double* __restrict__ th,
                               const double* __restrict__ spk, const int* __restrict__ sidx, const double* x, double* ws,
                               double& re_phi, double& im_phi, double& half) {
        double* da = ws + w.da;
        CgGen<double>::flow(b, m, th, x, da, true);
        int* perm = (int*)(ws + w.perm);
        const int N = m.n * m.dim;
        half = 0.5 * cg_lu_logabsdet(b, da + m.o_J, N, N, perm);
        slater(b, da + m.o_z, spk, sidx, m.n, m.dim, ws + w.Dm);
        double la, ar;
        cg_lu_logdet_complex(b, ws + w.Dm, m.n, m.n, perm, la, ar);
        re_phi = la - (double)m.n * (0.5 * m.dim) * log(m.L);
        im_phi = ar;
    }

    // grad / Laplacian of log Psi (all three modes of src/logpsi.py:55-172); see cg_derivs.hpp for the formulas
    static CG_DEVI void grad_laplacian(const CgBlk& b, const CgGenModel& m, const CgGenWs& w, const double* __restrict__ th,
                                       const double* __restrict__ spk, const int* __restrict__ sidx, const double* __restrict__ xg,
                                       int mode, const double* __restrict__ v, double* __restrict__ grad, double* __restrict__ lap,
                                       double* ws, double* lds) {
        const int n = m.n, d = m.dim, N = n * d;
        double* da = ws + w.da; double* x = ws + w.xc;
        for (int e = b.tid; e < N; e += b.nthr) x[e] = xg[e];
        b.sync();
        CgGen<double>::flow(b, m, th, x, da, true);
        double *Jc = ws + w.Jc, *Jinv = ws + w.Jinv, *Dm = ws + w.Dm, *Dc = ws + w.Dc, *Dinv = ws + w.Dinv;
        double *Ta = ws + w.Ta, *Kd = ws + w.Kd, *gz = ws + w.gz, *M = ws + w.M;
        int* perm = (int*)(ws + w.perm);
        for (int e = b.tid; e < N * N; e += b.nthr) Jc[e] = da[m.o_J + e];
        b.sync();
        (void)cg_inverse_real(b, Jc, N, N, Jinv, N, perm);
        slater(b, da + m.o_z, spk, sidx, n, d, Dm);
        for (int e = b.tid; e < 2 * n * n; e += b.nthr) Dc[e] = Dm[e];
        b.sync();
        double la, ar;
        cg_inverse_complex(b, Dc, n, n, Dinv, n, perm, la, ar);
        for (int e = b.tid; e < d * n * n; e += b.nthr) {            // T^a = D diag(i k^a) D^-1
            const int a = e / (n * n), r = e - a * n * n, i = r / n, l = r - i * n;
            double re = 0, im = 0;
            for (int j = 0; j < n; ++j) {
                const double ka = spk[(size_t)sidx[j] * d + a];
                const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + l)], Dinv[2 * (j * n + l) + 1]});
                re += -ka * p.im; im += ka * p.re;
            }
            Ta[2 * e] = re; Ta[2 * e + 1] = im;
        }
        for (int e = b.tid; e < d * d * n; e += b.nthr) {            // diag of K^ab = D diag(-k^a k^b) D^-1
            const int a = e / (d * n), r = e - a * d * n, bb = r / n, i = r - bb * n;
            double re = 0, im = 0;
            for (int j = 0; j < n; ++j) {
                const double kk = -spk[(size_t)sidx[j] * d + a] * spk[(size_t)sidx[j] * d + bb];
                const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + i)], Dinv[2 * (j * n + i) + 1]});
                re += kk * p.re; im += kk * p.im;
            }
            Kd[2 * e] = re; Kd[2 * e + 1] = im;
        }
        b.sync();
        for (int e = b.tid; e < N; e += b.nthr) {                    // g_ia = T^a_ii
            const int i = e / d, a = e - i * d;
            gz[2 * e] = Ta[2 * ((a * n + i) * n + i)]; gz[2 * e + 1] = Ta[2 * ((a * n + i) * n + i) + 1];
        }
        b.sync();
        Jet2* xj = (Jet2*)(ws + w.xj); Jet2* ja = (Jet2*)(ws + w.ja);
        double lap_re = 0.0, lap_im = 0.0;
        const int ndir = N + (mode == 0 ? 0 : 1);
        for (int dir = 0; dir < ndir; ++dir) {
            const bool probe = dir == N;
            for (int e = b.tid; e < N; e += b.nthr) xj[e] = Jet2(x[e], probe ? v[e] : (e == dir ? 1.0 : 0.0), 0.0);
            b.sync();
            CgGen<Jet2>::flow(b, m, th, xj, ja, true);
            const Jet2* zj = ja + m.o_z; const Jet2* Jj = ja + m.o_J;
            const bool want_phi2 = probe ? (mode == 1) : (mode == 0 || mode == 2);
            const bool want_jac2 = probe ? true : (mode == 0);
            double a_re = 0, a_im = 0, p_re = 0, p_im = 0, t1 = 0, t2 = 0, t3 = 0;
            for (int e = b.tid; e < N; e += b.nthr) {
                a_re += gz[2 * e] * zj[e].d; a_im += gz[2 * e + 1] * zj[e].d;
                if (want_phi2) {
                    p_re += gz[2 * e] * zj[e].dd; p_im += gz[2 * e + 1] * zj[e].dd;
                    const int i = e / d, a = e - i * d;
                    for (int bb = 0; bb < d; ++bb) {
                        const double zz = zj[e].d * zj[i * d + bb].d;
                        p_re += zz * Kd[2 * ((a * d + bb) * n + i)]; p_im += zz * Kd[2 * ((a * d + bb) * n + i) + 1];
                    }
                }
            }
            if (want_phi2)
                for (int e = b.tid; e < n * n; e += b.nthr) {
                    const int i = e / n, l = e - i * n;
                    CgCplx yil = {0, 0}, yli = {0, 0};
                    for (int a = 0; a < d; ++a) {
                        const double zi = zj[i * d + a].d, zl = zj[l * d + a].d;
                        yil.re += zi * Ta[2 * ((a * n + i) * n + l)]; yil.im += zi * Ta[2 * ((a * n + i) * n + l) + 1];
                        yli.re += zl * Ta[2 * ((a * n + l) * n + i)]; yli.im += zl * Ta[2 * ((a * n + l) * n + i) + 1];
                    }
                    const CgCplx pr = cmul(yil, yli);
                    p_re -= pr.re; p_im -= pr.im;
                }
            for (int e = b.tid; e < N * N; e += b.nthr) {
                const int al = e / N, ga = e - al * N;
                const double ji = Jinv[al * N + ga];
                t1 += ji * Jj[ga * N + al].d;
                if (want_jac2) t2 += ji * Jj[ga * N + al].dd;
            }
            if (want_jac2) {
                for (int e = b.tid; e < N * N; e += b.nthr) {
                    const int al = e / N, ga = e - al * N;
                    double mm = 0;
                    for (int k = 0; k < N; ++k) mm += Jinv[al * N + k] * Jj[k * N + ga].d;
                    M[e] = mm;
                }
                b.sync();
                for (int e = b.tid; e < N * N; e += b.nthr) { const int al = e / N, ga = e - al * N; t3 += M[al * N + ga] * M[ga * N + al]; }
            }
            a_re = cg_block_sum(b, a_re, lds); a_im = cg_block_sum(b, a_im, lds); t1 = cg_block_sum(b, t1, lds);
            if (want_phi2) { p_re = cg_block_sum(b, p_re, lds); p_im = cg_block_sum(b, p_im, lds); lap_re += p_re; lap_im += p_im; }
            if (want_jac2) { t2 = cg_block_sum(b, t2, lds); t3 = cg_block_sum(b, t3, lds); lap_re += 0.5 * (t2 - t3); }
            if (!probe && b.tid == 0) { grad[2 * dir] = a_re + 0.5 * t1; grad[2 * dir + 1] = a_im; }
            b.sync();
        }
        if (b.tid == 0) { lap[0] = lap_re; lap[1] = lap_im; }
    }
};
