// cg_generic.hpp -- log Psi, Metropolis chain and x-derivatives on top of the general-depth flow
// (cg_flow_generic.hpp).  Same mathematics as the fast path (cg_flow_fast.hpp / cg_derivs.hpp), runtime sizes,
// every array in a per-workgroup HBM workspace.
//
// theta-gradients (cg_param_vjp / cg_quantum_score, main.py:277-278 + src/logpsi.py:183-203) for any depth:
//     d/dtheta log phi(z)          = sum_a g_a dz_a/dtheta                      (g = d log phi / dz, as in cg_derivs.hpp)
//     d/dtheta 1/2 log|det J|      = 1/2 sum_b d/deps [ d/dtheta ( c_b . z(x + eps e_b; theta) ) ],  c_b = row b of J^-1
// i.e. one hand-written REVERSE pass of the primal flow per column of J, run on dual numbers (Jet2 with the x-tangent
// e_b): the dual part of the parameter gradient of c_b . z is  sum_a J^-1[b][a] dJ[a][b]/dtheta.  N + 1 cheap passes
// per walker (the primal flow is ~1/50 of a log Psi evaluation) instead of a reverse pass through the dense forward
// Jacobian; once per optimisation step.
#pragma once
#include "cg_flow_generic.hpp"
#include "cg_linalg.hpp"

// reverse-pass arena (offsets in units of the scalar type)
struct CgGenRev {
    size_t Ts, F, U, V, s, sbar, sbar2, tbar, tbar2, ubar, fbar, vbar, zbar, g, total;
};
static inline CgGenRev cg_gen_rev(const CgGenModel& m) {
    CgGenRev r; size_t t = 0;
    const size_t n = m.n, N = (size_t)m.n * m.dim, L = m.depth;
    auto take = [&](size_t c) { size_t q = t; t += c; return q; };
    r.Ts = take(L * n * n * m.wmax_t);      // Ts[0] = pair features, Ts[l+1] = pair stream after layer l
    r.F = take(L * n * m.fmax); r.U = take(L * n * m.hs); r.V = take(L * n * n * m.ht);
    r.s = take(n * m.wmax_s);
    r.sbar = take(n * m.wmax_s); r.sbar2 = take(n * m.wmax_s);
    r.tbar = take(n * n * m.wmax_t); r.tbar2 = take(n * n * m.wmax_t);
    r.ubar = take(n * m.hs); r.fbar = take(n * m.fmax); r.vbar = take(n * n * m.ht);
    r.zbar = take(N); r.g = take(m.nparam);
    r.total = t;
    return r;
}

struct CgGenWs {      // offsets in doubles
    size_t da, ja, Dm, Dc, Dinv, Jc, Jinv, M, Ta, Kd, gz, xj, xc, xp, perm, ra, gs, total;
    CgGenRev rev;
};
static inline CgGenWs cg_gen_ws(const CgGenModel& m, bool with_rev = false) {
    CgGenWs w; size_t t = 0;
    const size_t n = m.n, N = (size_t)m.n * m.dim, d = m.dim;
    auto take = [&](size_t c) { size_t r = t; t += (c + 1) & ~(size_t)1; return r; };
    w.da = take(m.total); w.ja = take(3 * m.total);
    w.Dm = take(2 * n * n); w.Dc = take(2 * n * n); w.Dinv = take(2 * n * n);
    w.Jc = take(N * N); w.Jinv = take(N * N); w.M = take(N * N);
    w.Ta = take(2 * d * n * n); w.Kd = take(2 * d * d * n); w.gz = take(2 * N);
    w.xj = take(3 * N); w.xc = take(N); w.xp = take(N); w.perm = take(N + 42);
    w.rev = cg_gen_rev(m);
    w.ra = w.gs = t;
    if (with_rev) { w.ra = take(3 * w.rev.total); w.gs = take(m.nparam); }
    w.total = t;
    return w;
}

// Primal flow with everything the reverse pass needs kept (T = Jet2: value + x-tangent), and its reverse pass.
template <class T>
struct CgGenRevK {
    static CG_DEVI void forward(const CgBlk& b, const CgGenModel& m, const CgGenRev& r, const double* __restrict__ th, const T* x, T* ra) {
        const int n = m.n, d = m.dim, hs = m.hs, ht = m.ht, P = 2 * d + 1, WS = m.wmax_s, WT = m.wmax_t;
        T *Ts = ra + r.Ts, *F = ra + r.F, *U = ra + r.U, *V = ra + r.V, *s = ra + r.s;
        const double rn = 1.0 / (double)n, c1 = 2.0 * CG_PI / m.L, ch = CG_PI / m.L;
        for (int e = b.tid; e < n * n; e += b.nthr) {                        // pair features (src/flow.py:20-26)
            const int i = e / n, j = e - i * n;
            T* t = Ts + (size_t)e * WT;
            T d2 = T(0.0);
            for (int a = 0; a < d; ++a) {
                const T rr = x[i * d + a] - x[j * d + a];
                T s2, c2, s1, c1v;
                cg_sincos(rr * c1, s2, c2); cg_sincos(rr * ch, s1, c1v);
                if (i == j) { c2 = T(1.0); s2 = T(0.0); }
                t[a] = c2; t[d + a] = s2; d2 += s1 * s1;
            }
            t[2 * d] = (i == j) ? T(0.0) : cg_sqrt(d2);
        }
        b.sync();
        for (int l = 0; l < m.depth; ++l) {
            const int wsp = l == 0 ? d : hs, wtp = l == 0 ? P : ht, fs = 2 * wsp + wtp;
            const T* tl = Ts + (size_t)l * n * n * WT;
            T* Fl = F + (size_t)l * n * m.fmax; T* Ul = U + (size_t)l * n * hs;
            for (int e = b.tid; e < n * fs; e += b.nthr) {                   // f_i = [s_i, mean_k s_k, mean_j t_ij]
                const int i = e / fs, c = e - i * fs;
                T v = T(0.0);
                if (c < wsp) { if (l > 0) v = s[i * WS + c]; }
                else if (c < 2 * wsp) { if (l > 0) { T a = T(0.0); for (int k = 0; k < n; ++k) a += s[k * WS + (c - wsp)]; v = a * rn; } }
                else { T a = T(0.0); for (int j = 0; j < n; ++j) a += tl[(size_t)(i * n + j) * WT + (c - 2 * wsp)]; v = a * rn; }
                Fl[i * m.fmax + c] = v;
            }
            b.sync();
            const double* W = th + m.sp_w[l]; const double* bb = th + m.sp_b[l];
            for (int e = b.tid; e < n * hs; e += b.nthr) {
                const int i = e / hs, h = e - i * hs;
                T a = T(bb[h]);
                for (int c = 0; c < fs; ++c) a += W[c * hs + h] * Fl[i * m.fmax + c];
                Ul[e] = a;
                const T spv = cg_softplus(a);
                s[i * WS + h] = (l == 0) ? spv : s[i * WS + h] + spv;      // F already holds the old s
            }
            if (l < m.depth - 1) {
                const double* Wt = th + m.tp_w[l]; const double* bt = th + m.tp_b[l];
                T* Vl = V + (size_t)l * n * n * ht; T* tn = Ts + (size_t)(l + 1) * n * n * WT;
                for (int e = b.tid; e < n * n * ht; e += b.nthr) {
                    const int pr = e / ht, h = e - pr * ht;
                    T a = T(bt[h]);
                    for (int c = 0; c < wtp; ++c) a += Wt[c * ht + h] * tl[(size_t)pr * WT + c];
                    Vl[e] = a;
                    const T spv = cg_softplus(a);
                    tn[(size_t)pr * WT + h] = (l == 0) ? spv : tl[(size_t)pr * WT + h] + spv;
                }
            }
            b.sync();
        }
    }

    // cotangent zbar (ra + r.zbar, N entries) -> parameter gradient g (ra + r.g, every entry written exactly once)
    static CG_DEVI void backward(const CgBlk& b, const CgGenModel& m, const CgGenRev& r, const double* __restrict__ th, T* ra) {
        const int n = m.n, d = m.dim, hs = m.hs, ht = m.ht, P = 2 * d + 1, WS = m.wmax_s, WT = m.wmax_t;
        const T *Ts = ra + r.Ts, *F = ra + r.F, *U = ra + r.U, *V = ra + r.V, *s = ra + r.s, *zbar = ra + r.zbar;
        T *sbar = ra + r.sbar, *sbar2 = ra + r.sbar2, *tbar = ra + r.tbar, *tbar2 = ra + r.tbar2, *ubar = ra + r.ubar,
          *fbar = ra + r.fbar, *vbar = ra + r.vbar, *g = ra + r.g;
        const double rn = 1.0 / (double)n;
        const double* Wf = th + m.fin_w;
        // z = x + s Wf + bf
        for (int e = b.tid; e < n * hs; e += b.nthr) {
            const int i = e / hs, h = e - i * hs;
            T a = T(0.0);
            for (int q = 0; q < d; ++q) a += Wf[h * d + q] * zbar[i * d + q];
            sbar[i * WS + h] = a;
        }
        for (int e = b.tid; e < d + hs * d; e += b.nthr) {
            T a = T(0.0);
            if (e < d) { for (int i = 0; i < n; ++i) a += zbar[i * d + e]; g[m.fin_b + e] = a; }
            else { const int q = e - d, h = q / d, c = q - h * d; for (int i = 0; i < n; ++i) a += s[i * WS + h] * zbar[i * d + c]; g[m.fin_w + q] = a; }
        }
        b.sync();
        for (int l = m.depth - 1; l >= 0; --l) {
            const int wsp = l == 0 ? d : hs, wtp = l == 0 ? P : ht, fs = 2 * wsp + wtp;
            const T* tl = Ts + (size_t)l * n * n * WT;
            const T* Fl = F + (size_t)l * n * m.fmax; const T* Ul = U + (size_t)l * n * hs;
            const bool has_t = l < m.depth - 1;
            // (A) pair-stream update of this layer: t_new = [t_old +] softplus(t_old Wt + bt); tbar is d/d t_new
            if (has_t) {
                const double* Wt = th + m.tp_w[l];
                const T* Vl = V + (size_t)l * n * n * ht;
                for (int e = b.tid; e < n * n * ht; e += b.nthr) {
                    const int pr = e / ht, h = e - pr * ht;
                    vbar[e] = tbar[(size_t)pr * WT + h] * cg_sigmoid(Vl[e]);
                }
                b.sync();
                for (int e = b.tid; e < ht + wtp * ht; e += b.nthr) {
                    T a = T(0.0);
                    if (e < ht) { for (int pr = 0; pr < n * n; ++pr) a += vbar[pr * ht + e]; g[m.tp_b[l] + e] = a; }
                    else { const int q = e - ht, c = q / ht, h = q - c * ht; for (int pr = 0; pr < n * n; ++pr) a += tl[(size_t)pr * WT + c] * vbar[pr * ht + h]; g[m.tp_w[l] + q] = a; }
                }
                if (l > 0)
                    for (int e = b.tid; e < n * n * ht; e += b.nthr) {       // d/d t_old (residual + through the layer)
                        const int pr = e / ht, c = e - pr * ht;
                        T a = tbar[(size_t)pr * WT + c];
                        for (int h = 0; h < ht; ++h) a += Wt[c * ht + h] * vbar[pr * ht + h];
                        tbar2[(size_t)pr * WT + c] = a;
                    }
            } else if (l > 0) {
                for (int e = b.tid; e < n * n * ht; e += b.nthr) { const int pr = e / ht, c = e - pr * ht; tbar2[(size_t)pr * WT + c] = T(0.0); }
            }
            // (B) one-particle update: s_new = [s_old +] softplus(f W + b)
            const double* W = th + m.sp_w[l];
            for (int e = b.tid; e < n * hs; e += b.nthr) { const int i = e / hs, h = e - i * hs; ubar[e] = sbar[i * WS + h] * cg_sigmoid(Ul[e]); }
            b.sync();
            for (int e = b.tid; e < hs + fs * hs; e += b.nthr) {
                T a = T(0.0);
                if (e < hs) { for (int i = 0; i < n; ++i) a += ubar[i * hs + e]; g[m.sp_b[l] + e] = a; }
                else { const int q = e - hs, c = q / hs, h = q - c * hs; for (int i = 0; i < n; ++i) a += Fl[i * m.fmax + c] * ubar[i * hs + h]; g[m.sp_w[l] + q] = a; }
            }
            if (l > 0) {
                for (int e = b.tid; e < n * fs; e += b.nthr) {
                    const int i = e / fs, c = e - i * fs;
                    T a = T(0.0);
                    for (int h = 0; h < hs; ++h) a += W[c * hs + h] * ubar[i * hs + h];
                    fbar[i * m.fmax + c] = a;
                }
                b.sync();
                for (int e = b.tid; e < n * hs; e += b.nthr) {               // d/d s_old: residual + own slot + mean slot
                    const int i = e / hs, c = e - i * hs;
                    T a = sbar[i * WS + c] + fbar[i * m.fmax + c];
                    T mm = T(0.0);
                    for (int k = 0; k < n; ++k) mm += fbar[k * m.fmax + wsp + c];
                    sbar2[i * WS + c] = a + mm * rn;
                }
                for (int e = b.tid; e < n * n * ht; e += b.nthr) {           // d/d t_old += (1/n) fbar_i[mean_j t slot]
                    const int pr = e / ht, c = e - pr * ht, i = pr / n;
                    tbar2[(size_t)pr * WT + c] += fbar[i * m.fmax + 2 * wsp + c] * rn;
                }
                b.sync();
                T* q1 = sbar; sbar = sbar2; sbar2 = q1;
                T* q2 = tbar; tbar = tbar2; tbar2 = q2;
            } else {
                b.sync();
            }
        }
    }
};

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
            {   // the seven sums of this direction in one reduction (2 barriers)
                double red[7] = {a_re, a_im, t1, p_re, p_im, t2, t3};
                cg_block_sum_n<7>(b, red, lds);
                a_re = red[0]; a_im = red[1]; t1 = red[2]; p_re = red[3]; p_im = red[4]; t2 = red[5]; t3 = red[6];
            }
            if (want_phi2) { lap_re += p_re; lap_im += p_im; }
            if (want_jac2) lap_re += 0.5 * (t2 - t3);
            if (!probe && b.tid == 0) { grad[2 * dir] = a_re + 0.5 * t1; grad[2 * dir + 1] = a_im; }
            b.sync();
        }
        if (b.tid == 0) { lap[0] = lap_re; lap[1] = lap_im; }
    }

    // sum_b w_re d/dtheta Re log Psi + w_im d/dtheta Im log Psi into gacc (+=), or per-sample scores (score[2p + {0,1}])
    static CG_DEVI void param_vjp(const CgBlk& b, const CgGenModel& m, const CgGenWs& w, const double* __restrict__ th,
                                  const double* __restrict__ spk, const int* __restrict__ sidx, const double* __restrict__ xg,
                                  double w_re, double w_im, double* __restrict__ gacc, double* __restrict__ score, double* ws) {
        const int n = m.n, d = m.dim, N = n * d, NP = m.nparam;
        const CgGenRev& r = w.rev;
        double* da = ws + w.da; double* x = ws + w.xc;
        for (int e = b.tid; e < N; e += b.nthr) x[e] = xg[e];
        b.sync();
        CgGen<double>::flow(b, m, th, x, da, true);
        double *Jc = ws + w.Jc, *Jinv = ws + w.Jinv, *Dm = ws + w.Dm, *Dc = ws + w.Dc, *Dinv = ws + w.Dinv, *gz = ws + w.gz;
        int* perm = (int*)(ws + w.perm);
        for (int e = b.tid; e < N * N; e += b.nthr) Jc[e] = da[m.o_J + e];
        b.sync();
        (void)cg_inverse_real(b, Jc, N, N, Jinv, N, perm);
        slater(b, da + m.o_z, spk, sidx, n, d, Dm);
        for (int e = b.tid; e < 2 * n * n; e += b.nthr) Dc[e] = Dm[e];
        b.sync();
        double la, ar;
        cg_inverse_complex(b, Dc, n, n, Dinv, n, perm, la, ar);
        for (int e = b.tid; e < N; e += b.nthr) {                    // g_ia = sum_j D_ij (i k_j^a) Dinv_ji
            const int i = e / d, a = e - i * d;
            double re = 0, im = 0;
            for (int j = 0; j < n; ++j) {
                const double ka = spk[(size_t)sidx[j] * d + a];
                const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + i)], Dinv[2 * (j * n + i) + 1]});
                re += -ka * p.im; im += ka * p.re;
            }
            gz[2 * e] = re; gz[2 * e + 1] = im;
        }
        b.sync();
        Jet2* xj = (Jet2*)(ws + w.xj); Jet2* ra = (Jet2*)(ws + w.ra);
        Jet2* zbar = ra + r.zbar; const Jet2* g = ra + r.g;
        double* gs = ws + w.gs;
        const int npass = score ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            const double wr = score ? (pass == 0 ? 1.0 : 0.0) : w_re;
            const double wi = score ? (pass == 0 ? 0.0 : 1.0) : w_im;
            // log phi part (value of the gradient; zero x-tangent)
            for (int e = b.tid; e < N; e += b.nthr) { xj[e] = Jet2(x[e]); zbar[e] = Jet2(wr * gz[2 * e] + wi * gz[2 * e + 1]); }
            b.sync();
            CgGenRevK<Jet2>::forward(b, m, r, th, xj, ra);
            CgGenRevK<Jet2>::backward(b, m, r, th, ra);
            for (int e = b.tid; e < NP; e += b.nthr) gs[e] = g[e].v;
            b.sync();
            if (wr != 0.0)
                for (int beta = 0; beta < N; ++beta) {               // 1/2 log|det J| part, one column of J per pass
                    for (int e = b.tid; e < N; e += b.nthr) {
                        xj[e] = Jet2(x[e], e == beta ? 1.0 : 0.0, 0.0);
                        zbar[e] = Jet2(0.5 * wr * Jinv[beta * N + e]);
                    }
                    b.sync();
                    CgGenRevK<Jet2>::forward(b, m, r, th, xj, ra);
                    CgGenRevK<Jet2>::backward(b, m, r, th, ra);
                    for (int e = b.tid; e < NP; e += b.nthr) gs[e] += g[e].d;
                    b.sync();
                }
            if (score) for (int e = b.tid; e < NP; e += b.nthr) score[2 * e + pass] = gs[e];
            else if (gacc) for (int e = b.tid; e < NP; e += b.nthr) gacc[e] += gs[e];
            b.sync();
        }
    }
};
