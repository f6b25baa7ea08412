// cg_derivs.hpp -- derivatives of log Psi for the local energy and the parameter gradient.
//
//   param_vjp      : sum_b w_re d/dtheta Re log Psi + w_im d/dtheta Im log Psi         src/VMC.py:69-76 + main.py:278
//                    and per-sample scores d log Psi / d theta                         src/logpsi.py:183-203
// (the x-derivatives, grad / Laplacian of log Psi, live in cg_lap.hpp)
//
// The reference takes jax.jacrev of log Psi (jacfwd + slogdet inside).  Here: a hand-written reverse pass of the
// structured forward code (adjoint of every phase of CgFast::primal / CgFast::jacobian), seeded with
// zbar = w_re Re g + w_im Im g (g_ia = d log phi / d z_ia = T^a_ii, T^a = D diag(i k^a) D^-1) and Jbar = w_re/2 J^-T.
//
// One workgroup per walker; intermediates live in LDS while they fit, otherwise in a per-workgroup HBM workspace (this
// kernel runs once per optimisation step, not per Metropolis step).
#pragma once
#include "cg_flow_fast.hpp"
#include "cg_lap.hpp"

template <int D, int HS, int HT>
struct CgDerivs {
    using F = CgFast<D, HS, HT>;
    using LP = CgLap<D, HS, HT>;
    using PairT = typename LP::PairT;
    static constexpr int PFS = LP::PFS;
    static constexpr int P = F::P;
    static constexpr int NP = F::NPARAM;

    struct Adj { size_t Jhat, Upb, Vb, Bb, Gb, sg1b, sg2b, Ub, Rb, s1b, s2b, u2b, u1b, m1b, gbb, pW0, pWt, su2, total; };
    static Adj adj_layout(int n) {
        const size_t N = (size_t)n * D;
        Adj a; size_t t = 0;
        auto take = [&](size_t c) { size_t r = t; t += (c + 1) & ~(size_t)1; return r; };
        a.Jhat = take(N * N);
        a.Upb = take(N * P); a.Vb = take(N * HT); a.Bb = take(N * HS); a.Gb = take((size_t)n * HS * D);
        a.sg1b = take((size_t)n * HS); a.sg2b = take((size_t)n * HS);
        a.Ub = take(N * HS); a.Rb = take(N * HS);
        a.s1b = take((size_t)n * HS); a.s2b = take((size_t)n * HS); a.u2b = take((size_t)n * HS); a.u1b = take((size_t)n * HS);
        a.m1b = take((size_t)n * HT); a.gbb = take(HS);
        a.pW0 = take((size_t)n * HS * P);            // per-(p,h) partials of W0bar from the G adjoint
        a.pWt = take((size_t)n * HT * (P + 1));      // per-(i,h) partials of Wtbar, btbar
        a.su2 = take(HS);
        a.total = t;
        return a;
    }
    static size_t adj_doubles(int n) { return adj_layout(n).total; }
    struct Ws {   // offsets in doubles into the per-workgroup workspace
        size_t da, x, Jc, Jinv, Dc, Dinv, Ta, Kd, gz, zbar, Jbar, perm, adj, gw, pt, kocc, total;
    };
    static Ws ws_layout(int n) {
        const size_t N = (size_t)n * D;
        const CgFastLds o = cg_fast_layout(n, D, HS, HT, false);
        Ws w; size_t t = 0;
        auto take = [&](size_t c) { size_t r = t; t += (c + 1) & ~(size_t)1; return r; };
        w.da = take(o.total);             // double arena (primal + jacobian)
        w.x = take(N);
        w.Jc = take(N * N); w.Jinv = take(N * N);
        w.Dc = take(2 * (size_t)n * n); w.Dinv = take(2 * (size_t)n * n);
        w.Ta = take(2 * (size_t)D * n * n); w.Kd = take(2 * (size_t)D * D * n);
        w.gz = take(2 * N); w.zbar = take(N); w.Jbar = take(N * N);
        w.perm = take(N + 42);
        w.adj = take(adj_doubles(n));
        w.gw = take(NP);
        w.pt = take((size_t)n * n * PFS);          // pair table (CgLap::pt_build)
        w.kocc = take(N);
        w.total = t;
        return w;
    }
    // vjp_fast / vjp_da: LDS scratch of the theta-VJP kernel (doubles; 0 = none) and whether the primal arena lives there.
    struct Layout { Ws w; Adj a; CgFastLds o; int vjp_fast; int vjp_da; int stage; unsigned mn, mN; };
    static constexpr size_t VJP_LDS_MAX_BYTES = (D == 2 ? 53 : 80) * 1024;      // keeps 3 (d=2) / 2 (d=3) workgroups per CU
    static CG_HD size_t inv_scratch_doubles(int n) { const size_t N = (size_t)n * D; return 2 * N * N + 4 * (size_t)n * n + N + 42; }
    static CG_HD size_t stage_doubles(int n, int nthr) { return cg_inv_panel_scratch(n * D, n, nthr); }    // LDS scratch of cg_inverse_panel_* (0: sizes they do not serve)
    static Layout layout(int n, int nthr = 256, size_t lds_max_bytes = VJP_LDS_MAX_BYTES) {
        Layout l; l.w = ws_layout(n); l.a = adj_layout(n); l.o = cg_fast_layout(n, D, HS, HT, false);
        l.mn = cg_div_magic((unsigned)n); l.mN = cg_div_magic((unsigned)(n * D));
        const size_t base = CG_TAB_DOUBLES + lds_doubles(n, nthr), inv = inv_scratch_doubles(n), NN = (size_t)n * D * n * D;
        l.vjp_fast = 0; l.vjp_da = 0; l.stage = 0;
        if (sizeof(double) * (base + l.o.total + inv - NN) <= lds_max_bytes) { l.vjp_fast = (int)(l.o.total + inv - NN); l.vjp_da = 1; }
        else if (sizeof(double) * (base + inv) <= lds_max_bytes) l.vjp_fast = (int)inv;
        else if (sizeof(double) * (base + stage_doubles(n, nthr)) <= lds_max_bytes) l.stage = (int)stage_doubles(n, nthr);   // register-tiled inverses, their panels staged in LDS
        return l;
    }
    static CG_HD size_t vjp_lds_doubles(const Layout& l) { return (size_t)(l.vjp_fast ? l.vjp_fast : l.stage); }
    static size_t ws_doubles(int n) { return ws_layout(n).total; }
    static CG_HD size_t lds_doubles(int n, int nthr) { (void)n; return (size_t)nthr + 16; }

    // ------------------------------------------------------------------------------------------------------
    // shared set-up: z, J, J^-1, D, D^-1, g_ia = d log phi / d z_ia.  Returns nothing; everything in ws.
    // ------------------------------------------------------------------------------------------------------
    static CG_DEVI void setup(const CgBlk& b, const double* __restrict__ th, const double* __restrict__ xg,
                              const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                              double* ws, const Ws& w, const CgFastLds& o, bool need_T,
                              double* fast = nullptr, size_t fast_cap = 0, bool da_fast = false,
                              double* stage = nullptr, unsigned mn = 0, unsigned mN = 0) {
        // fast: LDS scratch that is dead during the set-up (the Jet2 arena of the directional passes).  The two
        // Gauss-Jordan inversions (one barrier-separated step per column) and, when the caller does not need the
        // primal arena afterwards (da_fast), the primal + Jacobian evaluation run there instead of in the HBM workspace.
        const int N = n * D;
        const size_t NN = (size_t)N * N, nn2 = 2 * (size_t)n * n;
        da_fast = da_fast && fast && fast_cap >= (size_t)o.total + inv_scratch_doubles(n) - NN;
        const bool inv_lds = da_fast || (fast && fast_cap >= inv_scratch_doubles(n));
        double* da = da_fast ? fast : ws + w.da; double* x = ws + w.x;
        double* sc = inv_lds ? fast + (da_fast ? (size_t)o.total : 0) : nullptr;
        double* kocc = ws + w.kocc;             // wave vectors of the occupied orbitals, staged once (the loops below used to chase
        for (int e = b.tid; e < N; e += b.nthr) {                   // state_idx -> orbital table through global memory per term)
            x[e] = xg[e];
            const int j = e / D;
            kocc[e] = spk[(size_t)sidx[j] * D + (e - j * D)];
        }
        b.sync();
        typename F::WFrag wfrag;
        const typename F::WFrag* wf = F::frags(th, wfrag);                 // MFMA / DPP path of the sampler (device, 16 / 16), else the scalar one
        F::primal(b, th, (const double*)x, n, L, da, o, wf);
        LP::pt_build(b, da + o.sh, da + o.ch, n, mn ? mn : cg_div_magic((unsigned)n), ws + w.pt);
        F::jacobian(b, th, n, L, da, o, wf);
        // with the arena in LDS its J slot (dead after the set-up) is inverted in place
        double* Jc = da_fast ? da + o.J : (inv_lds ? sc : ws + w.Jc);
        if (inv_lds && !da_fast) sc += NN;
        double* Jinv = inv_lds ? sc : ws + w.Jinv;
        double* Dc = inv_lds ? sc + NN : ws + w.Dc;
        double* Dinv = inv_lds ? sc + NN + nn2 : ws + w.Dinv;
        int* perm = inv_lds ? (int*)(sc + NN + 2 * nn2) : (int*)(ws + w.perm);
        bool inverted = false;
#if defined(__HIP_DEVICE_COMPILE__)
        if (inv_lds && N <= 32 && n <= 16 && nn2 >= 128 && b.nthr >= 128) {
            // both inverses by wave-level Gauss-Jordan in registers, concurrently on two waves (cg_linalg.hpp); scratch: Dc
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            const int wave = b.tid >> 6;
            if (wave == 0) {
                if (N == 26) cg_wave_inverse_real<26>(da + o.J, N, N, Jinv, N, Dc);
                else cg_wave_inverse_real<32>(da + o.J, N, N, Jinv, N, Dc);
            } else if (wave == 1) {
                if (n == 13) cg_wave_inverse_complex<13>(da + o.Dm, n, n, Dinv, n, Dc + 64);
                else cg_wave_inverse_complex<16>(da + o.Dm, n, n, Dinv, n, Dc + 64);
            }
            b.sync();
            for (int e = b.tid; e < N * N; e += b.nthr) ws[w.Jinv + e] = Jinv[e];
            b.sync();
            inverted = true;
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        double* stg = stage ? stage : sc;       // LDS scratch of the tiled inverses: the staging area, or the old inversion scratch
        if (stg) stg = (double*)(((size_t)stg + 15) & ~(size_t)15);    // (16-byte LDS accesses)
        // the staging area is sized for them by layout(); the old inversion scratch holds them from n = 2 on (checked: n = 1 falls through)
        const size_t stg_need = cg_inv_panel_scratch(N, n, b.nthr) + 2;
        const bool stg_fits = stage || (inv_lds && (size_t)(sc - fast) + stg_need <= fast_cap);
        if (!inverted && stg && stg_need > 2 && stg_fits) {
            // larger systems: register-tiled, panel-blocked Gauss-Jordan (every thread a tile of the matrix, two barriers per panel)
            cg_inverse_panel_real(b, da + o.J, N, N, ws + w.Jinv, N, stg);
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            Dinv = ws + w.Dinv;
            cg_inverse_panel_complex(b, da + o.Dm, n, n, Dinv, n, stg);
            inverted = true;
        }
#elif !defined(__HIPCC__)
        if (!inverted) {     // host shim: in-place Gauss-Jordan on a copy
            if (!mN) { mn = cg_div_magic((unsigned)n); mN = cg_div_magic((unsigned)N); }
            std::vector<double> st((size_t)N * N + 4 * N + 64 + N);
            double* vec = st.data() + (size_t)N * N; double* scs = vec + 4 * N; int* rowsrc = (int*)(scs + 64);
            for (int e = 0; e < N * N; ++e) st[e] = da[o.J + e];
            cg_inverse_inplace_real(b, st.data(), N, N, vec, scs, rowsrc, mN);
            cg_inverse_scatter_real(b, st.data(), N, N, rowsrc, ws + w.Jinv, N, mN);
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            for (int e = 0; e < 2 * n * n; ++e) st[e] = da[o.Dm + e];
            cg_inverse_inplace_complex(b, st.data(), n, n, vec, scs, rowsrc, mn);
            Dinv = ws + w.Dinv;
            cg_inverse_scatter_complex(b, st.data(), n, n, rowsrc, Dinv, n, mn);
            inverted = true;
        }
#endif
        if (!inverted) {
            if (!da_fast) {
                for (int e = b.tid; e < N * N; e += b.nthr) Jc[e] = da[o.J + e];
                b.sync();
            }
            (void)cg_inverse_real(b, Jc, N, N, Jinv, N, perm);
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            for (int e = b.tid; e < 2 * n * n; e += b.nthr) Dc[e] = da[o.Dm + e];
            if (inv_lds) for (int e = b.tid; e < N * N; e += b.nthr) ws[w.Jinv + e] = Jinv[e];
            b.sync();
            double la, ar;
            cg_inverse_complex(b, Dc, n, n, Dinv, n, perm, la, ar);
        }
        const double* Dm = da + o.Dm;
        double* Ta = ws + w.Ta; double* Kd = ws + w.Kd; double* gz = ws + w.gz;
        // g_ia = T^a_ii = sum_j D_ij (i k_j^a) Dinv_ji
        for (int e = b.tid; e < N; e += b.nthr) {
            const int i = e / D, a = e - i * D;
            double re = 0, im = 0;
            for (int j = 0; j < n; ++j) {
                const double ka = kocc[j * D + a];
                const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + i)], Dinv[2 * (j * n + i) + 1]});
                re += -ka * p.im; im += ka * p.re;           // (i k) * p
            }
            gz[2 * e] = re; gz[2 * e + 1] = im;
        }
        if (need_T) {
            for (int e = b.tid; e < D * n * n; e += b.nthr) {
                const int a = e / (n * n), r = e - a * n * n, i = r / n, l = r - i * n;
                double re = 0, im = 0;
                for (int j = 0; j < n; ++j) {
                    const double ka = kocc[j * D + a];
                    const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + l)], Dinv[2 * (j * n + l) + 1]});
                    re += -ka * p.im; im += ka * p.re;
                }
                Ta[2 * e] = re; Ta[2 * e + 1] = im;
            }
            for (int e = b.tid; e < D * D * n; e += b.nthr) {
                const int a = e / (D * n), r = e - a * D * n, bb = r / n, i = r - bb * n;
                double re = 0, im = 0;
                for (int j = 0; j < n; ++j) {
                    const double kk = -kocc[j * D + a] * kocc[j * D + bb];
                    const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + i)], Dinv[2 * (j * n + i) + 1]});
                    re += kk * p.re; im += kk * p.im;
                }
                Kd[2 * e] = re; Kd[2 * e + 1] = im;
            }
        }
        b.sync();
    }

    // ------------------------------------------------------------------------------------------------------
    // parameter VJP (reverse pass).  Adjoint scratch layout:
    // ------------------------------------------------------------------------------------------------------
    // T_pq[f][b] non-zeros for pair features pf of r_pq
    struct TCol { double tc[D], ts[D], td[D]; };
    static CG_DEVI void tcols(const typename F::PairF& pf, double c1, double c2c, TCol& t) {
        const double rdel = 1.0 / pf.del;
#pragma unroll
        for (int bb = 0; bb < D; ++bb) { t.tc[bb] = -c1 * pf.s2[bb]; t.ts[bb] = c1 * pf.c2[bb]; t.td[bb] = c2c * pf.s2[bb] * rdel; }
    }

    // One reverse sweep for cotangents (zbar, Jbar); adds the parameter gradient into gw[NP] (gw zeroed by caller).
    static CG_DEVI void reverse(const CgBlk& b, const double* __restrict__ th, int n, double L, double* ws, const Ws& w,
                                const CgFastLds& o, const Adj& A, double* gw, const double* da,
                                double* jslot = nullptr, double* hot = nullptr, size_t hot_cap = 0, bool jac = true) {
        // jac = false: the Jacobian cotangent is zero (imaginary part: log|det J| is real) -- every adjoint that is linear in Jbar
        // is zero-filled instead of computed (J6 ... J2: two of the three pair passes and all the N x 16 contractions)
        // jslot / hot: LDS that is dead during the sweep (the arena's J slot, the set-up's inversion scratch); the adjoint
        // arrays of the pair loops (Jhat; Gbar, Bbar, Vbar, U'bar while they fit) live there instead of in the HBM workspace.
        const int N = n * D;
        double* ad = ws + w.adj;
        size_t hot_left = hot ? hot_cap : 0;
        auto pick = [&](double* dflt, size_t cnt) {
            cnt = (cnt + 1) & ~(size_t)1;
            if (hot_left < cnt) return dflt;
            double* r = hot; hot += cnt; hot_left -= cnt; return r;
        };
        const double* PT = ws + w.pt;
        const double *m0 = da + o.m0, *s1 = da + o.s1, *sg1 = da + o.sg1, *m1 = da + o.m1,
                     *gbar = da + o.gbar, *sg2 = da + o.sg2, *s2 = da + o.s2, *U = da + o.U, *V = da + o.V,
                     *Bm = da + o.Bm, *Up = da + o.Up, *G = da + o.G;
        const double* zbar = ws + w.zbar; const double* Jbar = ws + w.Jbar;
        double* Jhat = jslot ? jslot : ad + A.Jhat;
        double* Gb = pick(ad + A.Gb, (size_t)n * HS * D); double* Bb = pick(ad + A.Bb, (size_t)N * HS);
        double* Vb = pick(ad + A.Vb, (size_t)N * HT); double* Upb = pick(ad + A.Upb, (size_t)N * P);
        double *sg1b = ad + A.sg1b,
               *sg2b = ad + A.sg2b, *Ub = ad + A.Ub, *Rb = ad + A.Rb, *s1b = ad + A.s1b, *s2b = ad + A.s2b, *u2b = ad + A.u2b,
               *u1b = ad + A.u1b, *m1b = ad + A.m1b, *gbb = ad + A.gbb, *pW0 = ad + A.pW0, *pWt = ad + A.pWt, *su2 = ad + A.su2;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);

        if (!jac) {
            for (int e = b.tid; e < N * P; e += b.nthr) Upb[e] = 0.0;
            for (int e = b.tid; e < N * HS; e += b.nthr) { Bb[e] = 0.0; Ub[e] = 0.0; Rb[e] = 0.0; }
            for (int e = b.tid; e < N * HT; e += b.nthr) Vb[e] = 0.0;
            for (int e = b.tid; e < n * HS; e += b.nthr) sg1b[e] = 0.0;
            for (int e = b.tid; e < n * HS * P; e += b.nthr) pW0[e] = 0.0;
            for (int e = b.tid; e < n * HT * (P + 1); e += b.nthr) pWt[e] = 0.0;
            b.sync();
        } else {
        // (J6) J_ii = I - sum_{k!=i} J_ik  =>  Jhat_ik = Jbar_ik - Jbar_ii  (k != i)
        for (int e = b.tid; e < N * N; e += b.nthr) {
            const int r = e / N, c = e - r * N, i = r / D, a = r - i * D, k = c / D, bb = c - k * D;
            Jhat[e] = (i == k) ? 0.0 : Jbar[e] - Jbar[(i * D + a) * N + i * D + bb];
        }
        b.sync();
        // (J5) adjoints that are sums over k for fixed i (pair features from the pair table; the k = i terms vanish with Jhat_ii = 0)
        for (int e = b.tid; e < N * P; e += b.nthr) {              // Upbar_i[a][f] = -sum_k sum_b Jhat_ik[a][b] T_ik[f][b]
            const int r = e / P, f = e - r * P, i = r / D;
            const double* jr = Jhat + (size_t)r * N; const double* pr = PT + (size_t)i * n * PFS;
            double acc = 0;
            if (f < 2 * D) {
                const int bb = f < D ? f : f - D, off = f < D ? D + bb : bb;
                for (int k = 0; k < n; ++k) acc += jr[k * D + bb] * pr[k * PFS + off];
                acc *= f < D ? c1 : -c1;
            } else {
                for (int k = 0; k < n; ++k) {
                    double t = 0;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) t += jr[k * D + bb] * pr[k * PFS + D + bb];
                    acc += t * pr[k * PFS + 2 * D + 1];
                }
                acc *= -c2c;
            }
            Upb[e] = acc;
        }
        // Bbar_i[a][g] = sum_k sum_b Jhat_ik[a][b] G_k[g][b]  and  Gbar_k[g][b] = sum_i sum_a Jhat_ik[a][b] B_i[a][g]: (N x N)(N x 16) on MFMA
        cg_gemm_wg(b, N, HS, N, [&](int r, int c) { return Jhat[r * N + c]; }, [&](int c, int g) { return G[F::iG(c / D, g, c % D)]; },
                   [&](int r, int g, double v) { Bb[r * HS + g] = v; });
        cg_gemm_wg(b, N, HS, N, [&](int c, int r) { return Jhat[r * N + c]; }, [&](int r, int g) { return Bm[F::iB(r / D, r % D, g)]; },
                   [&](int c, int g, double v) { Gb[((c / D) * HS + g) * D + (c % D)] = v; });
        // (J5) pair pass in (i,h) layout: Vbar_i[:,h], and the sigma_t / q_t adjoints -> partial Wtbar / btbar
        for (int e = b.tid; e < n * HT; e += b.nthr) {
            const int i = e / HT, h = e - i * HT;
            double wt[P]; const double bt = th[F::o_t0b + h];
#pragma unroll
            for (int f = 0; f < P; ++f) wt[f] = th[F::o_t0w + f * HT + h];
            double vb[D], pw[P + 1];
#pragma unroll
            for (int a = 0; a < D; ++a) vb[a] = 0;
#pragma unroll
            for (int f = 0; f <= P; ++f) pw[f] = 0;
            for (int k = 0; k < n; ++k) {
                if (k == i) continue;
                PairT t; LP::pt_load(PT, i * n + k, c1, c2c, t);
                double u = bt + wt[2 * D] * t.del, q[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    u += wt[a] * t.c2[a] + wt[D + a] * t.s2[a];
                    q[a] = wt[a] * t.tc[a] + wt[D + a] * t.ts[a] + wt[2 * D] * t.td[a];
                }
                const double sg = sigmoid_only(u), sgp = sg * (1.0 - sg);
                double sgb = 0, qb[D];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) qb[bb] = 0;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double vih = V[F::iV(i, a, h)];
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        const double jh = Jhat[(i * D + a) * N + k * D + bb];
                        vb[a] -= jh * sg * q[bb];
                        sgb -= jh * vih * q[bb];
                        qb[bb] -= jh * vih * sg;
                    }
                }
                const double ub = sgb * sgp;          // adjoint of u_t (from sigma_t)
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    pw[a] += qb[a] * t.tc[a] + ub * t.c2[a];
                    pw[D + a] += qb[a] * t.ts[a] + ub * t.s2[a];
                    pw[2 * D] += qb[a] * t.td[a];
                }
                pw[2 * D] += ub * t.del;
                pw[P] += ub;
            }
#pragma unroll
            for (int a = 0; a < D; ++a) Vb[(i * D + a) * HT + h] = vb[a];
#pragma unroll
            for (int f = 0; f <= P; ++f) pWt[(size_t)e * (P + 1) + f] = pw[f];
        }
        b.sync();
        // (J4) G adjoint, item (p,h): sg1bar_p[h] (first part) and partial W0bar
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int p = e / HS, h = e - p * HS;
            double w_c[D], w_s[D];
#pragma unroll
            for (int a = 0; a < D; ++a) { w_c[a] = th[F::o_W0 + a * HS + h]; w_s[a] = th[F::o_W0 + (D + a) * HS + h]; }
            const double w_d = th[F::o_W0 + 2 * D * HS + h];
            const double sgp = sg1[e] * (rn * rn);
            double gp[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) gp[bb] = Gb[(p * HS + h) * D + bb];
            double sb = 0, pw[P];
#pragma unroll
            for (int f = 0; f < P; ++f) pw[f] = 0;
            for (int q = 0; q < n; ++q) {
                if (q == p) continue;
                PairT t; LP::pt_load(PT, p * n + q, c1, c2c, t);
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const double dG = gp[bb] - Gb[(q * HS + h) * D + bb];
                    sb += dG * (w_c[bb] * t.tc[bb] + w_s[bb] * t.ts[bb] + w_d * t.td[bb]);
                    const double qb = dG * sgp;        // adjoint of q0_pq[h][bb]
                    pw[bb] += qb * t.tc[bb]; pw[D + bb] += qb * t.ts[bb]; pw[2 * D] += qb * t.td[bb];
                }
            }
            sg1b[e] = sb * (rn * rn);
#pragma unroll
            for (int f = 0; f < P; ++f) pW0[(size_t)e * P + f] = pw[f];
        }
        b.sync();
        // (J3) Up_i[a][f] = (1/n) sum_g U_i[a][g] sg1_i[g] W0[f][g]
        for (int e = b.tid; e < N * HS; e += b.nthr) {             // Ubar_i[a][g]
            const int r = e / HS, g = e - r * HS, i = r / D;
            double acc = 0;
#pragma unroll
            for (int f = 0; f < P; ++f) acc += Upb[r * P + f] * th[F::o_W0 + f * HS + g];
            Ub[e] = acc * rn * sg1[i * HS + g];
        }
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // sg1bar_i[g] += (1/n) sum_{a,f} Upb U W0
            const int i = e / HS, g = e - i * HS;
            double acc = 0;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int f = 0; f < P; ++f) acc += Upb[(i * D + a) * P + f] * U[(i * D + a) * HS + g] * th[F::o_W0 + f * HS + g];
            sg1b[e] += acc * rn;
        }
        b.sync();
        // (J2) Rbar_i[a][h] = sum_g Ubar Wa[g][h] + Bbar Wb[g][h] + (1/n) Vbar Wc[g][h]
        cg_gemm_wg(b, N, HS, 2 * HS + HT,       // one (N x (2 HS + HT))((2 HS + HT) x HS) product: the rows of Wa, Wb, Wc are consecutive in theta
                   [&](int r, int k) { return k < HS ? Ub[r * HS + k] : k < 2 * HS ? Bb[r * HS + k - HS] : rn * Vb[r * HT + k - 2 * HS]; },
                   [&](int k, int h) { return th[F::o_Wa + k * HS + h]; }, [&](int r, int h, double v) { Rb[r * HS + h] = v; });
        b.sync();
        }   // jac
        // (J1) sg2bar_i[h] = sum_a Rbar_i[a][h] Wf[h][a];  (F8) s2bar_i[h] = sum_a Wf[h][a] zbar_i[a]
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            double sb = 0, s2 = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) { sb += Rb[(i * D + a) * HS + h] * th[F::o_fw + h * D + a]; s2 += th[F::o_fw + h * D + a] * zbar[i * D + a]; }
            sg2b[e] = sb; s2b[e] = s2;
            // (F7) u2bar = s2bar * sg2 + sg2bar * sg2'
            const double g2 = sg2[e];
            u2b[e] = s2 * g2 + sb * g2 * (1.0 - g2);
        }
        b.sync();
        for (int h = b.tid; h < HS; h += b.nthr) {                 // sum_i u2bar_i[h]
            double acc = 0;
            for (int i = 0; i < n; ++i) acc += u2b[i * HS + h];
            su2[h] = acc;
        }
        b.sync();
        for (int g = b.tid; g < HS; g += b.nthr) {                 // gbarbar[g] = sum_h Wb[g][h] su2[h]
            double acc = 0;
#pragma unroll
            for (int h = 0; h < HS; ++h) acc += th[F::o_Wb + g * HS + h] * su2[h];
            gbb[g] = acc;
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // s1bar, u1bar
            const int i = e / HS, g = e - i * HS;
            double acc = s2b[e] + rn * gbb[g];
#pragma unroll
            for (int h = 0; h < HS; ++h) acc += th[F::o_Wa + g * HS + h] * u2b[i * HS + h];
            s1b[e] = acc;
            const double g1 = sg1[e];
            u1b[e] = acc * g1 + sg1b[e] * g1 * (1.0 - g1);
        }
        for (int e = b.tid; e < n * HT; e += b.nthr) {             // m1bar_i[g] = sum_h Wc[g][h] u2bar_i[h]
            const int i = e / HT, g = e - i * HT;
            double acc = 0;
#pragma unroll
            for (int h = 0; h < HS; ++h) acc += th[F::o_Wc + g * HS + h] * u2b[i * HS + h];
            m1b[e] = acc;
        }
        b.sync();
        // (F4/F5) primal part of the pair stream: utbar_ij[h] = (1/n) m1bar_i[h] sig_t(u_ij[h]);  add to pWt
        for (int e = b.tid; e < n * HT; e += b.nthr) {
            const int i = e / HT, h = e - i * HT;
            double wt[P]; const double bt = th[F::o_t0b + h];
#pragma unroll
            for (int f = 0; f < P; ++f) wt[f] = th[F::o_t0w + f * HT + h];
            double pw[P + 1];
#pragma unroll
            for (int f = 0; f <= P; ++f) pw[f] = 0;
            const double mb = m1b[e] * rn;
            for (int j = 0; j < n; ++j) {
                const double* q = PT + (size_t)(i * n + j) * PFS;
                double u = bt + wt[2 * D] * q[2 * D];
#pragma unroll
                for (int a = 0; a < D; ++a) u += wt[a] * q[a] + wt[D + a] * q[D + a];
                const double ub = mb * sigmoid_only(u);
#pragma unroll
                for (int a = 0; a < D; ++a) { pw[a] += ub * q[a]; pw[D + a] += ub * q[D + a]; }
                pw[2 * D] += ub * q[2 * D]; pw[P] += ub;
            }
#pragma unroll
            for (int f = 0; f <= P; ++f) pWt[(size_t)e * (P + 1) + f] += pw[f];
        }
        b.sync();
        // ---- parameter gradients, one owner thread per parameter (fixed summation order)
        for (int e = b.tid; e < NP; e += b.nthr) {
            double acc = 0;
            if (e < F::o_fw) {                                          // final.b[a]
                const int a = e - F::o_fb;
                for (int i = 0; i < n; ++i) acc += zbar[i * D + a];
            } else if (e < F::o_s0b) {                                  // final.w[h][a]: (F8) + (J1) + direct term of (J2)
                const int r = e - F::o_fw, h = r / D, a = r - h * D;
                for (int i = 0; i < n; ++i)
                    acc += s2[i * HS + h] * zbar[i * D + a] + Rb[(i * D + a) * HS + h] * sg2[i * HS + h] + Ub[(i * D + a) * HS + h];
            } else if (e < F::o_s0w) {                                  // sp0.b[h]
                const int h = e - F::o_s0b;
                for (int i = 0; i < n; ++i) acc += u1b[i * HS + h];
            } else if (e < F::o_s1b) {                                  // sp0.w[f'][h]; rows < 2D multiply zeros
                const int r = e - F::o_s0w, fr = r / HS, h = r - fr * HS;
                if (fr >= 2 * D) {
                    const int f = fr - 2 * D;
                    for (int i = 0; i < n; ++i) {
                        acc += m0[i * P + f] * u1b[i * HS + h] + pW0[((size_t)i * HS + h) * P + f];
#pragma unroll
                        for (int a = 0; a < D; ++a) acc += rn * Upb[(i * D + a) * P + f] * U[(i * D + a) * HS + h] * sg1[i * HS + h];
                    }
                }
            } else if (e < F::o_s1w) {                                  // sp1.b[h]
                acc = su2[e - F::o_s1b];
            } else if (e < F::o_t0b) {                                  // sp1.w rows: Wa (HS), Wb (HS), Wc (HT)
                const int r = e - F::o_s1w, g = r / HS, h = r - g * HS;
                if (g < HS) {
                    for (int i = 0; i < n; ++i) {
                        acc += s1[i * HS + g] * u2b[i * HS + h];
#pragma unroll
                        for (int a = 0; a < D; ++a) acc += Ub[(i * D + a) * HS + g] * th[F::o_fw + h * D + a] * sg2[i * HS + h];
                    }
                } else if (g < 2 * HS) {
                    const int gg = g - HS;
                    acc = gbar[gg] * su2[h];
                    for (int i = 0; i < n; ++i)
#pragma unroll
                        for (int a = 0; a < D; ++a) acc += Bb[(i * D + a) * HS + gg] * th[F::o_fw + h * D + a] * sg2[i * HS + h];
                } else {
                    const int gg = g - 2 * HS;
                    for (int i = 0; i < n; ++i) {
                        acc += m1[i * HT + gg] * u2b[i * HS + h];
#pragma unroll
                        for (int a = 0; a < D; ++a) acc += rn * Vb[(i * D + a) * HT + gg] * th[F::o_fw + h * D + a] * sg2[i * HS + h];
                    }
                }
            } else if (e < F::o_t0w) {                                  // tp0.b[h]
                const int h = e - F::o_t0b;
                for (int i = 0; i < n; ++i) acc += pWt[((size_t)i * HT + h) * (P + 1) + P];
            } else {                                                    // tp0.w[f][h]
                const int r = e - F::o_t0w, f = r / HT, h = r - f * HT;
                for (int i = 0; i < n; ++i) acc += pWt[((size_t)i * HT + h) * (P + 1) + f];
            }
            gw[e] += acc;
        }
        b.sync();
    }

    static CG_DEVI void param_vjp(const CgBlk& b, const double* __restrict__ th, const double* __restrict__ xg,
                                  const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                                  double w_re, double w_im, double* __restrict__ gacc /*NP, nullable*/,
                                  double* __restrict__ score /*NP x 2, nullable*/, double* ws, double* lds, const Layout& lay) {
        const int N = n * D;
        const Ws& w = lay.w;
        const CgFastLds& o = lay.o;
        double* fast = lay.vjp_fast ? lds + lds_doubles(n, b.nthr) : nullptr;
        double* stage = lay.stage ? lds + lds_doubles(n, b.nthr) : nullptr;
        setup(b, th, xg, spk, sidx, n, L, ws, w, o, false, fast, (size_t)lay.vjp_fast, lay.vjp_da != 0, stage, lay.mn, lay.mN);
        const double* da = lay.vjp_da ? fast : ws + w.da;       // primal arena the reverse sweeps read
        const double* Jinv = ws + w.Jinv; const double* gz = ws + w.gz;
        double* zbar = ws + w.zbar; double* Jbar = ws + w.Jbar; double* gw = ws + w.gw;
        const int npass = score ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            const double wr = score ? (pass == 0 ? 1.0 : 0.0) : w_re;
            const double wi = score ? (pass == 0 ? 0.0 : 1.0) : w_im;
            const bool jac = wr != 0.0;                          // the Jacobian cotangent w_re / 2 J^-T vanishes for the imaginary part
            for (int e = b.tid; e < N; e += b.nthr) zbar[e] = wr * gz[2 * e] + wi * gz[2 * e + 1];
            if (jac) for (int e = b.tid; e < N * N; e += b.nthr) { const int al = e / N, be = e - al * N; Jbar[e] = 0.5 * wr * Jinv[be * N + al]; }
            for (int e = b.tid; e < NP; e += b.nthr) gw[e] = 0.0;
            b.sync();
            reverse(b, th, n, L, ws, w, o, lay.a, gw, da, lay.vjp_da ? fast + o.J : nullptr,
                    lay.vjp_da ? fast + o.total : nullptr, lay.vjp_da ? (size_t)lay.vjp_fast - o.total : 0, jac);
            if (score) for (int e = b.tid; e < NP; e += b.nthr) score[2 * e + pass] = gw[e];
            else if (gacc) for (int e = b.tid; e < NP; e += b.nthr) gacc[e] += gw[e];
            b.sync();
        }
    }
};
