// cg_score.hpp -- per-sample scores d log Psi / d theta of the depth-2 flow wave function, second generation (small n).
//
// Reference: make_quantum_score, src/logpsi.py:183-203 (jax.jacrev of log Psi w.r.t. the flow parameters, real and imaginary
// part); consumers: jax.jacrev(quantum_lossfn) (main.py:278) and the quantum Fisher matrix (src/sr.py:62-80).
//
// The first generation (cg_derivs.hpp, still the path of the larger systems) ran the hand-written reverse sweep twice -- once
// seeded with (Re g, 1/2 J^-T), once with (Im g, 0) -- out of a per-workgroup HBM workspace.  Here, for one walker per workgroup
// with everything in LDS:
//   * the imaginary part has NO Jacobian cotangent (log|det J| is real): only the short chain behind z (final layer -> last
//     one-particle layer -> first layer -> pair stream) exists for it, and it shares every loop -- and every sigmoid -- of that
//     chain with the real part (two right-hand sides of one linear map);
//   * pair features come from the pair table of the set-up (CgLap::pt_build) instead of being recomputed per (particle, unit);
//   * the per-(particle, unit) partial sums of the weight gradients are reduced over particles inside the wave (fixed-order
//     butterfly) instead of being parked in n-fold arrays;
//   * array lifetimes are planned (set-up scratch | Jacobian factors + Jhat | adjoints share LDS), ~76 KB at n = 13: two
//     workgroups per CU;
//   * the score row leaves as (re, im) pairs: 16-byte stores, the kernel's only HBM traffic besides x and state_idx.
#pragma once
#include "cg_lap.hpp"

template <int D, int HS, int HT>
struct CgScore {
    using F = CgFast<D, HS, HT>;
    using LP = CgLap<D, HS, HT>;
    using PairT = typename LP::PairT;
    static constexpr int P = F::P;
    static constexpr int NP = F::NPARAM;
    static constexpr int PFS = LP::PFS;
    static constexpr int KT = P + 1;                 // partial row of the two-particle layer: P weights + bias

    struct Lay {
        CgFastLds o;                                 // primal + Jacobian arena, every intermediate kept (offsets in doubles from the LDS base)
        unsigned mn, mN;
        int nw;                                      // waves per workgroup the partial arrays are sized for
        int th, x, kocc, gz, zb, pt, red;
        int Jinv, Dinv, scr;                         // set-up only
        int Jhat, Upb, Bb, Vb, Gb;                   // Jacobian cotangent, adjoints formed from it
        int sg1b, Ub, Rb, u2b, u1b, m1b, su2, gbb;   // real part of the chain
        int u2i, u1i, m1i, su2i, gbbi;               // imaginary part
        int pW0, pWtJ, pWtR, pWtI;                   // per-wave partial sums of the weight gradients
        unsigned total;                              // doubles
        int ok;
    };

    static Lay layout(int n, int nthr, size_t lds_budget_doubles) {
        const size_t N = (size_t)n * D, NN = N * N, nn2 = 2 * (size_t)n * n;
        Lay l; memset(&l, 0, sizeof(l));
        l.mn = cg_div_magic((unsigned)n); l.mN = cg_div_magic((unsigned)N);
        l.nw = nthr >= 64 ? nthr / 64 : 1;
        size_t t = 0;
        auto take = [&](size_t c) { const size_t r = t; t += (c + 1) & ~(size_t)1; return (int)r; };
        CgFastLds& o = l.o;
        // ---- K: alive from the set-up to the assembly
        l.red = take(16); l.th = take(NP);
        l.x = take(N); l.kocc = take(N); l.gz = take(2 * N); l.zb = take(2 * N);
        o.sg1 = take((size_t)n * HS); o.sg2 = take((size_t)n * HS); o.m0 = take((size_t)n * P); o.s1 = take((size_t)n * HS);
        o.m1 = take((size_t)n * HT); o.gbar = take(HS); o.s2 = take((size_t)n * HS); o.U = take(N * HS);
        o.wt = take(HT * (P + 1) + HS * D);
        l.pt = take((size_t)n * n * PFS);
        l.u2i = take((size_t)n * HS); l.u1i = take((size_t)n * HS); l.m1i = take((size_t)n * HT); l.su2i = take(HS); l.gbbi = take(HS);
        l.pW0 = take((size_t)l.nw * HS * P); l.pWtJ = take((size_t)l.nw * HT * KT); l.pWtR = take((size_t)l.nw * HT * KT); l.pWtI = take((size_t)l.nw * HT * KT);
        // ---- X1: set-up scratch; the adjoints formed from Jhat overlay it (Jinv is dead once Jhat exists)
        const size_t X1 = t;
        o.sh = take(N); o.ch = take(N); o.z = take(N); o.cb = take(HS); o.perm = take(4); o.Up = take(N * P);
        o.Dm = take(nn2); o.lus = take(64); l.Jinv = take(NN); l.Dinv = take(nn2); l.scr = take(128);
        size_t X1_end = t;
        t = X1;
        l.Upb = take(N * P); l.Bb = take(N * HS); l.Vb = take(N * HT); l.Gb = take((size_t)n * HS * D);
        X1_end = t > X1_end ? t : X1_end;
        // ---- X2: Jacobian factors and J (-> Jhat in J's slot); the rest of the chain overlays them after the Jhat phase
        t = X1_end;
        const size_t X2 = t;
        o.V = take((size_t)n * (HT * D + 2)); o.Bm = take((size_t)n * (HS * D + 2)); o.G = take((size_t)n * (HS * D + 2)); o.J = take(NN);
        l.Jhat = o.J;
        size_t X2_end = t;
        t = X2;
        l.sg1b = take((size_t)n * HS); l.Ub = take(N * HS); l.Rb = take(N * HS); l.u2b = take((size_t)n * HS); l.u1b = take((size_t)n * HS);
        l.m1b = take((size_t)n * HT); l.su2 = take(HS); l.gbb = take(HS);
        X2_end = t > X2_end ? t : X2_end;
        o.total = (int)X2_end; o.wave_lu = 0; o.dual = 0;
        l.total = (unsigned)X2_end;
        // wave-level inverses + MFMA set-up: N <= 32, n <= 16; index arithmetic by multiply-shift: N^2 < 65536
        l.ok = (N <= 32 && n <= 16 && nn2 >= 128 && nthr >= 128 && (nthr % 64) == 0 && l.total <= lds_budget_doubles) ? 1 : 0;
        return l;
    }

    // Per-wave partial sums of the weight gradients.  Device: a lane accumulates its (i, h) items in registers; after its loop the lanes of
    // a wave that hold the same unit h (lane = 16 (i mod 4) + h, spsize = tpsize = 16) are summed and lanes < 16 store the K values to
    // this wave's row (wave_rows_store).  Host build (one thread walks every item): each item is added to row 0 (item_flush; the rows
    // are zeroed by rows_zero), wave_rows_store has nothing left to do.
    template <int K>
    static CG_DEVI void rows_zero(double* part) {
        if (!CG_ON_DEVICE) for (int e = 0; e < 16 * K; ++e) part[e] = 0.0;
    }
    template <int K>
    static CG_DEVI void item_flush(double (&pw)[K], int h, double* part) {
        if (!CG_ON_DEVICE) for (int f = 0; f < K; ++f) { part[h * K + f] += pw[f]; pw[f] = 0; }
    }
    template <int K>
    static CG_DEVI void wave_rows_store(const CgBlk& b, double (&pw)[K], double* part) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int lane = b.tid & 63, wave = b.tid >> 6;
#pragma unroll
        for (int f = 0; f < K; ++f) {
            double v = pw[f];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if (lane < 16) part[(wave * 16 + lane) * K + f] = v;
        }
#else
        (void)b; (void)pw; (void)part;
#endif
    }

    // ------------------------------------------------------------------------------------------------------
    // set-up: z, every intermediate of the flow and of the Jacobian assembly, the pair table, J^-1, D^-1 -> g_ia = T^a_ii
    // ------------------------------------------------------------------------------------------------------
    static CG_DEVI void setup(const CgBlk& b, const double* th, const double* __restrict__ xg, const double* __restrict__ spk,
                              const int* __restrict__ sidx, int n, double L, double* lds, const Lay& l) {
        const int N = n * D;
        const CgFastLds& o = l.o;
        double* x = lds + l.x; double* kocc = lds + l.kocc;
        for (int e = b.tid; e < N; e += b.nthr) {
            x[e] = xg[e];
            const int j = e / D;
            kocc[e] = spk[(size_t)sidx[j] * D + (e - j * D)];
        }
        b.sync();
        typename F::WFrag wfrag;
        const typename F::WFrag* wf = F::frags(th, wfrag);                 // MFMA / DPP path of the sampler (device, 16 / 16), else the scalar one
        F::primal(b, th, (const double*)x, n, L, lds, o, wf);
        LP::pt_build(b, lds + o.sh, lds + o.ch, n, l.mn, lds + l.pt);
        F::jacobian(b, th, n, L, lds, o, wf);
        double* Jinv = lds + l.Jinv; double* Dinv = lds + l.Dinv; double* scr = lds + l.scr;
        F::slater_matrix(b, lds + o.z, kocc, nullptr, n, lds + o.Dm);
#if defined(__HIP_DEVICE_COMPILE__)
        {   // both inverses by wave-level Gauss-Jordan in registers, concurrently on two waves (no barriers, J and D intact)
            const int wave = b.tid >> 6;
            if (wave == 0) {
                if (N == 26) cg_wave_inverse_real<26>(lds + o.J, N, N, Jinv, N, scr);
                else cg_wave_inverse_real<32>(lds + o.J, N, N, Jinv, N, scr);
            } else if (wave == 1) {
                if (n == 13) cg_wave_inverse_complex<13>(lds + o.Dm, n, n, Dinv, n, scr + 64);
                else cg_wave_inverse_complex<16>(lds + o.Dm, n, n, Dinv, n, scr + 64);
            }
            b.sync();
        }
#elif !defined(__HIPCC__)
        {   // host shim: the LDS Gauss-Jordan of cg_linalg.hpp on copies (J's slot becomes Jhat, D is needed below)
            std::vector<double> Jc(lds + o.J, lds + o.J + (size_t)N * N), Dc(lds + o.Dm, lds + o.Dm + 2 * (size_t)n * n);
            std::vector<int> perm(N + 64);
            (void)cg_inverse_real(b, Jc.data(), N, N, Jinv, N, perm.data());
            double la, ar;
            cg_inverse_complex(b, Dc.data(), n, n, Dinv, n, perm.data(), la, ar);
        }
#endif
        LP::slater_g(b, n, lds + o.Dm, Dinv, kocc, lds + l.gz, lds + l.zb);      // g_ia = T^a_ii: cotangents of z, real and imaginary part
        b.sync();
    }
    // (fused kernel k_grad_lap2_scores) instead of setup(): the same arrays, read from what CgLap's set-up parked in the workspace.  The slot
    // mirrors this layout (stash_of), and what the sweep reads lies in three runs of it: [zb, sg1, sg2, m0, s1, m1, gbar, s2, U, (wt), pt],
    // [V, Bm, G] and J^-1 -- three flat copies with 16-byte accesses.
    static typename LP::Stash stash_of(const Lay& l) {
        typename LP::Stash st;
        const CgFastLds& o = l.o;
        st.m0 = o.m0; st.s1 = o.s1; st.sg1 = o.sg1; st.m1 = o.m1; st.gbar = o.gbar; st.sg2 = o.sg2; st.s2 = o.s2; st.U = o.U; st.V = o.V; st.Bm = o.Bm; st.G = o.G;
        st.pt = l.pt; st.Jinv = l.Jinv; st.zb = l.zb; st.total = l.total;
        return st;
    }
    static CG_DEVI void unstash(const CgBlk& b, int n, double* lds, const Lay& l, const double* stash) {
        const int N = n * D;
        const CgFastLds& o = l.o;
        auto ev = [](int v) { return (v + 1) & ~1; };
#if defined(__HIPCC__)
        typedef double d2_t __attribute__((ext_vector_type(2)));
        auto run = [&](int from, int to) {
            for (int e = from + 2 * b.tid; e < to; e += 2 * b.nthr) *(d2_t*)(lds + e) = *(const d2_t*)(stash + e);
        };
#else
        auto run = [&](int from, int to) { for (int e = from + b.tid; e < to; e += b.nthr) lds[e] = stash[e]; };
#endif
        run(l.zb, l.pt + ev(n * n * PFS));
        run(o.V, o.G + ev(n * (HS * D + 2)));
        run(l.Jinv, l.Jinv + ev(N * N));
        b.sync();
    }

    // ------------------------------------------------------------------------------------------------------
    // reverse sweep for both parts + assembly of the score row.  Phase names follow CgDerivs::reverse.
    // ------------------------------------------------------------------------------------------------------
    static CG_DEVI void sweep(const CgBlk& b, const double* th, int n, double L, double* lds, const Lay& l,
                              double* __restrict__ score /* NP x 2 */) {
        const int N = n * D;
        const CgFastLds& o = l.o;
        const double *m0 = lds + o.m0, *s1 = lds + o.s1, *sg1 = lds + o.sg1, *m1 = lds + o.m1, *gbar = lds + o.gbar, *sg2 = lds + o.sg2,
                     *s2 = lds + o.s2, *U = lds + o.U, *V = lds + o.V, *Bm = lds + o.Bm, *G = lds + o.G;
        const double* PT = lds + l.pt; const double* Jinv = lds + l.Jinv;
        const double* zr = lds + l.zb; const double* zi = zr + N;
        double *Jhat = lds + l.Jhat, *Upb = lds + l.Upb, *Bb = lds + l.Bb, *Vb = lds + l.Vb, *Gb = lds + l.Gb, *sg1b = lds + l.sg1b, *Ub = lds + l.Ub,
               *Rb = lds + l.Rb, *u2b = lds + l.u2b, *u1b = lds + l.u1b, *m1b = lds + l.m1b, *su2 = lds + l.su2, *gbb = lds + l.gbb,
               *u2i = lds + l.u2i, *u1i = lds + l.u1i, *m1i = lds + l.m1i, *su2i = lds + l.su2i, *gbbi = lds + l.gbbi,
               *pW0 = lds + l.pW0, *pWtJ = lds + l.pWtJ, *pWtR = lds + l.pWtR, *pWtI = lds + l.pWtI;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        const unsigned mN = l.mN;
        rows_zero<P>(pW0); rows_zero<KT>(pWtJ); rows_zero<KT>(pWtR); rows_zero<KT>(pWtI);          // (host build only)
        // (J6) J_ii = I - sum_{k!=i} J_ik  =>  Jhat_ik = Jbar_ik - Jbar_ii (k != i),  Jbar = 1/2 J^-T;  Jhat_ii = 0
        for (int e = b.tid; e < N * N; e += b.nthr) {
            const int r = cg_udiv(e, mN), c = e - r * N, i = r / D, k = c / D, bb = c - k * D;
            Jhat[e] = (i == k) ? 0.0 : 0.5 * (Jinv[c * N + r] - Jinv[(i * D + bb) * N + r]);
        }
        b.sync();
        // (J5) adjoints that are sums over k for fixed i
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
        {   // (J5) pair pass in (i,h) layout: Vbar_i[:,h] and the sigma_t / q_t adjoints -> partial Wtbar / btbar (Jacobian part)
            double pw[KT];
#pragma unroll
            for (int f = 0; f < KT; ++f) pw[f] = 0;
            for (int e = b.tid; e < n * HT; e += b.nthr) {
                const int i = e / HT, h = e - i * HT;
                double wt[P]; const double bt = th[F::o_t0b + h];
#pragma unroll
                for (int f = 0; f < P; ++f) wt[f] = th[F::o_t0w + f * HT + h];
                double vb[D];
#pragma unroll
                for (int a = 0; a < D; ++a) vb[a] = 0;
                double vi[D];
#pragma unroll
                for (int a = 0; a < D; ++a) vi[a] = V[F::iV(i, a, h)];
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
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) {
                            const double jh = Jhat[(i * D + a) * N + k * D + bb];
                            vb[a] -= jh * sg * q[bb];
                            sgb -= jh * vi[a] * q[bb];
                            qb[bb] -= jh * vi[a] * sg;
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
                item_flush<KT>(pw, h, pWtJ);
            }
            wave_rows_store<KT>(b, pw, pWtJ);
        }
        b.sync();
        {   // (J4) G adjoint, item (p,h): sg1bar_p[h] (first part) and partial W0bar;  (J3) second part of sg1bar
            double pw[P];
#pragma unroll
            for (int f = 0; f < P; ++f) pw[f] = 0;
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
                double sb = 0;
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
                double acc = 0;
#pragma unroll
                for (int a = 0; a < D; ++a)
#pragma unroll
                    for (int f = 0; f < P; ++f) acc += Upb[(p * D + a) * P + f] * U[(p * D + a) * HS + h] * th[F::o_W0 + f * HS + h];
                sg1b[e] = sb * (rn * rn) + acc * rn;
                item_flush<P>(pw, h, pW0);
            }
            wave_rows_store<P>(b, pw, pW0);
        }
        for (int e = b.tid; e < N * HS; e += b.nthr) {             // (J3) Ubar_i[a][g]
            const int r = e / HS, g = e - r * HS, i = r / D;
            double acc = 0;
#pragma unroll
            for (int f = 0; f < P; ++f) acc += Upb[r * P + f] * th[F::o_W0 + f * HS + g];
            Ub[e] = acc * rn * sg1[i * HS + g];
        }
        b.sync();
        // (J2) Rbar_i[a][h] = sum_g Ubar Wa[g][h] + Bbar Wb[g][h] + (1/n) Vbar Wc[g][h]
        cg_gemm_wg(b, N, HS, 2 * HS + HT,       // one (N x (2 HS + HT))((2 HS + HT) x HS) product: the rows of Wa, Wb, Wc are consecutive in theta
                   [&](int r, int k) { return k < HS ? Ub[r * HS + k] : k < 2 * HS ? Bb[r * HS + k - HS] : rn * Vb[r * HT + k - 2 * HS]; },
                   [&](int k, int h) { return th[F::o_Wa + k * HS + h]; }, [&](int r, int h, double v) { Rb[r * HS + h] = v; });
        b.sync();
        // (J1) sg2bar_i[h] = sum_a Rbar_i[a][h] Wf[h][a];  (F8) s2bar_i[h] = sum_a Wf[h][a] zbar_i[a];  (F7) u2bar = s2bar sg2 + sg2bar sg2'
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            double sb = 0, sr = 0, si = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const double wf = th[F::o_fw + h * D + a];
                sb += Rb[(i * D + a) * HS + h] * wf; sr += wf * zr[i * D + a]; si += wf * zi[i * D + a];
            }
            const double g2 = sg2[e];
            u2b[e] = sr * g2 + sb * g2 * (1.0 - g2);
            u2i[e] = si * g2;
        }
        b.sync();
        for (int h = b.tid; h < 2 * HS; h += b.nthr) {             // sum_i u2bar_i[h], both parts
            const double* src = h < HS ? u2b : u2i; const int hh = h < HS ? h : h - HS;
            double acc = 0;
            for (int i = 0; i < n; ++i) acc += src[i * HS + hh];
            (h < HS ? su2 : su2i)[hh] = acc;
        }
        b.sync();
        for (int g = b.tid; g < 2 * HS; g += b.nthr) {             // gbarbar[g] = sum_h Wb[g][h] su2[h]
            const double* src = g < HS ? su2 : su2i; const int gg = g < HS ? g : g - HS;
            double acc = 0;
#pragma unroll
            for (int h = 0; h < HS; ++h) acc += th[F::o_Wb + gg * HS + h] * src[h];
            (g < HS ? gbb : gbbi)[gg] = acc;
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // s1bar, u1bar
            const int i = e / HS, g = e - i * HS;
            double sr = 0, si = 0;                                  // s2bar_i[g] again (2 D multiply-adds instead of two arrays)
#pragma unroll
            for (int a = 0; a < D; ++a) { const double wf = th[F::o_fw + g * D + a]; sr += wf * zr[i * D + a]; si += wf * zi[i * D + a]; }
            double ar = sr + rn * gbb[g], ai = si + rn * gbbi[g];
#pragma unroll
            for (int h = 0; h < HS; ++h) { const double wa = th[F::o_Wa + g * HS + h]; ar += wa * u2b[i * HS + h]; ai += wa * u2i[i * HS + h]; }
            const double g1 = sg1[e];
            u1b[e] = ar * g1 + sg1b[e] * g1 * (1.0 - g1);
            u1i[e] = ai * g1;
        }
        for (int e = b.tid; e < n * HT; e += b.nthr) {             // m1bar_i[g] = sum_h Wc[g][h] u2bar_i[h]
            const int i = e / HT, g = e - i * HT;
            double ar = 0, ai = 0;
#pragma unroll
            for (int h = 0; h < HS; ++h) { const double wc = th[F::o_Wc + g * HS + h]; ar += wc * u2b[i * HS + h]; ai += wc * u2i[i * HS + h]; }
            m1b[e] = ar; m1i[e] = ai;
        }
        b.sync();
        {   // (F4/F5) primal part of the pair stream: utbar_ij[h] = (1/n) m1bar_i[h] sig_t(u_ij[h]), both parts from one sigmoid
            double pr[KT], pi[KT];
#pragma unroll
            for (int f = 0; f < KT; ++f) { pr[f] = 0; pi[f] = 0; }
            for (int e = b.tid; e < n * HT; e += b.nthr) {
                const int i = e / HT, h = e - i * HT;
                double wt[P]; const double bt = th[F::o_t0b + h];
#pragma unroll
                for (int f = 0; f < P; ++f) wt[f] = th[F::o_t0w + f * HT + h];
                const double mr = m1b[e] * rn, mi = m1i[e] * rn;
                double a[KT];                                       // sum_j sig_t(u_ij[h]) [features, 1]: the same for both parts
#pragma unroll
                for (int f = 0; f < KT; ++f) a[f] = 0;
                for (int j = 0; j < n; ++j) {
                    const double* q = PT + (size_t)(i * n + j) * PFS;
                    double u = bt + wt[2 * D] * q[2 * D];
#pragma unroll
                    for (int aa = 0; aa < D; ++aa) u += wt[aa] * q[aa] + wt[D + aa] * q[D + aa];
                    const double sg = sigmoid_only(u);
#pragma unroll
                    for (int aa = 0; aa < D; ++aa) { a[aa] += sg * q[aa]; a[D + aa] += sg * q[D + aa]; }
                    a[2 * D] += sg * q[2 * D]; a[P] += sg;
                }
#pragma unroll
                for (int f = 0; f < KT; ++f) { pr[f] += mr * a[f]; pi[f] += mi * a[f]; }
                item_flush<KT>(pr, h, pWtR); item_flush<KT>(pi, h, pWtI);
            }
            wave_rows_store<KT>(b, pr, pWtR); wave_rows_store<KT>(b, pi, pWtI);
        }
        b.sync();
        // ---- the score row, one owner thread per parameter (fixed summation order), real and imaginary part side by side
        const int nwp = b.waves();
        for (int e = b.tid; e < NP; e += b.nthr) {
            double ar = 0, ai = 0;
            if (e < F::o_fw) {                                          // final.b[a]
                const int a = e - F::o_fb;
                for (int i = 0; i < n; ++i) { ar += zr[i * D + a]; ai += zi[i * D + a]; }
            } else if (e < F::o_s0b) {                                  // final.w[h][a]: (F8) + (J1) + direct term of (J2)
                const int r = e - F::o_fw, h = r / D, a = r - h * D;
                for (int i = 0; i < n; ++i) {
                    const double sv = s2[i * HS + h];
                    ar += sv * zr[i * D + a] + Rb[(i * D + a) * HS + h] * sg2[i * HS + h] + Ub[(i * D + a) * HS + h];
                    ai += sv * zi[i * D + a];
                }
            } else if (e < F::o_s0w) {                                  // sp0.b[h]
                const int h = e - F::o_s0b;
                for (int i = 0; i < n; ++i) { ar += u1b[i * HS + h]; ai += u1i[i * HS + h]; }
            } else if (e < F::o_s1b) {                                  // sp0.w[f'][h]; rows < 2D multiply zeros
                const int r = e - F::o_s0w, fr = r / HS, h = r - fr * HS;
                if (fr >= 2 * D) {
                    const int f = fr - 2 * D;
                    for (int w = 0; w < nwp; ++w) ar += pW0[(w * HS + h) * P + f];
                    for (int i = 0; i < n; ++i) {
                        const double mv = m0[i * P + f];
                        ar += mv * u1b[i * HS + h]; ai += mv * u1i[i * HS + h];
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += rn * Upb[(i * D + a) * P + f] * U[(i * D + a) * HS + h] * sg1[i * HS + h];
                    }
                }
            } else if (e < F::o_s1w) {                                  // sp1.b[h]
                ar = su2[e - F::o_s1b]; ai = su2i[e - F::o_s1b];
            } else if (e < F::o_t0b) {                                  // sp1.w rows: Wa (HS), Wb (HS), Wc (HT)
                const int r = e - F::o_s1w, g = r / HS, h = r - g * HS;
                if (g < HS) {
                    for (int i = 0; i < n; ++i) {
                        const double sv = s1[i * HS + g];
                        ar += sv * u2b[i * HS + h]; ai += sv * u2i[i * HS + h];
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += Ub[(i * D + a) * HS + g] * th[F::o_fw + h * D + a] * sg2[i * HS + h];
                    }
                } else if (g < 2 * HS) {
                    const int gg = g - HS;
                    ar = gbar[gg] * su2[h]; ai = gbar[gg] * su2i[h];
                    for (int i = 0; i < n; ++i)
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += Bb[(i * D + a) * HS + gg] * th[F::o_fw + h * D + a] * sg2[i * HS + h];
                } else {
                    const int gg = g - 2 * HS;
                    for (int i = 0; i < n; ++i) {
                        const double mv = m1[i * HT + gg];
                        ar += mv * u2b[i * HS + h]; ai += mv * u2i[i * HS + h];
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += rn * Vb[(i * D + a) * HT + gg] * th[F::o_fw + h * D + a] * sg2[i * HS + h];
                    }
                }
            } else if (e < F::o_t0w) {                                  // tp0.b[h]
                const int h = e - F::o_t0b;
                for (int w = 0; w < nwp; ++w) { ar += pWtJ[(w * HT + h) * KT + P] + pWtR[(w * HT + h) * KT + P]; ai += pWtI[(w * HT + h) * KT + P]; }
            } else {                                                    // tp0.w[f][h]
                const int r = e - F::o_t0w, f = r / HT, h = r - f * HT;
                for (int w = 0; w < nwp; ++w) { ar += pWtJ[(w * HT + h) * KT + f] + pWtR[(w * HT + h) * KT + f]; ai += pWtI[(w * HT + h) * KT + f]; }
            }
            score[2 * e] = ar; score[2 * e + 1] = ai;
        }
    }

    static CG_DEVI void scores(const CgBlk& b, const double* th, const double* __restrict__ xg, const double* __restrict__ spk,
                               const int* __restrict__ sidx, int n, double L, double* __restrict__ score, double* lds, const Lay& l) {
        CG_STAMP_START(20)
        setup(b, th, xg, spk, sidx, n, L, lds, l);
        CG_STAMP_END(20)
        CG_STAMP_START(21)
        sweep(b, th, n, L, lds, l, score);
        CG_STAMP_END(21)
    }
};
