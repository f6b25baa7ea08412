// cg_lap.hpp -- grad_x log Psi and laplacian_x log Psi of the depth-2 flow wave function, second generation.
//
// Reference: make_logpsi_grad_laplacian, src/logpsi.py:55-172 (exact :77-106, Hutchinson :112-132, Hutchinson-split
// :134-164 -- the variant main.py:255-256 selects for every shipped run).  The reference nests AD transforms; the first
// generation of this kernel (cg_derivs.hpp) pushed one second-order jet per coordinate direction through the whole flow +
// Jacobian assembly (n d + 1 passes).  Here every piece is taken in the cheapest exact form:
//
//   grad_x log Psi      = J^T g  +  grad_x 1/2 log|det J|            g_ia = d log phi / d z_ia = T^a_ii   (App. A.3)
//                         the second term by ONE reverse sweep through the structured Jacobian assembly with the
//                         cotangent Jbar = 1/2 J^-T  (what jax.grad(logjacdet) does, src/logpsi.py:137-138)
//   lap_x log phi(z(x)) = tr(J^T H J) + g . lap_x z                  H = d2 log phi / dz dz (closed form, App. A.3)
//                         lap_x z by a forward-Laplacian pass through the flow: every hidden unit carries its value, the
//                         squared norm of its x-gradient and its Laplacian; for the depth-2 net only the last layer's
//                         pre-activations need their dense x-gradient (n HS x n d), which is a small GEMM
//   v^T hess(1/2 log|det J|) v : one second-order jet pass along the probe v (Hutchinson modes), or n d basis passes
//                         (exact mode; third derivatives of the flow are inherently that expensive)
//
// Modes (include/coulombgas.h): 0 exact, 1 Hutchinson (v^T hess(log Psi) v), 2 Hutchinson-split.
//
// Memory: one workgroup per walker.  The arrays are grouped into three blocks (Lay), each placed on the host either in
// LDS or in the per-workgroup HBM workspace.  When everything fits the LDS budget the kernel is instantiated with
// AL = true and every access is a ds_ instruction.
#pragma once
#include "cg_flow_fast.hpp"
// unroll factors of the pair loops (independent sigmoid chains per trip hide the LDS / transcendental latencies at 2 waves per SIMD)
#ifndef CG_UNR_K
#define CG_UNR_K 2
#endif
#ifndef CG_UNR_H
#define CG_UNR_H 2
#endif
#ifndef CG_UNR_Q
#define CG_UNR_Q 2
#endif
#ifndef CG_UNR_G
#define CG_UNR_G 4
#endif

template <int D, int HS, int HT>
struct CgLap {
    using F = CgFast<D, HS, HT>;
    static constexpr int P = F::P;
    static constexpr int NP = F::NPARAM;
    static constexpr int HM = HT > P ? HT : P;
    static constexpr int PFS = 2 * D + 2;        // doubles per ordered pair in the pair table: c2[D], s2[D], del, 1/del

    // Three blocks, each wholly in LDS or wholly in the per-workgroup HBM workspace (decided on the host):
    //   P  persistent: x, g, xbar, J^-1 (and T^a, K^ab in mode 1: the probe pass needs them)
    //   A  set-up, Slater part, reverse sweep, forward Laplacian (sub-phases overlay each other inside the block)
    //   B  jet passes; overlays A (the primal arena is dead by then)
    // Array offsets are doubles relative to their block.
    struct Lay {
        CgFastLds o;        // primal + Jacobian arena inside A, ordered by lifetime (R1 | R3 | R2, see layout())
        CgFastLds oj;       // Jet2 arena of the directional passes (aliased sampler layout) inside B
        int all_lds;        // 1: every block is in LDS
        int th_lds, th;     // 1: theta is copied to LDS at doubles offset th
        int P_lds, A_lds, B_lds;             // block placement
        unsigned P_off, A_off, B_off;        // block base (doubles) in its pool
        int stage_lds; unsigned stage;       // N > 32: scratch of the register-tiled inverses (published panels, pivot rows, bookkeeping)
        unsigned mn, mN;    // multiply-shift constants of e / n and e / (n D) (cg_div_magic)
        // P
        int red, x, gz, xbar, Jinv, Ta, Kd, TaKd_in_P;
        // A
        int da, pt, kocc, Jc, Dc, Dinv, perm, C;
        int Jhat, Upb, Vb, Bb, Gb, sg1b, sg2b, Ub, Rb, u2b, u1b, s1b, m1b, gbb, su2, m0b, rbar;
        int Lm0, gu1, Ls1, Lm1, Lgb, Am, Hk, SQ, Er, Su2, Ls2;
        int eLm0, egu1, eLm1;              // the forward Laplacian's pair sums when formed early (by the waves that idle during the inverses)
        // B
        int ja, xj, M, M_in_arena;
        unsigned lds_total, ws_total;      // doubles
    };

    // lds_budget_doubles: LDS doubles available to this kernel's arrays (excluding the exp/log tables).
    static Lay layout(int n, int nthr, int mode, size_t lds_budget_doubles, bool theta_in_lds = true) {
        const size_t N = (size_t)n * D, NN = N * N, nn2 = 2 * (size_t)n * n;
        Lay l; memset(&l, 0, sizeof(l));
        l.mn = cg_div_magic((unsigned)n); l.mN = cg_div_magic((unsigned)N);
        {   // jet arena: the aliased sampler layout (cg_fast_layout(alias = true)) with the weight scratch sized in doubles
            CgFastLds& j = l.oj; int t = 0;
            auto tk = [&](int cnt) { int r = t; t += (cnt + 1) & ~1; return r; };
            j.sh = tk(n * D); j.ch = tk(n * D); j.z = tk(n * D); j.sg1 = tk(n * HS); j.sg2 = tk(n * HS);
            j.perm = tk(2); j.wt = tk((HT * (P + 1) + 2) / 3 + 1);        // HT (P + 1) doubles inside an arena of Jet2 elements
            const int base = t;
            j.m0 = tk(n * P); j.s1 = tk(n * HS); j.m1 = tk(n * HT); j.gbar = tk(HS); j.cb = tk(HS); j.s2 = tk(n * HS);
            const int end_primal = t;
            t = base;
            j.V = tk(n * (HT * D + 2)); j.Bm = tk(n * (HS * D + 2)); j.Up = tk(n * D * P); j.G = tk(n * (HS * D + 2));
            const int end_jac = t;
            t = end_primal > end_jac ? end_primal : end_jac;
            j.J = tk(n * D * n * D);
            j.U = j.J;                                                    // U is dead once Up is formed
            if (n * D * HS > n * D * n * D) j.U = tk(n * D * HS);
            j.Dm = j.J; j.lus = base; j.wave_lu = 0;
            j.total = t;
        }
        auto ev = [](size_t c) { return (c + 1) & ~(size_t)1; };
        size_t t = 0;
        auto take = [&](size_t c) { const size_t r = t; t += ev(c); return (int)r; };
        // ---- P
        l.red = take(8 * (size_t)(nthr / 64 + 1)); l.x = take(N); l.gz = take(2 * N); l.xbar = take(N); l.Jinv = take(NN);
        l.TaKd_in_P = mode == 1 ? 1 : 0;
        if (l.TaKd_in_P) { l.Ta = take(2 * (size_t)D * n * n); l.Kd = take(2 * (size_t)D * D * n); }
        const size_t P_size = t;
        // ---- A: arena R1 (alive through the whole of A)
        t = 0;
        l.da = 0;
        CgFastLds& o = l.o;
        o.sh = take(N); o.ch = take(N); o.z = take(N); o.sg1 = take((size_t)n * HS); o.sg2 = take((size_t)n * HS);
        o.perm = take(2); o.wt = take(HT * (P + 1) + HS * D);
        o.U = take(N * HS); o.V = take((size_t)n * (HT * D + 2)); o.Bm = take((size_t)n * (HS * D + 2)); o.Up = take(N * P);
        o.G = take((size_t)n * (HS * D + 2));
        // pair table: the features of every ordered pair (i, k), computed ONCE per walker in the set-up; every pair loop of the
        // reverse sweep and of the forward Laplacian reads a row of it instead of recomputing ~90 instructions per pair
        l.pt = take((size_t)n * n * PFS);
        l.kocc = take(N);                                   // wave vectors of the walker's occupied orbitals (set-up, Slater part)
        const size_t A1 = t;
        // R3: J, Slater matrix (dead after the Slater part);  R2: primal temporaries;  then the set-up scratch
        o.J = take(NN); o.Dm = take(nn2); o.lus = take(2);
        const size_t A2 = t;
        o.m0 = take((size_t)n * P); o.s1 = take((size_t)n * HS); o.m1 = take((size_t)n * HT); o.gbar = take(HS); o.cb = take(HS); o.s2 = take((size_t)n * HS);
        o.total = (int)t; o.wave_lu = 0;
        l.Jc = take(NN); l.Dc = take(nn2); l.Dinv = take(nn2); l.perm = take(N + 42);
        // T^a, diag K^ab live from the set-up to the Slater part only (modes 0, 2): behind the set-up scratch, under the adjoints
        if (!l.TaKd_in_P) { l.Ta = take(2 * (size_t)D * n * n); l.Kd = take(2 * (size_t)D * D * n); }
        const size_t setup_end = t;
        size_t A_size = t;
        t = A2; l.C = take(NN); A_size = t > A_size ? t : A_size;
        t = A1;
        l.Jhat = take(NN);
        l.Upb = take(N * P); l.Vb = take(N * HT); l.Bb = take(N * HS); l.Gb = take((size_t)n * HS * D);
        l.sg1b = take((size_t)n * HS); l.sg2b = take((size_t)n * HS); l.Ub = take(N * HS); l.Rb = take(N * HS);
        l.u2b = take((size_t)n * HS); l.u1b = take((size_t)n * HS); l.s1b = take((size_t)n * HS); l.m1b = take((size_t)n * HT);
        l.gbb = take(HS); l.su2 = take(HS); l.m0b = take((size_t)n * P); l.rbar = take((size_t)n * n * D);
        A_size = t > A_size ? t : A_size;
        t = A1;
        l.Lm0 = take((size_t)n * P); l.gu1 = take((size_t)n * HS); l.Ls1 = take((size_t)n * HS); l.Lm1 = take((size_t)n * HT);
        l.Lgb = take(HS); l.Am = take((size_t)n * HS * P); l.Hk = take((size_t)n * HS * D);
        l.SQ = take((size_t)n * HT * D); l.Er = take((size_t)n * HS * D); l.Su2 = take((size_t)n * HS); l.Ls2 = take((size_t)n * HS);
        A_size = t > A_size ? t : A_size;
        // the early pair sums: behind everything the set-up, the Slater part AND the forward Laplacian use (the reverse sweep, which
        // runs last, overlays them)
        t = t > setup_end ? t : setup_end;
        l.eLm0 = take((size_t)n * P); l.egu1 = take((size_t)n * HS); l.eLm1 = take((size_t)n * HT);
        A_size = t > A_size ? t : A_size;
        // ---- B
        t = 0;
        l.xj = take(3 * N); l.ja = take(3 * (size_t)l.oj.total);
        // M = J^-1 J' (N x N doubles) fits the per-particle factor slots of the jet arena, dead once its J is assembled
        l.M_in_arena = 3 * (size_t)(l.oj.J - l.oj.m0) >= NN ? 1 : 0;
        l.M = l.M_in_arena ? l.ja + 3 * l.oj.m0 : take(NN);
        const size_t B_size = t;
        // ---- placement
        const size_t AB = A_size > B_size ? A_size : B_size;
        size_t lds = 0, ws = 0;
        auto place = [&](size_t sz, int& in_lds, unsigned& off) {
            if (lds + sz <= lds_budget_doubles) { in_lds = 1; off = (unsigned)lds; lds += sz; }
            else { in_lds = 0; off = (unsigned)ws; ws += sz; }
        };
        {   // N > 32: LDS scratch of the register-tiled Gauss-Jordan inverses (cg_inverse_panel_*)
            const bool wave_inv = N <= 32 && n <= 16 && nthr >= 128 && NN + nn2 >= 128;     // (set-up: register Gauss-Jordan, no staging)
            const size_t st = wave_inv ? 0 : ev(cg_inv_panel_scratch(N, n, nthr));   // panels of the register-tiled inverses (0: sizes they do not serve)
            if (st + ev(NP) <= lds_budget_doubles) { l.stage_lds = 1; l.stage = (unsigned)lds; lds += st; }
            else { l.stage_lds = 0; l.stage = (unsigned)ws; ws += st; }
        }
        place(P_size, l.P_lds, l.P_off);
        if (lds + AB <= lds_budget_doubles) { l.A_lds = l.B_lds = 1; l.A_off = l.B_off = (unsigned)lds; lds += AB; }
        else if (A_size <= B_size) {          // the larger block goes to the workspace; the smaller may still fit
            l.B_lds = 0; l.B_off = (unsigned)ws; ws += B_size;
            place(A_size, l.A_lds, l.A_off);
        } else {
            l.A_lds = 0; l.A_off = (unsigned)ws; ws += A_size;
            place(B_size, l.B_lds, l.B_off);
        }
        l.all_lds = (l.P_lds && l.A_lds && l.B_lds) ? 1 : 0;
        l.th_lds = 0; l.th = 0;
        if (theta_in_lds && lds + ev(NP) <= lds_budget_doubles) { l.th = (int)lds; lds += ev(NP); l.th_lds = 1; }
        l.lds_total = (unsigned)lds; l.ws_total = (unsigned)ws;
        return l;
    }

    // Fused kernel (k_grad_lap2_scores): what the score sweep of cg_score.hpp reads from the set-up -- the two set-ups are the same
    // computation -- parked in the walker's workspace slot while the rest of this kernel runs, then read into the score layout.  The slot
    // mirrors the score layout's LDS image, so that reading it back is three flat copies (one memory latency, not one per array).
    struct Stash { int m0, s1, sg1, m1, gbar, sg2, s2, U, V, Bm, G, pt, Jinv, zb; unsigned total; };     // (offsets: the score layout's own, see CgScore::stash_of)
    static CG_DEVI void copyw(const CgBlk& b, double* dst, const double* src, int count) {
        for (int e = b.tid; e < count; e += b.nthr) dst[e] = src[e];
    }
    // g_ia = T^a_ii = sum_j D_ij (i k_j^a) Dinv_ji, as the cotangents of z for the score sweep (zb: real parts, then imaginary parts) and,
    // where asked, interleaved (gz).  ONE piece of code for k_scores and for the fused kernel: their results agree bit for bit.
    static CG_DEVI void slater_g(const CgBlk& b, int n, const double* Dm, const double* Dinv, const double* kocc, double* gz, double* zb) {
        const int N = n * D;
        for (int e = b.tid; e < N; e += b.nthr) {
            const int i = e / D, a = e - i * D;
            double re = 0, im = 0;
            for (int j = 0; j < n; ++j) {
                const double ka = kocc[j * D + a];
                const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + i)], Dinv[2 * (j * n + i) + 1]});
                re += -ka * p.im; im += ka * p.re;
            }
            if (gz) { gz[2 * e] = re; gz[2 * e + 1] = im; }
            zb[e] = re; zb[N + e] = im;
        }
    }

    // block base pointers of one workgroup
    template <bool AL>
    struct Mem {
        double *p, *a, *b, *st;
        CG_DEVI Mem(double* lds, double* ws, const Lay& l) {
            if (AL) { p = lds + l.P_off; a = lds + l.A_off; b = lds + l.B_off; }
            else {
                p = (l.P_lds ? lds : ws) + l.P_off; a = (l.A_lds ? lds : ws) + l.A_off; b = (l.B_lds ? lds : ws) + l.B_off;
            }
            st = (l.stage_lds ? lds : ws) + l.stage;
        }
    };

    // One ordered pair (i, k) from the pair table: features of r_ik and the non-zeros of T_ik = d t0_ik / d r_ik
    //   tc_b = d cos(2 pi r_b / L) / d r_b,  ts_b = d sin(2 pi r_b / L) / d r_b,  td_b = d |sin(pi r / L)| / d r_b
    // Diagonal entries hold the exact diagonal feature [1.., 0.., 0] with 1/del = 0 (their T is never used: J_ii is not a pair term).
    struct PairT { double c2[D], s2[D], del, rdel, tc[D], ts[D], td[D]; };
    static CG_DEVI void pt_load(const double* pt, int e /* i n + k */, double c1, double c2c, PairT& p) {
        const double* q = pt + (size_t)e * PFS;
#pragma unroll
        for (int a = 0; a < D; ++a) { p.c2[a] = q[a]; p.s2[a] = q[D + a]; }
        p.del = q[2 * D]; p.rdel = q[2 * D + 1];
#pragma unroll
        for (int a = 0; a < D; ++a) { p.tc[a] = -c1 * p.s2[a]; p.ts[a] = c1 * p.c2[a]; p.td[a] = c2c * (p.s2[a] * p.rdel); }
    }
    // fills the pair table from the half-angle tables of primal() (sh, ch)
    static CG_DEVI void pt_build(const CgBlk& b, const double* sh, const double* ch, int n, unsigned mn, double* pt) {
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = cg_udiv(e, mn), k = e - i * n;
            double* q = pt + (size_t)e * PFS;
#if defined(__HIP_DEVICE_COMPILE__)
            typename F::PF6 f; F::own_pair(sh, ch, i, k, true, f);          // v_rsq_f64-based sqrt / reciprocal, as the sampler
#pragma unroll
            for (int a = 0; a < D; ++a) { q[a] = f.c2[a]; q[D + a] = f.s2[a]; }
            q[2 * D] = f.del; q[2 * D + 1] = f.rdel;
#else
            typename F::PairF f; F::pairfeat(sh, ch, i, k, f);
#pragma unroll
            for (int a = 0; a < D; ++a) { q[a] = f.c2[a]; q[D + a] = f.s2[a]; }
            q[2 * D] = f.del; q[2 * D + 1] = i == k ? 0.0 : 1.0 / f.del;
#endif
        }
    }

    // ------------------------------------------------------------------------------------------------------
    // set-up: z, J (kept), J^-1, D, D^-1 -> g, T^a, diag K^ab
    // ------------------------------------------------------------------------------------------------------
    template <bool AL>
    static CG_DEVI void setup(const CgBlk& b, const double* __restrict__ th, const double* __restrict__ xg,
                              const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                              const Mem<AL>& mem, const Lay& l, bool& early /* in: wanted; out: done */, bool& have_C,
                              double* stash = nullptr, const Stash* st = nullptr) {
        const int N = n * D;
        const CgFastLds& o = l.o;
        double* da = mem.a + l.da; double* x = mem.p + l.x;
        double* kocc = mem.a + l.kocc;       // k_j of the occupied orbitals, staged once: the T^a / K^ab loops below would chase
        for (int e = b.tid; e < N; e += b.nthr) {                  // state_idx -> orbital table through global memory per term
            x[e] = xg[e];
            const int j = e / D;
            kocc[e] = spk[(size_t)sidx[j] * D + (e - j * D)];
        }
        b.sync();
        typename F::WFrag wfrag;
        const typename F::WFrag* wf = F::frags(th, wfrag);                 // MFMA / DPP path of the sampler (device, 16 / 16), else the scalar one
        CG_STAMP_START(25)
        F::primal(b, th, (const double*)x, n, L, da, o, wf);
        CG_STAMP_END(25)
        CG_STAMP_START(26)
        pt_build(b, da + o.sh, da + o.ch, n, l.mn, mem.a + l.pt);          // (read after the barriers of the Jacobian assembly)
        F::jacobian(b, th, n, L, da, o, wf);
        CG_STAMP_END(26)
        if (stash) {        // (fused kernel) the arena as the score sweep reads it, before the inverses' early work reuses the primal temporaries
            b.sync();
            copyw(b, stash + st->m0, da + o.m0, n * P); copyw(b, stash + st->s1, da + o.s1, n * HS); copyw(b, stash + st->sg1, da + o.sg1, n * HS);
            copyw(b, stash + st->m1, da + o.m1, n * HT); copyw(b, stash + st->gbar, da + o.gbar, HS); copyw(b, stash + st->sg2, da + o.sg2, n * HS);
            copyw(b, stash + st->s2, da + o.s2, n * HS); copyw(b, stash + st->U, da + o.U, N * HS); copyw(b, stash + st->V, da + o.V, n * (HT * D + 2));
            copyw(b, stash + st->Bm, da + o.Bm, n * (HS * D + 2)); copyw(b, stash + st->G, da + o.G, n * (HS * D + 2));
            copyw(b, stash + st->pt, mem.a + l.pt, n * n * PFS);
            b.sync();
        }
        CG_STAMP_START(27)
        double* Jc = mem.a + l.Jc; double* Jinv = mem.p + l.Jinv; double* Dc = mem.a + l.Dc; double* Dinv = mem.a + l.Dinv;
        bool inverted = false, early_done = false;
        bool early_C = false;
#if defined(__HIP_DEVICE_COMPILE__)
        early = early && b.nthr >= 192;
        early_C = l.C + N * N <= l.Jc;
        if (N <= 32 && n <= 16 && b.nthr >= 128 && N * N + 2 * n * n >= 128) {      // (the Jc + Dc slots are the 128-double scratch)
            // both inverses by wave-level Gauss-Jordan in registers, concurrently on two waves (no barriers, J and D intact)
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            const int wave = b.tid >> 6;
            if (wave == 0) {
                if (N == 26) cg_wave_inverse_real<26>(da + o.J, N, N, Jinv, N, Jc);
                else cg_wave_inverse_real<32>(da + o.J, N, N, Jinv, N, Jc);
            } else if (wave == 1) {
                if (n == 13) cg_wave_inverse_complex<13>(da + o.Dm, n, n, Dinv, n, Jc + 64);
                else cg_wave_inverse_complex<16>(da + o.Dm, n, n, Dinv, n, Jc + 64);
            } else if (early) {
                // the other waves, meanwhile: what needs neither inverse -- the pair sums of the forward Laplacian and C = J J^T
                const CgBlk b2{b.tid - 128, b.nthr - 128};
                fwd_pair_sums(b2, th, n, L, mem.a + l.pt, mem.a + l.eLm1, mem.a + l.eLm0, mem.a + l.egu1);
                if (early_C) {                  // (only where C's slot stays clear of the inverses' scratch at Jc: small n D = 3 does not)
                    const double* J = da + o.J; double* C = mem.a + l.C;
                    cg_gemm_wg(b2, N, N, N, [&](int r, int k) { return J[r * N + k]; }, [&](int k, int c) { return J[c * N + k]; },
                               [&](int r, int c, double v) { C[r * N + c] = v; });
                }
            }
            b.sync();
            inverted = true;
            early_done = early;
        }
#endif
        early = early_done; have_C = early_done && early_C;
#if defined(__HIP_DEVICE_COMPILE__)
        if (!inverted && l.stage_lds && cg_inv_panel_scratch(N, n, b.nthr)) {
            // larger systems: register-tiled, panel-blocked Gauss-Jordan (every thread a tile of the matrix, two barriers per panel)
            cg_inverse_panel_real(b, da + o.J, N, N, Jinv, N, mem.st);
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            cg_inverse_panel_complex(b, da + o.Dm, n, n, Dinv, n, mem.st);
            inverted = true;
        }
#elif !defined(__HIPCC__)
        if (!inverted) {     // host shim: in-place Gauss-Jordan on a copy, the result scattered back through the row permutation
            std::vector<double> st((size_t)N * N + 4 * N + 64 + N);
            double* vec = st.data() + (size_t)N * N; double* sc = vec + 4 * N; int* rowsrc = (int*)(sc + 64);
            for (int e = 0; e < N * N; ++e) st[e] = da[o.J + e];
            cg_inverse_inplace_real(b, st.data(), N, N, vec, sc, rowsrc, l.mN);
            cg_inverse_scatter_real(b, st.data(), N, N, rowsrc, Jinv, N, l.mN);
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            for (int e = 0; e < 2 * n * n; ++e) st[e] = da[o.Dm + e];
            cg_inverse_inplace_complex(b, st.data(), n, n, vec, sc, rowsrc, l.mn);
            cg_inverse_scatter_complex(b, st.data(), n, n, rowsrc, Dinv, n, l.mn);
            inverted = true;
        }
#endif
        if (!inverted) {     // (beyond the tiled version's reach: [A | I] Gauss-Jordan in the workspace)
            int* perm = (int*)(mem.a + l.perm);
            for (int e = b.tid; e < N * N; e += b.nthr) Jc[e] = da[o.J + e];
            b.sync();
            (void)cg_inverse_real(b, Jc, N, N, Jinv, N, perm);
            F::slater_matrix(b, da + o.z, kocc, nullptr, n, da + o.Dm);
            for (int e = b.tid; e < 2 * n * n; e += b.nthr) Dc[e] = da[o.Dm + e];
            b.sync();
            double la, ar;
            cg_inverse_complex(b, Dc, n, n, Dinv, n, perm, la, ar);
        }
        CG_STAMP_END(27)
        CG_STAMP_START(28)
        const double* Dm = da + o.Dm;
        double* Ta = (l.TaKd_in_P ? mem.p : mem.a) + l.Ta; double* Kd = (l.TaKd_in_P ? mem.p : mem.a) + l.Kd; double* gz = mem.p + l.gz;
        const int nn = n * n;
        // T^a = D diag(i k^a) D^-1 for all directions as ONE real product on the matrix cores: with X^a = D diag(i k^a) and Y = D^-1,
        //   [Re T^a | Im T^a] = [X^a_re | X^a_im] [[Y_re, Y_im], [-Y_im, Y_re]]      rows (a, i), columns (part, q), K = 2 n;   g_ia = T^a_ii
        (void)nn;
        cg_gemm_wg(b, D * n, 2 * n, 2 * n,
                   [&](int r, int kk) {
                       const int a = r >= n ? (D > 2 && r >= 2 * n ? 2 : 1) : 0, i = r - a * n, j = kk < n ? kk : kk - n;
                       return kk < n ? -kocc[j * D + a] * Dm[2 * (i * n + j) + 1] : kocc[j * D + a] * Dm[2 * (i * n + j)];
                   },
                   [&](int kk, int c) {
                       const int part = c >= n ? 1 : 0, q = c - part * n, j = kk < n ? kk : kk - n;
                       const double yr = Dinv[2 * (j * n + q)], yi = Dinv[2 * (j * n + q) + 1];
                       return kk < n ? (part ? yi : yr) : (part ? yr : -yi);
                   },
                   [&](int r, int c, double v) {
                       const int a = r >= n ? (D > 2 && r >= 2 * n ? 2 : 1) : 0, i = r - a * n, part = c >= n ? 1 : 0, q = c - part * n;
                       Ta[2 * ((a * n + i) * n + q) + part] = v;
                       if (i == q) gz[2 * (i * D + a) + part] = v;
                   });
        for (int e = b.tid; e < D * D * n; e += b.nthr) {            // diag of K^ab = D diag(-k^a k^b) D^-1
            const int i = e / (D * D), r = e - i * D * D, a = r / D, bb = r - a * D;
            double re = 0, im = 0;
            for (int j = 0; j < n; ++j) {
                const double kk = -kocc[j * D + a] * kocc[j * D + bb];
                const CgCplx p = cmul({Dm[2 * (i * n + j)], Dm[2 * (i * n + j) + 1]}, {Dinv[2 * (j * n + i)], Dinv[2 * (j * n + i) + 1]});
                re += kk * p.re; im += kk * p.im;
            }
            Kd[2 * ((a * D + bb) * n + i)] = re; Kd[2 * ((a * D + bb) * n + i) + 1] = im;
        }
        if (stash) {        // J^-1 and the cotangents of z in the summation order of k_scores
            copyw(b, stash + st->Jinv, Jinv, N * N);
            slater_g(b, n, Dm, Dinv, kocc, nullptr, stash + st->zb);
        }
        b.sync();
        CG_STAMP_END(28)
    }

    // ------------------------------------------------------------------------------------------------------
    // tr(J^T H J) with H the Hessian of log phi in z:  H_(ia),(lb) = delta_il K^ab_ii - T^a_il T^b_li
    //   = sum_i sum_ab C_(ia),(ib) K^ab_ii - sum_il sum_ab C_(ia),(lb) T^a_il T^b_li ,   C = J J^T
    // and the first part of the gradient, grad_e = (J^T g)_e (the caller adds xbar after the reverse sweep).  Returns the two partial sums of this thread in (p_re, p_im).
    // ------------------------------------------------------------------------------------------------------
    template <bool AL>
    static CG_DEVI void slater_part(const CgBlk& b, int n, const Mem<AL>& mem, const Lay& l, bool want_lap,
                                    double* __restrict__ grad /*N x 2, global*/, double& p_re, double& p_im, bool have_C = false) {
        const int N = n * D;
        const double* J = mem.a + l.da + l.o.J; const double* gz = mem.p + l.gz;
        const double* Ta = (l.TaKd_in_P ? mem.p : mem.a) + l.Ta; const double* Kd = (l.TaKd_in_P ? mem.p : mem.a) + l.Kd;
        double* C = mem.a + l.C;
        for (int e = b.tid; e < N; e += b.nthr) {
            double re = 0, im = 0;
            for (int al = 0; al < N; ++al) { const double j = J[al * N + e]; re += gz[2 * al] * j; im += gz[2 * al + 1] * j; }
            grad[2 * e] = re; grad[2 * e + 1] = im;
        }
        p_re = 0; p_im = 0;
        if (!want_lap) return;
        if (!have_C) {
            cg_gemm_wg(b, N, N, N, [&](int r, int k) { return J[r * N + k]; }, [&](int k, int c) { return J[c * N + k]; },
                       [&](int r, int c, double v) { C[r * N + c] = v; });      // C = J J^T (matrix cores)
            b.sync();
        }
        for (int e = b.tid; e < n * D * D; e += b.nthr) {
            const int i = e / (D * D), r = e - i * D * D, a = r / D, bb = r - a * D;
            const double c = C[(i * D + a) * N + i * D + bb];
            p_re += c * Kd[2 * ((a * D + bb) * n + i)]; p_im += c * Kd[2 * ((a * D + bb) * n + i) + 1];
        }
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = e / n, q = e - i * n;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const double c = C[(i * D + a) * N + q * D + bb];
                    const CgCplx t1 = {Ta[2 * ((a * n + i) * n + q)], Ta[2 * ((a * n + i) * n + q) + 1]};
                    const CgCplx t2 = {Ta[2 * ((bb * n + q) * n + i)], Ta[2 * ((bb * n + q) * n + i) + 1]};
                    const CgCplx pr = cmul(t1, t2);
                    p_re -= c * pr.re; p_im -= c * pr.im;
                }
        }
    }

    // ------------------------------------------------------------------------------------------------------
    // xbar = grad_x 1/2 log|det J(x)|: reverse sweep of CgFast::primal / jacobian with Jbar = 1/2 J^-T, zbar = 0.
    // Same adjoint chain as CgDerivs::reverse (theta-gradient), continued down to the pair features and x.
    // ------------------------------------------------------------------------------------------------------
    template <bool AL>
    static CG_DEVI void reverse_x(const CgBlk& b, const double* __restrict__ th, int n, double L, const Mem<AL>& mem, const Lay& l) {
        const int N = n * D;
        const CgFastLds& o = l.o;
        const double* da = mem.a + l.da;
        const double *sg1 = da + o.sg1, *sg2 = da + o.sg2, *U = da + o.U, *V = da + o.V, *Bm = da + o.Bm, *Up = da + o.Up, *G = da + o.G;
        const double* PT = mem.a + l.pt;
        const double* Jinv = mem.p + l.Jinv;
        double *Jhat = mem.a + l.Jhat;
        double *Upb = mem.a + l.Upb, *Vb = mem.a + l.Vb, *Bb = mem.a + l.Bb, *Gb = mem.a + l.Gb, *sg1b = mem.a + l.sg1b,
               *sg2b = mem.a + l.sg2b, *Ub = mem.a + l.Ub, *Rb = mem.a + l.Rb, *u2b = mem.a + l.u2b, *u1b = mem.a + l.u1b, *s1b = mem.a + l.s1b,
               *m1b = mem.a + l.m1b, *gbb = mem.a + l.gbb, *su2 = mem.a + l.su2, *m0b = mem.a + l.m0b, *rbar = mem.a + l.rbar, *xbar = mem.p + l.xbar;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        const unsigned mn = l.mn, mN = l.mN;

        CG_STAMP_START(15)
        // (J6) J_ii = I - sum_{k!=i} J_ik  =>  Jhat_ik = Jbar_ik - Jbar_ii (k != i),  Jbar = 1/2 J^-T;  Jhat_ii = 0
        for (int e = b.tid; e < N * N; e += b.nthr) {
            const int r = cg_udiv(e, mN), c = e - r * N, i = r / D, k = c / D, bb = c - k * D;
            Jhat[e] = (i == k) ? 0.0 : 0.5 * (Jinv[c * N + r] - Jinv[(i * D + bb) * N + r]);
        }
        b.sync();
        // (J5) adjoints that are sums over k for fixed i (the k = i terms vanish with Jhat_ii = 0)
        for (int e = b.tid; e < N * P; e += b.nthr) {              // Upbar_i[a][f] = -sum_k sum_b Jhat_ik[a][b] T_ik[f][b]
            const int r = e / P, f = e - r * P, i = r / D;
            const double* jr = Jhat + (size_t)r * N; const double* pr = PT + (size_t)i * n * PFS;
            double acc = 0;
            if (f < 2 * D) {                                        // cos / sin feature of direction bb: T = -c1 s2 / c1 c2
                const int bb = f < D ? f : f - D, off = f < D ? D + bb : bb;
                for (int k = 0; k < n; ++k) acc += jr[k * D + bb] * pr[k * PFS + off];
                acc *= f < D ? c1 : -c1;
            } else {                                                // norm feature: T_b = c2c s2_b / del
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
        for (int e = b.tid; e < n * HT; e += b.nthr) {             // Vbar_i[a][h] = -sum_k sum_b Jhat_ik[a][b] sig_t(u_ik[h]) q_ik[h][b]
            const int i = e / HT, h = e - i * HT;
            double wt[P]; const double bt = th[F::o_t0b + h];
#pragma unroll
            for (int f = 0; f < P; ++f) wt[f] = th[F::o_t0w + f * HT + h];
            double vb[D];
#pragma unroll
            for (int a = 0; a < D; ++a) vb[a] = 0;
#pragma unroll CG_UNR_K
            for (int k = 0; k < n; ++k) {
                if (k == i) continue;
                PairT t; pt_load(PT, i * n + k, c1, c2c, t);
                double u = bt + wt[2 * D] * t.del, q[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    u += wt[a] * t.c2[a] + wt[D + a] * t.s2[a];
                    q[a] = wt[a] * t.tc[a] + wt[D + a] * t.ts[a] + wt[2 * D] * t.td[a];
                }
                const double sg = sigmoid_only(u);
#pragma unroll
                for (int a = 0; a < D; ++a)
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) vb[a] -= Jhat[(i * D + a) * N + k * D + bb] * (sg * q[bb]);
            }
#pragma unroll
            for (int a = 0; a < D; ++a) Vb[(i * D + a) * HT + h] = vb[a];
        }
        b.sync();
        CG_STAMP(15)
        // (J4) G adjoint -> sg1bar (first part);  (J3) Up_i = (1/n) (U_i diag sg1_i) W0^T -> Ubar, sg1bar (second part)
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int p = e / HS, h = e - p * HS;
            double w_c[D], w_s[D];
#pragma unroll
            for (int a = 0; a < D; ++a) { w_c[a] = th[F::o_W0 + a * HS + h]; w_s[a] = th[F::o_W0 + (D + a) * HS + h]; }
            const double w_d = th[F::o_W0 + 2 * D * HS + h];
            double gp[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) gp[bb] = Gb[(p * HS + h) * D + bb];
            double sb = 0;
#pragma unroll CG_UNR_Q
            for (int q = 0; q < n; ++q) {
                if (q == p) continue;
                PairT t; pt_load(PT, p * n + q, c1, c2c, t);
#pragma unroll
                for (int bb = 0; bb < D; ++bb)
                    sb += (gp[bb] - Gb[(q * HS + h) * D + bb]) * (w_c[bb] * t.tc[bb] + w_s[bb] * t.ts[bb] + w_d * t.td[bb]);
            }
            double acc = 0;
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int f = 0; f < P; ++f) acc += Upb[(p * D + a) * P + f] * U[(p * D + a) * HS + h] * th[F::o_W0 + f * HS + h];
            sg1b[e] = sb * (rn * rn) + acc * rn;
        }
        for (int e = b.tid; e < N * HS; e += b.nthr) {             // Ubar_i[a][g]
            const int r = e / HS, g = e - r * HS, i = r / D;
            double acc = 0;
#pragma unroll
            for (int f = 0; f < P; ++f) acc += Upb[r * P + f] * th[F::o_W0 + f * HS + g];
            Ub[e] = acc * rn * sg1[i * HS + g];
        }
        b.sync();
        // (J2) Rbar_i[a][h] = sum_g Ubar Wa[g][h] + Bbar Wb[g][h] + (1/n) Vbar Wc[g][h]: one (N x (2 HS + HT))((2 HS + HT) x HS) product
        // (the rows of Wa, Wb, Wc are consecutive in theta)
        cg_gemm_wg(b, N, HS, 2 * HS + HT,
                   [&](int r, int k) { return k < HS ? Ub[r * HS + k] : k < 2 * HS ? Bb[r * HS + k - HS] : rn * Vb[r * HT + k - 2 * HS]; },
                   [&](int k, int h) { return th[F::o_Wa + k * HS + h]; }, [&](int r, int h, double v) { Rb[r * HS + h] = v; });
        b.sync();
        // (J1) sg2bar_i[h] = sum_a Rbar_i[a][h] Wf[h][a];  u2bar = sg2bar sg2'   (zbar = 0: no s2bar)
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            double sb = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) sb += Rb[(i * D + a) * HS + h] * th[F::o_fw + h * D + a];
            sg2b[e] = sb;
            const double g2 = sg2[e];
            u2b[e] = sb * g2 * (1.0 - g2);
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
            double acc = rn * gbb[g];
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
        for (int e = b.tid; e < n * P; e += b.nthr) {              // m0bar_i[f] = sum_h W0[f][h] u1bar_i[h]
            const int i = e / P, f = e - i * P;
            double acc = 0;
#pragma unroll
            for (int h = 0; h < HS; ++h) acc += th[F::o_W0 + f * HS + h] * u1b[i * HS + h];
            m0b[e] = acc;
        }
        b.sync();
        CG_STAMP(16)
        // pair pass: adjoints of the features t0_ik (value: t0bar) and of their r-derivatives T_ik (Tc, Ts, Td), then rbar_ik
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = cg_udiv(e, mn), k = e - i * n;
            if (i == k) {
#pragma unroll
                for (int bb = 0; bb < D; ++bb) rbar[e * D + bb] = 0.0;
                continue;
            }
            PairT t; pt_load(PT, e, c1, c2c, t);
            double Jh[D][D];
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int bb = 0; bb < D; ++bb) Jh[a][bb] = Jhat[(i * D + a) * N + k * D + bb];
            double t0b[P], Tc[D], Ts[D], Td[D];
#pragma unroll
            for (int f = 0; f < P; ++f) t0b[f] = rn * m0b[i * P + f];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) {
                double c = 0, s = 0, d = 0;
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    c -= Jh[a][bb] * Up[(i * D + a) * P + bb];
                    s -= Jh[a][bb] * Up[(i * D + a) * P + D + bb];
                    d -= Jh[a][bb] * Up[(i * D + a) * P + 2 * D];
                }
                Tc[bb] = c; Ts[bb] = s; Td[bb] = d;
            }
#pragma unroll CG_UNR_G
            for (int g = 0; g < HS; ++g) {                          // G part: q0bar_ik[g][b] = sg1_i[g] (Gbar_i - Gbar_k)[g][b] / n^2
                const double s1g = sg1[i * HS + g] * rn * rn;
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const double q0b = s1g * (Gb[(i * HS + g) * D + bb] - Gb[(k * HS + g) * D + bb]);
                    Tc[bb] += th[F::o_W0 + bb * HS + g] * q0b;
                    Ts[bb] += th[F::o_W0 + (D + bb) * HS + g] * q0b;
                    Td[bb] += th[F::o_W0 + 2 * D * HS + g] * q0b;
                }
            }
#pragma unroll CG_UNR_H
            for (int h = 0; h < HT; ++h) {
                double wt[P];
#pragma unroll
                for (int f = 0; f < P; ++f) wt[f] = th[F::o_t0w + f * HT + h];
                double u = th[F::o_t0b + h] + wt[2 * D] * t.del, q[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    u += wt[a] * t.c2[a] + wt[D + a] * t.s2[a];
                    q[a] = wt[a] * t.tc[a] + wt[D + a] * t.ts[a] + wt[2 * D] * t.td[a];
                }
                const double sg = sigmoid_only(u), sgp = sg * (1.0 - sg);
                double sgb = 0, qb[D];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    double jv = 0;
#pragma unroll
                    for (int a = 0; a < D; ++a) jv += Jh[a][bb] * V[F::iV(i, a, h)];
                    sgb -= jv * q[bb];
                    qb[bb] = -jv * sg;
                }
                const double ub = sgb * sgp + rn * m1b[i * HT + h] * sg;     // adjoint of u_t,ik[h]: Jacobian part + primal part
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    t0b[a] += wt[a] * ub; t0b[D + a] += wt[D + a] * ub;
                    Tc[a] += wt[a] * qb[a]; Ts[a] += wt[D + a] * qb[a]; Td[a] += wt[2 * D] * qb[a];
                }
                t0b[2 * D] += wt[2 * D] * ub;
            }
            // r-derivatives: d tc_b / d r_b = -c1^2 c2_b, d ts_b / d r_b = -c1^2 s2_b,
            //                d td_b' / d r_b = delta_bb' (pi/L)^2 c2_b / del - td_b td_b' / del
            const double pl2 = 4.0 * c2c * c2c;
            double tdd = 0;
#pragma unroll
            for (int bb = 0; bb < D; ++bb) tdd += Td[bb] * t.td[bb];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) {
                double r = t0b[bb] * t.tc[bb] + t0b[D + bb] * t.ts[bb] + t0b[2 * D] * t.td[bb];
                r += -c1 * c1 * (Tc[bb] * t.c2[bb] + Ts[bb] * t.s2[bb]);
                r += t.rdel * (Td[bb] * pl2 * t.c2[bb] - t.td[bb] * tdd);
                rbar[e * D + bb] = r;
            }
        }
        b.sync();
        for (int e = b.tid; e < N; e += b.nthr) {                  // r_pq = x_p - x_q
            const int p = e / D, bb = e - p * D;
            double acc = 0;
            for (int q = 0; q < n; ++q) acc += rbar[(p * n + q) * D + bb] - rbar[(q * n + p) * D + bb];
            xbar[e] = acc;
        }
        b.sync();
        CG_STAMP_END(17)
    }

    // ------------------------------------------------------------------------------------------------------
    // Forward Laplacian of the flow: returns this thread's partial sums of  sum_ia g_ia lap_x z_ia  (complex).
    // Carries (value, |grad_x|^2, lap_x) through the hidden units (SURVEY App. A.1 for the layers):
    //   lap sp(u) = sig(u) lap u + sig'(u) |grad u|^2
    // Pair features depend on r = x_i - x_j only: lap_x f(r) = 2 lap_r f, |grad_x f|^2 = 2 |grad_r f|^2.
    // ------------------------------------------------------------------------------------------------------
    // The two pair sums of the forward Laplacian that depend on the pair table and the weights only (not on J^-1, D^-1): lap m1, lap m0
    // and |grad u1|^2.  Called by the whole workgroup, or -- early -- by the waves that would idle while two waves invert J and D.
    static CG_DEVI void fwd_pair_sums(const CgBlk& b, const double* th, int n, double L, const double* PT, double* Lm1, double* Lm0, double* gu1) {
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L), pl2 = 4.0 * c2c * c2c;
        // pair pass, item (i,h): lap m1_i[h], lap m0_i[f], |grad u1_i[h]|^2
        for (int e = b.tid; e < n * HM; e += b.nthr) {
            const int i = e / HM, h = e - i * HM;
            const bool do_t = h < HT;
            double wt[P], bt = 0.0;
#pragma unroll
            for (int f = 0; f < P; ++f) wt[f] = do_t ? th[F::o_t0w + f * HT + h] : 0.0;
            if (do_t) bt = th[F::o_t0b + h];
            double acc = 0.0, raw = 0.0;
#pragma unroll CG_UNR_K
            for (int j = 0; j < n; ++j) {
                if (j == i) continue;
                PairT t; pt_load(PT, i * n + j, c1, c2c, t);
                double l0d = 0.0;                                   // lap_r of the norm feature
#pragma unroll
                for (int a = 0; a < D; ++a) l0d += pl2 * t.c2[a] - t.td[a] * t.td[a];
                l0d *= t.rdel;
                if (do_t) {
                    double u = bt + wt[2 * D] * t.del, lu = wt[2 * D] * l0d, gsq = 0.0;
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const double wf = wt[a] * t.c2[a] + wt[D + a] * t.s2[a];
                        u += wf; lu -= c1 * c1 * wf;
                        const double q = wt[a] * t.tc[a] + wt[D + a] * t.ts[a] + wt[2 * D] * t.td[a];
                        gsq += q * q;
                    }
                    const double sg = sigmoid_only(u);
                    acc += 2.0 * (sg * lu + sg * (1.0 - sg) * gsq);
                }
                if (h < P) {
                    double fv = l0d;
#pragma unroll
                    for (int a = 0; a < D; ++a) { if (h == a) fv = -c1 * c1 * t.c2[a]; if (h == D + a) fv = -c1 * c1 * t.s2[a]; }
                    raw += 2.0 * fv;
                }
            }
            if (do_t) Lm1[i * HT + h] = acc * rn;
            if (h < P) Lm0[i * P + h] = raw * rn;
        }
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            double w_c[D], w_s[D];
#pragma unroll
            for (int a = 0; a < D; ++a) { w_c[a] = th[F::o_W0 + a * HS + h]; w_s[a] = th[F::o_W0 + (D + a) * HS + h]; }
            const double w_d = th[F::o_W0 + 2 * D * HS + h];
            double ssq = 0.0, sq[D];
#pragma unroll
            for (int a = 0; a < D; ++a) sq[a] = 0.0;
#pragma unroll CG_UNR_Q
            for (int k = 0; k < n; ++k) {
                if (k == i) continue;
                PairT t; pt_load(PT, i * n + k, c1, c2c, t);
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const double q0 = w_c[bb] * t.tc[bb] + w_s[bb] * t.ts[bb] + w_d * t.td[bb];
                    ssq += q0 * q0; sq[bb] += q0;
                }
            }
#pragma unroll
            for (int a = 0; a < D; ++a) ssq += sq[a] * sq[a];
            gu1[e] = ssq * rn * rn;
        }
    }

#if defined(__HIP_DEVICE_COMPILE__)
    // |grad_x u2_i[h]|^2 on the matrix cores (spsize = tpsize = 16).  For one particle i the dense x-gradient of the last layer's
    // pre-activations is  E_ik[h][b] = H_k[h][b] - (1/n) (A_i T_ik)[h][b] - (1/n) sum_g Wc[g][h] sig_t(u_ik[g]) q_ik[g][b]  (k != i),
    // E_ii = -sum_k E_ik.  The last term is a (n x 16)(16 x 16) product per direction b: rows = partner particle k, K = hidden unit g
    // of the two-particle stream, columns = h.  A operand sig q computed by the lane that feeds it (row k = lane & 15, g = 4 ks +
    // (lane >> 4)), B operand -Wc / n held in registers for all i, the first two terms are the C operand.  A wave owns particle i:
    // no barrier and no LDS round trip of sig q or E; sum_k |E_ik|^2 + |sum_k E_ik|^2 leaves the wave as 16 numbers.
    static __device__ __forceinline__ void su2_mfma(const CgBlk& b, const double* th, int n, double rn, double c1, double c2c,
                                                    const double* PT, const double* wt, const double* Am, const double* Hk, double* Su2) {
        using d4 = typename F::d4_t;
        const int lane = b.tid & 63, col = lane & 15, kq = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        double wg[4][P + 1], bw[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int g = 4 * ks + kq;
#pragma unroll
            for (int f = 0; f <= P; ++f) wg[ks][f] = wt[g * (P + 1) + f];
            bw[ks] = -rn * th[F::o_Wc + g * HS + col];
        }
        const int tiles = (n + 15) >> 4;
        for (int i = wave; i < n; i += nw) {
            double Ai[P];
#pragma unroll
            for (int f = 0; f < P; ++f) Ai[f] = Am[(i * HS + col) * P + f];
            double ssq = 0.0, sm[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) sm[bb] = 0.0;
            for (int kt = 0; kt < tiles; ++kt) {
                d4 c[D];
#pragma unroll
                for (int r = 0; r < 4; ++r) {                       // C operand: rows k = 16 kt + kq + 4 r of column h = col
                    const int k = 16 * kt + kq + 4 * r;
                    const bool ok = k < n && k != i;
                    PairT t; pt_load(PT, i * n + (ok ? k : i), c1, c2c, t);
#pragma unroll
                    for (int bb = 0; bb < D; ++bb)
                        c[bb][r] = ok ? Hk[((ok ? k : 0) * HS + col) * D + bb] - rn * (Ai[bb] * t.tc[bb] + Ai[D + bb] * t.ts[bb] + Ai[2 * D] * t.td[bb]) : 0.0;
                }
                const int ka = 16 * kt + col;                       // A operand: row k = 16 kt + col
                const bool oka = ka < n && ka != i;
                PairT ta; pt_load(PT, i * n + (oka ? ka : i), c1, c2c, ta);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    double u = wg[ks][0] + wg[ks][1 + 2 * D] * ta.del, q[D];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        u += wg[ks][1 + a] * ta.c2[a] + wg[ks][1 + D + a] * ta.s2[a];
                        q[a] = wg[ks][1 + a] * ta.tc[a] + wg[ks][1 + D + a] * ta.ts[a] + wg[ks][1 + 2 * D] * ta.td[a];
                    }
                    const double sg = oka ? sigmoid_only(u) : 0.0;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) c[bb] = F::mfma(sg * q[bb], bw[ks], c[bb]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) { ssq = fma(c[bb][r], c[bb][r], ssq); sm[bb] += c[bb][r]; }
            }
            // rows are spread over the four 16-lane groups of the wave (kq): fixed-order butterfly over kq
            ssq += __shfl_xor(ssq, 16); ssq += __shfl_xor(ssq, 32);
#pragma unroll
            for (int bb = 0; bb < D; ++bb) { sm[bb] += __shfl_xor(sm[bb], 16); sm[bb] += __shfl_xor(sm[bb], 32); ssq = fma(sm[bb], sm[bb], ssq); }
            if (kq == 0) Su2[i * HS + col] = ssq;
        }
    }
    // A_i = Wa^T diag(sg1_i) W0^T (HS x P) and H_k = Wb^T G_k (HS x D) on the matrix cores: 4 MFMAs per particle / per 16 rows (k, b)
    static __device__ __forceinline__ void am_hk_mfma(const CgBlk& b, const double* th, int n, const double* sg1, const double* G,
                                                      double* Am, double* Hk) {
        using d4 = typename F::d4_t;
        const int lane = b.tid & 63, col = lane & 15, kq = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        double wa[4], w0[4], wb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int g = 4 * ks + kq;
            wa[ks] = th[F::o_Wa + g * HS + col];                    // A[row h = col][k = g] = Wa[g][h] (times sg1_i[g])
            w0[ks] = col < P ? th[F::o_W0 + col * HS + g] : 0.0;    // B[k = g][col f] = W0[f][g]
            wb[ks] = th[F::o_Wb + g * HS + col];                    // B[k = g][col h] = Wb[g][h]
        }
        for (int i = wave; i < n; i += nw) {
            d4 c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = F::mfma(wa[ks] * sg1[i * HS + 4 * ks + kq], w0[ks], c);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (col < P) Am[(i * HS + kq + 4 * r) * P + col] = c[r];
        }
        const int N = n * D, tiles = (N + 15) >> 4;
        for (int t = wave; t < tiles; t += nw) {
            const int ra = 16 * t + col, ka = ra / D, ba = ra - ka * D;           // A row (k, b)
            d4 c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = F::mfma(ra < N ? G[F::iG(ka, 4 * ks + kq, ba)] : 0.0, wb[ks], c);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * t + kq + 4 * r;
                if (rr < N) Hk[((rr / D) * HS + col) * D + (rr % D)] = c[r];
            }
        }
    }
#else       // host builds: declared so that the `if constexpr (mfma_path)` branches of forward_laplacian parse; never instantiated
    static void su2_mfma(const CgBlk&, const double*, int, double, double, double, const double*, const double*, const double*, const double*, double*);
    static void am_hk_mfma(const CgBlk&, const double*, int, const double*, const double*, double*, double*);
#endif

    template <bool AL>
    static CG_DEVI void forward_laplacian(const CgBlk& b, const double* __restrict__ th, int n, double L, const Mem<AL>& mem,
                                          const Lay& l, double& q_re, double& q_im, bool pre = false) {
        const CgFastLds& o = l.o;
        const double* da = mem.a + l.da;
        const double *sg1 = da + o.sg1, *sg2 = da + o.sg2, *G = da + o.G;
        const double* PT = mem.a + l.pt;
        const double* gz = mem.p + l.gz;
        // pre: the pair sums were formed during the set-up (fwd_pair_sums by the idle waves) and wait behind the set-up scratch
        double *Lm0 = mem.a + (pre ? l.eLm0 : l.Lm0), *gu1 = mem.a + (pre ? l.egu1 : l.gu1), *Lm1 = mem.a + (pre ? l.eLm1 : l.Lm1);
        double *Ls1 = mem.a + l.Ls1, *Lgb = mem.a + l.Lgb, *Am = mem.a + l.Am, *Hk = mem.a + l.Hk, *Su2 = mem.a + l.Su2, *Ls2 = mem.a + l.Ls2;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L), pl2 = 4.0 * c2c * c2c;
        constexpr bool mfma_path = CG_ON_DEVICE && HS == 16 && HT == 16;      // (am_hk_mfma / su2_mfma: device code only)

        if (!pre) fwd_pair_sums(b, th, n, L, PT, Lm1, Lm0, gu1);
        // per-particle factors of the dense x-gradient of u2:  A_i = Wa^T diag(sg1_i) W0^T (HS x P),  H_k = Wb^T G_k (HS x D)
        if constexpr (mfma_path) {
            am_hk_mfma(b, th, n, sg1, G, Am, Hk);
        } else {
            for (int e = b.tid; e < n * HS * P; e += b.nthr) {
                const int i = e / (HS * P), r = e - i * HS * P, h = r / P, f = r - h * P;
                double acc = 0;
#pragma unroll
                for (int g = 0; g < HS; ++g) acc += th[F::o_Wa + g * HS + h] * sg1[i * HS + g] * th[F::o_W0 + f * HS + g];
                Am[e] = acc;
            }
            for (int e = b.tid; e < n * HS * D; e += b.nthr) {
                const int k = e / (HS * D), r = e - k * HS * D, h = r / D, bb = r - h * D;
                double acc = 0;
#pragma unroll
                for (int g = 0; g < HS; ++g) acc += th[F::o_Wb + g * HS + h] * G[F::iG(k, g, bb)];
                Hk[e] = acc;
            }
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // lap s1
            const int i = e / HS, h = e - i * HS;
            double lu = 0;
#pragma unroll
            for (int f = 0; f < P; ++f) lu += th[F::o_W0 + f * HS + h] * Lm0[i * P + f];
            const double g1 = sg1[e];
            Ls1[e] = g1 * lu + g1 * (1.0 - g1) * gu1[e];
        }
        // |grad_x u2_i[h]|^2:  E_ik[h][b] = d u2_i[h] / d x_kb  (k != i),  E_ii = -sum_k E_ik
        //   E_ik = -(1/n) A_i T_ik + H_k - (1/n) Wc^T diag(sig_t(u_ik)) Wt^T T_ik
        if constexpr (mfma_path) {
            su2_mfma(b, th, n, rn, c1, c2c, PT, da + o.wt, Am, Hk, Su2);
            b.sync();
            for (int h = b.tid; h < HS; h += b.nthr) {
                double a = 0;
                for (int i = 0; i < n; ++i) a += Ls1[i * HS + h];
                Lgb[h] = a * rn;
            }
        } else {
            double *SQ = mem.a + l.SQ, *Er = mem.a + l.Er;
            b.sync();
            for (int h = b.tid; h < HS; h += b.nthr) {
                double a = 0;
                for (int i = 0; i < n; ++i) a += Ls1[i * HS + h];
                Lgb[h] = a * rn;
            }
            for (int i = 0; i < n; ++i) {                           // one particle i at a time
                for (int e = b.tid; e < n * HT; e += b.nthr) {      // SQ[k][g][b] = sig_t(u_ik[g]) q_ik[g][b]
                    const int k = e / HT, g = e - k * HT;
                    if (k == i) {
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) SQ[e * D + bb] = 0.0;
                        continue;
                    }
                    PairT t; pt_load(PT, i * n + k, c1, c2c, t);
                    const double wd = th[F::o_t0w + 2 * D * HT + g];
                    double u = th[F::o_t0b + g] + wd * t.del, q[D];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const double wc = th[F::o_t0w + a * HT + g], ws_ = th[F::o_t0w + (D + a) * HT + g];
                        u += wc * t.c2[a] + ws_ * t.s2[a];
                        q[a] = wc * t.tc[a] + ws_ * t.ts[a] + wd * t.td[a];
                    }
                    const double sg = sigmoid_only(u);
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) SQ[e * D + bb] = sg * q[bb];
                }
                b.sync();
                for (int e = b.tid; e < n * HS; e += b.nthr) {
                    const int k = e / HS, h = e - k * HS;
                    if (k == i) {
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) Er[e * D + bb] = 0.0;
                        continue;
                    }
                    PairT t; pt_load(PT, i * n + k, c1, c2c, t);
                    const double* A = Am + ((size_t)i * HS + h) * P;
                    double ev[D];
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) ev[bb] = Hk[e * D + bb] - rn * (A[bb] * t.tc[bb] + A[D + bb] * t.ts[bb] + A[2 * D] * t.td[bb]);
                    for (int g = 0; g < HT; ++g) {
                        const double wc = rn * th[F::o_Wc + g * HS + h];
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) ev[bb] -= wc * SQ[(k * HT + g) * D + bb];
                    }
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) Er[e * D + bb] = ev[bb];
                }
                b.sync();
                for (int h = b.tid; h < HS; h += b.nthr) {
                    double ssq = 0.0, sm[D];
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) sm[bb] = 0.0;
                    for (int k = 0; k < n; ++k)
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) { const double v = Er[(k * HS + h) * D + bb]; ssq += v * v; sm[bb] += v; }
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) ssq += sm[bb] * sm[bb];
                    Su2[i * HS + h] = ssq;
                }
                // (the next particle's SQ pass is separated from this Er read by its own barrier pair)
            }
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // lap s2 = lap s1 + sg2 lap u2 + sg2' |grad u2|^2
            const int i = e / HS, h = e - i * HS;
            double lu = 0;
#pragma unroll
            for (int g = 0; g < HS; ++g) lu += th[F::o_Wa + g * HS + h] * Ls1[i * HS + g] + th[F::o_Wb + g * HS + h] * Lgb[g];
#pragma unroll
            for (int g = 0; g < HT; ++g) lu += th[F::o_Wc + g * HS + h] * Lm1[i * HT + g];
            const double g2 = sg2[e];
            Ls2[e] = Ls1[e] + g2 * lu + g2 * (1.0 - g2) * Su2[e];
        }
        b.sync();
        q_re = 0; q_im = 0;
        for (int e = b.tid; e < n * D; e += b.nthr) {              // lap z_ia = sum_h Wf[h][a] lap s2_i[h]
            const int i = e / D, a = e - i * D;
            double lz = 0;
#pragma unroll
            for (int h = 0; h < HS; ++h) lz += th[F::o_fw + h * D + a] * Ls2[i * HS + h];
            q_re += gz[2 * e] * lz; q_im += gz[2 * e + 1] * lz;
        }
    }

    // ------------------------------------------------------------------------------------------------------
    // second-order jet pass along dir (probe v, or basis direction `basis`): returns this thread's partial sums
    //   t2 = tr(J^-1 J''),  t3 = tr((J^-1 J')^2),  and (want_phi2) v^T hess(log phi) v through z', z''.
    // ------------------------------------------------------------------------------------------------------
    template <bool AL>
    static CG_DEVI void jet_part(const CgBlk& b, const double* __restrict__ th, int n, double L, const Mem<AL>& mem, const Lay& l,
                                 const double* __restrict__ dir, int basis, bool want_phi2, double (&red)[4]) {
        const int N = n * D;
        const CgFastLds& oj = l.oj;
        Jet2* xj = (Jet2*)(mem.b + l.xj); Jet2* ja = (Jet2*)(mem.b + l.ja);
        const double* x = mem.p + l.x; const double* Jinv = mem.p + l.Jinv; double* M = mem.b + l.M;
        const double* gz = mem.p + l.gz; const double* Ta = (l.TaKd_in_P ? mem.p : mem.a) + l.Ta; const double* Kd = (l.TaKd_in_P ? mem.p : mem.a) + l.Kd;
        for (int e = b.tid; e < N; e += b.nthr) xj[e] = Jet2(x[e], dir ? dir[e] : (e == basis ? 1.0 : 0.0), 0.0);
        b.sync();
        CG_STAMP_START(29)
        F::primal(b, th, (const Jet2*)xj, n, L, ja, oj, nullptr, dir ? -1 : basis / D);
        CG_STAMP(29)
        F::jacobian(b, th, n, L, ja, oj);
        CG_STAMP(30)
        const Jet2* zj = ja + oj.z; const Jet2* Jj = ja + oj.J;
        double p_re = 0, p_im = 0, t2 = 0, t3 = 0;
        if (want_phi2) {
            for (int e = b.tid; e < N; e += b.nthr) {
                p_re += gz[2 * e] * zj[e].dd; p_im += gz[2 * e + 1] * zj[e].dd;
                const int i = e / D, a = e - i * D;
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const double zz = zj[e].d * zj[i * D + bb].d;
                    p_re += zz * Kd[2 * ((a * D + bb) * n + i)]; p_im += zz * Kd[2 * ((a * D + bb) * n + i) + 1];
                }
            }
            for (int e = b.tid; e < n * n; e += b.nthr) {
                const int i = e / n, q = e - i * n;
                CgCplx yiq = {0, 0}, yqi = {0, 0};
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double zi = zj[i * D + a].d, zq = zj[q * D + a].d;
                    yiq.re += zi * Ta[2 * ((a * n + i) * n + q)]; yiq.im += zi * Ta[2 * ((a * n + i) * n + q) + 1];
                    yqi.re += zq * Ta[2 * ((a * n + q) * n + i)]; yqi.im += zq * Ta[2 * ((a * n + q) * n + i) + 1];
                }
                const CgCplx pr = cmul(yiq, yqi);
                p_re -= pr.re; p_im -= pr.im;
            }
        }
        for (int e = b.tid; e < N * N; e += b.nthr) {
            const int al = cg_udiv(e, l.mN), ga = e - al * N;
            t2 += Jinv[al * N + ga] * Jj[ga * N + al].dd;
        }
        cg_gemm_wg(b, N, N, N, [&](int r, int k) { return Jinv[r * N + k]; }, [&](int k, int c) { return Jj[k * N + c].d; },
                   [&](int r, int c, double v) { M[r * N + c] = v; });          // M = J^-1 J' (matrix cores)
        b.sync();
        for (int e = b.tid; e < N * N; e += b.nthr) {
            const int al = cg_udiv(e, l.mN), ga = e - al * N;
            t3 += M[al * N + ga] * M[ga * N + al];
        }
        red[0] = p_re; red[1] = p_im; red[2] = t2; red[3] = t3;
        CG_STAMP_END(31)
    }

    template <bool AL>
    static CG_DEVI void grad_laplacian(const CgBlk& b, const double* __restrict__ th, const double* __restrict__ xg,
                                       const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                                       int mode, const double* __restrict__ v, double* __restrict__ grad /*N x 2*/,
                                       double* __restrict__ lap /*2*/, double* lds, double* ws, const Lay& l,
                                       double* stash = nullptr, const Stash* st = nullptr) {
        const int N = n * D;
        const Mem<AL> mem(lds, ws, l);
        const bool exact_phi = mode != 1;
        bool early = exact_phi;                      // (set-up: granted on the wave-inverse path with waves to spare)
        CG_STAMP_START(20)
        bool have_C = false;
        setup<AL>(b, th, xg, spk, sidx, n, L, mem, l, early, have_C, stash, st);
        CG_STAMP_END(20)
        double s_re, s_im, q_re = 0, q_im = 0;
        CG_STAMP_START(21)
        slater_part<AL>(b, n, mem, l, exact_phi, grad, s_re, s_im, have_C);      // grad <- J^T g
        b.sync();                                                        // the forward Laplacian's arrays overlay J and C
        CG_STAMP_END(21)
        CG_STAMP_START(23)
        if (exact_phi) forward_laplacian<AL>(b, th, n, L, mem, l, q_re, q_im, early);
        double tot[4] = {s_re + q_re, s_im + q_im, 0.0, 0.0};
        b.sync();                                                        // the adjoints overlay the forward Laplacian's arrays
        CG_STAMP_END(23)
        CG_STAMP_START(22)
        reverse_x<AL>(b, th, n, L, mem, l);
        {
            const double* xbar = mem.p + l.xbar;
            for (int e = b.tid; e < N; e += b.nthr) grad[2 * e] += xbar[e];   // same thread wrote grad[2 e] above
        }
        b.sync();                                    // the jet arena overlays the primal arena
        CG_STAMP_END(22)
        CG_STAMP_START(24)
        if (mode == 0) {
            for (int dir = 0; dir < N; ++dir) {
                double r[4];
                jet_part<AL>(b, th, n, L, mem, l, nullptr, dir, false, r);
                tot[2] += r[2]; tot[3] += r[3];
                b.sync();
            }
        } else {
            double r[4];
            jet_part<AL>(b, th, n, L, mem, l, v, 0, mode == 1, r);
            tot[0] += r[0]; tot[1] += r[1]; tot[2] += r[2]; tot[3] += r[3];
        }
        CG_STAMP_END(24)
        cg_block_sum_n<4>(b, tot, mem.p + l.red);
        if (b.tid == 0) { lap[0] = tot[0] + 0.5 * (tot[2] - tot[3]); lap[1] = tot[1]; }
        b.sync();
    }
};
