// cg_big.hpp -- derivative kernels of the depth-2 flow wave function for the larger systems (n > 16: the production sizes of the
// reference, n = 29 / 49 / 57), second generation.  gfx950 only (device code; spsize = tpsize = 16).
//
// Reference: make_quantum_score (src/logpsi.py:183-203) and make_logpsi_grad_laplacian (src/logpsi.py:55-172).
//
// The first generation at these sizes (cg_derivs.hpp k_param_vjp, cg_lap.hpp k_grad_lap2 with AL = false) kept J^-1 in LDS and every
// other array in a per-workgroup HBM workspace that the (particle, unit) loops read 8 bytes per lane at a time: 28 ... 1300 x the
// algorithmic traffic and 70 - 80 % of the wave cycles waiting.  Here:
//   * a PLAN (CgPlan): every array of the kernel has a life time in phases and a priority; a first-fit interval allocator on the
//     host gives the hot ones LDS (overlaying arrays whose life times do not meet) and sends what does not fit -- the N x N
//     matrices that are written once and read once or twice with coalesced rows -- to the workgroup's workspace slot.  One code path
//     for every n: smaller systems simply keep more in LDS.
//   * the pair loops are ROW PASSES: a DPP row (16 lanes) owns particle i, lane h one hidden unit.  Per block of 16 partners every
//     lane forms the features of ONE pair (i, k) from the half-angle tables and fetches that pair's d x d block of the Jacobian
//     cotangent with one coalesced 16-byte access per row, parks both in a per-row LDS slot, and the row then walks the block with
//     broadcast reads.  No pair table (156 KB at n = 57), no materialised Jhat: J^-T is streamed exactly once per pass.
//   * J^-T (not J^-1) is what the inverse writes: Jhat_ik[a][b] = 1/2 (J^-T[(i,a)][(k,b)] - J^-T[(i,a)][(i,b)]) reads rows only.
//   * the contractions over the row index (Gbar) use the identity  sum_{i != k} Jhat_ik^T B_i = 1/2 (J^-1 B)_k - 1/2 S,
//     S = sum_i (J^-T)_ii^T B_i: a plain MFMA GEMM with J^-T plus a k-independent correction.
#pragma once
#include "cg_flow_fast.hpp"
#include "cg_lap.hpp"
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>

// ------------------------------------------------------------------------------------------------------------
// Host: life-time-aware placement.  Offsets are doubles; >= 0: LDS (relative to the kernel's LDS base behind the tables),
// < 0: workspace slot of the workgroup at ~off.
// ------------------------------------------------------------------------------------------------------------
struct CgPlan {
    struct Item { int* slot; unsigned size; int p0, p1, prio; bool must_lds; const char* name; };
    std::vector<Item> items;
    void add(int& slot, size_t size, int p0, int p1, int prio, bool must_lds = false, const char* name = "") {
        items.push_back({&slot, (unsigned)((size + 1) & ~(size_t)1), p0, p1, prio, must_lds, name});
    }
    // false: an array that must live in LDS does not fit
    bool solve(size_t lds_cap, unsigned& lds_total, unsigned& ws_total) {
        struct Placed { unsigned off, size; int p0, p1; };
        std::vector<Placed> L, W;
        std::vector<int> order(items.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            const Item &x = items[a], &y = items[b];
            if (x.must_lds != y.must_lds) return x.must_lds;
            if (x.must_lds) {                       // all of these must fit: longest-lived and largest first packs best
                const int lx = x.p1 - x.p0, ly = y.p1 - y.p0;
                if (lx != ly) return lx > ly;
                return x.size > y.size;
            }
            if (x.prio != y.prio) return x.prio > y.prio;
            return x.size > y.size;
        });
        auto fit = [](const std::vector<Placed>& pool, const Item& it) {
            std::vector<std::pair<unsigned, unsigned>> busy;
            for (const Placed& p : pool) if (!(p.p1 < it.p0 || it.p1 < p.p0)) busy.push_back({p.off, p.off + p.size});
            std::sort(busy.begin(), busy.end());
            unsigned pos = 0;
            for (const auto& iv : busy) {
                if (pos + it.size <= iv.first) break;
                if (iv.second > pos) pos = iv.second;
            }
            return pos;
        };
        lds_total = 0; ws_total = 0;
        bool ok = true;
        for (int idx : order) {
            const Item& it = items[idx];
            const unsigned pos = fit(L, it);
            if ((size_t)pos + it.size <= lds_cap) {
                L.push_back({pos, it.size, it.p0, it.p1}); *it.slot = (int)pos;
                lds_total = std::max(lds_total, pos + it.size);
            } else {
                if (it.must_lds) ok = false;
                if (getenv("CG_PLAN_DEBUG")) fprintf(stderr, "  plan: %s item #%d (%u doubles, phases %d-%d, prio %d) -> workspace\n", it.must_lds ? "MUST-LDS" : "", idx, it.size, it.p0, it.p1, it.prio);
                const unsigned wp = fit(W, it);
                W.push_back({wp, it.size, it.p0, it.p1}); *it.slot = ~(int)wp;
                ws_total = std::max(ws_total, wp + it.size);
            }
        }
        return ok;
    }
};

#if defined(__HIPCC__)
struct CgPl {
    double* lds; double* ws;
    __device__ __forceinline__ double* operator()(int off) const { return off >= 0 ? lds + off : ws + (size_t)(~off); }
};

template <int D, int HS, int HT>
struct CgBig {
    using F = CgFast<D, HS, HT>;
    static constexpr int P = F::P;
    static constexpr int NP = F::NPARAM;
    static constexpr int KT = P + 1;
    static constexpr int NF = 3 * D + 1;                         // per-pair features of a row pass: c2[D], s2[D], del, td[D]
    static constexpr int SLA = (NF + D * D + 1) & ~1;            // slot of the Jacobian-cotangent pass: + the pair's d x d block
    static constexpr int SLF = (NF + 1) & ~1;                    // feature-only slot
    static constexpr int RSA = 16 * SLA + 2, RSF = 16 * SLF + 2; // row strides (padded: the four rows of a wave on different banks)
    static constexpr int PFWAVE = 4 * (16 * (2 * D + 2) + 2) + 2 * D + 2;   // = CgFast::PFWAVE (per-wave scratch of the pair-primal pass)
    typedef double d2_t __attribute__((ext_vector_type(2)));
    typedef double d4_t __attribute__((ext_vector_type(4)));

    // phases (life times of the plan); scores: ... PASSA PASSB CHAIN PASSC ASM, grad / Laplacian: ... TK SLATER FWD PASSA PASSB CHAIN PAIR JET
    enum { PH_LOAD = 0, PH_PAIRS, PH_DENSE, PH_FACT, PH_JAC, PH_INV, PH_TK, PH_SLATER, PH_FWD, PH_PASSA, PH_PASSB, PH_CHAIN, PH_PASSC, PH_ASM,
           PH_PAIR = PH_PASSC,
           PH_JA = PH_ASM,      // jet pass: half-angle jets, pair sums
           PH_JB, PH_JC,        //           dense tangents; factor tangents, G pass
           PH_JD, PH_JE,        //           pair pass (J', t2); traces
           PH_JET = PH_JA, PH_END = PH_JE };
    static constexpr int SPGB = HS * D + 2;                      // per-particle stride of Gbar (padded like G: read with the particle on lanes)

    // ---- the part of the layout both kernels share: set-up (flow, Jacobian, the two inverses, g)
    struct LayC {
        CgFastLds o;                       // arena of CgFast::primal / jacobian (LDS offsets); o.J = the pair-primal scratch
        unsigned mn, mN; int nw;
        int x, kocc, zb, stage, rscr;      // LDS
        int J, JT, Dm, Dinv;               // placed
        int Kd;                            // diag K^ab (grad / Laplacian only; < 0 ... : see have_Kd)
        int have_Kd;
    };
    struct LayS {
        LayC c;
        int m0k, s1k, m1k, s2k, Uk;        // copies of the primal temporaries the score row needs at the very end; U behind the assembly
        int Upb, Bb, Vb, Gb, sg1b, Ub, Rb, u2b, u2i, u1b, u1i, m1b, m1i, sums /* su2, su2i, gbb, gbbi */;
        int pW0, pWtJ, pWtR, pWtI, pS;     // per-wave partial sums
        unsigned lds_total, ws_total; int ok;
    };
    struct LayG {
        LayC c;
        CgFastLds oj;                      // Jet2 arena of the directional pass (offsets in Jet2 elements from ja)
        int Ta, red, Uk, rscrF;
        int Lm0, gu1, Lm1, Am, Hk, Ls1, Su2, Lgb, Ls2;                     // forward Laplacian
        int Upb, Bb, Vb, Gb, sg1b, Ub, Rb, u2b, u1b, m1b, m0b, sums, pS;   // reverse sweep
        int xrow, colacc;                                                  // pair pass -> xbar
        int jp;                                                            // jet pass: pool of full jets (sh, ch, sg1 | m0, m1 / G), LDS
        int s1t, sg2t, s2t, gbt, zt, Ut, Vt, Bmt, Upt;                     //   (d, dd) tangents, two doubles per entry
        int Jp, M;                                                         //   J' and M = J^-1 J'
        unsigned lds_total, ws_total; int ok;
    };

    // What the score passes need from the set-up, parked in the workspace by the grad / Laplacian part of the fused kernel
    // (k_gradlap_scores_big) and read back into the score layout: the two kernels keep their own plans, the set-up runs once.
    struct Stash { int sh, ch, sg1, sg2, gbar, V, Bm, G, zb, Uk, s1, s2, m1, m0, JT /* only when the grad / Laplacian plan keeps J^-T in LDS */; unsigned total; };
    static Stash stash_layout(int n) {
        const int N = n * D;
        Stash st; int at = 0;
        auto take = [&](int& off, int size) { off = at; at += (size + 1) & ~1; };
        take(st.sh, N); take(st.ch, N); take(st.sg1, n * HS); take(st.sg2, n * HS); take(st.gbar, HS);
        take(st.V, n * F::SPV); take(st.Bm, n * F::SPB); take(st.G, n * F::SPG); take(st.zb, 2 * N); take(st.Uk, N * HS);
        take(st.s1, n * HS); take(st.s2, n * HS); take(st.m1, n * HT); take(st.m0, n * P); take(st.JT, N * N);
        st.total = (unsigned)at;
        return st;
    }

    static void plan_common(CgPlan& pl, LayC& c, int n, int nthr, int wt_last /* last phase that reads the staged two-particle weights */,
                            int x_last, int up_last) {
        const size_t N = (size_t)n * D;
        c.mn = cg_div_magic((unsigned)n); c.mN = cg_div_magic((unsigned)N); c.nw = nthr / 64;
        CgFastLds& o = c.o; memset(&o, 0, sizeof(o));
        const int HOT = 100;
        pl.add(c.x, N, PH_LOAD, x_last, HOT, true);
        pl.add(c.kocc, N, PH_LOAD, PH_TK, HOT, true);
        pl.add(o.sh, N, PH_LOAD, PH_END, HOT, true); pl.add(o.ch, N, PH_LOAD, PH_END, HOT, true);
        pl.add(o.z, N, PH_DENSE, PH_INV, HOT, true);
        pl.add(o.gbar, HS, PH_DENSE, PH_END, HOT, true); pl.add(o.cb, HS, PH_DENSE, PH_DENSE, HOT, true);
        pl.add(o.perm, 4, PH_LOAD, PH_INV, HOT, true);
        pl.add(o.wt, HT * (P + 1) + HS * D, PH_FACT, wt_last, HOT, true);
        pl.add(o.J, (size_t)c.nw * PFWAVE, PH_PAIRS, PH_PAIRS, HOT, true);     // pair-primal scratch (J itself: c.J)
        pl.add(o.Up, N * P, PH_FACT, up_last, HOT, true);
        pl.add(c.stage, cg_inv_panel_scratch((int)N, n, nthr) + 2, PH_INV, PH_INV, HOT, true);
        pl.add(c.Dm, 2 * (size_t)n * n, PH_INV, PH_TK, 1);
        pl.add(c.Dinv, 2 * (size_t)n * n, PH_INV, PH_TK, 1);
    }
    static bool shapes_ok(int n, int nthr) {
        const size_t N = (size_t)n * D;
        return cg_inv_panel_scratch((int)N, n, nthr) != 0 && (N & 1) == 0 && n >= 8 && N <= 128 && n <= 64 && (nthr % 64) == 0;
    }

    static LayS layout_scores(int n, int nthr, size_t lds_cap_doubles) {
        const size_t N = (size_t)n * D;
        LayS l; memset(&l, 0, sizeof(l));
        CgPlan pl;
        plan_common(pl, l.c, n, nthr, PH_JAC, PH_DENSE, PH_JAC);
        CgFastLds& o = l.c.o;
        const int nw = l.c.nw;
        l.c.have_Kd = 0;
        pl.add(l.c.J, N * N, PH_JAC, PH_INV, 0);
        // arena arrays of the flow: LDS while CgFast's code touches them
        pl.add(o.m0, (size_t)n * P, PH_PAIRS, PH_DENSE, 100, true); pl.add(o.m1, (size_t)n * HT, PH_PAIRS, PH_DENSE, 100, true);
        pl.add(o.s1, (size_t)n * HS, PH_DENSE, PH_DENSE, 100, true); pl.add(o.s2, (size_t)n * HS, PH_DENSE, PH_DENSE, 100, true);
        pl.add(o.sg1, (size_t)n * HS, PH_DENSE, PH_ASM, 90, true); pl.add(o.sg2, (size_t)n * HS, PH_DENSE, PH_ASM, 90, true);
        pl.add(o.U, N * HS, PH_FACT, PH_JAC, 100, true); pl.add(l.Uk, N * HS, PH_JAC, PH_ASM, 50);
        pl.add(o.V, (size_t)n * F::SPV, PH_FACT, PH_PASSA, 90, true); pl.add(o.Bm, (size_t)n * F::SPB, PH_FACT, PH_PASSA, 90, true);
        pl.add(o.G, (size_t)n * F::SPG, PH_FACT, PH_PASSA, 90, true);
        // kept copies (read once, by the score row)
        pl.add(l.m0k, (size_t)n * P, PH_DENSE, PH_ASM, 5); pl.add(l.s1k, (size_t)n * HS, PH_DENSE, PH_ASM, 5);
        pl.add(l.m1k, (size_t)n * HT, PH_DENSE, PH_ASM, 5); pl.add(l.s2k, (size_t)n * HS, PH_DENSE, PH_ASM, 5);
        pl.add(l.c.zb, 2 * N, PH_INV, PH_ASM, 95, true);
        pl.add(l.c.JT, N * N, PH_INV, PH_PASSA, 2);
        pl.add(l.c.rscr, (size_t)nw * 4 * RSA, PH_PASSA, PH_PASSC, 100, true);
        pl.add(l.Upb, N * P, PH_PASSA, PH_ASM, 85, true);
        pl.add(l.Bb, N * HS, PH_PASSA, PH_ASM, 40); pl.add(l.Vb, N * HT, PH_PASSA, PH_ASM, 40);
        pl.add(l.Gb, (size_t)n * SPGB, PH_PASSA, PH_PASSB, 88, true);
        pl.add(l.sg1b, (size_t)n * HS, PH_PASSB, PH_CHAIN, 60); pl.add(l.Ub, N * HS, PH_PASSB, PH_ASM, 40);
        pl.add(l.Rb, N * HS, PH_CHAIN, PH_ASM, 40);
        pl.add(l.u2b, (size_t)n * HS, PH_CHAIN, PH_ASM, 70); pl.add(l.u2i, (size_t)n * HS, PH_CHAIN, PH_ASM, 70);
        pl.add(l.u1b, (size_t)n * HS, PH_CHAIN, PH_ASM, 45); pl.add(l.u1i, (size_t)n * HS, PH_CHAIN, PH_ASM, 45);
        pl.add(l.m1b, (size_t)n * HT, PH_CHAIN, PH_PASSC, 75, true); pl.add(l.m1i, (size_t)n * HT, PH_CHAIN, PH_PASSC, 75, true);
        pl.add(l.sums, 4 * HS, PH_CHAIN, PH_ASM, 99, true);
        pl.add(l.pW0, (size_t)nw * HS * P, PH_PASSB, PH_ASM, 99, true); pl.add(l.pWtJ, (size_t)nw * HT * KT, PH_PASSA, PH_ASM, 99, true);
        pl.add(l.pWtR, (size_t)nw * HT * KT, PH_PASSC, PH_ASM, 99, true); pl.add(l.pWtI, (size_t)nw * HT * KT, PH_PASSC, PH_ASM, 99, true);
        pl.add(l.pS, (size_t)nw * D * HS, PH_PASSA, PH_PASSA, 99, true);
        const bool fits = pl.solve(lds_cap_doubles, l.lds_total, l.ws_total);
        l.ok = (fits && shapes_ok(n, nthr)) ? 1 : 0;
        return l;
    }

    // mode: CG_LAP_HUTCHINSON (1) or CG_LAP_HUTCHINSON_SPLIT (2); the exact mode keeps the first-generation kernel
    static LayG layout_gradlap(int n, int nthr, int mode, size_t lds_cap_doubles) {
        const size_t N = (size_t)n * D;
        LayG l; memset(&l, 0, sizeof(l));
        CgPlan pl;
        plan_common(pl, l.c, n, nthr, PH_FWD, PH_DENSE, PH_JD);
        CgFastLds& o = l.c.o;
        const int nw = l.c.nw;
        const bool phi2 = mode == 1;            // the probe pass needs g, T^a, diag K^ab (v^T hess(log phi) v through z', z'')
        l.c.have_Kd = 1;
        pl.add(l.c.J, N * N, PH_JAC, PH_SLATER, 0);
        pl.add(o.m0, (size_t)n * P, PH_PAIRS, PH_DENSE, 100, true); pl.add(o.m1, (size_t)n * HT, PH_PAIRS, PH_DENSE, 100, true);
        pl.add(o.s1, (size_t)n * HS, PH_DENSE, PH_DENSE, 100, true); pl.add(o.s2, (size_t)n * HS, PH_DENSE, PH_DENSE, 100, true);
        pl.add(o.sg1, (size_t)n * HS, PH_DENSE, PH_JB, 90, true); pl.add(o.sg2, (size_t)n * HS, PH_DENSE, PH_JB, 90, true);
        pl.add(o.U, N * HS, PH_FACT, PH_JAC, 100, true); pl.add(l.Uk, N * HS, PH_JAC, PH_JC, 50);
        pl.add(o.V, (size_t)n * F::SPV, PH_FACT, PH_JD, 90, true); pl.add(o.Bm, (size_t)n * F::SPB, PH_FACT, PH_JD, 90, true);
        pl.add(o.G, (size_t)n * F::SPG, PH_FACT, PH_PASSA, 90, true);
        pl.add(l.c.zb, 2 * N, PH_INV, PH_JE, 95, true);
        pl.add(l.c.Kd, 2 * (size_t)D * D * n, PH_INV, phi2 ? PH_JE : PH_SLATER, 95, true);
        pl.add(l.Ta, 2 * (size_t)D * n * n, PH_TK, phi2 ? PH_JE : PH_SLATER, 1);
        pl.add(l.red, 8 * (size_t)(nw + 1), PH_LOAD, PH_JE, 100, true);
        pl.add(l.c.JT, N * N, PH_INV, PH_JE, 2);
        pl.add(l.rscrF, (size_t)nw * 4 * RSF, PH_FWD, PH_FWD, 100, true);              // row scratch of the forward Laplacian (feature slots)
        pl.add(l.c.rscr, (size_t)nw * 4 * RSA, PH_PASSA, PH_PASSB, 100, true);           // ... of the reverse passes (features + the pair's block of Jhat)
        // forward Laplacian
        pl.add(l.Lm0, (size_t)n * P, PH_FWD, PH_FWD, 70); pl.add(l.gu1, (size_t)n * HS, PH_FWD, PH_FWD, 70); pl.add(l.Lm1, (size_t)n * HT, PH_FWD, PH_FWD, 70);
        pl.add(l.Am, (size_t)n * HS * P, PH_FWD, PH_FWD, 30); pl.add(l.Hk, (size_t)n * HS * D, PH_FWD, PH_FWD, 75);
        pl.add(l.Ls1, (size_t)n * HS, PH_FWD, PH_FWD, 70); pl.add(l.Su2, (size_t)n * HS, PH_FWD, PH_FWD, 70);
        pl.add(l.Lgb, HS, PH_FWD, PH_FWD, 99, true); pl.add(l.Ls2, (size_t)n * HS, PH_FWD, PH_FWD, 70);
        // reverse sweep
        pl.add(l.Upb, N * P, PH_PASSA, PH_PASSB, 85, true);
        pl.add(l.Bb, N * HS, PH_PASSA, PH_CHAIN, 40); pl.add(l.Vb, N * HT, PH_PASSA, PH_CHAIN, 40);
        pl.add(l.Gb, (size_t)n * SPGB, PH_PASSA, PH_PAIR, 88, true);
        pl.add(l.sg1b, (size_t)n * HS, PH_PASSB, PH_CHAIN, 60); pl.add(l.Ub, N * HS, PH_PASSB, PH_CHAIN, 40);
        pl.add(l.Rb, N * HS, PH_CHAIN, PH_CHAIN, 40);
        pl.add(l.u2b, (size_t)n * HS, PH_CHAIN, PH_CHAIN, 70); pl.add(l.u1b, (size_t)n * HS, PH_CHAIN, PH_CHAIN, 60);
        pl.add(l.m1b, (size_t)n * HT, PH_CHAIN, PH_PAIR, 80, true); pl.add(l.m0b, (size_t)n * P, PH_CHAIN, PH_PAIR, 80, true);
        pl.add(l.sums, 2 * HS, PH_CHAIN, PH_CHAIN, 99, true);
        pl.add(l.pS, (size_t)nw * D * HS, PH_PASSA, PH_PASSA, 99, true);
        pl.add(l.xrow, N, PH_PAIR, PH_PAIR, 99, true); pl.add(l.colacc, (size_t)nw * 64 * D, PH_PAIR, PH_PAIR, 99, true);
        // jet pass.  Full jets (value, d, dd) only where the DPP row passes of cg_flow_fast.hpp want them -- half-angle tables, pair sums,
        // sg1, G: one LDS pool addressed in Jet2 elements --, (d, dd) tangents next to the arrays of the primal arena everywhere else
        {
            CgFastLds& j = l.oj; memset(&j, 0, sizeof(j)); int t = 0;
            auto tk = [&](int cnt) { int r = t; t += (cnt + 1) & ~1; return r; };
            j.sh = tk(n * D); j.ch = tk(n * D); j.sg1 = tk(n * HS);
            const int base = t;
            j.m0 = tk(n * P); j.m1 = tk(n * HT);                          // dead once the dense tangents exist ...
            const int e1 = t;
            t = base; j.G = tk(n * (HS * D + 2));                         // ... G overlays them
            j.total = t > e1 ? t : e1;
        }
        pl.add(l.jp, 3 * (size_t)l.oj.total, PH_JA, PH_JD, 100, true);
        pl.add(l.s1t, 2 * (size_t)n * HS, PH_JB, PH_JB, 60); pl.add(l.s2t, 2 * (size_t)n * HS, PH_JB, PH_JB, 60);
        pl.add(l.sg2t, 2 * (size_t)n * HS, PH_JB, PH_JC, 60); pl.add(l.gbt, 2 * HS, PH_JB, PH_JB, 99, true);
        pl.add(l.zt, 2 * N, PH_JB, PH_JE, 99, true);
        pl.add(l.Ut, 2 * N * HS, PH_JC, PH_JC, 50);
        pl.add(l.Vt, 2 * (size_t)n * F::SPV, PH_JC, PH_JD, 80); pl.add(l.Bmt, 2 * (size_t)n * F::SPB, PH_JC, PH_JD, 70);
        pl.add(l.Upt, 2 * N * P, PH_JC, PH_JD, 85);
        pl.add(l.Jp, N * N, PH_JD, PH_JE, 0);
        const bool fits = pl.solve(lds_cap_doubles, l.lds_total, l.ws_total);
        l.ok = (fits && shapes_ok(n, nthr) && (mode == 1 || mode == 2)) ? 1 : 0;
        return l;
    }

#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(PFWAVE == F::PFWAVE, "pair-primal scratch size");
    // ------------------------------------------------------------------------------------------------------
    // small wave helpers
    // ------------------------------------------------------------------------------------------------------
    // sum over the 16 lanes of a DPP row; the total lands in lane 15 of the row (fixed order)
    static __device__ __forceinline__ double row_sum15(double v) {
        v += cg_dpp_f64<0x111>(v);      // row_shr:1 (bound_ctrl: lanes without a source add 0)
        v += cg_dpp_f64<0x112>(v);
        v += cg_dpp_f64<0x114>(v);
        v += cg_dpp_f64<0x118>(v);
        return v;
    }
    // sum over the four rows of a wave (lanes h, h + 16, h + 32, h + 48); every lane gets the total
    static __device__ __forceinline__ double rows_sum(double v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }
    // sum over the wave, wave-uniform result: DPP row sums + four readlanes (a ds_bpermute butterfly costs ~5x as much)
    static __device__ __forceinline__ double wave_sum(double v) {
        const double r = row_sum15(v);
        return (cg_readlane_f64(r, 15) + cg_readlane_f64(r, 31)) + (cg_readlane_f64(r, 47) + cg_readlane_f64(r, 63));
    }

    // sum over the lanes of a row group of the pair passes: the whole wave (two = false) or its half of 32 lanes (two = true)
    static __device__ __forceinline__ double group_sum(double v, bool two, int sub) {
        const double r = row_sum15(v);
        const double lo = cg_readlane_f64(r, 15) + cg_readlane_f64(r, 31), hi = cg_readlane_f64(r, 47) + cg_readlane_f64(r, 63);
        return two ? (sub ? hi : lo) : lo + hi;
    }

    struct Feat { double c2[D], s2[D], del, td[D]; };
    // features of the lane's own pair (i, k); !ok: finite filler nobody reads.  rdel_out: 1 / |sin| (0 for the diagonal pair)
    static __device__ __forceinline__ void own_feat(const double* sh, const double* ch, int i, int k, bool ok, double c2c, Feat& f, double* rdel_out = nullptr) {
        typename F::PF6 pf; F::own_pair(sh, ch, i, k, ok, pf);
#pragma unroll
        for (int a = 0; a < D; ++a) { f.c2[a] = pf.c2[a]; f.s2[a] = pf.s2[a]; f.td[a] = c2c * (pf.s2[a] * pf.rdel); }
        f.del = pf.del;
        if (rdel_out) *rdel_out = pf.rdel;
    }
    static __device__ __forceinline__ void put_feat(double* slot, const Feat& f) {
#pragma unroll
        for (int a = 0; a < D; ++a) { slot[a] = f.c2[a]; slot[D + a] = f.s2[a]; slot[2 * D + 1 + a] = f.td[a]; }
        slot[2 * D] = f.del;
    }
    static __device__ __forceinline__ void get_feat(const double* slot, Feat& f) {
#pragma unroll
        for (int a = 0; a < D; ++a) { f.c2[a] = slot[a]; f.s2[a] = slot[D + a]; f.td[a] = slot[2 * D + 1 + a]; }
        f.del = slot[2 * D];
    }

    // ------------------------------------------------------------------------------------------------------
    // set-up shared by both kernels: z and every intermediate of the flow, J (placed), J^-T, D, D^-1, g_ia = T^a_ii -> zb = [Re g | Im g],
    // and (grad / Laplacian) diag K^ab_ii = sum_j D_ij (-k_j^a k_j^b) Dinv_ji from the same products D_ij Dinv_ji
    // ------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void setup(const CgBlk& b, const double* __restrict__ th, const double* __restrict__ xg,
                                                 const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                                                 const CgPl& pl, const LayC& c, typename F::WFrag& wf) {
        const int N = n * D;
        double* lds = pl.lds;
        const CgFastLds& o = c.o;
        double* x = lds + c.x; double* kocc = lds + c.kocc;
        for (int e = b.tid; e < N; e += b.nthr) {
            x[e] = xg[e];
            const int j = e / D;
            kocc[e] = spk[(size_t)sidx[j] * D + (e - j * D)];
        }
        b.sync();
        F::load_frags(th, wf, true);
        wf.Jext = pl(c.J);
        CG_STAMP_START(25)
        F::primal(b, th, (const double*)x, n, L, lds, o, &wf);
        CG_STAMP_END(25)
    }
    static __device__ __forceinline__ void setup2(const CgBlk& b, const double* __restrict__ th, int n, double L, const CgPl& pl, const LayC& c,
                                                  typename F::WFrag& wf, double* Uk) {
        const int N = n * D;
        double* lds = pl.lds;
        const CgFastLds& o = c.o;
        const double* kocc = lds + c.kocc;
        CG_STAMP_START(26)
        F::jacobian(b, th, n, L, lds, o, &wf);
        for (int e = b.tid; e < N * HS; e += b.nthr) Uk[e] = lds[o.U + e];      // U leaves the arena (read again by pass B and the assembly)
        b.sync();
        CG_STAMP_END(26)
        CG_STAMP_START(27)
        double* J = pl(c.J); double* JT = pl(c.JT); double* Dm = pl(c.Dm); double* Dinv = pl(c.Dinv);
        double* stg = lds + c.stage;        // (16-byte aligned: even offset behind the 16-byte aligned LDS base; NO integer round trip -- the
                                            //  compiler must keep seeing an LDS pointer, or every panel access becomes a flat instruction: measured 2x)
        CG_STAMP_START(0)
        cg_inverse_panel_real(b, J, N, N, JT, N, stg, true);               // J^-T
        CG_STAMP_END(0)
        F::slater_matrix(b, lds + o.z, kocc, nullptr, n, Dm);
        cg_inverse_panel_complex(b, Dm, n, n, Dinv, n, stg);
        CG_STAMP_END(27)
        CG_STAMP_START(28)
        // a wave per row i, lanes over j
        double* zb = lds + c.zb; double* Kd = lds + c.Kd;
        const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
        for (int i = wave; i < n; i += nw) {
            double gr[D], gi[D], kr[D][D], ki[D][D];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                gr[a] = 0.0; gi[a] = 0.0;
#pragma unroll
                for (int bb = 0; bb < D; ++bb) { kr[a][bb] = 0.0; ki[a][bb] = 0.0; }
            }
            for (int j = lane; j < n; j += 64) {
                const d2_t dv = *(const d2_t*)(Dm + 2 * (i * n + j)), yv = *(const d2_t*)(Dinv + 2 * (j * n + i));
                const CgCplx p = cmul({dv[0], dv[1]}, {yv[0], yv[1]});
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double ka = kocc[j * D + a];
                    gr[a] = fma(-ka, p.im, gr[a]); gi[a] = fma(ka, p.re, gi[a]);
                    if (c.have_Kd) {
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) { const double kk = -ka * kocc[j * D + bb]; kr[a][bb] = fma(kk, p.re, kr[a][bb]); ki[a][bb] = fma(kk, p.im, ki[a][bb]); }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const double r = wave_sum(gr[a]), s = wave_sum(gi[a]);
                if (lane == 0) { zb[i * D + a] = r; zb[N + i * D + a] = s; }
                if (c.have_Kd) {
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        const double r2 = wave_sum(kr[a][bb]), s2 = wave_sum(ki[a][bb]);
                        if (lane == 0) { Kd[2 * ((a * D + bb) * n + i)] = r2; Kd[2 * ((a * D + bb) * n + i) + 1] = s2; }
                    }
                }
            }
        }
        b.sync();
        CG_STAMP_END(28)
    }

    // ------------------------------------------------------------------------------------------------------
    // the reverse sweep through the Jacobian assembly, shared by both kernels (SC: per-sample scores -- the parameter partials
    // are accumulated as well)
    // ------------------------------------------------------------------------------------------------------
    struct Rev {
        const double *sh, *ch, *sg1, *sg2, *U, *V, *Bm, *G, *JT;
        double *rscr, *Upb, *Bb, *Vb, *Gb, *sg1b, *Ub, *Rb, *pS, *pWtJ, *pW0;
    };

    // pass A (J5): U'bar_i, Bbar_i, Vbar_i; S for Gbar; (SC) the sigma_t / q_t adjoints -> partial Wtbar / btbar
    template <bool SC>
    static __device__ __forceinline__ void pass_a(const CgBlk& b, const double* __restrict__ th, int n, double L, const Rev& r) {
        const int N = n * D;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        const int lane = b.tid & 63, h = lane & 15, rg = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        const int grp = wave * 4 + rg, ngrp = nw * 4;
        const double *sh = r.sh, *ch = r.ch, *JT = r.JT, *V = r.V, *Bm = r.Bm, *G = r.G;
        double wc[D], wsn[D], qc[D], qs[D];
        const double bt = th[F::o_t0b + h], wd = th[F::o_t0w + 2 * D * HT + h];
#pragma unroll
        for (int a = 0; a < D; ++a) { wc[a] = th[F::o_t0w + a * HT + h]; wsn[a] = th[F::o_t0w + (D + a) * HT + h]; qc[a] = -c1 * wc[a]; qs[a] = c1 * wsn[a]; }
        double As2[D], Ac2[D], Bc2[D], Bs2[D], Atd = 0.0, Bdel = 0.0, B1 = 0.0, sS[D];
#pragma unroll
        for (int a = 0; a < D; ++a) { As2[a] = 0.0; Ac2[a] = 0.0; Bc2[a] = 0.0; Bs2[a] = 0.0; sS[a] = 0.0; }
        double* myrow = r.rscr + (size_t)grp * RSA;
        for (int i0 = 0; i0 < n; i0 += ngrp) {
            if (i0 + wave * 4 >= n) break;                  // (wave-uniform) nothing but filler rows left for this wave
            const int i = i0 + grp; const bool rowok = i < n; const int ic = rowok ? i : n - 1;
            double jd[D][D], vi[D];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                vi[a] = rowok ? V[F::iV(ic, a, h)] : 0.0;   // a filler row (i >= n, repeats row n - 1) adds exact zeros to the wave partials
#pragma unroll
                for (int bb = 0; bb < D; ++bb) jd[a][bb] = JT[(size_t)(ic * D + a) * N + ic * D + bb];
            }
            if (rowok) {
#pragma unroll
                for (int bb = 0; bb < D; ++bb)
#pragma unroll
                    for (int a = 0; a < D; ++a) sS[bb] = fma(jd[a][bb], Bm[F::iB(ic, a, h)], sS[bb]);
            }
            double vb[D], bbv[D], ups[D][D], upc[D][D], upd[D];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                vb[a] = 0.0; bbv[a] = 0.0; upd[a] = 0.0;
#pragma unroll
                for (int bb = 0; bb < D; ++bb) { ups[a][bb] = 0.0; upc[a][bb] = 0.0; }
            }
            for (int kb = 0; kb < n; kb += 16) {
                {   // own pair (ic, kb + h): features + its block of Jhat
                    const int k = kb + h; const bool ok = k < n; const int kc = ok ? k : ic;
                    Feat f; own_feat(sh, ch, ic, kc, ok, c2c, f);
                    double jh[D][D];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const double* row = JT + (size_t)(ic * D + a) * N + kc * D;
                        if (D == 2) { const d2_t t = *(const d2_t*)row; jh[a][0] = t[0]; jh[a][1] = t[1]; }
                        else {
#pragma unroll
                            for (int bb = 0; bb < D; ++bb) jh[a][bb] = row[bb];
                        }
                    }
                    const bool live = ok && kc != ic;
#pragma unroll
                    for (int a = 0; a < D; ++a)
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) {
                            jh[a][bb] = live ? 0.5 * (jh[a][bb] - jd[a][bb]) : 0.0;
                            ups[a][bb] = fma(jh[a][bb], f.s2[bb], ups[a][bb]);
                            upc[a][bb] = fma(jh[a][bb], f.c2[bb], upc[a][bb]);
                            upd[a] = fma(jh[a][bb], f.td[bb], upd[a]);
                        }
                    double* slot = myrow + h * SLA;
                    put_feat(slot, f);
#pragma unroll
                    for (int a = 0; a < D; ++a)
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) slot[NF + a * D + bb] = jh[a][bb];
                }
                asm volatile("" ::: "memory");             // cross-lane hand-off inside the wave (LDS executes a wave's accesses in order)
                const int jn = n - kb < 16 ? n - kb : 16;
                for (int jj = 0; jj < jn; ++jj) {
                    const double* slot = myrow + jj * SLA;
                    Feat f; get_feat(slot, f);
                    double jh[D][D];
#pragma unroll
                    for (int a = 0; a < D; ++a)
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) jh[a][bb] = slot[NF + a * D + bb];
                    double u = fma(wd, f.del, bt), q[D];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        u = fma(wc[a], f.c2[a], u); u = fma(wsn[a], f.s2[a], u);
                        q[a] = fma(qc[a], f.s2[a], fma(qs[a], f.c2[a], wd * f.td[a]));
                    }
                    const double sg = sigmoid_only(u);
                    double sgb = 0.0, qb[D];
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        const double sq = sg * q[bb];
#pragma unroll
                        for (int a = 0; a < D; ++a) vb[a] = fma(-jh[a][bb], sq, vb[a]);
                        if (SC) {
                            double jv = 0.0;
#pragma unroll
                            for (int a = 0; a < D; ++a) jv = fma(jh[a][bb], vi[a], jv);
                            sgb = fma(-jv, q[bb], sgb);
                            qb[bb] = -jv * sg;
                        }
                    }
                    const double* gk = G + F::iG(kb + jj, h, 0);
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        const double g = gk[bb];
#pragma unroll
                        for (int a = 0; a < D; ++a) bbv[a] = fma(jh[a][bb], g, bbv[a]);
                    }
                    if (SC) {
                        const double ub = sgb * (sg * (1.0 - sg));
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) {
                            As2[bb] = fma(qb[bb], f.s2[bb], As2[bb]); Ac2[bb] = fma(qb[bb], f.c2[bb], Ac2[bb]); Atd = fma(qb[bb], f.td[bb], Atd);
                            Bc2[bb] = fma(ub, f.c2[bb], Bc2[bb]); Bs2[bb] = fma(ub, f.s2[bb], Bs2[bb]);
                        }
                        Bdel = fma(ub, f.del, Bdel); B1 += ub;
                    }
                }
                asm volatile("" ::: "memory");             // the next block's stores stay behind these reads
            }
            if (rowok) {
#pragma unroll
                for (int a = 0; a < D; ++a) { r.Vb[(i * D + a) * HT + h] = vb[a]; r.Bb[(i * D + a) * HS + h] = bbv[a]; }
            }
#pragma unroll
            for (int a = 0; a < D; ++a) {
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const double s = row_sum15(ups[a][bb]), cc = row_sum15(upc[a][bb]);
                    if (rowok && h == 15) { r.Upb[(i * D + a) * P + bb] = c1 * s; r.Upb[(i * D + a) * P + D + bb] = -c1 * cc; }
                }
                const double dd = row_sum15(upd[a]);
                if (rowok && h == 15) r.Upb[(i * D + a) * P + 2 * D] = -dd;
            }
        }
        // wave partials: S, and (SC) Wtbar / btbar (Jacobian part)
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const double v3 = rows_sum(sS[a]);
            if (lane < 16) r.pS[(wave * D + a) * HS + h] = v3;
            if (SC) {
                const double v1 = rows_sum(fma(-c1, As2[a], Bc2[a])), v2 = rows_sum(fma(c1, Ac2[a], Bs2[a]));
                if (lane < 16) { r.pWtJ[(wave * HT + h) * KT + a] = v1; r.pWtJ[(wave * HT + h) * KT + D + a] = v2; }
            }
        }
        if (SC) {
            const double v1 = rows_sum(Atd + Bdel), v2 = rows_sum(B1);
            if (lane < 16) { r.pWtJ[(wave * HT + h) * KT + 2 * D] = v1; r.pWtJ[(wave * HT + h) * KT + P] = v2; }
        }
    }

    // Gbar_k[g][b] = sum_{i != k} sum_a Jhat_ik[a][b] B_i[a][g] = 1/2 ( (J^-T)^T B )[(k,b)][g] - 1/2 S[b][g]   (MFMA; J^-T streamed once)
    static __device__ __forceinline__ void gb_gemm(const CgBlk& b, int n, const Rev& r) {
        const int N = n * D;
        const int lane = b.tid & 63, col = lane & 15, kq = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        const int tiles = (N + 15) >> 4;
        const double *JT = r.JT, *Bm = r.Bm;
        for (int t = wave; t < tiles; t += nw) {
            const int row = 16 * t + col; const bool rok = row < N; const int rc = rok ? row : 0;
            d4_t acc = {0, 0, 0, 0};
#pragma unroll 8
            for (int k0 = 0; k0 < N; k0 += 4) {
                const int kk = k0 + kq; const bool kok = kk < N; const int rr = kok ? kk : 0;
                const double av = (rok && kok) ? JT[(size_t)rr * N + rc] : 0.0;
                const double bv = kok ? Bm[F::iB(rr / D, rr % D, col)] : 0.0;
                acc = F::mfma(av, bv, acc);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rr = 16 * t + kq + 4 * r4;
                if (rr < N) {
                    const int k = rr / D, bb = rr - k * D;
                    double s = 0.0;
                    for (int w = 0; w < nw; ++w) s += r.pS[(w * D + bb) * HS + col];
                    r.Gb[k * SPGB + col * D + bb] = 0.5 * (acc[r4] - s);
                }
            }
        }
    }

    // pass B (J4 + J3): sg1bar_p[h], Ubar_p; (SC) partial W0bar
    template <bool SC>
    static __device__ __forceinline__ void pass_b(const CgBlk& b, const double* __restrict__ th, int n, double L, const Rev& r) {
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        const int lane = b.tid & 63, h = lane & 15, rg = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        const int grp = wave * 4 + rg, ngrp = nw * 4;
        const double *sh = r.sh, *ch = r.ch, *Gb = r.Gb, *Upb = r.Upb, *U = r.U, *sg1 = r.sg1;
        double w_c[D], w_s[D];
#pragma unroll
        for (int a = 0; a < D; ++a) { w_c[a] = th[F::o_W0 + a * HS + h]; w_s[a] = th[F::o_W0 + (D + a) * HS + h]; }
        const double w_d = th[F::o_W0 + 2 * D * HS + h];
        double pw[P];
#pragma unroll
        for (int f = 0; f < P; ++f) pw[f] = 0.0;
        double* myrow = r.rscr + (size_t)grp * RSF;
        for (int p0 = 0; p0 < n; p0 += ngrp) {
            if (p0 + wave * 4 >= n) break;
            const int p = p0 + grp; const bool rowok = p < n; const int pc = rowok ? p : n - 1;
            double gp[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) gp[bb] = Gb[pc * SPGB + h * D + bb];
            double sb = 0.0, As2[D], Ac2[D], Atd = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) { As2[a] = 0.0; Ac2[a] = 0.0; }
            for (int qb0 = 0; qb0 < n; qb0 += 16) {
                {
                    const int q = qb0 + h; const bool ok = q < n;
                    Feat f; own_feat(sh, ch, pc, ok ? q : pc, ok, c2c, f);
                    put_feat(myrow + h * SLF, f);
                }
                asm volatile("" ::: "memory");
                const int jn = n - qb0 < 16 ? n - qb0 : 16;
                for (int jj = 0; jj < jn; ++jj) {
                    Feat f; get_feat(myrow + jj * SLF, f);
                    const double* gq = Gb + (qb0 + jj) * SPGB + h * D;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        const double dG = gp[bb] - gq[bb];                 // (q = p: exactly zero)
                        const double q0 = fma(-c1 * w_c[bb], f.s2[bb], fma(c1 * w_s[bb], f.c2[bb], w_d * f.td[bb]));
                        sb = fma(dG, q0, sb);
                        if (SC) { As2[bb] = fma(dG, f.s2[bb], As2[bb]); Ac2[bb] = fma(dG, f.c2[bb], Ac2[bb]); Atd = fma(dG, f.td[bb], Atd); }
                    }
                }
                asm volatile("" ::: "memory");
            }
            const double sg1p = sg1[pc * HS + h];
            if (SC) {
                const double sgp = rowok ? sg1p * (rn * rn) : 0.0;
#pragma unroll
                for (int bb = 0; bb < D; ++bb) { pw[bb] = fma(-c1 * sgp, As2[bb], pw[bb]); pw[D + bb] = fma(c1 * sgp, Ac2[bb], pw[D + bb]); }
                pw[2 * D] = fma(sgp, Atd, pw[2 * D]);
            }
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const double* up = Upb + (pc * D + a) * P;
                double t = w_d * up[2 * D];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) t = fma(up[bb], w_c[bb], fma(up[D + bb], w_s[bb], t));
                acc = fma(t, U[(pc * D + a) * HS + h], acc);
                if (rowok) r.Ub[(p * D + a) * HS + h] = t * rn * sg1p;
            }
            if (rowok) r.sg1b[p * HS + h] = sb * (rn * rn) + acc * rn;
        }
        if (SC) {
#pragma unroll
            for (int f = 0; f < P; ++f) { const double v = rows_sum(pw[f]); if (lane < 16) r.pW0[(wave * HS + h) * P + f] = v; }
        }
    }

    // (J2) Rbar_i[a][h] = sum_g Ubar Wa[g][h] + Bbar Wb[g][h] + (1/n) Vbar Wc[g][h]
    static __device__ __forceinline__ void rb_gemm(const CgBlk& b, const double* __restrict__ th, int n, const Rev& r) {
        const int N = n * D;
        const double rn = 1.0 / (double)n;
        const double *Ub = r.Ub, *Bb = r.Bb, *Vb = r.Vb; double* Rb = r.Rb;
        cg_gemm_wg(b, N, HS, 2 * HS + HT,
                   [&](int rr, int k) { return k < HS ? Ub[rr * HS + k] : k < 2 * HS ? Bb[rr * HS + k - HS] : rn * Vb[rr * HT + k - 2 * HS]; },
                   [&](int k, int hh) { return th[F::o_Wa + k * HS + hh]; }, [&](int rr, int hh, double v) { Rb[rr * HS + hh] = v; });
    }

    // ------------------------------------------------------------------------------------------------------
    // per-sample scores: set-up + ONE reverse sweep for both parts (the algorithm of cg_score.hpp; phase names follow it)
    // ------------------------------------------------------------------------------------------------------
    static __device__ __forceinline__ void scores(const CgBlk& b, const double* __restrict__ th, const double* __restrict__ xg,
                                                  const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                                                  double* __restrict__ score /* NP x 2 */, double* lds, double* ws, const LayS& l) {
        const int N = n * D;
        const CgPl pl{lds, ws};
        const LayC& c = l.c;
        const CgFastLds& o = c.o;
        typename F::WFrag wf;
        CG_STAMP_START(20)
        setup(b, th, xg, spk, sidx, n, L, pl, c, wf);
        {   // the primal temporaries the score row needs at the end leave the arena (their slots are reused from here on)
            double *m0k = pl(l.m0k), *s1k = pl(l.s1k), *m1k = pl(l.m1k), *s2k = pl(l.s2k);
            for (int e = b.tid; e < n * HS; e += b.nthr) { s1k[e] = lds[o.s1 + e]; s2k[e] = lds[o.s2 + e]; m1k[e] = lds[o.m1 + e]; }
            for (int e = b.tid; e < n * P; e += b.nthr) m0k[e] = lds[o.m0 + e];
            b.sync();
        }
        setup2(b, th, n, L, pl, c, wf, pl(l.Uk));
        CG_STAMP_END(20)
        score_passes(b, th, n, L, score, lds, ws, l, pl(c.JT));
    }
    // copy of `count` doubles (even, both sides 16-byte aligned) by the whole workgroup
    static __device__ __forceinline__ void copy2(const CgBlk& b, double* dst, const double* src, int count) {
        for (int e = 2 * b.tid; e < count; e += 2 * b.nthr) *(d2_t*)(dst + e) = *(const d2_t*)(src + e);
    }
    // the set-up results of the stash into the places the score layout gives them (fused kernel: instead of setup / setup2)
    static __device__ __forceinline__ void scores_unstash(const CgBlk& b, int n, double* lds, double* ws, const LayS& l, const double* stash, const Stash& st) {
        const int N = n * D;
        const CgPl pl{lds, ws};
        const CgFastLds& o = l.c.o;
        auto ev = [](int v) { return (v + 1) & ~1; };
        copy2(b, lds + o.sh, stash + st.sh, ev(N)); copy2(b, lds + o.ch, stash + st.ch, ev(N));
        copy2(b, lds + o.sg1, stash + st.sg1, n * HS); copy2(b, lds + o.sg2, stash + st.sg2, n * HS);
        copy2(b, lds + o.gbar, stash + st.gbar, HS);
        copy2(b, lds + o.V, stash + st.V, ev(n * F::SPV)); copy2(b, lds + o.Bm, stash + st.Bm, ev(n * F::SPB)); copy2(b, lds + o.G, stash + st.G, ev(n * F::SPG));
        copy2(b, lds + l.c.zb, stash + st.zb, 2 * N);
        copy2(b, pl(l.Uk), stash + st.Uk, N * HS);
        copy2(b, pl(l.s1k), stash + st.s1, n * HS); copy2(b, pl(l.s2k), stash + st.s2, n * HS);
        copy2(b, pl(l.m1k), stash + st.m1, n * HT); copy2(b, pl(l.m0k), stash + st.m0, ev(n * P));
        b.sync();
    }
    // everything after the set-up: reverse passes and the score row.  JT = J^-T (the score layout's own, or the one the grad / Laplacian
    // part of the fused kernel left in its workspace)
    static __device__ __forceinline__ void score_passes(const CgBlk& b, const double* __restrict__ th, int n, double L,
                                                        double* __restrict__ score /* NP x 2 */, double* lds, double* ws, const LayS& l, const double* JT) {
        const int N = n * D;
        const CgPl pl{lds, ws};
        const LayC& c = l.c;
        const CgFastLds& o = c.o;
        CG_STAMP_START(13)
        const double *sh = lds + o.sh, *ch = lds + o.ch, *sg1 = lds + o.sg1, *sg2 = lds + o.sg2, *U = pl(l.Uk), *gbar = lds + o.gbar;
        const double* zr = lds + c.zb; const double* zi = zr + N;
        const double *m0 = pl(l.m0k), *s1 = pl(l.s1k), *m1 = pl(l.m1k), *s2 = pl(l.s2k);
        double *Upb = lds + l.Upb, *Bb = pl(l.Bb), *Vb = pl(l.Vb), *sg1b = pl(l.sg1b), *Ub = pl(l.Ub), *Rb = pl(l.Rb),
               *u2b = pl(l.u2b), *u2i = pl(l.u2i), *u1b = pl(l.u1b), *u1i = pl(l.u1i), *m1b = lds + l.m1b, *m1i = lds + l.m1i,
               *su2 = lds + l.sums, *su2i = su2 + HS, *gbb = su2 + 2 * HS, *gbbi = su2 + 3 * HS,
               *pW0 = lds + l.pW0, *pWtJ = lds + l.pWtJ, *pWtR = lds + l.pWtR, *pWtI = lds + l.pWtI;
        const Rev rv{sh, ch, sg1, sg2, U, lds + o.V, lds + o.Bm, lds + o.G, JT, lds + c.rscr, Upb, Bb, Vb, lds + l.Gb, sg1b, Ub, Rb,
                     lds + l.pS, pWtJ, pW0};
        const double rn = 1.0 / (double)n;
        const double c2c = CG_PI / (2.0 * L);
        const int lane = b.tid & 63, h = lane & 15, rg = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        const int grp = wave * 4 + rg, ngrp = nw * 4;
        pass_a<true>(b, th, n, L, rv);
        b.sync();
        CG_STAMP_END(13)
        CG_STAMP_START(14)
        gb_gemm(b, n, rv);
        b.sync();
        CG_STAMP_END(14)
        CG_STAMP_START(19)
        pass_b<true>(b, th, n, L, rv);
        b.sync();
        CG_STAMP_END(19)
        CG_STAMP_START(21)
        rb_gemm(b, th, n, rv);
        b.sync();
        // ---- (J1) (F8) (F7) u2bar, both parts
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, hh = e - i * HS;
            double sb = 0, sr = 0, si = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const double wfv = th[F::o_fw + hh * D + a];
                sb += Rb[(i * D + a) * HS + hh] * wfv; sr += wfv * zr[i * D + a]; si += wfv * zi[i * D + a];
            }
            const double g2 = sg2[e];
            u2b[e] = sr * g2 + sb * g2 * (1.0 - g2);
            u2i[e] = si * g2;
        }
        b.sync();
        for (int hh = b.tid; hh < 2 * HS; hh += b.nthr) {          // sum_i u2bar_i[h], both parts
            const double* src = hh < HS ? u2b : u2i; const int h2 = hh < HS ? hh : hh - HS;
            double acc = 0;
            for (int i = 0; i < n; ++i) acc += src[i * HS + h2];
            (hh < HS ? su2 : su2i)[h2] = acc;
        }
        b.sync();
        for (int g = b.tid; g < 2 * HS; g += b.nthr) {             // gbarbar[g] = sum_h Wb[g][h] su2[h]
            const double* src = g < HS ? su2 : su2i; const int gg = g < HS ? g : g - HS;
            double acc = 0;
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) acc += th[F::o_Wb + gg * HS + hh] * src[hh];
            (g < HS ? gbb : gbbi)[gg] = acc;
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // s1bar, u1bar
            const int i = e / HS, g = e - i * HS;
            double sr = 0, si = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) { const double wfv = th[F::o_fw + g * D + a]; sr += wfv * zr[i * D + a]; si += wfv * zi[i * D + a]; }
            double ar = sr + rn * gbb[g], ai = si + rn * gbbi[g];
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) { const double wa = th[F::o_Wa + g * HS + hh]; ar += wa * u2b[i * HS + hh]; ai += wa * u2i[i * HS + hh]; }
            const double g1 = sg1[e];
            u1b[e] = ar * g1 + sg1b[e] * g1 * (1.0 - g1);
            u1i[e] = ai * g1;
        }
        for (int e = b.tid; e < n * HT; e += b.nthr) {             // m1bar_i[g] = sum_h Wc[g][h] u2bar_i[h]
            const int i = e / HT, g = e - i * HT;
            double ar = 0, ai = 0;
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) { const double wcv = th[F::o_Wc + g * HS + hh]; ar += wcv * u2b[i * HS + hh]; ai += wcv * u2i[i * HS + hh]; }
            m1b[e] = ar; m1i[e] = ai;
        }
        b.sync();
        CG_STAMP_END(21)
        CG_STAMP_START(22)
        // ---- pass C (F4/F5): primal part of the pair stream, both parts from one sigmoid
        {
            double wc[D], wsn[D];
            const double bt = th[F::o_t0b + h], wd = th[F::o_t0w + 2 * D * HT + h];
#pragma unroll
            for (int a = 0; a < D; ++a) { wc[a] = th[F::o_t0w + a * HT + h]; wsn[a] = th[F::o_t0w + (D + a) * HT + h]; }
            double pr[KT], pi[KT];
#pragma unroll
            for (int f = 0; f < KT; ++f) { pr[f] = 0.0; pi[f] = 0.0; }
            double* myrow = rv.rscr + (size_t)grp * RSF;
            for (int i0 = 0; i0 < n; i0 += ngrp) {
                if (i0 + wave * 4 >= n) break;
                const int i = i0 + grp; const bool rowok = i < n; const int ic = rowok ? i : n - 1;
                double a[KT];
#pragma unroll
                for (int f = 0; f < KT; ++f) a[f] = 0.0;
                for (int jb = 0; jb < n; jb += 16) {
                    {
                        const int j = jb + h; const bool ok = j < n;
                        Feat f; own_feat(sh, ch, ic, ok ? j : ic, ok, c2c, f);
                        put_feat(myrow + h * SLF, f);
                    }
                    asm volatile("" ::: "memory");
                    const int jn = n - jb < 16 ? n - jb : 16;
                    for (int jj = 0; jj < jn; ++jj) {
                        Feat f; get_feat(myrow + jj * SLF, f);
                        double u = fma(wd, f.del, bt);
#pragma unroll
                        for (int aa = 0; aa < D; ++aa) { u = fma(wc[aa], f.c2[aa], u); u = fma(wsn[aa], f.s2[aa], u); }
                        const double sg = sigmoid_only(u);
#pragma unroll
                        for (int aa = 0; aa < D; ++aa) { a[aa] = fma(sg, f.c2[aa], a[aa]); a[D + aa] = fma(sg, f.s2[aa], a[D + aa]); }
                        a[2 * D] = fma(sg, f.del, a[2 * D]); a[P] += sg;
                    }
                    asm volatile("" ::: "memory");
                }
                const double mr = rowok ? m1b[ic * HT + h] * rn : 0.0, mi = rowok ? m1i[ic * HT + h] * rn : 0.0;
#pragma unroll
                for (int f = 0; f < KT; ++f) { pr[f] = fma(mr, a[f], pr[f]); pi[f] = fma(mi, a[f], pi[f]); }
            }
#pragma unroll
            for (int f = 0; f < KT; ++f) {
                const double v1 = rows_sum(pr[f]), v2 = rows_sum(pi[f]);
                if (lane < 16) { pWtR[(wave * HT + h) * KT + f] = v1; pWtI[(wave * HT + h) * KT + f] = v2; }
            }
        }
        b.sync();
        CG_STAMP_END(22)
        CG_STAMP_START(23)
        // ---- the score row, one owner thread per parameter (fixed summation order), real and imaginary part side by side
        for (int e = b.tid; e < NP; e += b.nthr) {
            double ar = 0, ai = 0;
            if (e < F::o_fw) {                                          // final.b[a]
                const int a = e - F::o_fb;
                for (int i = 0; i < n; ++i) { ar += zr[i * D + a]; ai += zi[i * D + a]; }
            } else if (e < F::o_s0b) {                                  // final.w[h][a]: (F8) + (J1) + direct term of (J2)
                const int r = e - F::o_fw, hh = r / D, a = r - hh * D;
                for (int i = 0; i < n; ++i) {
                    const double sv = s2[i * HS + hh];
                    ar += sv * zr[i * D + a] + Rb[(i * D + a) * HS + hh] * sg2[i * HS + hh] + Ub[(i * D + a) * HS + hh];
                    ai += sv * zi[i * D + a];
                }
            } else if (e < F::o_s0w) {                                  // sp0.b[h]
                const int hh = e - F::o_s0b;
                for (int i = 0; i < n; ++i) { ar += u1b[i * HS + hh]; ai += u1i[i * HS + hh]; }
            } else if (e < F::o_s1b) {                                  // sp0.w[f'][h]; rows < 2D multiply zeros
                const int r = e - F::o_s0w, fr = r / HS, hh = r - fr * HS;
                if (fr >= 2 * D) {
                    const int f = fr - 2 * D;
                    for (int w = 0; w < nw; ++w) ar += pW0[(w * HS + hh) * P + f];
                    for (int i = 0; i < n; ++i) {
                        const double mv = m0[i * P + f];
                        ar += mv * u1b[i * HS + hh]; ai += mv * u1i[i * HS + hh];
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += rn * Upb[(i * D + a) * P + f] * U[(i * D + a) * HS + hh] * sg1[i * HS + hh];
                    }
                }
            } else if (e < F::o_s1w) {                                  // sp1.b[h]
                ar = su2[e - F::o_s1b]; ai = su2i[e - F::o_s1b];
            } else if (e < F::o_t0b) {                                  // sp1.w rows: Wa (HS), Wb (HS), Wc (HT)
                const int r = e - F::o_s1w, g = r / HS, hh = r - g * HS;
                if (g < HS) {
                    for (int i = 0; i < n; ++i) {
                        const double sv = s1[i * HS + g];
                        ar += sv * u2b[i * HS + hh]; ai += sv * u2i[i * HS + hh];
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += Ub[(i * D + a) * HS + g] * th[F::o_fw + hh * D + a] * sg2[i * HS + hh];
                    }
                } else if (g < 2 * HS) {
                    const int gg = g - HS;
                    ar = gbar[gg] * su2[hh]; ai = gbar[gg] * su2i[hh];
                    for (int i = 0; i < n; ++i)
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += Bb[(i * D + a) * HS + gg] * th[F::o_fw + hh * D + a] * sg2[i * HS + hh];
                } else {
                    const int gg = g - 2 * HS;
                    for (int i = 0; i < n; ++i) {
                        const double mv = m1[i * HT + gg];
                        ar += mv * u2b[i * HS + hh]; ai += mv * u2i[i * HS + hh];
#pragma unroll
                        for (int a = 0; a < D; ++a) ar += rn * Vb[(i * D + a) * HT + gg] * th[F::o_fw + hh * D + a] * sg2[i * HS + hh];
                    }
                }
            } else if (e < F::o_t0w) {                                  // tp0.b[h]
                const int hh = e - F::o_t0b;
                for (int w = 0; w < nw; ++w) { ar += pWtJ[(w * HT + hh) * KT + P] + pWtR[(w * HT + hh) * KT + P]; ai += pWtI[(w * HT + hh) * KT + P]; }
            } else {                                                    // tp0.w[f][h]
                const int r = e - F::o_t0w, f = r / HT, hh = r - f * HT;
                for (int w = 0; w < nw; ++w) { ar += pWtJ[(w * HT + hh) * KT + f] + pWtR[(w * HT + hh) * KT + f]; ai += pWtI[(w * HT + hh) * KT + f]; }
            }
            *(d2_t*)(score + 2 * e) = d2_t{ar, ai};
        }
        CG_STAMP_END(23)
    }

    // ------------------------------------------------------------------------------------------------------
    // grad_x log Psi and laplacian_x log Psi (Hutchinson / Hutchinson-split; the algorithm of cg_lap.hpp, phase by phase)
    // ------------------------------------------------------------------------------------------------------
    // T^a = D diag(i k^a) D^-1 for all directions as ONE real product on the matrix cores, operands straight from D and D^-1 with
    // 16-byte accesses: K runs over (j, re / im) so that a lane's four K steps are two consecutive complex entries of a row of D
    // (A operand) and of two rows of D^-1 (B operand).  rows (a, i), columns (part, q);  Ta[2 ((a n + i) n + q) + part]
    static __device__ __forceinline__ void ta_gemm(const CgBlk& b, int n, const double* Dm, const double* Dinv, const double* kocc, double* Ta) {
        const int lane = b.tid & 63, col = lane & 15, kq = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        const int M = D * n, NC = 2 * n, tm = (M + 15) >> 4, tn = (NC + 15) >> 4;
        for (int t = wave; t < tm * tn; t += nw) {
            const int ti = t / tn, tj = t - ti * tn;
            const int ra = 16 * ti + col; const bool aok = ra < M; const int rac = aok ? ra : 0;
            const int a = rac / n, i = rac - a * n;
            const int cb = 16 * tj + col; const bool bok = cb < NC; const int cbc = bok ? cb : 0;
            const int part = cbc >= n ? 1 : 0, q = cbc - part * n;
            d4_t acc = {0, 0, 0, 0};
            for (int j0 = 0; j0 < n; j0 += 8) {                         // 8 complex entries = 16 K steps per trip
                const int j = j0 + 2 * kq;
                double av[4], bv[4];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int jj = j + s2; const bool jok = jj < n; const int jc = jok ? jj : 0;
                    const d2_t dv = *(const d2_t*)(Dm + 2 * ((size_t)i * n + jc));
                    const d2_t yv = *(const d2_t*)(Dinv + 2 * ((size_t)jc * n + q));
                    const double ka = (aok && jok) ? kocc[jc * D + a] : 0.0;
                    av[2 * s2] = -ka * dv[1]; av[2 * s2 + 1] = ka * dv[0];
                    bv[2 * s2] = (bok && jok) ? (part ? yv[1] : yv[0]) : 0.0;
                    bv[2 * s2 + 1] = (bok && jok) ? (part ? yv[0] : -yv[1]) : 0.0;
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = F::mfma(av[s4], bv[s4], acc);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rr = 16 * ti + kq + 4 * r4;
                if (rr < M && bok) { const int a2 = rr / n, i2 = rr - a2 * n; Ta[2 * ((size_t)(a2 * n + i2) * n + q) + part] = acc[r4]; }
            }
        }
    }

    // grad <- J^T g (complex g; the caller adds xbar), and this thread's partial sums of
    //   tr(J^T H J) = sum_i sum_ab C_(ia),(ib) K^ab_ii - sum_il sum_ab C_(ia),(lb) T^a_il T^b_li ,   C = J J^T
    // C is never stored: its tiles (upper block triangle, MFMA, operands from J with 16-byte accesses) are contracted where they are formed.
    static __device__ __forceinline__ void slater_part(const CgBlk& b, int n, const double* J, const double* zb, const double* Ta, const double* Kd,
                                                       bool want_lap, double* __restrict__ grad, double& p_re, double& p_im) {
        const int N = n * D;
        const int lane = b.tid & 63, col = lane & 15, kq = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        {   // J^T g: four lanes per column e (rows al = part, part + 4, ...), fixed-order DPP sum inside the quad
            for (int e0 = (b.tid >> 6) << 6; e0 < 4 * N; e0 += b.nthr) {
                const int e = (e0 + lane) >> 2, part = lane & 3; const bool ok = e < N; const int ec = ok ? e : 0;
                double re = 0.0, im = 0.0;
#pragma unroll 4
                for (int al = part; al < N; al += 4) { const double jv = J[(size_t)al * N + ec]; re = fma(zb[al], jv, re); im = fma(zb[N + al], jv, im); }
                re += cg_dpp_f64<0xB1>(re); im += cg_dpp_f64<0xB1>(im);      // quad_perm [1,0,3,2]
                re += cg_dpp_f64<0x4E>(re); im += cg_dpp_f64<0x4E>(im);      // quad_perm [2,3,0,1]
                if (ok && part == 0) *(d2_t*)(grad + 2 * e) = d2_t{re, im};
            }
        }
        p_re = 0.0; p_im = 0.0;
        if (!want_lap) return;
        const int tiles = (N + 15) >> 4, nt = tiles * (tiles + 1) / 2;
        for (int t = wave; t < nt; t += nw) {
            int ti = 0, rem = t;
            while (rem >= tiles - ti) { rem -= tiles - ti; ++ti; }
            const int tj = ti + rem;
            const int ra = 16 * ti + col, cbx = 16 * tj + col;
            const bool aok = ra < N, bok = cbx < N;
            const double* ja = J + (size_t)(aok ? ra : 0) * N; const double* jb = J + (size_t)(bok ? cbx : 0) * N;
            d4_t acc = {0, 0, 0, 0};
            for (int k0 = 0; k0 < N; k0 += 16) {
                const int k = k0 + 4 * kq;
                double av[4], bv[4];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int kk = k + 2 * s2; const bool kok = kk < N;         // (N even: a pair is in or out as a whole)
                    const d2_t x1 = *(const d2_t*)(ja + (kok ? kk : 0)), x2 = *(const d2_t*)(jb + (kok ? kk : 0));
                    av[2 * s2] = (aok && kok) ? x1[0] : 0.0; av[2 * s2 + 1] = (aok && kok) ? x1[1] : 0.0;
                    bv[2 * s2] = (bok && kok) ? x2[0] : 0.0; bv[2 * s2 + 1] = (bok && kok) ? x2[1] : 0.0;
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = F::mfma(av[s4], bv[s4], acc);
            }
            const double wgt = ti == tj ? 1.0 : 2.0;                     // C and the weights are symmetric under (i a) <-> (l b)
            if (bok) {
                const int l = cbx / D, bb = cbx - l * D;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int rr = 16 * ti + kq + 4 * r4;
                    if (rr < N) {
                        const int i = rr / D, a = rr - i * D;
                        const d2_t t1 = *(const d2_t*)(Ta + 2 * ((size_t)(a * n + i) * n + l)), t2 = *(const d2_t*)(Ta + 2 * ((size_t)(bb * n + l) * n + i));
                        const CgCplx pr = cmul({t1[0], t1[1]}, {t2[0], t2[1]});
                        double wr = -pr.re, wi = -pr.im;
                        if (i == l) { wr += Kd[2 * ((a * D + bb) * n + i)]; wi += Kd[2 * ((a * D + bb) * n + i) + 1]; }
                        const double cv = wgt * acc[r4];
                        p_re = fma(cv, wr, p_re); p_im = fma(cv, wi, p_im);
                    }
                }
            }
        }
    }

    // |grad_x u2_i[h]|^2 on the matrix cores (CgLap::su2_mfma with the pair features formed where they are used instead of read
    // from a pair table): a wave owns particle i; rows = partner particle k, K = hidden unit g of the two-particle stream, columns = h.
    static __device__ __forceinline__ void su2_pass(const CgBlk& b, const double* th, int n, double L, const double* sh, const double* ch,
                                                    const double* wt, const double* Am, const double* Hk, double* Su2) {
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
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
                d4_t c[D];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {                      // C operand: rows k = 16 kt + kq + 4 r of column h = col
                    const int k = 16 * kt + kq + 4 * r4;
                    const bool ok = k < n && k != i;
                    Feat f; own_feat(sh, ch, i, ok ? k : i, ok, c2c, f);
#pragma unroll
                    for (int bb = 0; bb < D; ++bb)
                        c[bb][r4] = ok ? Hk[((ok ? k : 0) * HS + col) * D + bb] - rn * (c1 * (Ai[D + bb] * f.c2[bb] - Ai[bb] * f.s2[bb]) + Ai[2 * D] * f.td[bb]) : 0.0;
                }
                const int ka = 16 * kt + col;                         // A operand: row k = 16 kt + col
                const bool oka = ka < n && ka != i;
                Feat fa; own_feat(sh, ch, i, oka ? ka : i, oka, c2c, fa);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    double u = wg[ks][0] + wg[ks][1 + 2 * D] * fa.del, q[D];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        u += wg[ks][1 + a] * fa.c2[a] + wg[ks][1 + D + a] * fa.s2[a];
                        q[a] = c1 * (wg[ks][1 + D + a] * fa.c2[a] - wg[ks][1 + a] * fa.s2[a]) + wg[ks][1 + 2 * D] * fa.td[a];
                    }
                    const double sg = oka ? sigmoid_only(u) : 0.0;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) c[bb] = F::mfma(sg * q[bb], bw[ks], c[bb]);
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) { ssq = fma(c[bb][r4], c[bb][r4], ssq); sm[bb] += c[bb][r4]; }
            }
            ssq = rows_sum(ssq);
#pragma unroll
            for (int bb = 0; bb < D; ++bb) { sm[bb] = rows_sum(sm[bb]); ssq = fma(sm[bb], sm[bb], ssq); }
            if (kq == 0) Su2[i * HS + col] = ssq;
        }
    }

    // Forward Laplacian of the flow (CgLap::forward_laplacian): this thread's partial sums of sum_ia g_ia lap_x z_ia
    static __device__ __forceinline__ void forward_laplacian(const CgBlk& b, const double* __restrict__ th, int n, double L, const CgPl& pl, const LayG& l,
                                                             double& q_re, double& q_im) {
        double* lds = pl.lds;
        const CgFastLds& o = l.c.o;
        const int N = n * D;
        const double *sh = lds + o.sh, *ch = lds + o.ch, *sg1 = lds + o.sg1, *sg2 = lds + o.sg2, *G = lds + o.G;
        const double* zb = lds + l.c.zb;
        double *Lm0 = pl(l.Lm0), *gu1 = pl(l.gu1), *Lm1 = pl(l.Lm1), *Am = pl(l.Am), *Hk = pl(l.Hk), *Ls1 = pl(l.Ls1), *Su2 = pl(l.Su2),
               *Lgb = lds + l.Lgb, *Ls2 = pl(l.Ls2);
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L), pl2 = 4.0 * c2c * c2c;
        const int lane = b.tid & 63, h = lane & 15, rg = lane >> 4, wave = b.tid >> 6, nw = b.nthr >> 6;
        const int grp = wave * 4 + rg, ngrp = nw * 4;
        CG_STAMP_START(13)
        {   // row pass: lap m1_i[h], lap m0_i[f], |grad u1_i[h]|^2 (the two pair loops of CgLap::fwd_pair_sums in one walk over the pairs)
            double wc[D], wsn[D], w0c[D], w0s[D];
            const double bt = th[F::o_t0b + h], wd = th[F::o_t0w + 2 * D * HT + h], w0d = th[F::o_W0 + 2 * D * HS + h];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                wc[a] = th[F::o_t0w + a * HT + h]; wsn[a] = th[F::o_t0w + (D + a) * HT + h];
                w0c[a] = th[F::o_W0 + a * HS + h]; w0s[a] = th[F::o_W0 + (D + a) * HS + h];
            }
            double* myrow = lds + l.rscrF + (size_t)grp * RSF;
            for (int i0 = 0; i0 < n; i0 += ngrp) {
                if (i0 + wave * 4 >= n) break;
                const int i = i0 + grp; const bool rowok = i < n; const int ic = rowok ? i : n - 1;
                double acc = 0.0, raw = 0.0, ssq = 0.0, sq[D];
#pragma unroll
                for (int a = 0; a < D; ++a) sq[a] = 0.0;
                for (int jb = 0; jb < n; jb += 16) {
                    {
                        const int j = jb + h; const bool ok = j < n;
                        double rdel;
                        Feat f; own_feat(sh, ch, ic, ok ? j : ic, ok, c2c, f, &rdel);
                        double l0d = 0.0;                                   // lap_r of the norm feature
#pragma unroll
                        for (int a = 0; a < D; ++a) l0d += pl2 * f.c2[a] - f.td[a] * f.td[a];
                        double* slot = myrow + h * SLF;
                        put_feat(slot, f);
                        slot[NF] = l0d * rdel;
                    }
                    asm volatile("" ::: "memory");
                    const int jn = n - jb < 16 ? n - jb : 16;
                    for (int jj = 0; jj < jn; ++jj) {
                        if (jb + jj == ic) continue;                        // (per row: the j = i term is absent)
                        const double* slot = myrow + jj * SLF;
                        Feat f; get_feat(slot, f);
                        const double l0d = slot[NF];
                        double u = fma(wd, f.del, bt), lu = wd * l0d, gsq = 0.0;
#pragma unroll
                        for (int a = 0; a < D; ++a) {
                            const double wfv = fma(wc[a], f.c2[a], wsn[a] * f.s2[a]);
                            u += wfv; lu = fma(-c1 * c1, wfv, lu);
                            const double q = fma(c1, fma(wsn[a], f.c2[a], -wc[a] * f.s2[a]), wd * f.td[a]);
                            gsq = fma(q, q, gsq);
                            const double q0 = fma(c1, fma(w0s[a], f.c2[a], -w0c[a] * f.s2[a]), w0d * f.td[a]);
                            ssq = fma(q0, q0, ssq); sq[a] += q0;
                        }
                        const double sg = sigmoid_only(u);
                        acc += 2.0 * (sg * lu + sg * (1.0 - sg) * gsq);
                        double fv = l0d;
#pragma unroll
                        for (int a = 0; a < D; ++a) { if (h == a) fv = -c1 * c1 * f.c2[a]; if (h == D + a) fv = -c1 * c1 * f.s2[a]; }
                        raw += 2.0 * fv;
                    }
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int a = 0; a < D; ++a) ssq = fma(sq[a], sq[a], ssq);
                if (rowok) {
                    Lm1[i * HT + h] = acc * rn; gu1[i * HS + h] = ssq * rn * rn;
                    if (h < P) Lm0[i * P + h] = raw * rn;
                }
            }
        }
        CG_STAMP_END(13)
        // per-particle factors of the dense x-gradient of u2:  A_i = Wa^T diag(sg1_i) W0^T (HS x P),  H_k = Wb^T G_k (HS x D)
        CgLap<D, HS, HT>::am_hk_mfma(b, th, n, sg1, G, Am, Hk);
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // lap s1
            const int i = e / HS, hh = e - i * HS;
            double lu = 0;
#pragma unroll
            for (int f = 0; f < P; ++f) lu += th[F::o_W0 + f * HS + hh] * Lm0[i * P + f];
            const double g1 = sg1[e];
            Ls1[e] = g1 * lu + g1 * (1.0 - g1) * gu1[e];
        }
        CG_STAMP_START(14)
        su2_pass(b, th, n, L, sh, ch, lds + o.wt, Am, Hk, Su2);
        b.sync();
        CG_STAMP_END(14)
        for (int hh = b.tid; hh < HS; hh += b.nthr) {
            double a = 0;
            for (int i = 0; i < n; ++i) a += Ls1[i * HS + hh];
            Lgb[hh] = a * rn;
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // lap s2 = lap s1 + sg2 lap u2 + sg2' |grad u2|^2
            const int i = e / HS, hh = e - i * HS;
            double lu = 0;
#pragma unroll
            for (int g = 0; g < HS; ++g) lu += th[F::o_Wa + g * HS + hh] * Ls1[i * HS + g] + th[F::o_Wb + g * HS + hh] * Lgb[g];
#pragma unroll
            for (int g = 0; g < HT; ++g) lu += th[F::o_Wc + g * HS + hh] * Lm1[i * HT + g];
            const double g2 = sg2[e];
            Ls2[e] = Ls1[e] + g2 * lu + g2 * (1.0 - g2) * Su2[e];
        }
        b.sync();
        q_re = 0; q_im = 0;
        for (int e = b.tid; e < N; e += b.nthr) {                  // lap z_ia = sum_h Wf[h][a] lap s2_i[h]
            const int i = e / D, a = e - i * D;
            double lz = 0;
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) lz += th[F::o_fw + hh * D + a] * Ls2[i * HS + hh];
            q_re += zb[e] * lz; q_im += zb[N + e] * lz;
        }
    }

    // xbar = grad_x 1/2 log|det J(x)| added to grad: reverse sweep with Jbar = 1/2 J^-T, zbar = 0 (CgLap::reverse_x)
    static __device__ __forceinline__ void reverse_x(const CgBlk& b, const double* __restrict__ th, int n, double L, const CgPl& pl, const LayG& l,
                                                     double* __restrict__ grad) {
        double* lds = pl.lds;
        const CgFastLds& o = l.c.o;
        const int N = n * D;
        const double *sh = lds + o.sh, *ch = lds + o.ch, *sg1 = lds + o.sg1, *sg2 = lds + o.sg2, *V = lds + o.V, *Up = lds + o.Up;
        const double* JT = pl(l.c.JT);
        double *Upb = lds + l.Upb, *Bb = pl(l.Bb), *Vb = pl(l.Vb), *Gb = lds + l.Gb, *sg1b = pl(l.sg1b), *Ub = pl(l.Ub), *Rb = pl(l.Rb),
               *u2b = pl(l.u2b), *u1b = pl(l.u1b), *m1b = lds + l.m1b, *m0b = lds + l.m0b, *su2 = lds + l.sums, *gbb = su2 + HS,
               *xrow = lds + l.xrow, *colacc = lds + l.colacc;
        const Rev rv{sh, ch, sg1, sg2, pl(l.Uk), V, lds + o.Bm, lds + o.G, JT, lds + l.c.rscr, Upb, Bb, Vb, Gb, sg1b, Ub, Rb, lds + l.pS, nullptr, nullptr};
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L), pl2 = 4.0 * c2c * c2c;
        const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
        CG_STAMP_START(15)
        pass_a<false>(b, th, n, L, rv);
        b.sync();
        gb_gemm(b, n, rv);
        b.sync();
        pass_b<false>(b, th, n, L, rv);
        b.sync();
        CG_STAMP(15)
        rb_gemm(b, th, n, rv);
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // (J1) u2bar = sg2bar sg2'   (zbar = 0: no s2bar)
            const int i = e / HS, hh = e - i * HS;
            double sb = 0;
#pragma unroll
            for (int a = 0; a < D; ++a) sb += Rb[(i * D + a) * HS + hh] * th[F::o_fw + hh * D + a];
            const double g2 = sg2[e];
            u2b[e] = sb * g2 * (1.0 - g2);
        }
        b.sync();
        for (int hh = b.tid; hh < HS; hh += b.nthr) {
            double acc = 0;
            for (int i = 0; i < n; ++i) acc += u2b[i * HS + hh];
            su2[hh] = acc;
        }
        b.sync();
        for (int g = b.tid; g < HS; g += b.nthr) {
            double acc = 0;
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) acc += th[F::o_Wb + g * HS + hh] * su2[hh];
            gbb[g] = acc;
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {             // s1bar, u1bar
            const int i = e / HS, g = e - i * HS;
            double acc = rn * gbb[g];
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) acc += th[F::o_Wa + g * HS + hh] * u2b[i * HS + hh];
            const double g1 = sg1[e];
            u1b[e] = acc * g1 + sg1b[e] * g1 * (1.0 - g1);
        }
        for (int e = b.tid; e < n * HT; e += b.nthr) {             // m1bar_i[g] = sum_h Wc[g][h] u2bar_i[h]
            const int i = e / HT, g = e - i * HT;
            double acc = 0;
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) acc += th[F::o_Wc + g * HS + hh] * u2b[i * HS + hh];
            m1b[e] = acc;
        }
        b.sync();
        for (int e = b.tid; e < n * P; e += b.nthr) {              // m0bar_i[f] = sum_h W0[f][h] u1bar_i[h]
            const int i = e / P, f = e - i * P;
            double acc = 0;
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) acc += th[F::o_W0 + f * HS + hh] * u1b[i * HS + hh];
            m0b[e] = acc;
        }
        b.sync();
        CG_STAMP(16)
        // pair pass: a wave owns row i, lane = partner k (n <= 64): adjoints of the features t0_ik and of their r-derivatives, then
        // rbar_ik; x_p collects sum_q rbar_pq - rbar_qp (row sums by a wave reduction, column sums in per-lane accumulators)
        {
            double cacc[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) cacc[bb] = 0.0;
            // n <= 32: two rows per wave (32 lanes each), otherwise the wave is one row
            const bool two = n <= 32; const int sub = two ? lane >> 5 : 0, rpw = two ? 2 : 1;
            const int k = two ? lane & 31 : lane; const bool kin = k < n; const int kc = kin ? k : 0;
            for (int i0 = wave * rpw; i0 < n; i0 += nw * rpw) {
                const int ir = i0 + sub; const bool rowok = ir < n; const int i = rowok ? ir : n - 1;
                const bool ok = kin && rowok && k != i;
                typename F::PF6 pf; F::own_pair(sh, ch, i, ok ? kc : i, ok, pf);
                double tc[D], ts[D], td[D];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) { tc[bb] = -c1 * pf.s2[bb]; ts[bb] = c1 * pf.c2[bb]; td[bb] = c2c * (pf.s2[bb] * pf.rdel); }
                double Jh[D][D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double* row = JT + (size_t)(i * D + a) * N;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) Jh[a][bb] = ok ? 0.5 * (row[kc * D + bb] - row[i * D + bb]) : 0.0;
                }
                double t0b[P], Tc[D], Ts[D], Td[D];
#pragma unroll
                for (int f = 0; f < P; ++f) t0b[f] = rn * m0b[i * P + f];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    double cc = 0, ss = 0, dd = 0;
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        cc -= Jh[a][bb] * Up[(i * D + a) * P + bb];
                        ss -= Jh[a][bb] * Up[(i * D + a) * P + D + bb];
                        dd -= Jh[a][bb] * Up[(i * D + a) * P + 2 * D];
                    }
                    Tc[bb] = cc; Ts[bb] = ss; Td[bb] = dd;
                }
#pragma unroll 4
                for (int g = 0; g < HS; ++g) {                          // G part: q0bar_ik[g][b] = sg1_i[g] (Gbar_i - Gbar_k)[g][b] / n^2
                    const double s1g = sg1[i * HS + g] * rn * rn;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        const double q0b = s1g * (Gb[i * SPGB + g * D + bb] - Gb[kc * SPGB + g * D + bb]);
                        Tc[bb] = fma(th[F::o_W0 + bb * HS + g], q0b, Tc[bb]);
                        Ts[bb] = fma(th[F::o_W0 + (D + bb) * HS + g], q0b, Ts[bb]);
                        Td[bb] = fma(th[F::o_W0 + 2 * D * HS + g], q0b, Td[bb]);
                    }
                }
#pragma unroll 2
                for (int hh = 0; hh < HT; ++hh) {
                    double wt[P];
#pragma unroll
                    for (int f = 0; f < P; ++f) wt[f] = th[F::o_t0w + f * HT + hh];
                    double u = th[F::o_t0b + hh] + wt[2 * D] * pf.del, q[D];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        u += wt[a] * pf.c2[a] + wt[D + a] * pf.s2[a];
                        q[a] = wt[a] * tc[a] + wt[D + a] * ts[a] + wt[2 * D] * td[a];
                    }
                    const double sg = sigmoid_only(u), sgp = sg * (1.0 - sg);
                    double sgb = 0, qb[D];
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        double jv = 0;
#pragma unroll
                        for (int a = 0; a < D; ++a) jv += Jh[a][bb] * V[F::iV(i, a, hh)];
                        sgb -= jv * q[bb];
                        qb[bb] = -jv * sg;
                    }
                    const double ub = sgb * sgp + rn * m1b[i * HT + hh] * sg;     // adjoint of u_t,ik[h]: Jacobian part + primal part
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        t0b[a] += wt[a] * ub; t0b[D + a] += wt[D + a] * ub;
                        Tc[a] += wt[a] * qb[a]; Ts[a] += wt[D + a] * qb[a]; Td[a] += wt[2 * D] * qb[a];
                    }
                    t0b[2 * D] += wt[2 * D] * ub;
                }
                double tdd = 0;
#pragma unroll
                for (int bb = 0; bb < D; ++bb) tdd += Td[bb] * td[bb];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    double rr = t0b[bb] * tc[bb] + t0b[D + bb] * ts[bb] + t0b[2 * D] * td[bb];
                    rr += -c1 * c1 * (Tc[bb] * pf.c2[bb] + Ts[bb] * pf.s2[bb]);
                    rr += pf.rdel * (Td[bb] * pl2 * pf.c2[bb] - td[bb] * tdd);
                    rr = ok ? rr : 0.0;
                    cacc[bb] -= rr;
                    const double tot = group_sum(rr, two, sub);
                    if (rowok && k == 0) xrow[i * D + bb] = tot;
                }
            }
#pragma unroll
            for (int bb = 0; bb < D; ++bb) colacc[(wave * 64 + lane) * D + bb] = cacc[bb];
        }
        b.sync();
        for (int e = b.tid; e < N; e += b.nthr) {
            const int p = e / D, bb = e - p * D;
            double acc = xrow[e];
            for (int w = 0; w < nw; ++w) {
                acc += colacc[(w * 64 + p) * D + bb];
                if (n <= 32) acc += colacc[(w * 64 + 32 + p) * D + bb];
            }
            grad[2 * e] += acc;
        }
        b.sync();
        CG_STAMP_END(17)
    }

    // ------------------------------------------------------------------------------------------------------
    // second-order jet pass along the probe v: this thread's partial sums of t2 = tr(J^-1 J''), t3 = tr((J^-1 J')^2) and (want_phi2)
    // v^T hess(log phi) v through z', z''.
    // The first generation pushed Jet2 numbers through the generic flow code out of a 528 KB workspace arena.  Here the VALUES are the
    // ones the set-up left in the primal arena (no transcendental is evaluated twice: the jets of the one-particle activations need
    // sigma only), the per-particle quantities carry (d, dd) tangents next to them, the pair passes are the DPP row passes of
    // cg_flow_fast.hpp on a small LDS pool of full jets, J'' is contracted with J^-T where its blocks are formed
    //     t2 = sum_(i != k) sum_ab (J^-T[(i,a)][(k,b)] - J^-T[(i,a)][(i,b)]) J''_ik[a][b]      (J''_ii = -sum_k J''_ik)
    // and only J' (N x N doubles) is written out.
    // ------------------------------------------------------------------------------------------------------
    typedef d2_t T2;
    static __device__ __forceinline__ Jet2 mk(double v, T2 t) { return Jet2(v, t[0], t[1]); }
    static __device__ __forceinline__ T2 tg(const Jet2& j) { return T2{j.d, j.dd}; }
    // tangent of f(u) for a one-particle activation whose sigma(u) = g is known: f' = f1, f'' = f2
    static __device__ __forceinline__ T2 chain(T2 u, double f1, double f2) { return T2{f1 * u[0], fma(f1, u[1], f2 * (u[0] * u[0]))}; }

    static __device__ __forceinline__ void jet_part(const CgBlk& b, const double* __restrict__ th, int n, double L, const CgPl& pl, const LayG& l,
                                                    const double* __restrict__ dir, bool want_phi2, double (&red)[4]) {
        double* lds = pl.lds;
        const int N = n * D;
        const CgFastLds& o = l.c.o; const CgFastLds& oj = l.oj;
        const double *sh = lds + o.sh, *ch = lds + o.ch, *sg1 = lds + o.sg1, *sg2 = lds + o.sg2, *V = lds + o.V, *Bm = lds + o.Bm, *Up = lds + o.Up;
        const double* U = pl(l.Uk);
        const double* JT = pl(l.c.JT); double* Jp = pl(l.Jp);
        const double* zb = lds + l.c.zb; const double* Ta = pl(l.Ta); const double* Kd = lds + l.c.Kd;
        Jet2* jp = (Jet2*)(lds + l.jp);
        Jet2 *shj = jp + oj.sh, *chj = jp + oj.ch, *m0j = jp + oj.m0, *m1j = jp + oj.m1, *sg1j = jp + oj.sg1, *Gj = jp + oj.G;
        T2 *s1t = (T2*)pl(l.s1t), *s2t = (T2*)pl(l.s2t), *sg2t = (T2*)pl(l.sg2t), *gbt = (T2*)(lds + l.gbt), *zt = (T2*)(lds + l.zt),
           *Ut = (T2*)pl(l.Ut), *Vt = (T2*)pl(l.Vt), *Bmt = (T2*)pl(l.Bmt), *Upt = (T2*)pl(l.Upt);
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        const int lane = b.tid & 63, wave = b.tid >> 6, nw = b.nthr >> 6;
        CG_STAMP_START(29)
        // (JA) half-angle jets: a = pi x / L, a' = pi v / L, a'' = 0
        for (int e = b.tid; e < N; e += b.nthr) {
            const double a1 = dir[e] * (CG_PI / L), sv = sh[e], cv = ch[e];
            shj[e] = Jet2(sv, cv * a1, -sv * (a1 * a1)); chj[e] = Jet2(cv, -sv * a1, -cv * (a1 * a1));
        }
        b.sync();
        F::primal_pairs_jet_dpp(b, th, n, jp, oj);                  // m0, m1 as full jets (one sigmoid per (pair, unit))
        b.sync();
        // (JB) one-particle layers: tangents only, sigma from the arena
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            T2 u = {0.0, 0.0};
#pragma unroll
            for (int f = 0; f < P; ++f) u += th[F::o_W0 + f * HS + h] * tg(m0j[i * P + f]);
            const double g = sg1[e], g1 = g * (1.0 - g), g2 = g1 * (1.0 - 2.0 * g);
            s1t[e] = chain(u, g, g1);
            sg1j[e] = mk(g, chain(u, g1, g2));
        }
        b.sync();
        for (int h = b.tid; h < HS; h += b.nthr) {
            T2 a = {0.0, 0.0};
            for (int i = 0; i < n; ++i) a += s1t[i * HS + h];
            gbt[h] = a * rn;
        }
        b.sync();
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            T2 u = {0.0, 0.0};
#pragma unroll
            for (int g = 0; g < HS; ++g) u += th[F::o_Wa + g * HS + h] * s1t[i * HS + g] + th[F::o_Wb + g * HS + h] * gbt[g];
#pragma unroll
            for (int g = 0; g < HT; ++g) u += th[F::o_Wc + g * HS + h] * tg(m1j[i * HT + g]);
            const double g = sg2[e], g1 = g * (1.0 - g), g2 = g1 * (1.0 - 2.0 * g);
            s2t[e] = s1t[e] + chain(u, g, g1);
            sg2t[e] = chain(u, g1, g2);
        }
        b.sync();
        if (want_phi2) {
            for (int e = b.tid; e < N; e += b.nthr) {               // z' = v + Wf^T s2', z'' = Wf^T s2''
                const int i = e / D, a = e - i * D;
                T2 z = {dir[e], 0.0};
#pragma unroll
                for (int h = 0; h < HS; ++h) z += th[F::o_fw + h * D + a] * s2t[i * HS + h];
                zt[e] = z;
            }
        }
        CG_STAMP(29)
        // (JC) tangents of the per-particle left factors (linear in sg2'), of U' (product of U and sg1), and G as full jets
        for (int e = b.tid; e < N * HS; e += b.nthr) {
            const int r = e / HS, g = e - r * HS, i = r / D, a = r - i * D;
            T2 ua = {0.0, 0.0}, ub = {0.0, 0.0}, vc = {0.0, 0.0};
#pragma unroll
            for (int h = 0; h < HS; ++h) {
                const T2 rih = th[F::o_fw + h * D + a] * sg2t[i * HS + h];
                ua += th[F::o_Wa + g * HS + h] * rih; ub += th[F::o_Wb + g * HS + h] * rih; vc += th[F::o_Wc + g * HS + h] * rih;
            }
            Ut[e] = ua; Bmt[F::iB(i, a, g)] = ub; Vt[F::iV(i, a, g)] = vc * rn;
        }
        F::g_pass_jet_dpp(b, th, n, L, jp, oj);                     // G_k (full jets) over the dead pair sums
        b.sync();
        for (int e = b.tid; e < N * P; e += b.nthr) {
            const int r = e / P, f = e - r * P, i = r / D;
            T2 acc = {0.0, 0.0};
#pragma unroll
            for (int g = 0; g < HS; ++g) acc += th[F::o_W0 + f * HS + g] * tg(mk(U[r * HS + g], Ut[r * HS + g]) * sg1j[i * HS + g]);
            Upt[e] = acc * rn;
        }
        b.sync();
        CG_STAMP(30)
        // (JD) pair pass: a wave owns row i, lane = partner k.  J_ik = -U'_i T_ik - V_i diag(sig_t(u_ik)) Wt^T T_ik + B_i G_k as jets
        double t2 = 0.0;
        {
            const bool two = n <= 32; const int sub = two ? lane >> 5 : 0, rpw = two ? 2 : 1;
            const int k = two ? lane & 31 : lane; const bool kin = k < n; const int kc = kin ? k : 0;
            for (int i0 = wave * rpw; i0 < n; i0 += nw * rpw) {
                const int ir = i0 + sub; const bool rowok = ir < n; const int i = rowok ? ir : n - 1;
                const bool ok = kin && rowok && k != i;
                typename F::JPF pf; F::template jet_own_pair<true>(shj, chj, i, kc, ok, pf);
                Jet2 tc[D], ts[D], td[D];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) { tc[bb] = -c1 * pf.s2[bb]; ts[bb] = c1 * pf.c2[bb]; td[bb] = c2c * pf.sr[bb]; }
                Jet2 Jb[D][D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const int r = i * D + a;
                    const Jet2 upd = mk(Up[r * P + 2 * D], Upt[r * P + 2 * D]);
#pragma unroll
                    for (int bb = 0; bb < D; ++bb)
                        Jb[a][bb] = -(mk(Up[r * P + bb], Upt[r * P + bb]) * tc[bb] + mk(Up[r * P + D + bb], Upt[r * P + D + bb]) * ts[bb] + upd * td[bb]);
                }
#pragma unroll 2
                for (int h = 0; h < HT; ++h) {
                    const double wd = th[F::o_t0w + 2 * D * HT + h];
                    Jet2 u = th[F::o_t0b + h] + wd * pf.del, q[D];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const double wc = th[F::o_t0w + a * HT + h], wsn = th[F::o_t0w + (D + a) * HT + h];
                        u += wc * pf.c2[a] + wsn * pf.s2[a];
                        q[a] = wc * tc[a] + wsn * ts[a] + wd * td[a];
                    }
                    const Jet2 sg = cg_sigmoid(u);
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const Jet2 vs = mk(V[F::iV(i, a, h)], Vt[F::iV(i, a, h)]) * sg;
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) Jb[a][bb] -= vs * q[bb];
                    }
                }
#pragma unroll 4
                for (int g = 0; g < HS; ++g) {
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const Jet2 bg = mk(Bm[F::iB(i, a, g)], Bmt[F::iB(i, a, g)]);
#pragma unroll
                        for (int bb = 0; bb < D; ++bb) Jb[a][bb] += bg * Gj[F::iG(kc, g, bb)];
                    }
                }
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double* row = JT + (size_t)(i * D + a) * N;
                    double* out = Jp + (size_t)(i * D + a) * N;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) {
                        const double jd = ok ? Jb[a][bb].d : 0.0, jdd = ok ? Jb[a][bb].dd : 0.0;
                        t2 = fma(ok ? row[kc * D + bb] - row[i * D + bb] : 0.0, jdd, t2);
                        const double ds = group_sum(jd, two, sub);      // J'_ii = -sum_(k != i) J'_ik
                        if (kin && rowok) out[k * D + bb] = k == i ? -ds : jd;
                    }
                }
            }
        }
        b.sync();
        CG_STAMP_END(31)
        CG_STAMP_START(19)
        // (JE) traces
        double p_re = 0, p_im = 0, t3 = 0;
        if (want_phi2) {
            for (int e = b.tid; e < N; e += b.nthr) {
                p_re += zb[e] * zt[e][1]; p_im += zb[N + e] * zt[e][1];
                const int i = e / D, a = e - i * D;
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const double zz = zt[e][0] * zt[i * D + bb][0];
                    p_re += zz * Kd[2 * ((a * D + bb) * n + i)]; p_im += zz * Kd[2 * ((a * D + bb) * n + i) + 1];
                }
            }
            for (int e = b.tid; e < n * n; e += b.nthr) {
                const int i = e / n, q = e - i * n;
                CgCplx yiq = {0, 0}, yqi = {0, 0};
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double zi = zt[i * D + a][0], zq = zt[q * D + a][0];
                    yiq.re += zi * Ta[2 * ((size_t)(a * n + i) * n + q)]; yiq.im += zi * Ta[2 * ((size_t)(a * n + i) * n + q) + 1];
                    yqi.re += zq * Ta[2 * ((size_t)(a * n + q) * n + i)]; yqi.im += zq * Ta[2 * ((size_t)(a * n + q) * n + i) + 1];
                }
                const CgCplx pr = cmul(yiq, yqi);
                p_re -= pr.re; p_im -= pr.im;
            }
        }
        // t3 = tr(M M), M = J^-1 J' = (J^-T)^T J', without storing M: a wave forms the tile M_IJ and, in the same lane layout, the transposed
        // tile (M_JI)^T  (A operand: columns of J' in place of those of J^-T, B operand the other way round), multiplies them lane by lane and
        // sums; I <= J only, the off-diagonal pairs count twice.  Operands straight from the two N x N matrices (rows of 16 lanes contiguous).
        {
            const int col = lane & 15, kq = lane >> 4;
            const int tiles = (N + 15) >> 4, npair = tiles * (tiles + 1) / 2;
            for (int tp = wave; tp < npair; tp += nw) {
                int ti = 0, rem = tp;
                while (rem >= tiles - ti) { rem -= tiles - ti; ++ti; }
                const int tj = ti + rem;
                const int ri = 16 * ti + col, cj = 16 * tj + col;
                const bool iok = ri < N, jok = cj < N;
                const int ric = iok ? ri : 0, cjc = jok ? cj : 0;
                d4_t a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
#pragma unroll 4
                for (int k0 = 0; k0 < N; k0 += 4) {
                    const int k = k0 + kq; const bool kok = k < N; const size_t ko = (size_t)(kok ? k : 0) * N;
                    const double ja = (iok && kok) ? JT[ko + ric] : 0.0, pb = (jok && kok) ? Jp[ko + cjc] : 0.0;
                    const double pa = (iok && kok) ? Jp[ko + ric] : 0.0, jb = (jok && kok) ? JT[ko + cjc] : 0.0;
                    a1 = F::mfma(ja, pb, a1);                  // M[16 ti + r][16 tj + c]
                    a2 = F::mfma(pa, jb, a2);                  // M[16 tj + c][16 ti + r]
                }
                const double wgt = ti == tj ? 1.0 : 2.0;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) t3 = fma(wgt * a1[r4], a2[r4], t3);
            }
        }
        red[0] = p_re; red[1] = p_im; red[2] = t2; red[3] = t3;
        CG_STAMP_END(19)
    }

    static __device__ __forceinline__ void grad_laplacian(const CgBlk& b, const double* __restrict__ th, const double* __restrict__ xg,
                                                          const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                                                          int mode, const double* __restrict__ v, double* __restrict__ grad /*N x 2*/,
                                                          double* __restrict__ lap /*2*/, double* lds, double* ws, const LayG& l,
                                                          double* stash = nullptr, const Stash* st = nullptr) {
        const CgPl pl{lds, ws};
        const LayC& c = l.c;
        typename F::WFrag wf;
        const bool exact_phi = mode != 1;
        CG_STAMP_START(20)
        setup(b, th, xg, spk, sidx, n, L, pl, c, wf);
        auto ev = [](int v) { return (v + 1) & ~1; };
        if (stash) {        // (fused kernel) the primal temporaries of the score row, before the arena is reused
            const CgFastLds& o = c.o;
            copy2(b, stash + st->s1, lds + o.s1, n * HS); copy2(b, stash + st->s2, lds + o.s2, n * HS);
            copy2(b, stash + st->m1, lds + o.m1, n * HT); copy2(b, stash + st->m0, lds + o.m0, ev(n * P));
            b.sync();
        }
        setup2(b, th, n, L, pl, c, wf, pl(l.Uk));
        if (stash) {        // ... and the rest of what the score passes read
            const CgFastLds& o = c.o; const int N = n * D;
            copy2(b, stash + st->sh, lds + o.sh, ev(N)); copy2(b, stash + st->ch, lds + o.ch, ev(N));
            copy2(b, stash + st->sg1, lds + o.sg1, n * HS); copy2(b, stash + st->sg2, lds + o.sg2, n * HS);
            copy2(b, stash + st->gbar, lds + o.gbar, HS);
            copy2(b, stash + st->V, lds + o.V, ev(n * F::SPV)); copy2(b, stash + st->Bm, lds + o.Bm, ev(n * F::SPB)); copy2(b, stash + st->G, lds + o.G, ev(n * F::SPG));
            copy2(b, stash + st->zb, lds + c.zb, 2 * N);
            copy2(b, stash + st->Uk, pl(l.Uk), N * HS);
        }
        CG_STAMP_START(18)
        ta_gemm(b, n, pl(c.Dm), pl(c.Dinv), lds + c.kocc, pl(l.Ta));
        b.sync();
        CG_STAMP_END(18)
        CG_STAMP_END(20)
        double s_re, s_im, q_re = 0, q_im = 0;
        CG_STAMP_START(21)
        slater_part(b, n, pl(c.J), lds + c.zb, pl(l.Ta), lds + c.Kd, exact_phi, grad, s_re, s_im);      // grad <- J^T g
        b.sync();
        CG_STAMP_END(21)
        CG_STAMP_START(23)
        if (exact_phi) forward_laplacian(b, th, n, L, pl, l, q_re, q_im);
        double tot[4] = {s_re + q_re, s_im + q_im, 0.0, 0.0};
        b.sync();
        CG_STAMP_END(23)
        CG_STAMP_START(22)
        reverse_x(b, th, n, L, pl, l, grad);
        CG_STAMP_END(22)
        CG_STAMP_START(24)
        {
            double r[4];
            jet_part(b, th, n, L, pl, l, v, mode == 1, r);
            tot[0] += r[0]; tot[1] += r[1]; tot[2] += r[2]; tot[3] += r[3];
        }
        CG_STAMP_END(24)
        cg_block_sum_n<4>(b, tot, lds + l.red);
        if (b.tid == 0) { lap[0] = tot[0] + 0.5 * (tot[2] - tot[3]); lap[1] = tot[1]; }
    }
#endif      // __HIP_DEVICE_COMPILE__
};
#endif
