// cg_flow_fast.hpp -- depth-2 FermiNet flow: value z(x), structured Jacobian J = dz/dx, log Psi.
//
// Reference semantics: src/flow.py:16-55 (FermiNet), src/logpsi.py:7-33 (logpsi),
// src/slater.py:4-19 (logslaterdet).  The reference obtains J with jax.jacfwd (n*d dense
// tangents through the (n,n,tpsize) pair tensor, src/logpsi.py:27-28).  Here J is assembled
// from its block structure with every d x d block contracted from the
// *left* first, which removes all h x h work per pair:
//
//   J_ik = -U'_i T_ik - V_i diag(sig_t(u_ik)) Wt^T T_ik + B_i G_k         (k != i)
//   J_ii = I - sum_{k != i} J_ik                                          (translation equivariance)
//
//   T_ik  = d t0_ik / d r_ik                     (p x d, 3 non-zeros per column)
//   R_i   = Wf^T diag(sig(u2_i))                 (d x hs)
//   U'_i  = (1/n) (Wf^T + R_i Wa^T) diag(sig(u1_i)) W0^T      (d x p)
//   V_i   = (1/n) R_i Wc^T                       (d x ht)
//   B_i   = R_i Wb^T                             (d x hs)
//   G_k   = d gbar / d x_k = (1/n^2) sum_{l != k} [ sig1_k (W0^T T_kl) - sig1_l (W0^T T_lk) ]   (hs x d)
//
// Work-item layouts (deterministic: no atomics, every sum has a fixed order):
//   pair-primal  : item (i,h), serial over j      -> m1_i[h] = mean_j softplus(u_ij[h]),  m0_i
//   G pass       : item (k,h), serial over l
//   Jacobian pass: item (i,k), serial over h      -> d x d block, written once
//
// theta is the flat ravel_pytree vector of the Haiku tree:
//   [final.b (d), final.w (hs,d), sp0.b (hs), sp0.w (4d+1,hs), sp1.b (hs), sp1.w (2hs+ht,hs), tp0.b (ht), tp0.w (2d+1,ht)]
#pragma once
#include "cg_common.hpp"
#include "cg_linalg.hpp"
#include "cg_jet.hpp"

// Offsets (in doubles) into the per-walker LDS arena; filled by the host (cg_layout.hpp).
struct CgFastLds {
    int sh, ch, m0, s1, sg1, m1, gbar, cb, sg2, s2, z, U, V, Bm, Up, G, J, Dm, perm, wt, lus, total;
    int wave_lu;      // 1: both determinants by the wave-level register LU (N <= 32, n <= 16, Dm not on J)
    int dual;         // 1: (large n, sampler layout) the Slater matrix has LDS of its own while J is factored: both LUs run concurrently
};

template <int D, int HS, int HT>
struct CgFast {
    static constexpr int P = 2 * D + 1;
    static constexpr int o_fb = 0;
    static constexpr int o_fw = o_fb + D;
    static constexpr int o_s0b = o_fw + HS * D;
    static constexpr int o_s0w = o_s0b + HS;
    static constexpr int o_s1b = o_s0w + (4 * D + 1) * HS;
    static constexpr int o_s1w = o_s1b + HS;
    static constexpr int o_t0b = o_s1w + (2 * HS + HT) * HS;
    static constexpr int o_t0w = o_t0b + HT;
    static constexpr int NPARAM = o_t0w + P * HT;
    static constexpr int o_W0 = o_s0w + 2 * D * HS;        // rows of sp0.w that see mean_j t0_ij
    static constexpr int o_Wa = o_s1w;
    static constexpr int o_Wb = o_s1w + HS * HS;
    static constexpr int o_Wc = o_s1w + 2 * HS * HS;

    // per-particle strides of V, Bm, G padded by 2 doubles: the Jacobian pass reads them with the particle index
    // varying across lanes, and an unpadded 32-double stride maps every particle to the same LDS banks
    static constexpr int SPV = HT * D + 2, SPB = HS * D + 2, SPG = HS * D + 2;
    static CG_HD int iV(int i, int a, int h) { return i * SPV + h * D + a; }
    static CG_HD int iB(int i, int a, int g) { return i * SPB + g * D + a; }
    static CG_HD int iG(int k, int g, int bb) { return k * SPG + g * D + bb; }

    template <class T> struct PairFT { T s2[D], c2[D], del; };
    using PairF = PairFT<double>;

    // features of r_ij = x_i - x_j from per-particle half-angle tables:
    //   sin(pi r/L) = sh_i ch_j - ch_i sh_j ; cos(2 pi r/L) = 1 - 2 sin^2 ; sin(2 pi r/L) = 2 sin cos
    template <class T>
    static CG_DEVI void pairfeat(const T* sh, const T* ch, int i, int j, PairFT<T>& f) {
        T d2 = T(0.0);
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const T si = sh[i * D + a], ci = ch[i * D + a], sj = sh[j * D + a], cj = ch[j * D + a];
            const T s = si * cj - ci * sj, c = ci * cj + si * sj;
            f.s2[a] = 2.0 * (s * c); f.c2[a] = 1.0 - 2.0 * (s * s); d2 += s * s;
        }
        if (i == j) {       // exact diagonal feature [1..1, 0..0, 0]  (src/flow.py:25: "* (1 - eye)")
#pragma unroll
            for (int a = 0; a < D; ++a) { f.s2[a] = T(0.0); f.c2[a] = T(1.0); }
            f.del = T(0.0);
        } else {
            f.del = cg_sqrt(d2);
        }
    }

    // Value-only features of a pair whose coordinates carry no tangent.  In a directional pass along e_(p,a) only the
    // 2n - 1 pairs that involve particle p have non-zero derivatives; for all others the features, the two-particle
    // pre-activations and their softplus are plain doubles (exactly: the skipped jet parts are zeros).
    template <class T>
    static CG_DEVI void pairfeat_val(const T* sh, const T* ch, int i, int j, PairFT<double>& f) {
        double d2 = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const double si = cg_val(sh[i * D + a]), ci = cg_val(ch[i * D + a]), sj = cg_val(sh[j * D + a]), cj = cg_val(ch[j * D + a]);
            const double s = si * cj - ci * sj, c = ci * cj + si * sj;
            f.s2[a] = 2.0 * (s * c); f.c2[a] = 1.0 - 2.0 * (s * s); d2 += s * s;
        }
        if (i == j) {
#pragma unroll
            for (int a = 0; a < D; ++a) { f.s2[a] = 0.0; f.c2[a] = 1.0; }
            f.del = 0.0;
        } else {
            f.del = sqrt(d2);
        }
    }

    // ---------------------------------------------------------------------------------------
    // Dense layers on the matrix cores (gfx950, spsize = tpsize = 16, T = double only).
    // The per-particle layers are tiny GEMMs against the flow weights: (n x 5)(5 x 16), (n x 48)(48 x 16),
    // (n x 16)(16 x d), (n d x 16)(16 x 16) x 3, (n d x 16)(16 x 5).  As scalar loops each lane chases its own
    // weight row through L1 (measured: ~45 % of a logp evaluation).  With v_mfma_f64_16x16x4_f64 the weight
    // operand of every k-step is ONE double per lane, loaded once per kernel (WFrag) and kept in VGPRs for the
    // whole Metropolis chain; activations are read from LDS in the A-operand layout.
    //   A: lane l holds A[row = l & 15][k = l >> 4];  B: B[k = l >> 4][col = l & 15];
    //   C/D: 4 doubles per lane, D[row = (l >> 4) + 4 r][col = l & 15]   (f64 layout, not the f32 one)
    // ---------------------------------------------------------------------------------------
    // Per-lane weight columns kept in VGPRs for the whole chain (the pair passes use them in every iteration).
    // The MFMA B-operand fragments are (re)loaded from theta right before each dense phase instead: they are the
    // same 8.6 KB for every wave (L2/L1 hits), whereas keeping them live across the LUs made the compiler spill
    // them to per-wave scratch (measured: 6 GB of memory-side traffic per launch).
    struct WFrag {
        const double* th;
        double tw[P + 1];  // two-particle layer column h = lane & 15: bias, then P weights  (pair-primal pass)
        bool inline_libm;  // true: keep sincos inline (derivative kernels: see CG_OUTLINE in cg_common.hpp)
        bool wt_resident;  // true: the caller staged the two-particle weights into o.wt once (stage_wt) and nothing overwrites that
                           // slot between evaluations (Metropolis chain at small n); false: jacobian_mfma stages them itself
        double* Jext;      // not null (large-n derivative kernels, cg_big.hpp): J is assembled there (any memory) instead of at lds + o.J,
                           // which then only serves as the pair-primal scratch
    };
    // The MFMA / DPP code paths of primal() / jacobian() take their weights through a WFrag; a null pointer selects the scalar statement
    // of the same arithmetic (host builds of these headers; widths other than 16 / 16).  One place decides: frags().
    static CG_DEVI void load_frags(const double* __restrict__ th, WFrag& w, bool inline_libm = false) { w.th = th; w.inline_libm = inline_libm; w.wt_resident = false; w.Jext = nullptr; }
    static CG_DEVI const WFrag* frags(const double* __restrict__ th, WFrag& w, bool inline_libm = true) {
        if (!(CG_ON_DEVICE && HS == 16 && HT == 16)) return nullptr;
        load_frags(th, w, inline_libm);
        return &w;
    }
    struct DenseP { double w0[2], b0, b2, wacb[12], wf[4], bf; };     // primal dense layers
    struct DenseJ { double ja[4], jb[4], jc[4]; };                     // R_i W_x^T
    struct DenseU { double w0t[4]; };
#if defined(__HIP_DEVICE_COMPILE__)
    typedef double d4_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ d4_t mfma(double a, double bb, d4_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void load_pair_cols(const double* __restrict__ th_in, WFrag& w) {
        const double* th = th_in;
        asm volatile("" : "+s"(th));      // opaque to LICM (see load_dense_p)
        const int col = threadIdx.x & 15;
        w.th = th_in;
        w.tw[0] = th[o_t0b + col];
#pragma unroll
        for (int f = 0; f < P; ++f) w.tw[1 + f] = th[o_t0w + f * HT + col];
    }
    static __device__ __forceinline__ void load_dense_p(const double* __restrict__ th_in, DenseP& w) {
        const double* th = th_in;
        asm volatile("" : "+s"(th));      // opaque to LICM: keep these loads inside the evaluation, not hoisted + spilled
        const int l = threadIdx.x & 63, col = l & 15, kq = l >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { const int f = 4 * ks + kq; w.w0[ks] = f < P ? th[o_W0 + f * HS + col] : 0.0; }
        w.b0 = th[o_s0b + col]; w.b2 = th[o_s1b + col];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + kq;
            w.wacb[ks] = th[o_Wa + k * HS + col];
            w.wacb[4 + ks] = th[o_Wc + k * HS + col];
            w.wacb[8 + ks] = th[o_Wb + k * HS + col];
            w.wf[ks] = col < D ? th[o_fw + k * D + col] : 0.0;
        }
        w.bf = col < D ? th[o_fb + col] : 0.0;
    }
    static __device__ __forceinline__ void load_dense_j(const double* __restrict__ th_in, DenseJ& w) {
        const double* th = th_in;
        asm volatile("" : "+s"(th));      // opaque to LICM: keep these loads inside the evaluation, not hoisted + spilled
        const int l = threadIdx.x & 63, col = l & 15, kq = l >> 4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + kq;
            w.ja[ks] = th[o_Wa + col * HS + k];            // B[k = h][col = g] = Wa[g][h]
            w.jb[ks] = th[o_Wb + col * HS + k];
            w.jc[ks] = th[o_Wc + col * HS + k];
        }
    }
    static __device__ __forceinline__ void load_dense_u(const double* __restrict__ th_in, DenseU& w) {
        const double* th = th_in;
        asm volatile("" : "+s"(th));      // opaque to LICM: keep these loads inside the evaluation, not hoisted + spilled
        const int l = threadIdx.x & 63, col = l & 15, kq = l >> 4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w.w0t[ks] = col < P ? th[o_W0 + col * HS + 4 * ks + kq] : 0.0;   // B[k = g][col = f] = W0[f][g]
    }
    // broadcast lane LANE of every 16-lane DPP row to the whole row (row_newbcast, gfx90a+)
    template <int LANE>
    static __device__ __forceinline__ double row_bcast(double v) {
        const long long u = __double_as_longlong(v);
        // (bound_ctrl set: every lane has a source, and the destination needs no zero-initialising move -- two instructions fewer per broadcast)
        const int lo = __builtin_amdgcn_update_dpp(0, (int)u, 0x150 + LANE, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), 0x150 + LANE, 0xf, 0xf, true);
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    }
    struct PF6 { double c2[D], s2[D], del, rdel; };
    template <int LANE>
    static __device__ __forceinline__ void pf_bcast(const PF6& mine, PF6& out) {
#pragma unroll
        for (int a = 0; a < D; ++a) { out.c2[a] = row_bcast<LANE>(mine.c2[a]); out.s2[a] = row_bcast<LANE>(mine.s2[a]); }
        out.del = row_bcast<LANE>(mine.del); out.rdel = row_bcast<LANE>(mine.rdel);
    }
    // Pair-primal pass with shared pair features: a DPP row (16 lanes) owns particle i, lane h = hidden unit.
    // Each lane computes the features of ONE pair (i, j = 16 c + h); the row then walks j = 16 c + jj and every
    // lane receives pair (i, j) by row broadcast -- 12 v_mov_dpp instead of recomputing ~45 instructions.
    template <int JJ>
    static __device__ __forceinline__ void primal_pair_step(const PF6& mine, const WFrag& w, int i, int jbase, int n, int h,
                                                            double& acc, double& raw) {
        if constexpr (JJ < 16) {
            if (jbase + JJ < n) {                                  // wave-uniform
                PF6 pf; pf_bcast<JJ>(mine, pf);                    // (the owner lane already holds the exact diagonal feature)
                double u = w.tw[0] + w.tw[1 + 2 * D] * pf.del;
#pragma unroll
                for (int a = 0; a < D; ++a) u += w.tw[1 + a] * pf.c2[a] + w.tw[1 + D + a] * pf.s2[a];
                acc += softplus_only(u);
                double fv = pf.del;
#pragma unroll
                for (int a = 0; a < D; ++a) { if (h == a) fv = pf.c2[a]; if (h == D + a) fv = pf.s2[a]; }
                raw += fv;
                primal_pair_step<JJ + 1>(mine, w, i, jbase, n, h, acc, raw);
            }
        }
    }
    static __device__ __forceinline__ void own_pair(const double* sh, const double* ch, int i, int j, bool ok, PF6& f) {
        double d2 = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const double si = ok ? sh[i * D + a] : 0.0, ci = ok ? ch[i * D + a] : 1.0, sj = ok ? sh[j * D + a] : 1.0, cj = ok ? ch[j * D + a] : 0.0;
            const double s = si * cj - ci * sj, c = ci * cj + si * sj;
            f.s2[a] = 2.0 * (s * c); f.c2[a] = 1.0 - 2.0 * (s * s); d2 += s * s;
        }
        double rd;
        cg_fast_sqrt_rsqrt(d2, f.del, rd);
        f.rdel = (ok && i != j) ? rd : 0.0;
        if (i == j) {                                          // exact diagonal feature [1..1, 0..0, 0] (src/flow.py:25)
#pragma unroll
            for (int a = 0; a < D; ++a) { f.c2[a] = 1.0; f.s2[a] = 0.0; }
            f.del = 0.0;
        }
    }
    static __device__ __forceinline__ void primal_pairs_dpp(const CgBlk& b, const WFrag& wfr, int n, double* lds, const CgFastLds& o) {
        WFrag w; load_pair_cols(wfr.th, w);
        const double *sh = lds + o.sh, *ch = lds + o.ch;
        double *m0 = lds + o.m0, *m1 = lds + o.m1;
        const double rn = 1.0 / (double)n;
        const int h = b.tid & 15;
        for (int e0 = (b.tid >> 6) << 6; e0 < n * 16; e0 += b.nthr) {      // whole waves stay together
            const int i = (e0 + (b.tid & 63)) >> 4;
            const bool rowok = i < n;
            double acc = 0.0, raw = 0.0;
            for (int jb = 0; jb < n; jb += 16) {
                PF6 mine; own_pair(sh, ch, rowok ? i : 0, jb + h, rowok && jb + h < n, mine);
                primal_pair_step<0>(mine, w, i, jb, n, h, acc, raw);
            }
            if (rowok) { m1[i * HT + h] = acc * rn; if (h < P) m0[i * P + h] = raw * rn; }
        }
    }
    // Pair-primal pass, features handed off through LDS.  Same work split as primal_pairs_dpp (a 16-lane row owns particle
    // i, lane h = hidden unit; each lane computes the features of ONE pair (i, 16 c + h)), but the row then walks j by
    // reading pair (i, j) back from an LDS scratch (broadcast reads: no VALU issue slots) instead of 10 v_mov_dpp per
    // step.  The scratch is J's slot, dead until the Jacobian assembly.  The means of the cos / sin features are taken
    // from their closed form  sum_j cos(t_i - t_j) = cos t_i C + sin t_i S,  sum_j sin(t_i - t_j) = sin t_i C - cos t_i S
    // (t = 2 pi x / L, C = sum_j cos t_j, S = sum_j sin t_j; the diagonal feature [1, 0] is the j = i term), so only the
    // norm feature is accumulated in the pair loop.
    static constexpr int PFS = 2 * D + 2;                  // doubles per pair in the scratch: c2[D], s2[D], del, pad
    static constexpr int PFROW = 16 * PFS + 2;             // row stride (padded against bank conflicts between the 4 rows)
    static constexpr int PFWAVE = 4 * PFROW + 2 * D + 2;   // per wave: 4 rows + C[D], S[D]
    static __device__ __forceinline__ bool primal_pairs_lds_fits(const CgBlk& b, int n) {
        return (b.nthr >> 6) * PFWAVE <= n * D * n * D;
    }
    static __device__ __forceinline__ void primal_pairs_lds(const CgBlk& b, const WFrag& wfr, int n, double* lds, const CgFastLds& o) {
        WFrag w; load_pair_cols(wfr.th, w);
        const double *sh = lds + o.sh, *ch = lds + o.ch;
        double *m0 = lds + o.m0, *m1 = lds + o.m1;
        const double rn = 1.0 / (double)n;
        const int lane = b.tid & 63, h = lane & 15, rr = lane >> 4;
        double* scr = lds + o.J + (b.tid >> 6) * PFWAVE;   // this wave's scratch
        double* cs = scr + 4 * PFROW;                      // C[a] at cs[a], S[a] at cs[D + a]
        if (lane < 2 * D) {
            const int a = lane < D ? lane : lane - D;
            double acc = 0.0;
            for (int j = 0; j < n; ++j) {
                const double sj = sh[j * D + a], cj = ch[j * D + a];
                acc += lane < D ? 1.0 - 2.0 * (sj * sj) : 2.0 * (sj * cj);
            }
            cs[lane] = acc;
        }
        for (int e0 = (b.tid >> 6) << 6; e0 < n * 16; e0 += b.nthr) {      // whole waves stay together
            const int i = (e0 + lane) >> 4;
            const bool rowok = i < n;
            double acc = 0.0, rawd = 0.0;
            for (int jb = 0; jb < n; jb += 16) {
                PF6 mine; own_pair(sh, ch, rowok ? i : 0, jb + h, rowok && jb + h < n, mine);
                double* slot = scr + rr * PFROW + h * PFS;
#pragma unroll
                for (int a = 0; a < D; ++a) { slot[a] = mine.c2[a]; slot[D + a] = mine.s2[a]; }
                slot[2 * D] = mine.del;
                asm volatile("" ::: "memory");             // cross-lane hand-off (LDS executes one wave's accesses in order)
                const int jn = n - jb < 16 ? n - jb : 16;
                const double* row = scr + rr * PFROW;
                // two independent softplus chains per trip: each has three LDS round trips (features, exp table, log
                // table) on its critical path, and one wave alone cannot hide them behind a single chain
                auto uof = [&](const double* pf) {
                    double u = w.tw[0] + w.tw[1 + 2 * D] * pf[2 * D];
#pragma unroll
                    for (int a = 0; a < D; ++a) u += w.tw[1 + a] * pf[a] + w.tw[1 + D + a] * pf[D + a];
                    return u;
                };
                // sum_j softplus(u_j) = sum_j max(u_j, 0) + log prod_j (1 + e^-|u_j|): the factors lie in [1, 2], so the
                // product of a block of <= 16 cannot overflow and ONE logarithm per block replaces one per pair (the
                // logarithm is ~40 % of a softplus); two independent exp chains per trip hide the LDS round trips
                double pra = 1.0, prb = 1.0;
                int jj = 0;
                for (; jj + 1 < jn; jj += 2) {
                    const double* pa = row + jj * PFS; const double* pb = pa + PFS;
                    const double ua = uof(pa), ub = uof(pb);
                    rawd += pa[2 * D] + pb[2 * D];
                    acc += fmax(ua, 0.0) + fmax(ub, 0.0);
                    pra *= 1.0 + cg_exp_nonpos(-fabs(ua)); prb *= 1.0 + cg_exp_nonpos(-fabs(ub));
                }
                if (jj < jn) {
                    const double* pa = row + jj * PFS;
                    const double ua = uof(pa);
                    rawd += pa[2 * D];
                    acc += fmax(ua, 0.0);
                    pra *= 1.0 + cg_exp_nonpos(-fabs(ua));
                }
                acc += cg_log_ge1(pra * prb);
                asm volatile("" ::: "memory");             // the next block's stores stay behind these reads
            }
            if (rowok) {
                m1[i * HT + h] = acc * rn;
                if (h < 2 * D) {
                    const int a = h < D ? h : h - D;
                    const double si = sh[i * D + a], ci = ch[i * D + a];
                    const double ct = 1.0 - 2.0 * (si * si), st = 2.0 * (si * ci);
                    m0[i * P + h] = (h < D ? ct * cs[a] + st * cs[D + a] : st * cs[a] - ct * cs[D + a]) * rn;
                } else if (h == 2 * D) {
                    m0[i * P + h] = rawd * rn;
                }
            }
        }
        asm volatile("" ::: "memory");
    }
    // G pass on the matrix cores.  With the pair-feature matrices (zero diagonal, b = direction)
    //     C_b[k][l] = cos(2 pi r_kl,b / L),   S_b[k][l] = sin(2 pi r_kl,b / L),   R_b[k][l] = S_b[k][l] / |sin(pi r_kl / L)|
    // the sum over l of  sg1_k (odd + evn) - sg1_l (evn - odd)  (see the scalar G pass in jacobian()) becomes
    //     n^2 G_k[h][b] =   c1 W0s[b][h] ( sg1_k[h] rowsum C_b[k] - (C_b sg1)[k][h] )
    //                     - c1 W0c[b][h] ( sg1_k[h] rowsum S_b[k] + (S_b sg1)[k][h] )
    //                     + c2c W0d[h]   ( sg1_k[h] rowsum R_b[k] + (R_b sg1)[k][h] )
    // i.e. 3 D products (n x n)(n x 16) plus their row sums (the same A operand against a ones B operand).
    // A operand: lane (row k = l & 15, l' = 4 ks + (l >> 4)) computes the features of ITS pair once -- no 16-fold
    // redundancy and no DPP broadcast; the pass costs ~4 pair-feature evaluations per lane and tile instead of
    // n/4 x 13 x ~30 VALU instructions.
    static __device__ __forceinline__ void g_pass_mfma(const CgBlk& b, const WFrag& wfr, int n, double L, double* lds, const CgFastLds& o) {
        const double* th = wfr.th;
        asm volatile("" : "+s"(th));      // opaque to LICM (see load_dense_p)
        const double *sh = lds + o.sh, *ch = lds + o.ch, *sg1 = lds + o.sg1;
        double* G = lds + o.G;
        const int l = b.tid & 63, col = l & 15, kq = l >> 4;
        const int wave = b.tid >> 6, nw = b.nthr >> 6;
        const int tiles = (n + 15) >> 4;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        double kc[D], ksn[D];                                         // -c1 W0c[b][h], c1 W0s[b][h] of column h = col
#pragma unroll
        for (int a = 0; a < D; ++a) { kc[a] = -c1 * th[o_W0 + a * HS + col]; ksn[a] = c1 * th[o_W0 + (D + a) * HS + col]; }
        const double kd = c2c * th[o_W0 + 2 * D * HS + col];
        for (int kt = wave; kt < tiles; kt += nw) {
            const int k = 16 * kt + col;                              // A row of this lane
            d4_t pC[D], pS[D], pR[D], sC[D], sS[D], sR[D];            // products with sg1 / row sums
#pragma unroll
            for (int a = 0; a < D; ++a) {
                pC[a] = d4_t{0, 0, 0, 0}; pS[a] = pC[a]; pR[a] = pC[a]; sC[a] = pC[a]; sS[a] = pC[a]; sR[a] = pC[a];
            }
            for (int lt = 0; lt < tiles; ++lt) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int lp = 16 * lt + 4 * ks + kq;             // k-dimension index of this lane
                    const bool inr = lp < n;
                    PF6 pf; own_pair(sh, ch, k, lp, k < n && inr, pf);
                    const double bS = inr ? sg1[lp * HS + col] : 0.0; // B[l'][h]; rows l' >= n contribute nothing
                    const double bO = inr ? 1.0 : 0.0;
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const double xC = (lp == k) ? 0.0 : pf.c2[a];  // own_pair's exact diagonal feature is [1, 0, 0]
                        const double xS = pf.s2[a], xR = pf.s2[a] * pf.rdel;
                        pC[a] = mfma(xC, bS, pC[a]); sC[a] = mfma(xC, bO, sC[a]);
                        pS[a] = mfma(xS, bS, pS[a]); sS[a] = mfma(xS, bO, sS[a]);
                        pR[a] = mfma(xR, bS, pR[a]); sR[a] = mfma(xR, bO, sR[a]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * kt + kq + 4 * r;                  // C/D row of this lane
                if (kk < n) {
                    const double sgk = sg1[kk * HS + col];
#pragma unroll
                    for (int a = 0; a < D; ++a) {
                        const double v = ksn[a] * fma(sgk, sC[a][r], -pC[a][r]) + kc[a] * fma(sgk, sS[a][r], pS[a][r])
                                         + kd * fma(sgk, sR[a][r], pR[a][r]);
                        G[iG(kk, col, a)] = v * rn * rn;
                    }
                }
            }
        }
    }
    // dense part of primal(): needs m0, m1 in LDS; fills s1 sg1 sg2 s2 z.  Executed by wave 0; others wait.
    static __device__ __forceinline__ void primal_dense_mfma(const CgBlk& b, const WFrag& wfr, const double* x, int n,
                                                             double* lds, const CgFastLds& o) {
        DenseP w; load_dense_p(wfr.th, w);
        double *m0 = lds + o.m0, *s1 = lds + o.s1, *sg1 = lds + o.sg1, *m1 = lds + o.m1, *gbar = lds + o.gbar,
               *sg2 = lds + o.sg2, *s2 = lds + o.s2, *z = lds + o.z;
        const int l = b.tid & 63, col = l & 15, kq = l >> 4;
        const int wave = b.tid >> 6, nw = b.nthr >> 6;
        const int tiles = (n + 15) >> 4;
        const double rn = 1.0 / (double)n;
        // layer 0: u1 = m0 W0 + b0
        for (int t = wave; t < tiles; t += nw) {
            const int ia = 16 * t + col;                     // A row of this lane
            d4_t c = {w.b0, w.b0, w.b0, w.b0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int f = 4 * ks + kq;
                const double a = (ia < n && f < P) ? m0[ia * P + f] : 0.0;
                c = mfma(a, w.w0[ks], c);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * t + kq + 4 * r;
                if (i < n) { double sp, sg; softplus_sigmoid(c[r], sp, sg); s1[i * HS + col] = sp; sg1[i * HS + col] = sg; }
            }
        }
        b.sync();
        if (b.tid < HS) {                                     // gbar = mean_i s1_i
            double a = 0.0;
            for (int i = 0; i < n; ++i) a += s1[i * HS + b.tid];
            gbar[b.tid] = a * rn;
        }
        b.sync();
        // last layer: u2 = s1 Wa + m1 Wc + gbar Wb + b2;  s2 = s1 + softplus(u2)
        for (int t = wave; t < tiles; t += nw) {
            const int ia = 16 * t + col;
            d4_t c = {w.b2, w.b2, w.b2, w.b2};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 4 * ks + kq;
                const double a1 = ia < n ? s1[ia * HS + k] : 0.0;
                const double a2 = ia < n ? m1[ia * HT + k] : 0.0;
                const double a3 = ia < n ? gbar[k] : 0.0;
                c = mfma(a1, w.wacb[ks], c);
                c = mfma(a2, w.wacb[4 + ks], c);
                c = mfma(a3, w.wacb[8 + ks], c);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * t + kq + 4 * r;
                if (i < n) { double sp, sg; softplus_sigmoid(c[r], sp, sg); sg2[i * HS + col] = sg; s2[i * HS + col] = s1[i * HS + col] + sp; }
            }
        }
        b.sync();
        // z = x + s2 Wf + bf
        for (int t = wave; t < tiles; t += nw) {
            const int ia = 16 * t + col;
            d4_t c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * t + kq + 4 * r;
                c[r] = (i < n && col < D) ? x[i * D + col] + w.bf : 0.0;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double a = ia < n ? s2[ia * HS + 4 * ks + kq] : 0.0;
                c = mfma(a, w.wf[ks], c);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * t + kq + 4 * r;
                if (i < n && col < D) z[i * D + col] = c[r];
            }
        }
        b.sync();
    }
    // U, Bm (which = 0) or V (which = 1) of jacobian(): rows r = (i,a).  wfl: Wf (HS x D) staged in LDS.
    template <int WHICH>
    static __device__ __forceinline__ void jac_factors_mfma(const CgBlk& b, const WFrag& wfr, int n, double* lds,
                                                            const CgFastLds& o, const double* wfl) {
        DenseJ w; load_dense_j(wfr.th, w);
        const double* sg2 = lds + o.sg2;
        double *U = lds + o.U, *V = lds + o.V, *Bm = lds + o.Bm;
        const int l = b.tid & 63, col = l & 15, kq = l >> 4;
        const int wave = b.tid >> 6, nw = b.nthr >> 6;
        const int N = n * D, tiles = (N + 15) >> 4;
        const double rn = 1.0 / (double)n;
        for (int t = wave; t < tiles; t += nw) {
            const int ra = 16 * t + col, ia = ra / D, aa = ra - ia * D;
            d4_t cu = {0, 0, 0, 0}, cbm = {0, 0, 0, 0};
            if (WHICH == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * t + kq + 4 * r;
                    cu[r] = rr < N ? wfl[col * D + (rr % D)] : 0.0;      // direct term Wf[g][a]
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int h = 4 * ks + kq;
                const double a = ra < N ? wfl[h * D + aa] * sg2[ia * HS + h] : 0.0;   // R_i[a][h]
                if (WHICH == 0) { cu = mfma(a, w.ja[ks], cu); cbm = mfma(a, w.jb[ks], cbm); }
                else cu = mfma(a, w.jc[ks], cu);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * t + kq + 4 * r;
                if (rr < N) {
                    const int ii = rr / D, ar = rr - ii * D;
                    if (WHICH == 0) { U[rr * HS + col] = cu[r]; Bm[iB(ii, ar, col)] = cbm[r]; }
                    else V[iV(ii, ar, col)] = cu[r] * rn;
                }
            }
        }
    }
    // J[r][c] = sum_g Bm[r][g] G[g][c]  (r = (i,a), c = (k,b)): the rank-16 term B_i G_k of every block at once
    static __device__ __forceinline__ void jac_bg_mfma(const CgBlk& b, int n, double* lds, const CgFastLds& o, double* Jext = nullptr) {
        const double *Bm = lds + o.Bm, *G = lds + o.G;
        double* J = Jext ? Jext : lds + o.J;
        const int l = b.tid & 63, col = l & 15, kq = l >> 4;
        const int wave = b.tid >> 6, nw = b.nthr >> 6;
        const int N = n * D, tiles = (N + 15) >> 4;
        for (int tt = wave; tt < tiles * tiles; tt += nw) {
            const int t = tt / tiles, u = tt - t * tiles;
            const int ra = 16 * t + col, ia = ra / D, aa = ra - ia * D;       // A row
            const int cb = 16 * u + col, kb = cb / D, bb = cb - kb * D;       // B column
            d4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int g = 4 * ks + kq;
                const double av = ra < N ? Bm[iB(ia, aa, g)] : 0.0;
                const double bv = cb < N ? G[iG(kb, g, bb)] : 0.0;
                c = mfma(av, bv, c);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * t + kq + 4 * r;
                if (rr < N && cb < N) J[rr * N + cb] = c[r];
            }
        }
    }
    // Up = (1/n) (U diag sg1) W0^T
    static __device__ __forceinline__ void jac_up_mfma(const CgBlk& b, const WFrag& wfr, int n, double* lds, const CgFastLds& o) {
        DenseU w; load_dense_u(wfr.th, w);
        const double *sg1 = lds + o.sg1, *U = lds + o.U;
        double* Up = lds + o.Up;
        const int l = b.tid & 63, col = l & 15, kq = l >> 4;
        const int wave = b.tid >> 6, nw = b.nthr >> 6;
        const int N = n * D, tiles = (N + 15) >> 4;
        const double rn = 1.0 / (double)n;
        for (int t = wave; t < tiles; t += nw) {
            const int ra = 16 * t + col, ia = ra / D;
            d4_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int g = 4 * ks + kq;
                const double a = ra < N ? U[ra * HS + g] * sg1[ia * HS + g] : 0.0;
                c = mfma(a, w.w0t[ks], c);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 16 * t + kq + 4 * r;
                if (rr < N && col < P) Up[rr * P + col] = c[r] * rn;
            }
        }
    }
    // two-particle layer weights [h][bias, P weights] and Wf (HS x D) from the parameter vector into the LDS slot o.wt
    static __device__ __forceinline__ void stage_wt(const CgBlk& b, const double* __restrict__ th, double* lds, const CgFastLds& o) {
        double* wt = lds + o.wt;
        double* wfl = wt + HT * (P + 1);
        for (int e = b.tid; e < HT * (P + 1); e += b.nthr) {
            const int h = e / (P + 1), f = e - h * (P + 1);
            wt[e] = f == 0 ? th[o_t0b + h] : th[o_t0w + (f - 1) * HT + h];
        }
        for (int e = b.tid; e < HS * D; e += b.nthr) wfl[e] = th[o_fw + e];
        b.sync();
    }
    // The whole Jacobian assembly on the MFMA / DPP path (double, spsize = tpsize = 16).  Order chosen so that
    // V can reuse Bm's LDS slot:  U,Bm | G  ->  Up  ->  J = Bm G  ->  V  ->  J += pair part  ->  diagonal blocks.
    static __device__ __forceinline__ void jacobian_mfma(const CgBlk& b, const double* __restrict__ th, const WFrag& w, int n,
                                                         double L, double* lds, const CgFastLds& o) {
        const double *sh = lds + o.sh, *ch = lds + o.ch;
        double *V = lds + o.V, *Up = lds + o.Up, *J = w.Jext ? w.Jext : lds + o.J;
        const int N = n * D;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        double* wt = lds + o.wt;
        double* wfl = wt + HT * (P + 1);
        if (!w.wt_resident) stage_wt(b, th, lds, o);
        CG_STAMP(4)
#if !defined(CG_EXP_NO_DENSE)       /* timing experiment (garbage numbers): the MFMA dense phases free -- the bound on pipelining them */
        jac_factors_mfma<0>(b, w, n, lds, o, wfl);
#endif
        CG_STAMP(5)
        g_pass_mfma(b, w, n, L, lds, o);
        b.sync();
        CG_STAMP(6)
#if !defined(CG_EXP_NO_DENSE)
        jac_up_mfma(b, w, n, lds, o);
        b.sync();
        CG_STAMP(7)
        jac_bg_mfma(b, n, lds, o, w.Jext);
        b.sync();
        CG_STAMP(8)
        jac_factors_mfma<1>(b, w, n, lds, o, wfl);
#endif
        b.sync();
        CG_STAMP(9)
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = e / n, k = e - i * n;
            if (i == k) continue;
            PF6 pf; own_pair(sh, ch, i, k, true, pf);
            const double rdel = pf.rdel;
            double tc[D], ts[D], td[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) { tc[bb] = -c1 * pf.s2[bb]; ts[bb] = c1 * pf.c2[bb]; td[bb] = c2c * (pf.s2[bb] * rdel); }
            double Jb[D][D];
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int bb = 0; bb < D; ++bb)
                    Jb[a][bb] = J[(i * D + a) * N + k * D + bb]
                                - (Up[(i * D + a) * P + bb] * tc[bb] + Up[(i * D + a) * P + D + bb] * ts[bb] + Up[(i * D + a) * P + 2 * D] * td[bb]);
#pragma unroll 4
            for (int h = 0; h < HT; ++h) {
                const double* wh = wt + h * (P + 1);
                const double wd = wh[1 + 2 * D];
                double u = wh[0] + wd * pf.del;
                double q[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double wc = wh[1 + a], ws = wh[1 + D + a];
                    u += wc * pf.c2[a] + ws * pf.s2[a];
                    q[a] = wc * tc[a] + ws * ts[a] + wd * td[a];
                }
#if defined(CG_EXP_FREE_SIGMA)      /* timing experiment (garbage numbers): what a CACHED sigmoid would cost -- an upper bound on what sharing
                                       the exponential of the primal pass with this pass can gain (profiles/r03*_experiments.txt) */
                const double sg = 0.25 + 1e-3 * u;
#else
                const double sg = sigmoid_only(u);
#endif
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double vs = V[iV(i, a, h)] * sg;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) Jb[a][bb] -= vs * q[bb];
                }
            }
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int bb = 0; bb < D; ++bb) J[(i * D + a) * N + k * D + bb] = Jb[a][bb];
        }
        b.sync();
        CG_STAMP(10)
        if (w.Jext) {
            // J outside LDS: eight lanes share an entry of the diagonal block (partial sums over k = lane, lane + 8, ..., then a
            // fixed-order DPP reduction inside the group of eight) instead of n dependent loads per thread
            const int items = n * D * D;
            for (int e0 = (b.tid >> 6) << 6; e0 < items * 8; e0 += b.nthr) {
                const int e = (e0 + (b.tid & 63)) >> 3, sub = b.tid & 7;
                const bool ok = e < items;
                const int ec = ok ? e : 0, i = ec / (D * D), r = ec - i * (D * D), a = r / D, bb = r - a * D;
                double v = 0.0;
                for (int k = sub; k < n; k += 8)
                    if (k != i) v -= J[(i * D + a) * N + k * D + bb];
                { const double t = cg_dpp_f64<0x111>(v); v += sub >= 1 ? t : 0.0; }      // row_shr:1 / 2 / 4 restricted to the group of eight
                { const double t = cg_dpp_f64<0x112>(v); v += sub >= 2 ? t : 0.0; }
                { const double t = cg_dpp_f64<0x114>(v); v += sub >= 4 ? t : 0.0; }
                if (ok && sub == 7) J[(i * D + a) * N + i * D + bb] = v + ((a == bb) ? 1.0 : 0.0);
            }
        } else
        for (int e = b.tid; e < n * D * D; e += b.nthr) {
            const int i = e / (D * D), r = e - i * (D * D), a = r / D, bb = r - a * D;
            double v = (a == bb) ? 1.0 : 0.0;
            for (int k = 0; k < n; ++k)
                if (k != i) v -= J[(i * D + a) * N + k * D + bb];
            J[(i * D + a) * N + i * D + bb] = v;
        }
        b.sync();
        CG_STAMP(11)
    }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
    // ---------------------------------------------------------------------------------------
    // Second-order jets (directional passes of the Laplacian), dense probe: the two pair loops whose work items are (particle, unit)
    // used to recompute the Jet2 features of pair (i, j) in each of the 16 unit lanes (~150 instructions, 16-fold redundant).  As in
    // primal_pairs_dpp: a DPP row (16 lanes) owns particle i, each lane computes the features of ONE pair (i, 16 c + lane) and the row
    // walks j by row broadcasts (2 v_mov_dpp per double).  spsize = tpsize = 16.
    // ---------------------------------------------------------------------------------------
    template <int LANE>
    static __device__ __forceinline__ Jet2 jet_row_bcast(const Jet2& v) { return Jet2(row_bcast<LANE>(v.v), row_bcast<LANE>(v.d), row_bcast<LANE>(v.dd)); }
    struct JPF { Jet2 c2[D], s2[D], del, sr[D]; };        // sr = s2 / del (G pass); del (primal pass)
    // features of the lane's own pair; ok = false, or i == j in the G-pass flavour: all-zero jets (contribute nothing)
    template <bool GPASS>
    static __device__ __forceinline__ void jet_own_pair(const Jet2* sh, const Jet2* ch, int i, int j, bool ok, JPF& f) {
        Jet2 d2(0.0);
#pragma unroll
        for (int a = 0; a < D; ++a) {
            const Jet2 si = ok ? sh[i * D + a] : Jet2(0.0), ci = ok ? ch[i * D + a] : Jet2(1.0), sj = ok ? sh[j * D + a] : Jet2(1.0), cj = ok ? ch[j * D + a] : Jet2(0.0);
            const Jet2 sn = si * cj - ci * sj, cs = ci * cj + si * sj;
            f.s2[a] = 2.0 * (sn * cs); f.c2[a] = 1.0 - 2.0 * (sn * sn); d2 += sn * sn;
        }
        const bool diag = i == j, live = ok && !diag;
        if (live) {
            f.del = cg_sqrt(d2);
            if (GPASS) { const Jet2 rd = cg_rcp(f.del);
#pragma unroll
                for (int a = 0; a < D; ++a) f.sr[a] = f.s2[a] * rd; }
        } else {
            f.del = Jet2(0.0);
#pragma unroll
            for (int a = 0; a < D; ++a) {
                f.sr[a] = Jet2(0.0);
                if (GPASS || !ok) { f.s2[a] = Jet2(0.0); f.c2[a] = Jet2(0.0); }          // G pass: the l = k term is absent
                else { f.s2[a] = Jet2(0.0); f.c2[a] = Jet2(1.0); }                        // primal: exact diagonal feature [1.., 0.., 0]
            }
        }
    }
    template <int JJ>
    static __device__ __forceinline__ void jet_primal_step(const JPF& mine, const double (&wt)[P], double bt, int jbase, int n, int h, Jet2& acc, Jet2& raw) {
        if constexpr (JJ < 16) {
            if (jbase + JJ < n) {                                  // wave-uniform
                Jet2 u(bt);
                Jet2 fv(0.0);
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const Jet2 c2 = jet_row_bcast<JJ>(mine.c2[a]), s2 = jet_row_bcast<JJ>(mine.s2[a]);
                    u += wt[a] * c2 + wt[D + a] * s2;
                    if (h == a) fv = c2;
                    if (h == D + a) fv = s2;
                }
                const Jet2 del = jet_row_bcast<JJ>(mine.del);
                u += wt[2 * D] * del;
                if (h == 2 * D) fv = del;
                acc += cg_softplus(u);
                raw += fv;
                jet_primal_step<JJ + 1>(mine, wt, bt, jbase, n, h, acc, raw);
            }
        }
    }
    static __device__ __forceinline__ void primal_pairs_jet_dpp(const CgBlk& b, const double* th, int n, Jet2* lds, const CgFastLds& o) {
        const Jet2 *sh = lds + o.sh, *ch = lds + o.ch;
        Jet2 *m0 = lds + o.m0, *m1 = lds + o.m1;
        const double rn = 1.0 / (double)n;
        const int h = b.tid & 15;
        double wt[P]; const double bt = th[o_t0b + h];
#pragma unroll
        for (int f = 0; f < P; ++f) wt[f] = th[o_t0w + f * HT + h];
        for (int e0 = (b.tid >> 6) << 6; e0 < n * 16; e0 += b.nthr) {      // whole waves stay together
            const int i = (e0 + (b.tid & 63)) >> 4;
            const bool rowok = i < n;
            Jet2 acc(0.0), raw(0.0);
            for (int jb = 0; jb < n; jb += 16) {
                JPF mine; jet_own_pair<false>(sh, ch, rowok ? i : 0, jb + h < n ? jb + h : 0, rowok && jb + h < n, mine);
                jet_primal_step<0>(mine, wt, bt, jb, n, h, acc, raw);
            }
            if (rowok) { m1[i * HT + h] = acc * rn; if (h < P) m0[i * P + h] = raw * rn; }
        }
    }
    template <int JJ>
    static __device__ __forceinline__ void jet_g_step(const JPF& mine, const Jet2* sg1, const Jet2& sgk, const double (&kc)[D], const double (&ksn)[D], double kd,
                                                      int lbase, int n, int h, Jet2 (&acc)[D]) {
        if constexpr (JJ < 16) {
            if (lbase + JJ < n) {
                const Jet2 sgl = sg1[(lbase + JJ) * HS + h];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const Jet2 s2 = jet_row_bcast<JJ>(mine.s2[bb]), c2 = jet_row_bcast<JJ>(mine.c2[bb]), sr = jet_row_bcast<JJ>(mine.sr[bb]);
                    const Jet2 odd = kc[bb] * s2 + kd * sr;          // odd in r:  (-c1 w_c) s2 + (c2c w_d) s2 / del
                    const Jet2 evn = ksn[bb] * c2;                   // even in r: (c1 w_s) c2
                    acc[bb] += sgk * (odd + evn) - sgl * (evn - odd);
                }
                jet_g_step<JJ + 1>(mine, sg1, sgk, kc, ksn, kd, lbase, n, h, acc);
            }
        }
    }
    static __device__ __forceinline__ void g_pass_jet_dpp(const CgBlk& b, const double* th, int n, double L, Jet2* lds, const CgFastLds& o) {
        const Jet2 *sh = lds + o.sh, *ch = lds + o.ch, *sg1 = lds + o.sg1;
        Jet2* G = lds + o.G;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        const int h = b.tid & 15;
        double kc[D], ksn[D];
#pragma unroll
        for (int a = 0; a < D; ++a) { kc[a] = -c1 * th[o_W0 + a * HS + h]; ksn[a] = c1 * th[o_W0 + (D + a) * HS + h]; }
        const double kd = c2c * th[o_W0 + 2 * D * HS + h];
        for (int e0 = (b.tid >> 6) << 6; e0 < n * 16; e0 += b.nthr) {
            const int k = (e0 + (b.tid & 63)) >> 4;
            const bool rowok = k < n;
            const Jet2 sgk = rowok ? sg1[k * HS + h] : Jet2(0.0);
            Jet2 acc[D];
#pragma unroll
            for (int a = 0; a < D; ++a) acc[a] = Jet2(0.0);
            for (int lb = 0; lb < n; lb += 16) {
                JPF mine; jet_own_pair<true>(sh, ch, rowok ? k : 0, lb + h < n ? lb + h : 0, rowok && lb + h < n, mine);
                jet_g_step<0>(mine, sg1, sgk, kc, ksn, kd, lb, n, h, acc);
            }
            if (rowok) {
#pragma unroll
                for (int bb = 0; bb < D; ++bb) G[iG(k, h, bb)] = acc[bb] * (rn * rn);
            }
        }
    }
#else       // host builds: declarations only, so that the `if constexpr (CG_ON_DEVICE && ...)` branches of primal() / jacobian() parse
    static bool primal_pairs_lds_fits(const CgBlk&, int);
    static void primal_pairs_lds(const CgBlk&, const WFrag&, int, double*, const CgFastLds&);
    static void primal_pairs_dpp(const CgBlk&, const WFrag&, int, double*, const CgFastLds&);
    static void primal_dense_mfma(const CgBlk&, const WFrag&, const double*, int, double*, const CgFastLds&);
    static void jacobian_mfma(const CgBlk&, const double*, const WFrag&, int, double, double*, const CgFastLds&);
    static void primal_pairs_jet_dpp(const CgBlk&, const double*, int, Jet2*, const CgFastLds&);
    static void g_pass_jet_dpp(const CgBlk&, const double*, int, double, Jet2*, const CgFastLds&);
#endif

    // ---------------------------------------------------------------------------------------
    // primal pass: fills sh,ch,m0,s1,sg1,m1,gbar,cb,sg2,s2,z in LDS.
    // ---------------------------------------------------------------------------------------
    template <class T>
    // hot >= 0 (jet types only): the particle whose coordinate carries the tangent of this directional pass.
    static CG_DEVI void primal(const CgBlk& b, const double* __restrict__ th, const T* x /*n*D*/,
                               int n, double L, T* lds, const CgFastLds& o, const WFrag* wf = nullptr, int hot = -1) {
        T *sh = lds + o.sh, *ch = lds + o.ch, *m0 = lds + o.m0, *s1 = lds + o.s1, *sg1 = lds + o.sg1,
               *m1 = lds + o.m1, *gbar = lds + o.gbar, *cb = lds + o.cb, *sg2 = lds + o.sg2, *s2 = lds + o.s2,
               *z = lds + o.z;
        const double rn = 1.0 / (double)n;
        for (int e = b.tid; e < n * D; e += b.nthr) {
            T s, c; cg_sincos(x[e] * (CG_PI / L), s, c, wf != nullptr && !wf->inline_libm);    // out of line in the sampler kernels only
            sh[e] = s; ch[e] = c;
        }
        b.sync();
        CG_STAMP(1)
        bool pairs_done = false;
        if constexpr (CG_ON_DEVICE && sizeof(T) == sizeof(double) && HS == 16 && HT == 16) {
            if (wf) {
                if (primal_pairs_lds_fits(b, n)) primal_pairs_lds(b, *wf, n, (double*)lds, o);
                else primal_pairs_dpp(b, *wf, n, (double*)lds, o);
                pairs_done = true;
            }
        }
        // pair-primal: item (i,h)
        constexpr int HM = HT > P ? HT : P;          // lanes h < P also carry one raw-feature mean
        if constexpr (CG_ON_DEVICE && CgIsJet<T>::value && HS == 16 && HT == 16) {
            if (!pairs_done && hot < 0 && (b.nthr & 63) == 0) { primal_pairs_jet_dpp(b, th, n, (Jet2*)lds, o); pairs_done = true; }   // dense probe
        }
        if constexpr (CgIsJet<T>::value) {
            // Sparse tangent of a basis-direction pass: rows i != hot see one jet pair (j = hot) and n - 1 double pairs;
            // the hot row's n x HM (j, h) units are spread over the whole workgroup through a scratch in J's slot (dead
            // until the Jacobian assembly) and summed in the original order, so the results are bitwise those of the
            // dense loop below.  ~2.7x shorter critical path of the largest phase of a pass.
            if (!pairs_done && hot >= 0 && 2 * n * HM <= n * D * n * D) {
                T* spl = lds + o.J; T* rwl = spl + n * HM;
                for (int e = b.tid; e < n * HM; e += b.nthr) {
                    const int i = e / HM, h = e - i * HM;
                    if (i == hot) continue;
                    double wt[P], bt = 0.0;
                    const bool do_t = h < HT;
#pragma unroll
                    for (int f = 0; f < P; ++f) wt[f] = do_t ? th[o_t0w + f * HT + h] : 0.0;
                    if (do_t) bt = th[o_t0b + h];
                    T acc = T(0.0), raw = T(0.0);
                    for (int j = 0; j < n; ++j) {
                        if (j != hot) {
                            PairFT<double> pd; pairfeat_val(sh, ch, i, j, pd);
                            double ud = bt + wt[2 * D] * pd.del;
#pragma unroll
                            for (int a = 0; a < D; ++a) ud += wt[a] * pd.c2[a] + wt[D + a] * pd.s2[a];
                            if (do_t) acc += T(softplus_only(ud));
                            if (h < P) {
                                double fv = pd.del;
#pragma unroll
                                for (int a = 0; a < D; ++a) { if (h == a) fv = pd.c2[a]; if (h == D + a) fv = pd.s2[a]; }
                                raw += T(fv);
                            }
                        } else {
                            PairFT<T> pf; pairfeat(sh, ch, i, j, pf);
                            T u = T(bt);
#pragma unroll
                            for (int a = 0; a < D; ++a) u += wt[a] * pf.c2[a] + wt[D + a] * pf.s2[a];
                            u += wt[2 * D] * pf.del;
                            if (do_t) acc += cg_softplus(u);
                            if (h < P) {
                                T fv = pf.del;
#pragma unroll
                                for (int a = 0; a < D; ++a) { if (h == a) fv = pf.c2[a]; if (h == D + a) fv = pf.s2[a]; }
                                raw += fv;
                            }
                        }
                    }
                    if (do_t) m1[i * HT + h] = acc * rn;
                    if (h < P) m0[i * P + h] = raw * rn;
                }
                for (int e = b.tid; e < n * HM; e += b.nthr) {          // hot row: unit (j, h)
                    const int j = e / HM, h = e - j * HM;
                    const bool do_t = h < HT;
                    PairFT<T> pf; pairfeat(sh, ch, hot, j, pf);
                    T u = T(do_t ? th[o_t0b + h] : 0.0);
#pragma unroll
                    for (int a = 0; a < D; ++a) u += (do_t ? th[o_t0w + a * HT + h] : 0.0) * pf.c2[a] + (do_t ? th[o_t0w + (D + a) * HT + h] : 0.0) * pf.s2[a];
                    u += (do_t ? th[o_t0w + 2 * D * HT + h] : 0.0) * pf.del;
                    spl[e] = do_t ? cg_softplus(u) : T(0.0);
                    T fv = pf.del;
#pragma unroll
                    for (int a = 0; a < D; ++a) { if (h == a) fv = pf.c2[a]; if (h == D + a) fv = pf.s2[a]; }
                    rwl[e] = fv;
                }
                b.sync();
                for (int h = b.tid; h < HM; h += b.nthr) {
                    T acc = T(0.0), raw = T(0.0);
                    for (int j = 0; j < n; ++j) { acc += spl[j * HM + h]; raw += rwl[j * HM + h]; }
                    if (h < HT) m1[hot * HT + h] = acc * rn;
                    if (h < P) m0[hot * P + h] = raw * rn;
                }
                pairs_done = true;
            }
        }
        if (!pairs_done)
        for (int e = b.tid; e < n * HM; e += b.nthr) {
            const int i = e / HM, h = e - i * HM;
            double wt[P], bt = 0.0;
            const bool do_t = h < HT;
#pragma unroll
            for (int f = 0; f < P; ++f) wt[f] = do_t ? th[o_t0w + f * HT + h] : 0.0;
            if (do_t) bt = th[o_t0b + h];
            T acc = T(0.0), raw = T(0.0);
#pragma unroll 2
            for (int j = 0; j < n; ++j) {
                PairFT<T> pf; pairfeat(sh, ch, i, j, pf);
                T u = T(bt);
#pragma unroll
                for (int a = 0; a < D; ++a) u += wt[a] * pf.c2[a] + wt[D + a] * pf.s2[a];
                u += wt[2 * D] * pf.del;
                if (do_t) acc += cg_softplus(u);
                if (h < P) {
                    T fv = pf.del;
#pragma unroll
                    for (int a = 0; a < D; ++a) { if (h == a) fv = pf.c2[a]; if (h == D + a) fv = pf.s2[a]; }
                    raw += fv;
                }
            }
            if (do_t) m1[i * HT + h] = acc * rn;
            if (h < P) m0[i * P + h] = raw * rn;
        }
        b.sync();
        CG_STAMP(2)
        if constexpr (CG_ON_DEVICE && sizeof(T) == sizeof(double) && HS == 16 && HT == 16) {
#if defined(CG_EXP_NO_DENSE)
            if (wf) { for (int e = b.tid; e < n * D; e += b.nthr) z[e] = x[e] + 1e-3 * m1[e]; b.sync(); CG_STAMP(3) return; }
#endif
            if (wf) { primal_dense_mfma(b, *wf, (const double*)x, n, (double*)lds, o); CG_STAMP(3) return; }
        }
        // layer 0 of the one-particle stream: u1_i = W0^T m0_i + b0 (s0 = 0, src/flow.py:16-18,45)
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            T u = T(th[o_s0b + h]);
#pragma unroll
            for (int f = 0; f < P; ++f) u += th[o_W0 + f * HS + h] * m0[i * P + f];
            T sp, sg; cg_softplus_sigmoid(u, sp, sg);
            s1[e] = sp; sg1[e] = sg;
        }
        b.sync();
        for (int h = b.tid; h < HS; h += b.nthr) {
            T a = T(0.0);
            for (int i = 0; i < n; ++i) a += s1[i * HS + h];
            gbar[h] = a * rn;
        }
        b.sync();
        for (int h = b.tid; h < HS; h += b.nthr) {
            T a = T(th[o_s1b + h]);
#pragma unroll
            for (int g = 0; g < HS; ++g) a += th[o_Wb + g * HS + h] * gbar[g];
            cb[h] = a;
        }
        b.sync();
        // last one-particle layer (residual, src/flow.py:51-52)
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int i = e / HS, h = e - i * HS;
            T u = cb[h];
#pragma unroll
            for (int g = 0; g < HS; ++g) u += th[o_Wa + g * HS + h] * s1[i * HS + g];
#pragma unroll
            for (int g = 0; g < HT; ++g) u += th[o_Wc + g * HS + h] * m1[i * HT + g];
            T sp, sg; cg_softplus_sigmoid(u, sp, sg);
            sg2[e] = sg; s2[e] = s1[e] + sp;
        }
        b.sync();
        for (int e = b.tid; e < n * D; e += b.nthr) {     // z = x + final(s2), src/flow.py:54-55
            const int i = e / D, a = e - i * D;
            T v = x[e] + th[o_fb + a];
#pragma unroll
            for (int h = 0; h < HS; ++h) v += th[o_fw + h * D + a] * s2[i * HS + h];
            z[e] = v;
        }
        b.sync();
        CG_STAMP(3)
    }

    // ---------------------------------------------------------------------------------------
    // Jacobian assembly (needs primal() results in LDS).  Writes J (N x N, N = n*D, row-major).
    // ---------------------------------------------------------------------------------------
    template <class T>
    static CG_DEVI void jacobian(const CgBlk& b, const double* __restrict__ th, int n, double L,
                                 T* lds, const CgFastLds& o, const WFrag* wf = nullptr) {
        const T *sh = lds + o.sh, *ch = lds + o.ch, *sg1 = lds + o.sg1, *sg2 = lds + o.sg2;
        T *U = lds + o.U, *V = lds + o.V, *Bm = lds + o.Bm, *Up = lds + o.Up, *G = lds + o.G, *J = lds + o.J;
        const int N = n * D;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / L, c2c = CG_PI / (2.0 * L);
        if constexpr (CG_ON_DEVICE && sizeof(T) == sizeof(double) && HS == 16 && HT == 16) {
            if (wf) { jacobian_mfma(b, th, *wf, n, L, (double*)lds, o); return; }
        }
        // two-particle layer weights -> arena, [h][bias, w_0..w_{P-1}]: the Jacobian pass reads them with broadcast loads
        double* wt = (double*)(lds + o.wt);
        for (int e = b.tid; e < HT * (P + 1); e += b.nthr) {
            const int h = e / (P + 1), f = e - h * (P + 1);
            wt[e] = f == 0 ? th[o_t0b + h] : th[o_t0w + (f - 1) * HT + h];
        }
        // per-particle left factors: item (i,a,g)
        for (int e = b.tid; e < n * D * HS; e += b.nthr) {
            const int i = e / (D * HS), r = e - i * (D * HS), a = r / HS, g = r - a * HS;
            T ua = T(th[o_fw + g * D + a]), ub = T(0.0);
#pragma unroll
            for (int h = 0; h < HS; ++h) {
                const T rih = th[o_fw + h * D + a] * sg2[i * HS + h];
                ua += rih * th[o_Wa + g * HS + h];
                ub += rih * th[o_Wb + g * HS + h];
            }
            U[e] = ua; Bm[iB(i, a, g)] = ub;
        }
        for (int e = b.tid; e < n * D * HT; e += b.nthr) {
            const int i = e / (D * HT), r = e - i * (D * HT), a = r / HT, g = r - a * HT;
            T v = T(0.0);
#pragma unroll
            for (int h = 0; h < HS; ++h) v += (th[o_fw + h * D + a] * th[o_Wc + g * HS + h]) * sg2[i * HS + h];
            V[iV(i, a, g)] = v * rn;
        }
        // G pass: item (k,h)
        bool g_done = false;
        if constexpr (CG_ON_DEVICE && CgIsJet<T>::value && HS == 16 && HT == 16) {
            if ((b.nthr & 63) == 0) { g_pass_jet_dpp(b, th, n, L, (Jet2*)lds, o); g_done = true; }
        }
        if (!g_done)
        for (int e = b.tid; e < n * HS; e += b.nthr) {
            const int k = e / HS, h = e - k * HS;
            double w_c[D], w_s[D];
#pragma unroll
            for (int a = 0; a < D; ++a) { w_c[a] = th[o_W0 + a * HS + h]; w_s[a] = th[o_W0 + (D + a) * HS + h]; }
            const double w_d = th[o_W0 + 2 * D * HS + h];
            const T sgk = sg1[k * HS + h];
            T acc[D];
#pragma unroll
            for (int a = 0; a < D; ++a) acc[a] = T(0.0);
#pragma unroll 2
            for (int l = 0; l < n; ++l) {
                if (l == k) continue;
                PairFT<T> pf; pairfeat(sh, ch, k, l, pf);
                const T rdel = cg_rcp(pf.del);
                const T sgl = sg1[l * HS + h];
#pragma unroll
                for (int bb = 0; bb < D; ++bb) {
                    const T odd = (-c1 * w_c[bb]) * pf.s2[bb] + (c2c * w_d) * (pf.s2[bb] * rdel);   // odd in r
                    const T evn = (c1 * w_s[bb]) * pf.c2[bb];                                       // even in r
                    // (W0^T T_kl)[h,bb] = odd + evn ;  (W0^T T_lk)[h,bb] = -odd + evn
                    acc[bb] += sgk * (odd + evn) - sgl * (evn - odd);
                }
            }
#pragma unroll
            for (int bb = 0; bb < D; ++bb) G[iG(k, h, bb)] = acc[bb] * rn * rn;
        }
        b.sync();
        CG_STAMP(4)
        for (int e = b.tid; e < n * D * P; e += b.nthr) {      // U'_i: item (i,a,f)
            const int i = e / (D * P), r = e - i * (D * P), a = r / P, f = r - a * P;
            T v = T(0.0);
#pragma unroll
            for (int g = 0; g < HS; ++g) v += th[o_W0 + f * HS + g] * (U[(i * D + a) * HS + g] * sg1[i * HS + g]);
            Up[e] = v * rn;
        }
        b.sync();
        CG_STAMP(5)
        // Jacobian pass: item (i,k), k != i
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = e / n, k = e - i * n;
            if (i == k) continue;
            PairFT<T> pf; pairfeat(sh, ch, i, k, pf);
            const T rdel = cg_rcp(pf.del);
            T tc[D], ts[D], td[D];
#pragma unroll
            for (int bb = 0; bb < D; ++bb) { tc[bb] = -c1 * pf.s2[bb]; ts[bb] = c1 * pf.c2[bb]; td[bb] = c2c * (pf.s2[bb] * rdel); }
            T Jb[D][D];
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int bb = 0; bb < D; ++bb)
                    Jb[a][bb] = -(Up[(i * D + a) * P + bb] * tc[bb] + Up[(i * D + a) * P + D + bb] * ts[bb] +
                                  Up[(i * D + a) * P + 2 * D] * td[bb]);
#pragma unroll 4
            for (int h = 0; h < HT; ++h) {
                const double* wh = wt + h * (P + 1);
                const double wd = wh[1 + 2 * D];
                T u = wh[0] + wd * pf.del;
                T q[D];
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const double wc = wh[1 + a], ws = wh[1 + D + a];
                    u += wc * pf.c2[a] + ws * pf.s2[a];
                    q[a] = wc * tc[a] + ws * ts[a] + wd * td[a];
                }
                const T sg = cg_sigmoid(u);
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const T vs = V[iV(i, a, h)] * sg;
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) Jb[a][bb] -= vs * q[bb];
                }
            }
#pragma unroll 4
            for (int g = 0; g < HS; ++g) {
#pragma unroll
                for (int a = 0; a < D; ++a) {
                    const T bg = Bm[iB(i, a, g)];
#pragma unroll
                    for (int bb = 0; bb < D; ++bb) Jb[a][bb] += bg * G[iG(k, g, bb)];
                }
            }
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int bb = 0; bb < D; ++bb) J[(i * D + a) * N + k * D + bb] = Jb[a][bb];
        }
        b.sync();
        CG_STAMP(6)
        // diagonal blocks from sum_k J_ik = I
        for (int e = b.tid; e < n * D * D; e += b.nthr) {
            const int i = e / (D * D), r = e - i * (D * D), a = r / D, bb = r - a * D;
            T v = T((a == bb) ? 1.0 : 0.0);
            for (int k = 0; k < n; ++k)
                if (k != i) v -= J[(i * D + a) * N + k * D + bb];
            J[(i * D + a) * N + i * D + bb] = v;
        }
        b.sync();
        CG_STAMP(7)
    }

    // ---------------------------------------------------------------------------------------
    // Slater matrix D_ij = exp(i k_j . z_i) (the L^{-d/2} factor is added analytically)
    // k_j = 2 pi / L * sp_indices[state_idx[j]]    (src/slater.py:14-17, src/logpsi.py:23)
    // ---------------------------------------------------------------------------------------
    static CG_DEVI void slater_matrix(const CgBlk& b, const double* z, const double* __restrict__ spk /*M x D, already * 2pi/L*/,
                                      const int* __restrict__ sidx, int n, double* Dm, bool ool = false) {
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = e / n, j = e - i * n;
            const double* k = sidx ? spk + (size_t)sidx[j] * D : spk + j * D;      // sidx == nullptr: spk already holds the n occupied k
            double ph = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) ph += k[a] * z[i * D + a];
            double sn, cs; cg_sincos(ph, sn, cs, ool);
            Dm[2 * e] = cs; Dm[2 * e + 1] = sn;
        }
        b.sync();
    }

    // log Psi(x) = log phi(z(x)) + 1/2 log|det J|  ->  out = [Re, Im]  (src/logpsi.py:30-31)
    // also returns the two pieces separately (make_logphi_logjacdet, src/logpsi.py:35-53).
    static CG_DEVI void logpsi(const CgBlk& b, const double* __restrict__ th, const double* x /*LDS*/,
                               const double* __restrict__ spk, const int* __restrict__ sidx, int n, double L,
                               double* lds, const CgFastLds& o, double& re_phi, double& im_phi, double& half_logdetJ,
                               const WFrag* wf = nullptr) {
        primal(b, th, x, n, L, lds, o, wf);
        jacobian(b, th, n, L, lds, o, wf);
        int* perm = (int*)(lds + o.perm);
        double la, ar;
#if defined(__HIP_DEVICE_COMPILE__)
        if (o.wave_lu) {
            // Slater matrix first (its slot does not overlap J), then the two LUs run barrier-free in registers:
            // wave 0 the real Jacobian, wave 1 (if the workgroup has one) the complex Slater matrix, concurrently.
            slater_matrix(b, lds + o.z, spk, sidx, n, lds + o.Dm, true);
            CG_STAMP(12)
            double* res = (double*)perm;
            const int wave = b.tid >> 6, cw = b.nthr > 64 ? 1 : 0;
            if (b.nthr == 64 && (n == 13 || n == 16) && n * D == 2 * n) {
                // single-wave workgroup: both factorisations interleaved in one instruction stream
                double lr, l2, a2;
#if defined(CG_EXP_NO_LU)           /* timing experiment (garbage numbers): both factorisations free -- the bound on any LU speed-up */
                lr = lds[o.J] + lds[o.J + 27]; l2 = lds[o.Dm]; a2 = lds[o.Dm + 1];
#else
                if (n == 13) cg_wave_lu2_both<26, 13>(lds + o.J, 2 * n, 2 * n, lds + o.lus, lds + o.Dm, n, n, lds + o.lus + 32, lr, l2, a2);
                else cg_wave_lu2_both<32, 16>(lds + o.J, 2 * n, 2 * n, lds + o.lus, lds + o.Dm, n, n, lds + o.lus + 32, lr, l2, a2);
#endif
                half_logdetJ = 0.5 * lr; la = l2; ar = a2;
                CG_STAMP(13)
                CG_STAMP(14)
            } else {
            if (wave == 0) {
                const int NN = n * D;
                const double v = NN == 26 ? cg_wave_lu2_logabsdet<26>(lds + o.J, NN, NN, lds + o.lus)
                                          : cg_wave_lu2_logabsdet<32>(lds + o.J, NN, NN, lds + o.lus);
                if (b.tid == 0) res[0] = v;
            }
            CG_STAMP(13)
            __builtin_amdgcn_sched_barrier(0);         // keep the two register-resident factorisations apart (spills otherwise)
            if (wave == cw) {
                double l2, a2;
                if (n == 13) cg_wave_lu2_logdet_complex<13>(lds + o.Dm, n, n, lds + o.lus + 32, l2, a2);
                else cg_wave_lu2_logdet_complex<16>(lds + o.Dm, n, n, lds + o.lus + 32, l2, a2);
                if ((b.tid & 63) == 0) { res[1] = l2; res[2] = a2; }
            }
            b.sync();
            half_logdetJ = 0.5 * res[0]; la = res[1]; ar = res[2];
            b.sync();
            CG_STAMP(14)
            }
        } else {
            // larger sizes: workgroup-wide blocked LUs (8-column register panels, MFMA trailing updates).  With both matrices in
            // LDS (o.dual) they are factored concurrently; otherwise the Slater matrix shares J's LDS (o.Dm == o.J) and is
            // formed after the real factorisation.
            double* res = (double*)perm;
            if (o.dual && b.nthr >= 192 && b.nthr <= 1024 && n * D <= 128 && n <= 64) {
                // both matrices have LDS of their own: the Slater matrix is formed first and the two LUs run concurrently
                slater_matrix(b, lds + o.z, spk, sidx, n, lds + o.Dm, true);
                CG_STAMP(12)                           // (diagnostic builds, this branch: 12 = Slater matrix, 13 = both LUs)
                double lr;
                // second generation (rows never move, GEMM-only helpers) wherever its 16-byte row accesses are aligned: every D = 2 system
                if (((n * D) & 1) == 0 && n >= 8) cg_blocked_lu_dual2(b, lds + o.J, n * D, n * D, lds + o.Dm, n, n, res, lr, la, ar);
                else cg_blocked_lu_dual(b, lds + o.J, n * D, n * D, lds + o.Dm, n, n, res, lr, la, ar);
                half_logdetJ = 0.5 * lr;
                CG_STAMP(13)
                CG_STAMP(14)
            } else {
                half_logdetJ = 0.5 * cg_blocked_lu_logabsdet(b, lds + o.J, n * D, n * D, res);
                CG_STAMP(12)                           // (diagnostic builds, this branch: 12 = real LU, 13 = Slater matrix, 14 = complex LU)
                slater_matrix(b, lds + o.z, spk, sidx, n, lds + o.Dm, true);
                CG_STAMP(13)
                cg_blocked_lu_logdet_complex(b, lds + o.Dm, n, n, res, la, ar);
                CG_STAMP(14)
            }
        }
#else
        {
            half_logdetJ = 0.5 * cg_lu_logabsdet(b, lds + o.J, n * D, n * D, perm, nullptr, true);
            slater_matrix(b, lds + o.z, spk, sidx, n, lds + o.Dm, true);
            cg_lu_logdet_complex(b, lds + o.Dm, n, n, perm, la, ar, true);
        }
#endif
        re_phi = la - (double)n * (0.5 * D) * cg_log_pos(L);      // (table logarithm: this was the last libm call of an evaluation)
        im_phi = ar;
    }
};

// host-side layout of the LDS arena.
//   alias = false: every array has its own slot (derivative kernels keep all intermediates for the reverse pass).
//   alias = true : sampler layout.  Lifetimes inside CgFast::logpsi:
//       persistent      sh ch z sg1 sg2 perm
//       primal only     m0 s1 m1 gbar cb s2           (dead once z is formed)        -> share a slot with V Bm Up G
//       jacobian        U (dead once Up is formed)                                    -> lives inside J
//                       V Bm Up G J
//       Slater matrix   Dm: after the LU of J                                          -> on top of V Bm Up G, or on J
static CG_HD CgFastLds cg_fast_layout(int n, int D, int HS, int HT, bool alias, bool mfma = false) {
    CgFastLds o; int P = 2 * D + 1, t = 0;
    auto take = [&](int cnt) { int r = t; t += (cnt + 1) & ~1; return r; };
    const bool small = n * D <= 32 && n <= 16;
    o.dual = 0;
    int dead0 = 0;                                      // (large n, sampler) start of what is dead once J is assembled
    if (alias && !small) {
        // large n: everything but z and the LU scratch is dead after the Jacobian assembly; kept contiguous so that the
        // Slater matrix fits over it (n = 57: 6626 doubles against 2 n^2 = 6498) and is factored WHILE J is factored.
        o.z = take(n * D); o.perm = take(164);           // CG_LU_DUAL_DOUBLES: flags and per-panel records (pivot rows, live row tiles) of both LUs
        dead0 = t;
        o.sh = take(n * D); o.ch = take(n * D); o.sg1 = take(n * HS); o.sg2 = take(n * HS);
        o.wt = take(HT * (P + 1) + HS * D);
    } else {
        o.sh = take(n * D); o.ch = take(n * D); o.z = take(n * D);
        o.sg1 = take(n * HS); o.sg2 = take(n * HS);
        // results of the wave-level LUs (3 doubles) or, on the LDS LU path, its argmax scratch (>= 40 doubles)
        o.perm = take(small ? 4 : 40);
        o.wt = take(HT * (P + 1) + HS * D);             // two-particle layer weights [h][bias, P weights], then Wf (HS x D)
    }
    if (!alias) {
        o.m0 = take(n * P); o.s1 = take(n * HS); o.m1 = take(n * HT); o.gbar = take(HS); o.cb = take(HS); o.s2 = take(n * HS);
        o.U = take(n * D * HS); o.V = take(n * (HT * D + 2)); o.Bm = take(n * (HS * D + 2)); o.Up = take(n * D * P); o.G = take(n * (HS * D + 2));
        o.J = take(n * D * n * D);
        o.Dm = take(2 * n * n);
        o.lus = take(64);
        o.total = t;
        o.wave_lu = (n * D <= 32 && n <= 16) ? 1 : 0;
        return o;
    }
    const int base = t;
    o.m0 = take(n * P); o.s1 = take(n * HS); o.m1 = take(n * HT); o.gbar = take(HS); o.cb = take(HS); o.s2 = take(n * HS);
    const int end_primal = t;
    t = base;
    // MFMA path (device, spsize = tpsize = 16): V is formed after the B.G product and reuses Bm's slot
    if (mfma && HS == HT) { o.Bm = take(n * (HS * D + 2)); o.V = o.Bm; }
    else { o.V = take(n * (HT * D + 2)); o.Bm = take(n * (HS * D + 2)); }
    o.Up = take(n * D * P); o.G = take(n * (HS * D + 2));
    const int end_jac = t;
    t = end_primal > end_jac ? end_primal : end_jac;
    o.J = take(n * D * n * D);
    o.U = o.J;                                          // n*D*HS <= (n*D)^2 whenever HS <= n*D
    if (n * D * HS > n * D * n * D) { o.U = take(n * D * HS); }
    const int end_dead = end_primal > end_jac ? end_primal : end_jac;
    if (end_jac - base >= 2 * n * n) o.Dm = base;       // Slater matrix over the dead per-particle factors
    else if (!small && end_dead - dead0 >= 2 * n * n) o.Dm = dead0;   // ... over everything that is dead after the assembly
    else o.Dm = o.J;                                    // over J after its LU (2 n^2 <= (n D)^2)
    o.dual = (!small && o.Dm != o.J) ? 1 : 0;
    // pivot-row scratch of the wave-level LUs (2 x 32 doubles): behind the Slater matrix, still inside the dead factors
    o.lus = base + ((2 * n * n + 1) & ~1);
    o.wave_lu = (n * D <= 32 && n <= 16 && o.Dm != o.J && o.lus + 64 <= end_jac) ? 1 : 0;
    o.total = t;
    return o;
}
