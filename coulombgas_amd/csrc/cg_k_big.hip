// cg_k_big.hip -- derivative kernels of the larger systems (n > 16), second generation (cg_big.hpp): planned LDS / workspace
// placement, row passes.  Instantiated for the (dim 2, 16, 16) flow of every shipped run; the other configurations keep the first
// generation (cg_k_derivs_*.hip).
#if defined(CG_INV_TRACE)      /* diagnostic builds: cycles of thread 0 per phase of the real panel inverse (panel, barrier, pivot rows, barrier, update) */
#include <hip/hip_runtime.h>
__device__ unsigned long long cg_inv_ph[8];
#define CG_INV_T_DECL unsigned long long tprev_ = 0;
#define CG_INV_T(i) { if (b.tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); if (i > 0) atomicAdd(&cg_inv_ph[i], t_ - tprev_); tprev_ = t_; } }
#endif
#include "cg_host.hpp"
#include "cg_big.hpp"
#if defined(CG_INV_TRACE)
extern "C" int cg_debug_inv_trace(cg_ctx* c, unsigned long long* out8, int clear) {
    CG_HIP(c, hipStreamSynchronize(c->stream));
    CG_HIP(c, hipMemcpyFromSymbol(out8, HIP_SYMBOL(cg_inv_ph), sizeof(unsigned long long) * 8));
    if (clear) { unsigned long long z[8] = {0}; CG_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(cg_inv_ph), z, sizeof(z))); }
    return CG_OK;
}
#endif

// One walker per workgroup; the workspace slot of a workgroup is indexed by blockIdx.x, the batch goes in launches of `gridDim.x` walkers.
template <int D, int HS, int HT, int NT>
__global__ void __launch_bounds__(NT, NT <= 256 ? 2 : 1) k_scores_big(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab,
                             const double* __restrict__ x, const int* __restrict__ sidx, int B, int w0, double* __restrict__ score,
                             double* ws, typename CgBig<D, HS, HT>::LayS lay) {
#if defined(__HIP_DEVICE_COMPILE__)
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    CG_STAMP_INIT
    constexpr int NP = CgFast<D, HS, HT>::NPARAM;
    const int n = m.n, N = n * D, w = w0 + blockIdx.x;
    if (w < B) CgBig<D, HS, HT>::scores(b, theta, x + (size_t)w * N, spk, sidx + (size_t)w * n, n, m.L, score + (size_t)w * NP * 2, lds,
                                        ws + (size_t)blockIdx.x * lay.ws_total, lay);
    CG_STAMP_FLUSH
#endif
}

// scores of B walkers by the planned kernel of the larger systems: 1 launched, 0 not served, < 0 error
int cg_big_scores(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, double* score) {
    constexpr int D = 2, HS = 16, HT = 16;
    if (c->dim != D || c->hs != HS || c->ht != HT || c->n <= 16) return 0;      // (n <= 16: k_scores, everything in LDS)
    {
        int rc;
        const int n = c->n;
        if (cg_env_int("CG_BIG", 1) == 0) return 0;
        const int bnt = cg_env_int("CG_BIG_NT", n * D <= 64 ? 256 : 512);
        const int per_cu = bnt == 256 ? cg_env_int("CG_BIG_PER_CU", 2) : 1;
        const size_t capb = (size_t)cg_env_int("CG_BIG_LDS_KB", per_cu == 2 ? 79 : 159) * 1024;
        const auto bl = CgBig<D, HS, HT>::layout_scores(n, bnt, capb / sizeof(double) - CG_TAB_DOUBLES);
        if (cg_env_int("CG_BIG_DEBUG", 0))
            fprintf(stderr, "cg_big_scores n=%d nt=%d ok=%d lds %u doubles, ws %u doubles per workgroup; J %d JT %d Dm %d Dinv %d s1k %d s2k %d m1k %d Bb %d Vb %d Ub %d Rb %d u1b %d u1i %d sg1b %d\n",
                    n, bnt, bl.ok, bl.lds_total, bl.ws_total, bl.c.J, bl.c.JT, bl.c.Dm, bl.c.Dinv, bl.s1k, bl.s2k, bl.m1k, bl.Bb, bl.Vb, bl.Ub, bl.Rb, bl.u1b, bl.u1i, bl.sg1b);
        if (!bl.ok) return 0;
        const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + (size_t)bl.lds_total);
        const int chunk = std::min(B, c->cu_count * per_cu * cg_env_int("CG_BIG_ROUNDS", 4));
        if ((rc = ensure_ws(c, sizeof(double) * ((size_t)bl.ws_total * chunk + 8)))) return rc;
        auto go = [&](auto ntc) -> int {
            constexpr int NT = decltype(ntc)::value;
            if (int r = set_lds(c, k_scores_big<D, HS, HT, NT>, lds)) return r;
            for (int w0 = 0; w0 < B; w0 += chunk)
                hipLaunchKernelGGL((k_scores_big<D, HS, HT, NT>), dim3(std::min(chunk, B - w0)), dim3(NT), lds, c->stream, m, (const double*)c->d_theta,
                                   (const double*)c->d_spk, (const double*)c->d_tab, x, sidx, B, w0, score, (double*)c->ws, bl);
            return 0;
        };
        if ((rc = bnt == 256 ? go(std::integral_constant<int, 256>{}) : go(std::integral_constant<int, 512>{}))) return rc;
        return 1;
    }
}

template <int D, int HS, int HT, int NT>
__global__ void __launch_bounds__(NT, NT <= 256 ? 2 : 1) k_gradlap_big(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab,
                              const double* __restrict__ x, const int* __restrict__ sidx, int B, int w0, int mode, const double* __restrict__ v,
                              double* __restrict__ grad, double* __restrict__ lap, double* ws, typename CgBig<D, HS, HT>::LayG lay) {
#if defined(__HIP_DEVICE_COMPILE__)
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    CG_STAMP_INIT
    const int n = m.n, N = n * D, w = w0 + blockIdx.x;
    if (w < B) CgBig<D, HS, HT>::grad_laplacian(b, theta, x + (size_t)w * N, spk, sidx + (size_t)w * n, n, m.L, mode, v + (size_t)w * N,
                                                grad + (size_t)w * N * 2, lap + 2 * w, lds, ws + (size_t)blockIdx.x * lay.ws_total, lay);
    CG_STAMP_FLUSH
#endif
}

// grad / Laplacian of B walkers (Hutchinson modes) by the planned kernel of the larger systems: 1 launched, 0 not served, < 0 error
int cg_big_grad_lap(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, int mode, const double* v, double* grad, double* lap) {
    constexpr int D = 2, HS = 16, HT = 16;
    if (c->dim != D || c->hs != HS || c->ht != HT || (mode != 1 && mode != 2) || !v || c->n <= 16) return 0;   // (n <= 16: k_grad_lap2 in every mode)
    int rc;
    const int n = c->n;
    if (cg_env_int("CG_BIG", 1) == 0 || cg_env_int("CG_BIG_LAP", 1) == 0) return 0;
    const int bnt = cg_env_int("CG_BIG_NT", n * D <= 64 ? 256 : 512);
    const int per_cu = bnt == 256 ? cg_env_int("CG_BIG_PER_CU", 2) : 1;
    const size_t capb = (size_t)cg_env_int("CG_BIG_LDS_KB", per_cu == 2 ? 79 : 159) * 1024;
    const auto bl = CgBig<D, HS, HT>::layout_gradlap(n, bnt, mode, capb / sizeof(double) - CG_TAB_DOUBLES);
    if (cg_env_int("CG_BIG_DEBUG", 0))
        fprintf(stderr, "cg_big_grad_lap n=%d nt=%d mode=%d ok=%d lds %u doubles, ws %u doubles per workgroup; J %d JT %d Dm %d Dinv %d Ta %d Am %d Hk %d Bb %d Vb %d Ub %d Rb %d jp %d Vt %d Bmt %d Upt %d Jp %d\n",
                n, bnt, mode, bl.ok, bl.lds_total, bl.ws_total, bl.c.J, bl.c.JT, bl.c.Dm, bl.c.Dinv, bl.Ta, bl.Am, bl.Hk, bl.Bb, bl.Vb, bl.Ub, bl.Rb, bl.jp, bl.Vt, bl.Bmt, bl.Upt, bl.Jp);
    if (!bl.ok) return 0;
    const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + (size_t)bl.lds_total);
    const int chunk = std::min(B, c->cu_count * per_cu * cg_env_int("CG_BIG_ROUNDS", 4));
    if ((rc = ensure_ws(c, sizeof(double) * ((size_t)bl.ws_total * chunk + 8)))) return rc;
    auto go = [&](auto ntc) -> int {
        constexpr int NT = decltype(ntc)::value;
        if (int r = set_lds(c, k_gradlap_big<D, HS, HT, NT>, lds)) return r;
        for (int w0 = 0; w0 < B; w0 += chunk)
            hipLaunchKernelGGL((k_gradlap_big<D, HS, HT, NT>), dim3(std::min(chunk, B - w0)), dim3(NT), lds, c->stream, m, (const double*)c->d_theta,
                               (const double*)c->d_spk, (const double*)c->d_tab, x, sidx, B, w0, mode, v, grad, lap, (double*)c->ws, bl);
        return 0;
    };
    if ((rc = bnt == 256 ? go(std::integral_constant<int, 256>{}) : go(std::integral_constant<int, 512>{}))) return rc;
    return 1;
}

// Both at once for the optimisation step (src/VMC.py:35 and the jacrev of main.py:278 on the same walkers): the set-up -- flow, Jacobian,
// the two inverses, g: 60 % of k_scores_big -- runs once; the grad / Laplacian part parks what the score passes need in the workspace
// (CgBig::Stash), then the score passes run in their own layout.  Results: those of the two kernels, bit for bit.
template <int D, int HS, int HT, int NT>
__global__ void __launch_bounds__(NT, NT <= 256 ? 2 : 1) k_gradlap_scores_big(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab,
                              const double* __restrict__ x, const int* __restrict__ sidx, int B, int w0, int mode, const double* __restrict__ v,
                              double* __restrict__ grad, double* __restrict__ lap, double* __restrict__ score, double* ws,
                              typename CgBig<D, HS, HT>::LayG lg, typename CgBig<D, HS, HT>::LayS ls, typename CgBig<D, HS, HT>::Stash st) {
#if defined(__HIP_DEVICE_COMPILE__)
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    CG_STAMP_INIT
    constexpr int NP = CgFast<D, HS, HT>::NPARAM;
    const int n = m.n, N = n * D, w = w0 + blockIdx.x;
    if (w < B) {
        double* wg = ws + (size_t)blockIdx.x * ((size_t)lg.ws_total + st.total + ls.ws_total);
        double* stash = wg + lg.ws_total;
        CgBig<D, HS, HT>::grad_laplacian(b, theta, x + (size_t)w * N, spk, sidx + (size_t)w * n, n, m.L, mode, v + (size_t)w * N,
                                         grad + (size_t)w * N * 2, lap + 2 * w, lds, wg, lg, stash, &st);
        b.sync();
        const double* JT = wg + (size_t)(~lg.c.JT);
        if (lg.c.JT >= 0) {         // (small systems: J^-T sat in LDS, where the score layout is about to put its own arrays)
            CgBig<D, HS, HT>::copy2(b, stash + st.JT, lds + lg.c.JT, N * N);
            JT = stash + st.JT;
            b.sync();
        }
        CgBig<D, HS, HT>::scores_unstash(b, n, lds, stash + st.total, ls, stash, st);
        CgBig<D, HS, HT>::score_passes(b, theta, n, m.L, score + (size_t)w * NP * 2, lds, stash + st.total, ls, JT);
    }
    CG_STAMP_FLUSH
#endif
}

// 1 launched, 0 not served (the caller runs the two kernels one after the other), < 0 error
int cg_big_grad_lap_scores(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, int mode, const double* v, double* grad, double* lap,
                           double* score) {
    constexpr int D = 2, HS = 16, HT = 16;
    typedef CgBig<D, HS, HT> Big;
    if (c->dim != D || c->hs != HS || c->ht != HT || (mode != 1 && mode != 2) || !v || c->n <= 16) return 0;
    int rc;
    const int n = c->n;
    if (cg_env_int("CG_BIG", 1) == 0 || cg_env_int("CG_BIG_LAP", 1) == 0 || cg_env_int("CG_BIG_FUSED", 1) == 0) return 0;
    const int bnt = cg_env_int("CG_BIG_NT", n * D <= 64 ? 256 : 512);
    const int per_cu = bnt == 256 ? cg_env_int("CG_BIG_PER_CU", 2) : 1;
    const size_t capb = (size_t)cg_env_int("CG_BIG_LDS_KB", per_cu == 2 ? 79 : 159) * 1024;
    const auto lg = Big::layout_gradlap(n, bnt, mode, capb / sizeof(double) - CG_TAB_DOUBLES);
    const auto ls = Big::layout_scores(n, bnt, capb / sizeof(double) - CG_TAB_DOUBLES);
    if (!lg.ok || !ls.ok) return 0;
    const auto st = Big::stash_layout(n);
    const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + (size_t)std::max(lg.lds_total, ls.lds_total));
    const size_t per_wg = (size_t)lg.ws_total + st.total + ls.ws_total;
    const int chunk = std::min(B, c->cu_count * per_cu * cg_env_int("CG_BIG_ROUNDS", 4));
    if ((rc = ensure_ws(c, sizeof(double) * (per_wg * chunk + 8)))) return rc;
    auto go = [&](auto ntc) -> int {
        constexpr int NT = decltype(ntc)::value;
        if (int r = set_lds(c, k_gradlap_scores_big<D, HS, HT, NT>, lds)) return r;
        for (int w0 = 0; w0 < B; w0 += chunk)
            hipLaunchKernelGGL((k_gradlap_scores_big<D, HS, HT, NT>), dim3(std::min(chunk, B - w0)), dim3(NT), lds, c->stream, m, (const double*)c->d_theta,
                               (const double*)c->d_spk, (const double*)c->d_tab, x, sidx, B, w0, mode, v, grad, lap, score, (double*)c->ws, lg, ls, st);
        return 0;
    };
    if ((rc = bnt == 256 ? go(std::integral_constant<int, 256>{}) : go(std::integral_constant<int, 512>{}))) return rc;
    return 1;
}

#if defined(CG_STAMPS)
CG_STAMP_READER(cg_debug_stamps_big)         /* diagnostic builds only: the per-phase cycle counters of this unit's kernels */
#endif
