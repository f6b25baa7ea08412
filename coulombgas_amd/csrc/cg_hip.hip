// cg_hip.hip -- gfx950 kernels + the C-ABI of include/coulombgas.h.
// One workgroup per walker; all per-walker intermediates live in LDS (layout: cg_fast_layout).
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#include "../../include/coulombgas.h"
#include "cg_common.hpp"
#include "cg_linalg.hpp"
#include "cg_flow_fast.hpp"
#include "cg_dispatch.hpp"
#include "cg_rng.hpp"
#include "cg_ewald.hpp"
#include "cg_derivs.hpp"
#include "cg_generic.hpp"

// ------------------------------------------------------------------------------------------
// device-side model descriptor (passed by value to every kernel)
// ------------------------------------------------------------------------------------------
struct CgDev {
    int n;
    double L;
    CgFastLds lay;
};

// minimum waves per SIMD the sampler kernels are register-allocated for (2 -> <= 256 VGPRs, 4 -> <= 128)
#ifndef CG_WAVES_PER_EU
#define CG_WAVES_PER_EU 2
#endif

enum { CG_MODE_LOGPSI = 0, CG_MODE_FLOW = 1, CG_MODE_JAC = 2 };

template <int D, int HS, int HT, int MAXT>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? CG_WAVES_PER_EU : 1)) k_logpsi(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab, const double* __restrict__ x, const int* __restrict__ sidx, int B, int mode,
                         double* __restrict__ logphi, double* __restrict__ hld, double* __restrict__ logpsi_out,
                         double* __restrict__ logp_out, double* __restrict__ z_out, double* __restrict__ J_out) {
    using F = CgFast<D, HS, HT>;
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    const int n = m.n, N = n * D;
    double* xs = lds + m.lay.total;
    typename F::WFrag wfrag; const typename F::WFrag* wf = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (HS == 16 && HT == 16) { F::load_frags(theta, wfrag); wf = &wfrag; }
#endif
    for (int w = blockIdx.x; w < B; w += gridDim.x) {
        for (int e = b.tid; e < N; e += b.nthr) xs[e] = x[(size_t)w * N + e];
        b.sync();
        if (mode == CG_MODE_LOGPSI) {
            double re, im, h;
            F::logpsi(b, theta, xs, spk, sidx + (size_t)w * n, n, m.L, lds, m.lay, re, im, h, wf);
            if (b.tid == 0) {
                if (logphi) { logphi[2 * w] = re; logphi[2 * w + 1] = im; }
                if (hld) hld[w] = h;
                if (logpsi_out) { logpsi_out[2 * w] = re + h; logpsi_out[2 * w + 1] = im; }
                if (logp_out) logp_out[w] = 2.0 * (re + h);
            }
        } else {
            F::primal(b, theta, xs, n, m.L, lds, m.lay, wf);
            if (z_out)
                for (int e = b.tid; e < N; e += b.nthr) z_out[(size_t)w * N + e] = lds[m.lay.z + e];
            if (mode == CG_MODE_JAC) {
                F::jacobian(b, theta, n, m.L, lds, m.lay, wf);
                for (int e = b.tid; e < N * N; e += b.nthr) J_out[(size_t)w * N * N + e] = lds[m.lay.J + e];
            }
        }
        b.sync();
    }
}

// Batched Metropolis chain: src/MCMC.py:22-39.  One workgroup owns one walker for all mc_steps;
// x is read once and written once, the proposal/accept state never leaves the CU.
// NS > 0: specialised on n = NS and on a workgroup of exactly MAXT threads (CG_MCMC_SPECIALS).
template <int D, int HS, int HT, int MAXT, int NS = 0>
__global__ void __launch_bounds__(MAXT, (MAXT <= 256 ? CG_WAVES_PER_EU : 1)) k_mcmc(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab, double* __restrict__ x, const int* __restrict__ sidx, int B, int steps, double stddev,
                       uint64_t seed, uint64_t walker_offset, const double* __restrict__ noise,
                       const double* __restrict__ unif, double* __restrict__ logp_out,
                       unsigned long long* __restrict__ n_accept) {
    using F = CgFast<D, HS, HT>;
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, NS > 0 ? MAXT : (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    const int n = NS > 0 ? NS : m.n, N = n * D;
    CgFastLds lay_s = m.lay;
    if constexpr (NS > 0) lay_s = cg_fast_layout(NS, D, HS, HT, true, HS == 16 && HT == 16);   // folds to constants
    const CgFastLds& lay = lay_s;
    double* xc = lds + lay.total;          // current configuration
    double* xp = xc + ((N + 1) & ~1);        // proposal
    int* flag = (int*)(xp + ((N + 1) & ~1));
    CG_STAMP_INIT
    typename F::WFrag wfrag; const typename F::WFrag* wf = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (HS == 16 && HT == 16) { F::load_frags(theta, wfrag); wf = &wfrag; }
#endif
    for (int w = blockIdx.x; w < B; w += gridDim.x) {
        const int* si = sidx + (size_t)w * n;
        for (int e = b.tid; e < N; e += b.nthr) xc[e] = x[(size_t)w * N + e];
        b.sync();
        double logp = 0.0;
        unsigned int nacc = 0;
        // step -1 evaluates logp of the initial configuration (src/MCMC.py:36) through the SAME call site as the
        // proposals, so that the (large, unrolled) log Psi code exists once in the instruction stream.
        for (int s = -1; s < steps; ++s) {
            CG_STAMP_START(0)
            for (int e = b.tid; e < N; e += b.nthr) {
                double g = 0.0;
                if (s >= 0) g = noise ? noise[((size_t)s * B + w) * N + e]
                                      : cg_philox_normal_ool(seed, walker_offset + w, (uint32_t)s, (uint32_t)e);
                xp[e] = xc[e] + stddev * g;
            }
            b.sync();
            CG_STAMP(0)
            double re, im, h;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CG_NO_OPAQUE_TID)
            // the lane id is made opaque once per evaluation: everything derived from it (LDS addresses, tile indices)
            // is then recomputed inside the evaluation instead of being hoisted out of the chain loop and spilled
            int tid_o = b.tid; asm volatile("" : "+v"(tid_o));
            const CgBlk be{tid_o, b.nthr};
#else
            const CgBlk& be = b;
#endif
            F::logpsi(be, theta, xp, spk, si, n, m.L, lds, lay, re, im, h, wf);
            const double lp = 2.0 * (re + h);
            if (b.tid == 0) {
                int acc = 1;
                if (s >= 0) {
                    const double u = unif ? unif[(size_t)s * B + w] : cg_philox_uniform_ool(seed, walker_offset + w, (uint32_t)s);
                    const double ratio = cg_exp_ool(lp - logp);
                    acc = (u < ratio) ? 1 : 0;            // NaN -> reject, +inf -> accept (src/MCMC.py:28-29)
                }
                *flag = acc;
            }
            b.sync();
            const int acc = *flag;
            if (acc) {
                for (int e = b.tid; e < N; e += b.nthr) xc[e] = xp[e];
                logp = lp;
                if (s >= 0) ++nacc;
            }
            b.sync();
            CG_STAMP_END(15)
        }
        for (int e = b.tid; e < N; e += b.nthr) x[(size_t)w * N + e] = xc[e];
        if (b.tid == 0) {
            if (logp_out) logp_out[w] = logp;
            if (n_accept && nacc) atomicAdd(n_accept, (unsigned long long)nacc);
        }
        b.sync();
    }
    CG_STAMP_FLUSH
}

template <int D>
__global__ void k_ewald(const double* __restrict__ x, int B, int n, double L, double kappa, double rs,
                        const int* __restrict__ G, const double* __restrict__ gk, int nG, int Gmax, double g0,
                        double* __restrict__ V) {
    extern __shared__ double lds[];
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    const int N = n * D;
    double* xs = lds;
    double* rest = lds + ((N + 1) & ~1);
    for (int w = blockIdx.x; w < B; w += gridDim.x) {
        for (int e = b.tid; e < N; e += b.nthr) xs[e] = x[(size_t)w * N + e];
        b.sync();
        const double v = cg_ewald_walker<D>(b, xs, n, L, kappa, rs, G, gk, nG, Gmax, g0, rest);
        if (b.tid == 0) V[w] = v;
        b.sync();
    }
}

__global__ void k_wrap(double* __restrict__ x, size_t count, double L) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { const double v = x[i]; x[i] = v - L * floor(v / L); }
}

// grad / Laplacian of log Psi w.r.t. x (cg_derivs.hpp); per-walker workspace in HBM.
// Register budget: 3 waves/SIMD (168 VGPRs) for d = 2.  For d = 3 the jets of the d x d blocks need ~480 spilled VGPRs at
// that budget and the spill-heavy code hipcc (ROCm 7.2) generates returns wrong jets for every direction but the first
// (deterministic; parity test test_grad_laplacian_all_modes[case1]); at 2 waves/SIMD it is correct and no slower.
#ifndef CG_DERIV_WAVES
#define CG_DERIV_WAVES 3
#endif
#define CG_DERIV_WAVES_OF(D) ((D) == 2 ? CG_DERIV_WAVES : (CG_DERIV_WAVES < 2 ? CG_DERIV_WAVES : 2))
// JLDS: the Jet2 arena of the directional passes lives in LDS (2 workgroups per CU, so 2 waves/SIMD of registers).
template <int D, int HS, int HT, bool JLDS>
__global__ void __launch_bounds__(256, (JLDS ? (CG_DERIV_WAVES_OF(D) < 2 ? CG_DERIV_WAVES_OF(D) : 2) : CG_DERIV_WAVES_OF(D))) k_grad_lap(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab, const double* __restrict__ x, const int* __restrict__ sidx, int B, int mode,
                           const double* __restrict__ v, double* __restrict__ grad, double* __restrict__ lap,
                           double* ws, size_t ws_per_walker, typename CgDerivs<D, HS, HT>::Layout lay) {
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    CG_STAMP_INIT
    const int n = m.n, N = n * D;
    const double* th = theta;
    if (JLDS && lay.theta_lds) {         // per-lane weight reads of the jet passes from LDS instead of the vector L1
        double* th_l = lds + CgDerivs<D, HS, HT>::lds_doubles(n, b.nthr) + CgDerivs<D, HS, HT>::jet_lds_doubles(lay);
        for (int e = b.tid; e < CgFast<D, HS, HT>::NPARAM; e += b.nthr) th_l[e] = theta[e];
        __syncthreads();
        th = th_l;
    }
    for (int w = blockIdx.x; w < B; w += gridDim.x) {
        CgDerivs<D, HS, HT>::grad_laplacian(b, th, x + (size_t)w * N, spk, sidx + (size_t)w * n, n, m.L, mode,
                                            v ? v + (size_t)w * N : nullptr, grad + (size_t)w * N * 2, lap + 2 * w,
                                            ws + (size_t)blockIdx.x * ws_per_walker, lds, lay);
        b.sync();
    }
    CG_STAMP_FLUSH
}

template <int D, int HS, int HT>
__global__ void __launch_bounds__(256, CG_DERIV_WAVES_OF(D)) k_param_vjp(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab, const double* __restrict__ x, const int* __restrict__ sidx, int B,
                            const double* __restrict__ w_re, const double* __restrict__ w_im,
                            double* __restrict__ partial /* gridDim.x x P */, double* __restrict__ score /* nullable B x P x 2 */,
                            double* ws, size_t ws_per_walker, typename CgDerivs<D, HS, HT>::Layout lay) {
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    const int n = m.n, N = n * D;
    constexpr int P = CgFast<D, HS, HT>::NPARAM;
    double* gacc = partial ? partial + (size_t)blockIdx.x * P : nullptr;
    if (gacc) for (int e = b.tid; e < P; e += b.nthr) gacc[e] = 0.0;
    b.sync();
    for (int w = blockIdx.x; w < B; w += gridDim.x) {
        CgDerivs<D, HS, HT>::param_vjp(b, theta, x + (size_t)w * N, spk, sidx + (size_t)w * n, n, m.L,
                                       w_re ? w_re[w] : 1.0, w_im ? w_im[w] : 0.0, gacc,
                                       score ? score + (size_t)w * P * 2 : nullptr,
                                       ws + (size_t)blockIdx.x * ws_per_walker, lds, lay);
        b.sync();
    }
}

// deterministic second-stage reduction of per-workgroup partial gradients: out[p] = sum_g partial[g][p]
__global__ void k_reduce_rows(const double* __restrict__ partial, int rows, int P, double* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double a = 0.0;
    for (int r = 0; r < rows; ++r) a += partial[(size_t)r * P + p];
    out[p] = a;
}

// Quantum Fisher matrix of stochastic reconfiguration (src/sr.py:74-76):  F[p][q] = (1/B) sum_b Re( conj(S[b][p]) S[b][q] )
// = (1/B) sum_b ( Sr[b][p] Sr[b][q] + Si[b][p] Si[b][q] ),  S = per-sample scores (B x P, complex interleaved).
// One wave per 16 x 16 tile of the upper triangle (mirrored on store); the batch axis is the K of v_mfma_f64_16x16x4.
__global__ void __launch_bounds__(256) k_fisher(const double* __restrict__ S, int B, int P, double* __restrict__ F) {
    typedef double d4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles = (P + 15) >> 4;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= tiles * tiles) return;
    const int ti = tile / tiles, tj = tile - ti * tiles;
    if (tj < ti) return;
    const int col = lane & 15, kq = lane >> 4;
    const int p = 16 * ti + col, q = 16 * tj + col;
    const bool pok = p < P, qok = q < P;
    d4_t acc = {0, 0, 0, 0};
    for (int b1 = 0; b1 < B; b1 += 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {                 // four independent load groups in flight per trip
            const int b = b1 + 4 * u + kq;
            const bool bok = b < B;
            const double* sa = S + ((size_t)b * P + p) * 2;
            const double* sb = S + ((size_t)b * P + q) * 2;
            const double a_re = (bok && pok) ? sa[0] : 0.0, a_im = (bok && pok) ? sa[1] : 0.0;
            const double b_re = (bok && qok) ? sb[0] : 0.0, b_im = (bok && qok) ? sb[1] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_re, b_re, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_im, b_im, acc, 0, 0, 0);
        }
    }
    const double rb = 1.0 / (double)B;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int pr = 16 * ti + kq + 4 * r;
        if (pr < P && qok) {
            const double v = acc[r] * rb;
            F[(size_t)pr * P + q] = v;
            if (ti != tj) F[(size_t)q * P + pr] = v;
        }
    }
}
// mean over the batch of the complex scores (src/sr.py:70): out[2 p + c] = (1/B) sum_b S[b][p][c]; fixed summation order
// Column sums of the resident score matrix over one slice of the batch (blockIdx.y): out[slice][c] = sum_{b in slice} S[b][c].
// The slices are summed in fixed order by k_reduce_rows (deterministic), the 1/B of the mean is applied afterwards.
__global__ void __launch_bounds__(256) k_score_mean(const double* __restrict__ S, int B, int P2 /* 2 P */, int chunk, double* __restrict__ out) {
    __shared__ double part[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
    double a = 0.0;
    if (c < P2) for (int b = b0 + rg; b < b1; b += 4) a += S[(size_t)b * P2 + c];
    part[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0 && c < P2) out[(size_t)blockIdx.y * P2 + c] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
}

// out[slice][p] = sum_{b in slice} ( w_re[b] Sre[b][p] + w_im[b] Sim[b][p] ): the theta-VJP from resident scores
__global__ void __launch_bounds__(256) k_score_gemv(const double* __restrict__ S, const double* __restrict__ w_re,
                                                    const double* __restrict__ w_im, int B, int P, int chunk, double* __restrict__ out) {
    __shared__ double part[256];
    const int p = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
    double a = 0.0;
    if (p < P)
        for (int b = b0 + rg; b < b1; b += 4) {
            const double* s = S + ((size_t)b * P + p) * 2;
            a += w_re[b] * s[0] + w_im[b] * s[1];
        }
    part[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0 && p < P) out[(size_t)blockIdx.y * P + p] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
}

__global__ void k_scale(double* __restrict__ buf, size_t count, double s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) buf[i] *= s;
}



// ---- general-depth path (cg_generic.hpp): any FermiNet depth / widths; workspace in HBM --------------------------
__global__ void __launch_bounds__(256) k_gen_logpsi(CgGenModel m, CgGenWs w, const double* __restrict__ theta,
                                                    const double* __restrict__ spk, const double* __restrict__ tab,
                                                    const double* __restrict__ x, const int* __restrict__ sidx, int B, int mode,
                                                    double* __restrict__ logphi, double* __restrict__ hld, double* __restrict__ logpsi_out,
                                                    double* __restrict__ logp_out, double* __restrict__ z_out, double* __restrict__ J_out,
                                                    double* wsall) {
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    double* ws = wsall + (size_t)blockIdx.x * w.total;
    const int n = m.n, N = n * m.dim;
    for (int q = blockIdx.x; q < B; q += gridDim.x) {
        if (mode == CG_MODE_LOGPSI) {
            double re, im, h;
            CgGenK::logpsi(b, m, w, theta, spk, sidx + (size_t)q * n, x + (size_t)q * N, ws, re, im, h);
            if (b.tid == 0) {
                if (logphi) { logphi[2 * q] = re; logphi[2 * q + 1] = im; }
                if (hld) hld[q] = h;
                if (logpsi_out) { logpsi_out[2 * q] = re + h; logpsi_out[2 * q + 1] = im; }
                if (logp_out) logp_out[q] = 2.0 * (re + h);
            }
        } else {
            CgGen<double>::flow(b, m, theta, x + (size_t)q * N, ws + w.da, mode == CG_MODE_JAC);
            if (z_out) for (int e = b.tid; e < N; e += b.nthr) z_out[(size_t)q * N + e] = ws[w.da + m.o_z + e];
            if (mode == CG_MODE_JAC) for (int e = b.tid; e < N * N; e += b.nthr) J_out[(size_t)q * N * N + e] = ws[w.da + m.o_J + e];
        }
        b.sync();
    }
}

__global__ void __launch_bounds__(256) k_gen_mcmc(CgGenModel m, CgGenWs w, const double* __restrict__ theta,
                                                  const double* __restrict__ spk, const double* __restrict__ tab, double* __restrict__ x,
                                                  const int* __restrict__ sidx, int B, int steps, double stddev, uint64_t seed,
                                                  uint64_t walker_offset, const double* __restrict__ noise, const double* __restrict__ unif,
                                                  double* __restrict__ logp_out, unsigned long long* __restrict__ n_accept,
                                                  double* wsall) {
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    int* flag = (int*)(cg_dyn_lds + CG_TAB_DOUBLES);
    double* ws = wsall + (size_t)blockIdx.x * w.total;
    double *xc = ws + w.xc, *xp = ws + w.xp;
    const int n = m.n, N = n * m.dim;
    for (int q = blockIdx.x; q < B; q += gridDim.x) {
        const int* si = sidx + (size_t)q * n;
        for (int e = b.tid; e < N; e += b.nthr) xc[e] = x[(size_t)q * N + e];
        b.sync();
        double logp = 0.0;
        unsigned int nacc = 0;
        for (int s = -1; s < steps; ++s) {
            CG_STAMP_START(0)
            for (int e = b.tid; e < N; e += b.nthr) {
                double g = 0.0;
                if (s >= 0) g = noise ? noise[((size_t)s * B + q) * N + e] : cg_philox_normal(seed, walker_offset + q, (uint32_t)s, (uint32_t)e);
                xp[e] = xc[e] + stddev * g;
            }
            b.sync();
            double re, im, h;
            CgGenK::logpsi(b, m, w, theta, spk, si, xp, ws, re, im, h);
            const double lp = 2.0 * (re + h);
            if (b.tid == 0) {
                int acc = 1;
                if (s >= 0) {
                    const double u = unif ? unif[(size_t)s * B + q] : cg_philox_uniform(seed, walker_offset + q, (uint32_t)s);
                    acc = (u < exp(lp - logp)) ? 1 : 0;
                }
                *flag = acc;
            }
            b.sync();
            const int acc = *flag;
            if (acc) { for (int e = b.tid; e < N; e += b.nthr) xc[e] = xp[e]; logp = lp; if (s >= 0) ++nacc; }
            b.sync();
        }
        for (int e = b.tid; e < N; e += b.nthr) x[(size_t)q * N + e] = xc[e];
        if (b.tid == 0) { if (logp_out) logp_out[q] = logp; if (n_accept && nacc) atomicAdd(n_accept, (unsigned long long)nacc); }
        b.sync();
    }
}

__global__ void __launch_bounds__(256) k_gen_param_vjp(CgGenModel m, CgGenWs w, const double* __restrict__ theta,
                                                       const double* __restrict__ spk, const double* __restrict__ tab,
                                                       const double* __restrict__ x, const int* __restrict__ sidx, int B,
                                                       const double* __restrict__ w_re, const double* __restrict__ w_im,
                                                       double* __restrict__ partial /* gridDim.x x P */, double* __restrict__ score /* nullable B x P x 2 */,
                                                       double* wsall) {
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    double* ws = wsall + (size_t)blockIdx.x * w.total;
    const int n = m.n, N = n * m.dim, P = m.nparam;
    double* gacc = partial ? partial + (size_t)blockIdx.x * P : nullptr;
    if (gacc) for (int e = b.tid; e < P; e += b.nthr) gacc[e] = 0.0;
    b.sync();
    for (int q = blockIdx.x; q < B; q += gridDim.x) {
        CgGenK::param_vjp(b, m, w, theta, spk, sidx + (size_t)q * n, x + (size_t)q * N, w_re ? w_re[q] : 1.0, w_im ? w_im[q] : 0.0,
                          gacc, score ? score + (size_t)q * P * 2 : nullptr, ws);
        b.sync();
    }
}

__global__ void __launch_bounds__(256) k_gen_grad_lap(CgGenModel m, CgGenWs w, const double* __restrict__ theta,
                                                      const double* __restrict__ spk, const double* __restrict__ tab,
                                                      const double* __restrict__ x, const int* __restrict__ sidx, int B, int mode,
                                                      const double* __restrict__ v, double* __restrict__ grad, double* __restrict__ lap,
                                                      double* wsall) {
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    double* ws = wsall + (size_t)blockIdx.x * w.total;
    const int n = m.n, N = n * m.dim;
    for (int q = blockIdx.x; q < B; q += gridDim.x) {
        CgGenK::grad_laplacian(b, m, w, theta, spk, sidx + (size_t)q * n, x + (size_t)q * N, mode, v ? v + (size_t)q * N : nullptr,
                               grad + (size_t)q * N * 2, lap + 2 * q, ws, lds);
        b.sync();
    }
}

// fp64 peak micro-benchmarks (roofline denominators for bench.py; /opt/skills/guides has no f64 row)
__global__ void __launch_bounds__(256) k_peak_fma64(double* out, int iters, double a, double b) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (double)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}
typedef double cg_d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_peak_mfma64(double* out, int iters, double a, double b) {
    cg_d4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
    const double av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
    }
    double s = c0[0] + c1[1] + c2[2] + c3[3];
    if (s == 12345.678) out[0] = s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

struct Chunk { void* p; size_t cap; };

struct cg_ctx {
    int device = 0, n = 0, dim = 0, depth = 0, hs = 0, ht = 0, M = 0, P = 0;
    double L = 0;
    bool fast = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void* bounce = nullptr; size_t bounce_cap = 0;     // pinned staging buffer for large device-to-host results
    double* d_theta = nullptr;
    double* d_spk = nullptr;
    double* d_tab = nullptr;     // exp / log tables of cg_common.hpp
    bool have_theta = false;
    // ewald
    bool have_ewald = false;
    double kappa = 0, rs = 0, g0 = 0;
    int nG = 0, Gmax = 0;
    int* d_G = nullptr;
    double* d_gk = nullptr;
    int ptr_mode = CG_PTR_HOST;
    int block_threads = 0;
    int cu_count = 256;
    CgFastLds lay;
    CgGenModel gm;               // general-depth path (fast == false)
    CgGenWs gw;
    CgGenWs gwv;      // the same + the reverse-pass arena of the theta-VJP
    unsigned long long* d_accept = nullptr;
    // staging arena for host-pointer mode + internal workspaces
    std::vector<Chunk> chunks;
    size_t cur = 0, off = 0;
    // persistent workspace (derivative kernels)
    void* ws = nullptr; size_t ws_cap = 0;
    double* d_scores = nullptr; size_t scores_cap = 0; int scores_B = 0;     // resident per-sample scores (cg_scores_*)
    std::string err;
};

#define CG_FAIL(ctx, code, ...)                                         \
    do {                                                                \
        char _b[512]; snprintf(_b, sizeof(_b), __VA_ARGS__);            \
        if (ctx) (ctx)->err = _b;                                       \
        g_last_error = _b;                                              \
        return (code);                                                  \
    } while (0)

#define CG_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) CG_FAIL(ctx, CG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

static int arena_reset(cg_ctx* c) {
    if (c->chunks.size() > 1) {
        size_t tot = 0;
        for (auto& ch : c->chunks) { tot += ch.cap; (void)hipFree(ch.p); }
        c->chunks.clear();
        void* p = nullptr;
        if (hipMalloc(&p, tot) != hipSuccess) return CG_ERR_HIP;
        c->chunks.push_back({p, tot});
    }
    c->cur = 0; c->off = 0;
    return CG_OK;
}
static void* arena_take(cg_ctx* c, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    while (c->cur < c->chunks.size()) {
        if (c->off + bytes <= c->chunks[c->cur].cap) { void* p = (char*)c->chunks[c->cur].p + c->off; c->off += bytes; return p; }
        ++c->cur; c->off = 0;
    }
    size_t cap = std::max(bytes, (size_t)(c->chunks.empty() ? (1u << 20) : 2 * c->chunks.back().cap));
    void* p = nullptr;
    if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
    c->chunks.push_back({p, cap});
    c->cur = c->chunks.size() - 1; c->off = bytes;
    return p;
}

// An argument that is an input and/or output array in either pointer mode.
struct Arg {
    void* user; void* dev; size_t bytes; bool in, out;
};
static int stage(cg_ctx* c, Arg& a) {
    if (!a.user) { a.dev = nullptr; return CG_OK; }
    if (c->ptr_mode == CG_PTR_DEVICE) { a.dev = a.user; return CG_OK; }
    a.dev = arena_take(c, a.bytes);
    if (!a.dev) CG_FAIL(c, CG_ERR_HIP, "device staging allocation of %zu bytes failed", a.bytes);
    if (a.in) CG_HIP(c, hipMemcpyAsync(a.dev, a.user, a.bytes, hipMemcpyHostToDevice, c->stream));
    return CG_OK;
}
static int unstage(cg_ctx* c, Arg& a) {
    if (!a.user || c->ptr_mode == CG_PTR_DEVICE || !a.out) return CG_OK;
    if (a.bytes >= ((size_t)1 << 20)) {
        // Large results (Fisher matrices, score blocks) go through a pinned buffer: a device-to-host copy into pageable
        // memory ran at ~1 GB/s on part of the pool (9 MB Fisher matrix: 10 ms), DMA into pinned memory + memcpy does not.
        if (c->bounce_cap < a.bytes) {
            if (c->bounce) { (void)hipHostFree(c->bounce); c->bounce = nullptr; c->bounce_cap = 0; }
            if (hipHostMalloc(&c->bounce, a.bytes, hipHostMallocDefault) == hipSuccess) c->bounce_cap = a.bytes;
            else { c->bounce = nullptr; (void)hipGetLastError(); }
        }
        if (c->bounce) {
            CG_HIP(c, hipMemcpyAsync(c->bounce, a.dev, a.bytes, hipMemcpyDeviceToHost, c->stream));
            CG_HIP(c, hipStreamSynchronize(c->stream));
            memcpy(a.user, c->bounce, a.bytes);
            return CG_OK;
        }
    }
    CG_HIP(c, hipMemcpyAsync(a.user, a.dev, a.bytes, hipMemcpyDeviceToHost, c->stream));
    return CG_OK;
}
static int finish(cg_ctx* c) {
    CG_HIP(c, hipGetLastError());
    if (c->ptr_mode == CG_PTR_HOST) CG_HIP(c, hipStreamSynchronize(c->stream));
    return CG_OK;
}

static int auto_threads(int n) {
    if (n <= 16) return 64;
    if (n <= 24) return 128;
    if (n <= 40) return 256;
    if (n <= 64) return 512;
    return 1024;
}
static int threads_of(const cg_ctx* c) { return c->block_threads > 0 ? c->block_threads : auto_threads(c->n); }

static CgDev make_dev(const cg_ctx* c) {
    CgDev m; m.n = c->n; m.L = c->L; m.lay = c->lay;
    return m;
}

template <class K>
static int set_lds(cg_ctx* c, K kernel, size_t bytes) {
    if (bytes > 160 * 1024) CG_FAIL(c, CG_ERR_UNSUPPORTED, "workgroup needs %zu bytes of LDS (> 160 KiB): n too large for the LDS-resident path", bytes);
    if (bytes > 48 * 1024) CG_HIP(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return CG_OK;
}

extern "C" {

const char* cg_last_error(const cg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int cg_create(cg_ctx** out, int device, int n, int dim, int depth, int spsize, int tpsize, double L,
              const double* sp_indices, int M) {
    if (!out) CG_FAIL((cg_ctx*)nullptr, CG_ERR_ARG, "cg_create: out is NULL");
    *out = nullptr;
    if (n < 1 || (dim != 2 && dim != 3) || depth < 2 || spsize < 1 || tpsize < 1 || !(L > 0) || !sp_indices || M < n)
        CG_FAIL((cg_ctx*)nullptr, CG_ERR_ARG, "cg_create: bad argument (n=%d dim=%d depth=%d spsize=%d tpsize=%d L=%g M=%d); depth >= 2 "
                "(src/flow.py:52 is ill-formed for depth 1), dim in {2,3}, M >= n", n, dim, depth, spsize, tpsize, L, M);
    const bool fast_ok = cg_fast_supported(depth, dim, spsize, tpsize);
    if (!fast_ok && (depth > CG_GEN_MAXDEPTH || spsize > 256 || tpsize > 256))
        CG_FAIL((cg_ctx*)nullptr, CG_ERR_UNSUPPORTED, "cg_create: depth=%d spsize=%d tpsize=%d exceeds the general path's limits "
                "(depth <= %d, widths <= 256)", depth, spsize, tpsize, CG_GEN_MAXDEPTH);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) CG_FAIL((cg_ctx*)nullptr, CG_ERR_HIP, "cg_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) CG_FAIL((cg_ctx*)nullptr, CG_ERR_ARG, "cg_create: device %d out of range [0,%d)", device, ndev);
    cg_ctx* c = new cg_ctx();
    c->device = device; c->n = n; c->dim = dim; c->depth = depth; c->hs = spsize; c->ht = tpsize; c->M = M; c->L = L;
    c->fast = fast_ok;
    cg_gen_model_init(c->gm, n, dim, depth, spsize, tpsize, L);
    c->gw = cg_gen_ws(c->gm);
    c->gwv = cg_gen_ws(c->gm, true);
    c->P = c->gm.nparam;
    memset(&c->lay, 0, sizeof(c->lay));
    if (fast_ok) {
#define CG_X(D, HS, HT) if (dim == D && spsize == HS && tpsize == HT) c->P = CgFast<D, HS, HT>::NPARAM;
        CG_FAST_CONFIGS(CG_X)
#undef CG_X
        c->lay = cg_fast_layout(n, dim, spsize, tpsize, true, spsize == 16 && tpsize == 16);
        // maximum size of the LDS-resident path: J (n d)^2 + the per-particle factors must fit 160 KiB.  Beyond it (n > ~64
        // at d = 2) the same model runs on the general path (HBM workspace), which provides every entry point.
        const size_t NN = (size_t)n * dim;
        if (sizeof(double) * (CG_TAB_DOUBLES + (size_t)c->lay.total + 2 * ((NN + 1) & ~(size_t)1) + 2) > 160 * 1024) {
            c->fast = false;
            c->P = c->gm.nparam;
        }
    }
    auto fail = [&](const char* what, hipError_t err) {
        g_last_error = std::string("cg_create: ") + what + ": " + hipGetErrorString(err);
        delete c; return CG_ERR_HIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipMalloc((void**)&c->d_theta, sizeof(double) * c->P)) != hipSuccess) return fail("hipMalloc theta", e);
    if ((e = hipMalloc((void**)&c->d_spk, sizeof(double) * (size_t)M * dim)) != hipSuccess) return fail("hipMalloc orbitals", e);
    if ((e = hipMalloc((void**)&c->d_accept, sizeof(unsigned long long))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void**)&c->d_tab, sizeof(double) * CG_TAB_DOUBLES)) != hipSuccess) return fail("hipMalloc tables", e);
    { double tabh[CG_TAB_DOUBLES]; cg_tab_fill(tabh);
      if ((e = hipMemcpy(c->d_tab, tabh, sizeof(tabh), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy tables", e); }
    std::vector<double> spk((size_t)M * dim);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);     // src/slater.py:14
    if ((e = hipMemcpy(c->d_spk, spk.data(), sizeof(double) * spk.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy orbitals", e);
    if ((e = hipMemset(c->d_accept, 0, sizeof(unsigned long long))) != hipSuccess) return fail("hipMemset", e);
    *out = c;
    return CG_OK;
}

void cg_destroy(cg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& ch : c->chunks) (void)hipFree(ch.p);
    if (c->ws) (void)hipFree(c->ws);
    if (c->bounce) (void)hipHostFree(c->bounce);
    if (c->d_scores) (void)hipFree(c->d_scores);
    if (c->d_theta) (void)hipFree(c->d_theta);
    if (c->d_spk) (void)hipFree(c->d_spk);
    if (c->d_tab) (void)hipFree(c->d_tab);
    if (c->d_G) (void)hipFree(c->d_G);
    if (c->d_gk) (void)hipFree(c->d_gk);
    if (c->d_accept) (void)hipFree(c->d_accept);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int cg_set_pointer_mode(cg_ctx* c, int mode) {
    if (!c) return CG_ERR_ARG;
    if (mode != CG_PTR_HOST && mode != CG_PTR_DEVICE) CG_FAIL(c, CG_ERR_ARG, "cg_set_pointer_mode: mode %d", mode);
    c->ptr_mode = mode; return CG_OK;
}
int cg_sync(cg_ctx* c) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    return CG_OK;
}
int cg_num_params(const cg_ctx* c) { return c ? c->P : CG_ERR_ARG; }

int cg_set_flow_params(cg_ctx* c, const double* theta) {
    if (!c || !theta) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemcpyAsync(c->d_theta, theta, sizeof(double) * c->P, hipMemcpyHostToDevice, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    c->have_theta = true;
    return CG_OK;
}

int cg_set_ewald(cg_ctx* c, double kappa, const int64_t* G, int nG, double rs) {
    if (!c || !G || nG < 1 || !(kappa > 0)) CG_FAIL(c, CG_ERR_ARG, "cg_set_ewald: bad argument");
    CG_HIP(c, hipSetDevice(c->device));
    const int D = c->dim;
    std::vector<int> g32((size_t)nG * D);
    std::vector<double> gk(nG);
    int gmax = 0;
    for (int g = 0; g < nG; ++g) {
        double g2 = 0;
        for (int a = 0; a < D; ++a) {
            int64_t v = G[(size_t)g * D + a];
            if (v > 4096 || v < -4096) CG_FAIL(c, CG_ERR_ARG, "cg_set_ewald: |G| component %lld too large", (long long)v);
            g32[(size_t)g * D + a] = (int)v; gmax = std::max(gmax, (int)std::llabs(v)); g2 += (double)v * (double)v;
        }
        if (g2 == 0) CG_FAIL(c, CG_ERR_ARG, "cg_set_ewald: G = 0 must not be in the list (src/potential.py:16)");
        const double gn = std::sqrt(g2);
        // src/potential.py:54-59
        gk[g] = (D == 3) ? std::exp(-CG_PI * CG_PI * g2 / (kappa * kappa)) / (CG_PI * g2) : std::erfc(CG_PI * gn / kappa) / gn;
    }
    c->g0 = (D == 3) ? -CG_PI / (kappa * kappa) : -2.0 * std::sqrt(CG_PI) / kappa;
    if (c->d_G) { (void)hipFree(c->d_G); c->d_G = nullptr; }
    if (c->d_gk) { (void)hipFree(c->d_gk); c->d_gk = nullptr; }
    CG_HIP(c, hipMalloc((void**)&c->d_G, sizeof(int) * g32.size()));
    CG_HIP(c, hipMalloc((void**)&c->d_gk, sizeof(double) * nG));
    CG_HIP(c, hipMemcpy(c->d_G, g32.data(), sizeof(int) * g32.size(), hipMemcpyHostToDevice));
    CG_HIP(c, hipMemcpy(c->d_gk, gk.data(), sizeof(double) * nG, hipMemcpyHostToDevice));
    c->kappa = kappa; c->rs = rs; c->nG = nG; c->Gmax = gmax; c->have_ewald = true;
    return CG_OK;
}

int cg_dev_alloc(cg_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMalloc(dptr, bytes ? bytes : 8));
    return CG_OK;
}
int cg_dev_free(cg_ctx* c, void* dptr) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    CG_HIP(c, hipFree(dptr));
    return CG_OK;
}
int cg_memcpy_h2d(cg_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    return CG_OK;
}
int cg_memcpy_d2h(cg_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    return CG_OK;
}
int cg_memset(cg_ctx* c, void* dst, int value, size_t bytes) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return CG_OK;
}
int cg_timer_start(cg_ctx* c) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipEventRecord(c->ev0, c->stream));
    return CG_OK;
}
int cg_timer_stop(cg_ctx* c, float* ms) {
    if (!c || !ms) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipEventRecord(c->ev1, c->stream));
    CG_HIP(c, hipEventSynchronize(c->ev1));
    CG_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return CG_OK;
}
int cg_set_block_threads(cg_ctx* c, int threads) {
    if (!c) return CG_ERR_ARG;
    if (threads != 0 && (threads < 64 || threads > 1024 || threads % 64)) CG_FAIL(c, CG_ERR_ARG, "cg_set_block_threads: %d is not 0 or a multiple of 64 in [64,1024]", threads);
    c->block_threads = threads; return CG_OK;
}
int cg_get_launch_info(cg_ctx* c, int64_t* info) {
    if (!c || !info) return CG_ERR_ARG;
    const int N = c->n * c->dim;
    info[0] = threads_of(c);
    info[1] = (int64_t)sizeof(double) * (CG_TAB_DOUBLES + c->lay.total + 2 * ((N + 1) & ~1) + 2);
    info[2] = c->cu_count; info[3] = c->P; info[4] = c->fast ? 1 : 0; info[5] = info[6] = info[7] = 0;
    return CG_OK;
}

static int check_ready(cg_ctx* c, const char* fn, int B) {
    if (!c) return CG_ERR_ARG;
    if (B < 0) CG_FAIL(c, CG_ERR_ARG, "%s: negative batch", fn);
    if (!c->have_theta) CG_FAIL(c, CG_ERR_STATE, "%s: cg_set_flow_params has not been called", fn);
    CG_HIP(c, hipSetDevice(c->device));
    return CG_OK;
}

static int ensure_ws(cg_ctx* c, size_t bytes) {
    if (bytes <= c->ws_cap) return CG_OK;
    CG_HIP(c, hipStreamSynchronize(c->stream));
    if (c->ws) { (void)hipFree(c->ws); c->ws = nullptr; c->ws_cap = 0; }
    CG_HIP(c, hipMalloc(&c->ws, bytes));
    c->ws_cap = bytes;
    return CG_OK;
}

static int run_logpsi(cg_ctx* c, const char* fn, const double* x, const int32_t* sidx, int B, int mode,
                      double* logphi, double* hld, double* logpsi_out, double* logp_out, double* z_out, double* J_out) {
    int rc = check_ready(c, fn, B); if (rc) return rc;
    if (B == 0) return CG_OK;
    if (!x || (mode == CG_MODE_LOGPSI && !sidx)) CG_FAIL(c, CG_ERR_ARG, "%s: NULL input", fn);
    const int n = c->n, N = n * c->dim;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "%s: arena", fn);
    Arg ax{(void*)x, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg as{(void*)sidx, nullptr, sizeof(int32_t) * (size_t)B * n, true, false};
    Arg a1{logphi, nullptr, sizeof(double) * 2 * (size_t)B, false, true};
    Arg a2{hld, nullptr, sizeof(double) * (size_t)B, false, true};
    Arg a3{logpsi_out, nullptr, sizeof(double) * 2 * (size_t)B, false, true};
    Arg a4{logp_out, nullptr, sizeof(double) * (size_t)B, false, true};
    Arg a5{z_out, nullptr, sizeof(double) * (size_t)B * N, false, true};
    Arg a6{J_out, nullptr, sizeof(double) * (size_t)B * N * N, false, true};
    Arg* all[] = {&ax, &as, &a1, &a2, &a3, &a4, &a5, &a6};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    if (!c->fast) {
        const int grid = std::min(B, c->cu_count * 4);
        if ((rc = ensure_ws(c, sizeof(double) * c->gw.total * grid))) return rc;
        hipLaunchKernelGGL(k_gen_logpsi, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 8), c->stream, c->gm, c->gw,
                           (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev,
                           (const int*)as.dev, B, mode, (double*)a1.dev, (double*)a2.dev, (double*)a3.dev, (double*)a4.dev,
                           (double*)a5.dev, (double*)a6.dev, (double*)c->ws);
        for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
        return finish(c);
    }
    const int nt = threads_of(c);
    const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + c->lay.total + ((N + 1) & ~1));
    const CgDev m = make_dev(c);
    bool launched = false;
#define CG_X(D, HS, HT)                                                                                              \
    if (!launched && c->dim == D && c->hs == HS && c->ht == HT) {                                                   \
        if (nt <= 256) {                                                                                             \
            if ((rc = set_lds(c, k_logpsi<D, HS, HT, 256>, lds))) return rc;                                         \
            hipLaunchKernelGGL((k_logpsi<D, HS, HT, 256>), dim3(B), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev, \
                               (const int*)as.dev, B, mode, (double*)a1.dev, (double*)a2.dev, (double*)a3.dev,       \
                               (double*)a4.dev, (double*)a5.dev, (double*)a6.dev);                                   \
        } else {                                                                                                     \
            if ((rc = set_lds(c, k_logpsi<D, HS, HT, 1024>, lds))) return rc;                                        \
            hipLaunchKernelGGL((k_logpsi<D, HS, HT, 1024>), dim3(B), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev, \
                               (const int*)as.dev, B, mode, (double*)a1.dev, (double*)a2.dev, (double*)a3.dev,       \
                               (double*)a4.dev, (double*)a5.dev, (double*)a6.dev);                                   \
        }                                                                                                            \
        launched = true;                                                                                             \
    }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    if (!launched) CG_FAIL(c, CG_ERR_UNSUPPORTED, "%s: configuration not instantiated", fn);
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}

int cg_flow_forward(cg_ctx* c, const double* x, int B, double* z) {
    if (c && !z) CG_FAIL(c, CG_ERR_ARG, "cg_flow_forward: z is NULL");
    return run_logpsi(c, "cg_flow_forward", x, nullptr, B, CG_MODE_FLOW, nullptr, nullptr, nullptr, nullptr, z, nullptr);
}
int cg_flow_jacobian(cg_ctx* c, const double* x, int B, double* J) {
    if (c && !J) CG_FAIL(c, CG_ERR_ARG, "cg_flow_jacobian: J is NULL");
    return run_logpsi(c, "cg_flow_jacobian", x, nullptr, B, CG_MODE_JAC, nullptr, nullptr, nullptr, nullptr, nullptr, J);
}
int cg_logpsi(cg_ctx* c, const double* x, const int32_t* sidx, int B, double* out) {
    if (c && !out) CG_FAIL(c, CG_ERR_ARG, "cg_logpsi: out is NULL");
    return run_logpsi(c, "cg_logpsi", x, sidx, B, CG_MODE_LOGPSI, nullptr, nullptr, out, nullptr, nullptr, nullptr);
}
int cg_logphi_logjacdet(cg_ctx* c, const double* x, const int32_t* sidx, int B, double* logphi, double* hld) {
    return run_logpsi(c, "cg_logphi_logjacdet", x, sidx, B, CG_MODE_LOGPSI, logphi, hld, nullptr, nullptr, nullptr, nullptr);
}
int cg_logp(cg_ctx* c, const double* x, const int32_t* sidx, int B, double* logp) {
    if (c && !logp) CG_FAIL(c, CG_ERR_ARG, "cg_logp: logp is NULL");
    return run_logpsi(c, "cg_logp", x, sidx, B, CG_MODE_LOGPSI, nullptr, nullptr, nullptr, logp, nullptr, nullptr);
}

int cg_mcmc(cg_ctx* c, double* x, const int32_t* sidx, int B, int mc_steps, double mc_stddev, uint64_t seed,
            uint64_t walker_offset, const double* noise, const double* unif, double* logp_out, int64_t* n_accept) {
    int rc = check_ready(c, "cg_mcmc", B); if (rc) return rc;
    if (mc_steps < 0) CG_FAIL(c, CG_ERR_ARG, "cg_mcmc: mc_steps < 0");
    if ((noise == nullptr) != (unif == nullptr)) CG_FAIL(c, CG_ERR_ARG, "cg_mcmc: noise and unif must both be given or both be NULL");
    if (n_accept) *n_accept = 0;
    if (B == 0) return CG_OK;
    if (!x || !sidx) CG_FAIL(c, CG_ERR_ARG, "cg_mcmc: NULL input");
    const int n = c->n, N = n * c->dim;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_mcmc: arena");
    Arg ax{x, nullptr, sizeof(double) * (size_t)B * N, true, true};
    Arg as{(void*)sidx, nullptr, sizeof(int32_t) * (size_t)B * n, true, false};
    Arg an{(void*)noise, nullptr, sizeof(double) * (size_t)mc_steps * B * N, true, false};
    Arg au{(void*)unif, nullptr, sizeof(double) * (size_t)mc_steps * B, true, false};
    Arg al{logp_out, nullptr, sizeof(double) * (size_t)B, false, true};
    Arg* all[] = {&ax, &as, &an, &au, &al};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    CG_HIP(c, hipMemsetAsync(c->d_accept, 0, sizeof(unsigned long long), c->stream));
    if (!c->fast) {
        const int grid = std::min(B, c->cu_count * 4);
        if ((rc = ensure_ws(c, sizeof(double) * c->gw.total * grid))) return rc;
        hipLaunchKernelGGL(k_gen_mcmc, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 8), c->stream, c->gm, c->gw,
                           (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (double*)ax.dev,
                           (const int*)as.dev, B, mc_steps, mc_stddev, seed, walker_offset, (const double*)an.dev,
                           (const double*)au.dev, (double*)al.dev, c->d_accept, (double*)c->ws);
        for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
        if ((rc = finish(c))) return rc;
        if (n_accept) return cg_mcmc_accepts(c, n_accept);
        return CG_OK;
    }
    const int nt = threads_of(c);
    const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + c->lay.total + 2 * ((N + 1) & ~1) + 2);
    const CgDev m = make_dev(c);
    bool launched = false;
#define CG_X(D, HS, HT, NS, NT)                                                                                     \
    if (!launched && c->dim == D && c->hs == HS && c->ht == HT && c->n == NS && nt == NT) {                        \
        if ((rc = set_lds(c, k_mcmc<D, HS, HT, NT, NS>, lds))) return rc;                                           \
        hipLaunchKernelGGL((k_mcmc<D, HS, HT, NT, NS>), dim3(B), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (double*)ax.dev,     \
                           (const int*)as.dev, B, mc_steps, mc_stddev, seed, walker_offset,                         \
                           (const double*)an.dev, (const double*)au.dev, (double*)al.dev, c->d_accept);             \
        launched = true;                                                                                            \
    }
    CG_MCMC_SPECIALS(CG_X)
#undef CG_X
#define CG_X(D, HS, HT)                                                                                             \
    if (!launched && c->dim == D && c->hs == HS && c->ht == HT) {                                                  \
        if (nt <= 256) {                                                                                            \
            if ((rc = set_lds(c, k_mcmc<D, HS, HT, 256>, lds))) return rc;                                          \
            hipLaunchKernelGGL((k_mcmc<D, HS, HT, 256>), dim3(B), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (double*)ax.dev,     \
                               (const int*)as.dev, B, mc_steps, mc_stddev, seed, walker_offset,                     \
                               (const double*)an.dev, (const double*)au.dev, (double*)al.dev, c->d_accept);         \
        } else {                                                                                                    \
            if ((rc = set_lds(c, k_mcmc<D, HS, HT, 1024>, lds))) return rc;                                         \
            hipLaunchKernelGGL((k_mcmc<D, HS, HT, 1024>), dim3(B), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (double*)ax.dev,    \
                               (const int*)as.dev, B, mc_steps, mc_stddev, seed, walker_offset,                     \
                               (const double*)an.dev, (const double*)au.dev, (double*)al.dev, c->d_accept);         \
        }                                                                                                           \
        launched = true;                                                                                            \
    }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    if (!launched) CG_FAIL(c, CG_ERR_UNSUPPORTED, "cg_mcmc: configuration not instantiated");
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    if ((rc = finish(c))) return rc;
    if (n_accept) return cg_mcmc_accepts(c, n_accept);
    return CG_OK;
}

int cg_mcmc_accepts(cg_ctx* c, int64_t* n_accept) {
    if (!c || !n_accept) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    unsigned long long v = 0;
    CG_HIP(c, hipMemcpyAsync(&v, c->d_accept, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    *n_accept = (int64_t)v;
    return CG_OK;
}

int cg_wrap(cg_ctx* c, double* x, int B) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (B == 0) return CG_OK;
    if (!x) CG_FAIL(c, CG_ERR_ARG, "cg_wrap: x is NULL");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_wrap: arena");
    const size_t cnt = (size_t)B * c->n * c->dim;
    Arg ax{x, nullptr, sizeof(double) * cnt, true, true};
    if ((rc = stage(c, ax))) return rc;
    hipLaunchKernelGGL(k_wrap, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, (double*)ax.dev, cnt, c->L);
    if ((rc = unstage(c, ax))) return rc;
    return finish(c);
}

int cg_ewald(cg_ctx* c, const double* x, int B, double* V) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (!c->have_ewald) CG_FAIL(c, CG_ERR_STATE, "cg_ewald: cg_set_ewald has not been called");
    if (B == 0) return CG_OK;
    if (!x || !V) CG_FAIL(c, CG_ERR_ARG, "cg_ewald: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_ewald: arena");
    const int n = c->n, D = c->dim, N = n * D;
    Arg ax{(void*)x, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg av{V, nullptr, sizeof(double) * (size_t)B, false, true};
    if ((rc = stage(c, ax)) || (rc = stage(c, av))) return rc;
    const int nt = 256;
    const size_t lds = sizeof(double) * (((N + 1) & ~1) + (size_t)N * (c->Gmax + 1) * 2 + nt);
    if (D == 2) {
        if ((rc = set_lds(c, k_ewald<2>, lds))) return rc;
        hipLaunchKernelGGL((k_ewald<2>), dim3(B), dim3(nt), lds, c->stream, (const double*)ax.dev, B, n, c->L, c->kappa,
                           c->rs, (const int*)c->d_G, (const double*)c->d_gk, c->nG, c->Gmax, c->g0, (double*)av.dev);
    } else {
        if ((rc = set_lds(c, k_ewald<3>, lds))) return rc;
        hipLaunchKernelGGL((k_ewald<3>), dim3(B), dim3(nt), lds, c->stream, (const double*)ax.dev, B, n, c->L, c->kappa,
                           c->rs, (const int*)c->d_G, (const double*)c->d_gk, c->nG, c->Gmax, c->g0, (double*)av.dev);
    }
    if ((rc = unstage(c, av))) return rc;
    return finish(c);
}


int cg_grad_laplacian(cg_ctx* c, const double* x, const int32_t* sidx, int B, int mode, const double* v,
                      double* grad, double* lap) {
    int rc = check_ready(c, "cg_grad_laplacian", B); if (rc) return rc;
    if (mode < 0 || mode > 2) CG_FAIL(c, CG_ERR_ARG, "cg_grad_laplacian: mode %d", mode);
    if (mode != CG_LAP_EXACT && !v) CG_FAIL(c, CG_ERR_ARG, "cg_grad_laplacian: Hutchinson modes need v");
    if (B == 0) return CG_OK;
    if (!x || !sidx || !grad || !lap) CG_FAIL(c, CG_ERR_ARG, "cg_grad_laplacian: NULL argument");
    const int n = c->n, N = n * c->dim;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_grad_laplacian: arena");
    Arg ax{(void*)x, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg as{(void*)sidx, nullptr, sizeof(int32_t) * (size_t)B * n, true, false};
    Arg av{(void*)v, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg ag{grad, nullptr, sizeof(double) * (size_t)B * N * 2, false, true};
    Arg al{lap, nullptr, sizeof(double) * (size_t)B * 2, false, true};
    Arg* all[] = {&ax, &as, &av, &ag, &al};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    const int nt = 256;
    const int grid = std::min(B, c->cu_count * 2 * CG_DERIV_WAVES);
    if (!c->fast) {
        if ((rc = ensure_ws(c, sizeof(double) * c->gw.total * grid))) return rc;
        hipLaunchKernelGGL(k_gen_grad_lap, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 256 + 16), c->stream, c->gm, c->gw,
                           (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev,
                           (const int*)as.dev, B, mode, (const double*)av.dev, (double*)ag.dev, (double*)al.dev, (double*)c->ws);
        for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
        return finish(c);
    }
    const CgDev m = make_dev(c);
    bool launched = false;
#define CG_X(D, HS, HT)                                                                                              \
    if (!launched && c->dim == D && c->hs == HS && c->ht == HT) {                                                   \
        const size_t wsw = CgDerivs<D, HS, HT>::ws_doubles(n);                                                      \
        const auto dl = CgDerivs<D, HS, HT>::layout(n, nt);                                                         \
        const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + CgDerivs<D, HS, HT>::lds_doubles(n, nt) + CgDerivs<D, HS, HT>::jet_lds_doubles(dl) + CgDerivs<D, HS, HT>::theta_lds_doubles(dl)); \
        if ((rc = ensure_ws(c, sizeof(double) * wsw * grid))) return rc;                                            \
        if (dl.jets_in_lds) {                                                                                       \
            if ((rc = set_lds(c, k_grad_lap<D, HS, HT, true>, lds))) return rc;                                     \
            hipLaunchKernelGGL((k_grad_lap<D, HS, HT, true>), dim3(grid), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev, \
                               (const int*)as.dev, B, mode, (const double*)av.dev, (double*)ag.dev, (double*)al.dev, \
                               (double*)c->ws, wsw, dl);                                                            \
        } else {                                                                                                    \
            if ((rc = set_lds(c, k_grad_lap<D, HS, HT, false>, lds))) return rc;                                    \
            hipLaunchKernelGGL((k_grad_lap<D, HS, HT, false>), dim3(grid), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev, \
                               (const int*)as.dev, B, mode, (const double*)av.dev, (double*)ag.dev, (double*)al.dev, \
                               (double*)c->ws, wsw, dl);                                                            \
        }                                                                                                           \
        launched = true;                                                                                            \
    }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    if (!launched) CG_FAIL(c, CG_ERR_UNSUPPORTED, "cg_grad_laplacian: configuration not instantiated");
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}

// Batch reductions over the resident score matrix, sliced over the batch so that the whole chip streams it (one slice
// per ~64 walkers, at most 64 slices), then summed in fixed order: mean over b (w_re == nullptr, count = 2P, scaled by 1/B)
// or the weighted sum of cg_scores_vjp (count = P).
static int score_reduce(cg_ctx* c, const double* S, const double* w_re, const double* w_im, int B, int count, double* out) {
    const int nsl = std::max(1, std::min(64, (B + 63) / 64)), chunk = (B + nsl - 1) / nsl;
    double* partial = (double*)arena_take(c, sizeof(double) * (size_t)nsl * count);
    if (!partial) CG_FAIL(c, CG_ERR_HIP, "score reduction: workspace allocation failed");
    if (w_re) hipLaunchKernelGGL(k_score_gemv, dim3((count + 63) / 64, nsl), dim3(256), 0, c->stream, S, w_re, w_im, B, count, chunk, partial);
    else hipLaunchKernelGGL(k_score_mean, dim3((count + 63) / 64, nsl), dim3(256), 0, c->stream, S, B, count, chunk, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3((count + 127) / 128), dim3(128), 0, c->stream, (const double*)partial, nsl, count, out);
    if (!w_re) hipLaunchKernelGGL(k_scale, dim3((count + 255) / 256), dim3(256), 0, c->stream, out, (size_t)count, 1.0 / (double)B);
    return CG_OK;
}

static int run_vjp(cg_ctx* c, const char* fn, const double* x, const int32_t* sidx, int B, const double* w_re,
                   const double* w_im, double* g_theta, double* score, double* fisher = nullptr, double* smean = nullptr,
                   bool keep_scores = false) {
    int rc = check_ready(c, fn, B); if (rc) return rc;
    const int n = c->n, N = n * c->dim, P = c->P;
    if (B == 0) {
        if (g_theta && c->ptr_mode == CG_PTR_HOST) memset(g_theta, 0, sizeof(double) * P);
        else if (g_theta) CG_HIP(c, hipMemsetAsync(g_theta, 0, sizeof(double) * P, c->stream));
        return CG_OK;
    }
    if (!x || !sidx) CG_FAIL(c, CG_ERR_ARG, "%s: NULL argument", fn);
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "%s: arena", fn);
    Arg ax{(void*)x, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg as{(void*)sidx, nullptr, sizeof(int32_t) * (size_t)B * n, true, false};
    Arg awr{(void*)w_re, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg awi{(void*)w_im, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg ag{g_theta, nullptr, sizeof(double) * (size_t)P, false, true};
    Arg asc{score, nullptr, sizeof(double) * (size_t)B * P * 2, false, true};
    Arg afi{fisher, nullptr, sizeof(double) * (size_t)P * P, false, true};
    Arg asm_{smean, nullptr, sizeof(double) * (size_t)P * 2, false, true};
    Arg* all[] = {&ax, &as, &awr, &awi, &ag, &asc, &afi, &asm_};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    if ((fisher || keep_scores) && !asc.dev) {        // the scores stay on the device, in the context's resident buffer
        if (c->scores_cap < asc.bytes) {
            if (c->d_scores) { CG_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_scores); c->d_scores = nullptr; c->scores_cap = 0; }
            if (hipMalloc((void**)&c->d_scores, asc.bytes) != hipSuccess)
                CG_FAIL(c, CG_ERR_HIP, "%s: %zu bytes for the per-sample scores could not be allocated", fn, asc.bytes);
            c->scores_cap = asc.bytes;
        }
        asc.dev = c->d_scores; c->scores_B = B;
    }
    const int nt = 256;
    const int grid = std::min(B, c->cu_count * 2 * CG_DERIV_WAVES);
    double* partial = g_theta ? (double*)arena_take(c, sizeof(double) * (size_t)grid * P) : nullptr;
    if (g_theta && !partial) CG_FAIL(c, CG_ERR_HIP, "%s: workspace allocation failed", fn);
    const CgDev m = make_dev(c);
    bool launched = false;
    if (!c->fast) {       // any depth / widths: dual-number reverse passes of the primal flow (cg_generic.hpp)
        if ((rc = ensure_ws(c, sizeof(double) * c->gwv.total * grid))) return rc;
        hipLaunchKernelGGL(k_gen_param_vjp, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 8), c->stream, c->gm, c->gwv,
                           (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev,
                           (const int*)as.dev, B, (const double*)awr.dev, (const double*)awi.dev, partial, (double*)asc.dev, (double*)c->ws);
        launched = true;
    }
#define CG_X(D, HS, HT)                                                                                               \
    if (!launched && c->dim == D && c->hs == HS && c->ht == HT) {                                                    \
        const size_t wsw = CgDerivs<D, HS, HT>::ws_doubles(n);                                                       \
        const auto dl = CgDerivs<D, HS, HT>::layout(n, nt);                                                          \
        const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + CgDerivs<D, HS, HT>::lds_doubles(n, nt) + CgDerivs<D, HS, HT>::vjp_lds_doubles(dl)); \
        if ((rc = ensure_ws(c, sizeof(double) * wsw * grid))) return rc;                                             \
        if ((rc = set_lds(c, k_param_vjp<D, HS, HT>, lds))) return rc;                                               \
        hipLaunchKernelGGL((k_param_vjp<D, HS, HT>), dim3(grid), dim3(nt), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, (const double*)ax.dev, \
                           (const int*)as.dev, B, (const double*)awr.dev, (const double*)awi.dev, partial,           \
                           (double*)asc.dev, (double*)c->ws, wsw, dl);                                               \
        launched = true;                                                                                             \
    }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    if (!launched) CG_FAIL(c, CG_ERR_UNSUPPORTED, "%s: configuration not instantiated", fn);
    if (g_theta)
        hipLaunchKernelGGL(k_reduce_rows, dim3((P + 127) / 128), dim3(128), 0, c->stream, (const double*)partial, grid, P, (double*)ag.dev);
    if (fisher) {
        const int tiles = (P + 15) / 16;
        hipLaunchKernelGGL(k_fisher, dim3((tiles * tiles + 3) / 4), dim3(256), 0, c->stream, (const double*)asc.dev, B, P, (double*)afi.dev);
        if (smean && (rc = score_reduce(c, (const double*)asc.dev, nullptr, nullptr, B, 2 * P, (double*)asm_.dev))) return rc;
    }
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}

int cg_param_vjp(cg_ctx* c, const double* x, const int32_t* sidx, int B, const double* w_re, const double* w_im, double* g_theta) {
    if (c && !g_theta) CG_FAIL(c, CG_ERR_ARG, "cg_param_vjp: g_theta is NULL");
    if (c && (!w_re || !w_im) && B > 0) CG_FAIL(c, CG_ERR_ARG, "cg_param_vjp: weights are NULL");
    return run_vjp(c, "cg_param_vjp", x, sidx, B, w_re, w_im, g_theta, nullptr);
}
int cg_quantum_score(cg_ctx* c, const double* x, const int32_t* sidx, int B, double* score) {
    if (c && !score) CG_FAIL(c, CG_ERR_ARG, "cg_quantum_score: score is NULL");
    return run_vjp(c, "cg_quantum_score", x, sidx, B, nullptr, nullptr, nullptr, score);
}
/* Resident per-sample scores: computed once per (x, state_idx, theta), then reused for the theta-VJPs of the loss
 * (weights 2 Re/Im E_clip / B and 2 / B, main.py:278) and for the Fisher matrix -- 2 reverse sweeps instead of 6. */
int cg_scores_compute(cg_ctx* c, const double* x, const int32_t* sidx, int B) {
    if (c && B <= 0) CG_FAIL(c, CG_ERR_ARG, "cg_scores_compute: empty batch");
    return run_vjp(c, "cg_scores_compute", x, sidx, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true);
}
int cg_scores_vjp(cg_ctx* c, const double* w_re, const double* w_im, double* g_theta) {
    if (!c) return CG_ERR_ARG;
    if (!c->d_scores || c->scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_scores_vjp: cg_scores_compute has not been called");
    if (!w_re || !w_im || !g_theta) CG_FAIL(c, CG_ERR_ARG, "cg_scores_vjp: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    int rc; const int B = c->scores_B, P = c->P;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_scores_vjp: arena");
    Arg awr{(void*)w_re, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg awi{(void*)w_im, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg ag{g_theta, nullptr, sizeof(double) * (size_t)P, false, true};
    Arg* all[] = {&awr, &awi, &ag};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    if ((rc = score_reduce(c, (const double*)c->d_scores, (const double*)awr.dev, (const double*)awi.dev, B, P, (double*)ag.dev))) return rc;
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}
int cg_scores_fisher(cg_ctx* c, double* fisher, double* score_mean) {
    if (!c) return CG_ERR_ARG;
    if (!c->d_scores || c->scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_scores_fisher: cg_scores_compute has not been called");
    if (!fisher || !score_mean) CG_FAIL(c, CG_ERR_ARG, "cg_scores_fisher: NULL output");
    CG_HIP(c, hipSetDevice(c->device));
    int rc; const int B = c->scores_B, P = c->P;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_scores_fisher: arena");
    Arg afi{fisher, nullptr, sizeof(double) * (size_t)P * P, false, true};
    Arg asm_{score_mean, nullptr, sizeof(double) * (size_t)P * 2, false, true};
    Arg* all[] = {&afi, &asm_};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    const int tiles = (P + 15) / 16;
    hipLaunchKernelGGL(k_fisher, dim3((tiles * tiles + 3) / 4), dim3(256), 0, c->stream, (const double*)c->d_scores, B, P, (double*)afi.dev);
    if ((rc = score_reduce(c, (const double*)c->d_scores, nullptr, nullptr, B, 2 * P, (double*)asm_.dev))) return rc;
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}
int cg_quantum_fisher(cg_ctx* c, const double* x, const int32_t* sidx, int B, double* fisher, double* score_mean) {
    if (c && (!fisher || !score_mean)) CG_FAIL(c, CG_ERR_ARG, "cg_quantum_fisher: NULL output");
    if (c && B <= 0) CG_FAIL(c, CG_ERR_ARG, "cg_quantum_fisher: empty batch");
    return run_vjp(c, "cg_quantum_fisher", x, sidx, B, nullptr, nullptr, nullptr, nullptr, fisher, score_mean);
}


/* which: 0 = v_fma_f64 (VALU), 1 = v_mfma_f64_16x16x4_f64.  Returns achieved TFLOP/s (HIP-event timed). */
int cg_microbench_fp64(cg_ctx* c, int which, double* tflops) {
    if (!c || !tflops) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_microbench_fp64: arena");
    double* out = (double*)arena_take(c, 64);
    const int blocks = c->cu_count * 8, iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        CG_HIP(c, hipEventRecord(c->ev0, c->stream));
        if (which == 0) hipLaunchKernelGGL(k_peak_fma64, dim3(blocks), dim3(256), 0, c->stream, out, iters, 0.999999, 1e-9);
        else hipLaunchKernelGGL(k_peak_mfma64, dim3(blocks), dim3(256), 0, c->stream, out, iters, 0.5, 0.25);
        CG_HIP(c, hipEventRecord(c->ev1, c->stream));
        CG_HIP(c, hipEventSynchronize(c->ev1));
    }
    float ms = 0; CG_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    const double flops = which == 0 ? (double)blocks * 256 * iters * 16 * 2 : (double)blocks * 4 /*waves*/ * iters * 4 * (16.0 * 16 * 4 * 2);
    *tflops = flops / (ms * 1e-3) / 1e12;
    return CG_OK;
}

#if defined(CG_STAMPS)
/* diagnostic builds only: read (and clear) the per-phase cycle counters of cg_common.hpp */
int cg_debug_stamps(cg_ctx* c, unsigned long long* out64, int clear) {
    if (!c || !out64) return CG_ERR_ARG;
    CG_HIP(c, hipStreamSynchronize(c->stream));
    CG_HIP(c, hipMemcpyFromSymbol(out64, HIP_SYMBOL(cg_stamp_acc), sizeof(unsigned long long) * 64));
    if (clear) { unsigned long long z[64] = {0}; CG_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(cg_stamp_acc), z, sizeof(z))); }
    return CG_OK;
}
#endif

int cg_scale_dev(cg_ctx* c, double* buf, size_t count, double s) {
    if (!c) return CG_ERR_ARG;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, buf, count, s);
    CG_HIP(c, hipGetLastError());
    return CG_OK;
}

}  // extern "C"

#include "cg_comm.inc"
#include "cg_solve.inc"
