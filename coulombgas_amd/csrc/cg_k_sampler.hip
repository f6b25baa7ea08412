// cg_k_sampler.hip -- log Psi and the fused Metropolis chain of the depth-2 fast path (cg_flow_fast.hpp) + their entry points.
#include "cg_host.hpp"
#include "cg_rng.hpp"

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

extern "C" {

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
        if ((rc = cg_gen_run_logpsi(c, (const double*)ax.dev, (const int*)as.dev, B, mode, (double*)a1.dev, (double*)a2.dev, (double*)a3.dev,
                                    (double*)a4.dev, (double*)a5.dev, (double*)a6.dev))) return rc;
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
        if ((rc = cg_gen_run_mcmc(c, (double*)ax.dev, (const int*)as.dev, B, mc_steps, mc_stddev, seed, walker_offset, (const double*)an.dev,
                                  (const double*)au.dev, (double*)al.dev))) return rc;
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

}  // extern "C"
