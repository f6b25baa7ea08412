// cg_k_sampler_a.hip -- log Psi / Metropolis kernels of the (2, 16, 16) flow (every shipped run) + the entry points of the
// sampler family.  The other instantiations are compiled in cg_k_sampler_b.hip.
#include "cg_host.hpp"
#include "cg_rng.hpp"

#define CG_UNIT_CONFIGS(X) CG_FAST_CONFIGS_A(X)
#define CG_UNIT_SPECIALS(X) CG_MCMC_SPECIALS(X)
#define CG_UNIT_NAME(f) cg_sampler_a_##f
#include "cg_k_sampler.inc"

int cg_sampler_b_logpsi(cg_ctx* c, int nt, size_t lds, const CgDev& m, const double* x, const int* sidx, int B, int mode,
                        double* logphi, double* hld, double* logpsi_out, double* logp_out, double* z_out, double* J_out);
int cg_sampler_b_mcmc(cg_ctx* c, int nt, size_t lds, const CgDev& m, double* x, const int* sidx, int B, int mc_steps, double mc_stddev,
                      uint64_t seed, uint64_t walker_offset, const double* noise, const double* unif, double* logp_out);

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
    if ((rc = cg_sampler_a_logpsi(c, nt, lds, m, (const double*)ax.dev, (const int*)as.dev, B, mode, (double*)a1.dev, (double*)a2.dev,
                                  (double*)a3.dev, (double*)a4.dev, (double*)a5.dev, (double*)a6.dev)) < 0) return rc;
    if (rc == 0 && (rc = cg_sampler_b_logpsi(c, nt, lds, m, (const double*)ax.dev, (const int*)as.dev, B, mode, (double*)a1.dev, (double*)a2.dev,
                                             (double*)a3.dev, (double*)a4.dev, (double*)a5.dev, (double*)a6.dev)) < 0) return rc;
    launched = rc == 1;
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
    const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + c->lay.total + 3 * ((N + 1) & ~1) + 2);   // + x, proposal, flag, k-vectors
    const CgDev m = make_dev(c);
    bool launched = false;
    if ((rc = cg_sampler_a_mcmc(c, nt, lds, m, (double*)ax.dev, (const int*)as.dev, B, mc_steps, mc_stddev, seed, walker_offset,
                                (const double*)an.dev, (const double*)au.dev, (double*)al.dev)) < 0) return rc;
    if (rc == 0 && (rc = cg_sampler_b_mcmc(c, nt, lds, m, (double*)ax.dev, (const int*)as.dev, B, mc_steps, mc_stddev, seed, walker_offset,
                                           (const double*)an.dev, (const double*)au.dev, (double*)al.dev)) < 0) return rc;
    launched = rc == 1;
    if (!launched) CG_FAIL(c, CG_ERR_UNSUPPORTED, "cg_mcmc: configuration not instantiated");
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    if ((rc = finish(c))) return rc;
    if (n_accept) return cg_mcmc_accepts(c, n_accept);
    return CG_OK;
}

}  // extern "C"

#if defined(CG_STAMPS)
CG_STAMP_READER(cg_debug_stamps)      /* diagnostic builds only (tools/stamps*.py): the per-phase cycle counters of this unit's kernels */
#endif
