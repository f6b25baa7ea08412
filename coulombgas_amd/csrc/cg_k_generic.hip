// cg_k_generic.hip -- general-depth path (cg_generic.hpp): any FermiNet depth / widths; workspace in HBM.
#include "cg_host.hpp"
#include "cg_rng.hpp"

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

// ---- launches (called from the entry points in cg_k_sampler.hip / cg_k_derivs.hip) -------------------------------------
int cg_gen_run_logpsi(cg_ctx* c, const double* x, const int* sidx, int B, int mode, double* logphi, double* hld, double* logpsi_out,
                      double* logp_out, double* z_out, double* J_out) {
    const int grid = std::min(B, c->cu_count * 4);
    int rc;
    if ((rc = ensure_ws(c, sizeof(double) * c->gw.total * grid))) return rc;
    hipLaunchKernelGGL(k_gen_logpsi, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 8), c->stream, c->gm, c->gw,
                       (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, x, sidx, B, mode, logphi, hld,
                       logpsi_out, logp_out, z_out, J_out, (double*)c->ws);
    return CG_OK;
}
int cg_gen_run_mcmc(cg_ctx* c, double* x, const int* sidx, int B, int steps, double stddev, uint64_t seed, uint64_t walker_offset,
                    const double* noise, const double* unif, double* logp_out) {
    const int grid = std::min(B, c->cu_count * 4);
    int rc;
    if ((rc = ensure_ws(c, sizeof(double) * c->gw.total * grid))) return rc;
    hipLaunchKernelGGL(k_gen_mcmc, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 8), c->stream, c->gm, c->gw,
                       (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, x, sidx, B, steps, stddev, seed,
                       walker_offset, noise, unif, logp_out, c->d_accept, (double*)c->ws);
    return CG_OK;
}
int cg_gen_run_param_vjp(cg_ctx* c, int grid, const double* x, const int* sidx, int B, const double* w_re, const double* w_im,
                         double* partial, double* score) {
    int rc;
    if ((rc = ensure_ws(c, sizeof(double) * c->gwv.total * grid))) return rc;
    hipLaunchKernelGGL(k_gen_param_vjp, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 8), c->stream, c->gm, c->gwv,
                       (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, x, sidx, B, w_re, w_im, partial,
                       score, (double*)c->ws);
    return CG_OK;
}
int cg_gen_run_grad_lap(cg_ctx* c, int grid, const double* x, const int* sidx, int B, int mode, const double* v, double* grad, double* lap) {
    int rc;
    if ((rc = ensure_ws(c, sizeof(double) * c->gw.total * grid))) return rc;
    hipLaunchKernelGGL(k_gen_grad_lap, dim3(grid), dim3(256), sizeof(double) * (CG_TAB_DOUBLES + 256 + 16), c->stream, c->gm, c->gw,
                       (const double*)c->d_theta, (const double*)c->d_spk, (const double*)c->d_tab, x, sidx, B, mode, v, grad, lap,
                       (double*)c->ws);
    return CG_OK;
}
