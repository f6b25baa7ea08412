// tests/host_emul/cg_emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the *same* per-walker device code as the HIP library (coulombgas_amd/csrc/*.hpp) for
// the host with a 1-thread workgroup shim, so that the kernel arithmetic can be compared with
// the oracle in the GPU-less build container (pytest -m "not gpu").  It is never loaded by the
// coulombgas_amd package: the product path is the HIP library only.
#include <vector>
#include <cstring>
#include <algorithm>
#include <cmath>
#include "../../coulombgas_amd/csrc/cg_common.hpp"
#include "../../coulombgas_amd/csrc/cg_linalg.hpp"
#include "../../coulombgas_amd/csrc/cg_flow_fast.hpp"
#include "../../coulombgas_amd/csrc/cg_dispatch.hpp"

template <int D, int HS, int HT>
static void emu_logpsi_t(int n, double L, const double* theta, const double* sp_indices, const int* sidx,
                         const double* x, int B, double* logphi, double* hld, double* z_out, double* J_out, int M) {
    using F = CgFast<D, HS, HT>;
    CgFastLds o = cg_fast_layout(n, D, HS, HT, false);
    std::vector<double> lds(o.total + 8), xs(n * D), spk((size_t)M * D);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    for (int w = 0; w < B; ++w) {
        memcpy(xs.data(), x + (size_t)w * n * D, sizeof(double) * n * D);
        F::primal(b, theta, xs.data(), n, L, lds.data(), o);
        F::jacobian(b, theta, n, L, lds.data(), o);
        if (z_out) memcpy(z_out + (size_t)w * n * D, lds.data() + o.z, sizeof(double) * n * D);
        if (J_out) memcpy(J_out + (size_t)w * n * D * n * D, lds.data() + o.J, sizeof(double) * n * D * n * D);
        double re, im, h;
        F::logpsi(b, theta, xs.data(), spk.data(), sidx + (size_t)w * n, n, L, lds.data(), o, re, im, h);
        logphi[2 * w] = re; logphi[2 * w + 1] = im; hld[w] = h;
    }
}

extern "C" int emu_logpsi(int n, int dim, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                          const int* sidx, const double* x, int B, double* logphi, double* hld, double* z_out, double* J_out) {
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) { emu_logpsi_t<D, HS, HT>(n, L, theta, sp_indices, sidx, x, B, logphi, hld, z_out, J_out, M); return 0; }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return -1;
}

// ---- derivative kernels (cg_derivs.hpp) on the host shim ----
#include "../../coulombgas_amd/csrc/cg_derivs.hpp"

// ---- grad / Laplacian (cg_lap.hpp): reverse sweep + forward Laplacian + jet pass(es); lds_budget (doubles) selects
// which blocks of the kernel's memory would live in LDS on the GPU (here it only changes the layout) ----
#include "../../coulombgas_amd/csrc/cg_lap.hpp"
template <int D, int HS, int HT>
static void emu_gradlap_t(int n, double L, const double* theta, const double* sp_indices, int M, const int* sidx,
                           const double* x, int B, int mode, const double* v, double* grad, double* lap, long lds_budget) {
    using G = CgLap<D, HS, HT>;
    const auto lay = G::layout(n, 1, mode, (size_t)lds_budget);
    std::vector<double> ws(lay.ws_total + 8), lds(lay.lds_total + 8), spk((size_t)M * D);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    const double* th = theta;
    if (lay.th_lds) { memcpy(lds.data() + lay.th, theta, sizeof(double) * G::NP); th = lds.data() + lay.th; }
    for (int w = 0; w < B; ++w) {
        if (lay.all_lds)
            G::template grad_laplacian<true>(b, th, x + (size_t)w * n * D, spk.data(), sidx + (size_t)w * n, n, L, mode,
                                             v ? v + (size_t)w * n * D : nullptr, grad + (size_t)w * n * D * 2, lap + 2 * w, lds.data(), ws.data(), lay);
        else
            G::template grad_laplacian<false>(b, th, x + (size_t)w * n * D, spk.data(), sidx + (size_t)w * n, n, L, mode,
                                              v ? v + (size_t)w * n * D : nullptr, grad + (size_t)w * n * D * 2, lap + 2 * w, lds.data(), ws.data(), lay);
    }
}
extern "C" int emu_grad_laplacian(int n, int dim, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                                   const int* sidx, const double* x, int B, int mode, const double* v, double* grad, double* lap, long lds_budget) {
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) { emu_gradlap_t<D, HS, HT>(n, L, theta, sp_indices, M, sidx, x, B, mode, v, grad, lap, lds_budget); return 0; }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return -1;
}

template <int D, int HS, int HT>
static void emu_vjp_t(int n, double L, const double* theta, const double* sp_indices, int M, const int* sidx,
                      const double* x, int B, const double* w_re, const double* w_im, double* g, double* score) {
    using G = CgDerivs<D, HS, HT>;
    const auto lay = G::layout(n, 1);              // same LDS scratch decisions as the GPU launch
    std::vector<double> ws(G::ws_doubles(n) + 8), lds(G::lds_doubles(n, 1) + G::vjp_lds_doubles(lay) + 8), spk((size_t)M * D);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    if (g) for (int e = 0; e < G::NP; ++e) g[e] = 0.0;
    for (int w = 0; w < B; ++w)
        G::param_vjp(b, theta, x + (size_t)w * n * D, spk.data(), sidx + (size_t)w * n, n, L, w_re ? w_re[w] : 1.0,
                     w_im ? w_im[w] : 0.0, g, score ? score + (size_t)w * G::NP * 2 : nullptr, ws.data(), lds.data(), lay);
}
extern "C" int emu_param_vjp(int n, int dim, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                             const int* sidx, const double* x, int B, const double* w_re, const double* w_im, double* g, double* score) {
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) { emu_vjp_t<D, HS, HT>(n, L, theta, sp_indices, M, sidx, x, B, w_re, w_im, g, score); return 0; }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return -1;
}

// ---- per-sample scores, second generation (cg_score.hpp) on the host shim ----
#include "../../coulombgas_amd/csrc/cg_score.hpp"
template <int D, int HS, int HT>
static int emu_scores2_t(int n, double L, const double* theta, const double* sp_indices, int M, const int* sidx,
                         const double* x, int B, double* score) {
    using G = CgScore<D, HS, HT>;
    const auto lay = G::layout(n, 1, (size_t)1 << 30);
    std::vector<double> lds(lay.total + 8), spk((size_t)M * D);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    memcpy(lds.data() + lay.th, theta, sizeof(double) * G::NP);
    for (int w = 0; w < B; ++w)
        G::scores(b, lds.data() + lay.th, x + (size_t)w * n * D, spk.data(), sidx + (size_t)w * n, n, L, score + (size_t)w * G::NP * 2, lds.data(), lay);
    return 0;
}
extern "C" int emu_scores2(int n, int dim, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                           const int* sidx, const double* x, int B, double* score) {
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) { if (HS != 16 || HT != 16) return -2; return emu_scores2_t<D, HS, HT>(n, L, theta, sp_indices, M, sidx, x, B, score); }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return -1;
}

// ---- sampler and Ewald on the host shim (supplied noise only) ----
#include "../../coulombgas_amd/csrc/cg_ewald.hpp"

template <int D, int HS, int HT>
static long emu_mcmc_t(int n, double L, const double* theta, const double* sp_indices, int M, const int* sidx, double* x, int B,
                       int steps, double stddev, const double* noise, const double* unif, double* logp_out) {
    using F = CgFast<D, HS, HT>;
    const int N = n * D;
    CgFastLds o = cg_fast_layout(n, D, HS, HT, true);       // the aliased layout the GPU kernel uses
    std::vector<double> lds(o.total + 8), xc(N), xp(N), spk((size_t)M * D);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    long total = 0;
    for (int w = 0; w < B; ++w) {
        memcpy(xc.data(), x + (size_t)w * N, sizeof(double) * N);
        double re, im, h;
        F::logpsi(b, theta, xc.data(), spk.data(), sidx + (size_t)w * n, n, L, lds.data(), o, re, im, h);
        double logp = 2.0 * (re + h);
        for (int s = 0; s < steps; ++s) {
            for (int e = 0; e < N; ++e) xp[e] = xc[e] + stddev * noise[((size_t)s * B + w) * N + e];
            F::logpsi(b, theta, xp.data(), spk.data(), sidx + (size_t)w * n, n, L, lds.data(), o, re, im, h);
            const double lp = 2.0 * (re + h);
            if (unif[(size_t)s * B + w] < exp(lp - logp)) { xc = xp; logp = lp; ++total; }
        }
        memcpy(x + (size_t)w * N, xc.data(), sizeof(double) * N);
        if (logp_out) logp_out[w] = logp;
    }
    return total;
}
extern "C" long emu_mcmc(int n, int dim, int hs, int ht, double L, const double* theta, const double* sp_indices, int M, const int* sidx,
                         double* x, int B, int steps, double stddev, const double* noise, const double* unif, double* logp_out) {
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) return emu_mcmc_t<D, HS, HT>(n, L, theta, sp_indices, M, sidx, x, B, steps, stddev, noise, unif, logp_out);
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return -1;
}

extern "C" int emu_ewald(int n, int dim, double L, double kappa, double rs, const long* G, int nG, const double* x, int B, double* V) {
    std::vector<int> g32((size_t)nG * dim); std::vector<double> gk(nG);
    int gmax = 0;
    for (int g = 0; g < nG; ++g) {
        double g2 = 0;
        for (int a = 0; a < dim; ++a) { g32[g * dim + a] = (int)G[g * dim + a]; gmax = std::max(gmax, std::abs((int)G[g * dim + a])); g2 += (double)G[g * dim + a] * G[g * dim + a]; }
        gk[g] = dim == 3 ? exp(-CG_PI * CG_PI * g2 / (kappa * kappa)) / (CG_PI * g2) : erfc(CG_PI * sqrt(g2) / kappa) / sqrt(g2);
    }
    const double g0 = dim == 3 ? -CG_PI / (kappa * kappa) : -2.0 * sqrt(CG_PI) / kappa;
    std::vector<double> lds((size_t)n * dim * (gmax + 1) * 2 + 16);
    CgBlk b{0, 1};
    for (int w = 0; w < B; ++w)
        V[w] = dim == 2 ? cg_ewald_walker<2>(b, x + (size_t)w * n * dim, n, L, kappa, rs, g32.data(), gk.data(), nG, gmax, g0, lds.data())
                        : cg_ewald_walker<3>(b, x + (size_t)w * n * dim, n, L, kappa, rs, g32.data(), gk.data(), nG, gmax, g0, lds.data());
    return 0;
}

// ---- general-depth path (cg_generic.hpp) on the host shim ----
#include "../../coulombgas_amd/csrc/cg_generic.hpp"
extern "C" int emu_gen_logpsi(int n, int dim, int depth, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                              const int* sidx, const double* x, int B, double* out3, double* z_out, double* J_out) {
    CgGenModel m; cg_gen_model_init(m, n, dim, depth, hs, ht, L);
    CgGenWs w = cg_gen_ws(m);
    std::vector<double> ws(w.total + 8), spk((size_t)M * dim);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    const int N = n * dim;
    for (int q = 0; q < B; ++q) {
        if (z_out || J_out) {
            CgGen<double>::flow(b, m, theta, x + (size_t)q * N, ws.data() + w.da, true);
            if (z_out) memcpy(z_out + (size_t)q * N, ws.data() + w.da + m.o_z, sizeof(double) * N);
            if (J_out) memcpy(J_out + (size_t)q * N * N, ws.data() + w.da + m.o_J, sizeof(double) * N * N);
        }
        CgGenK::logpsi(b, m, w, theta, spk.data(), sidx + (size_t)q * n, x + (size_t)q * N, ws.data(), out3[3 * q], out3[3 * q + 1], out3[3 * q + 2]);
    }
    return m.nparam;
}
extern "C" int emu_gen_grad_laplacian(int n, int dim, int depth, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                                      const int* sidx, const double* x, int B, int mode, const double* v, double* grad, double* lap) {
    CgGenModel m; cg_gen_model_init(m, n, dim, depth, hs, ht, L);
    CgGenWs w = cg_gen_ws(m);
    std::vector<double> ws(w.total + 8), lds(64), spk((size_t)M * dim);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    const int N = n * dim;
    for (int q = 0; q < B; ++q)
        CgGenK::grad_laplacian(b, m, w, theta, spk.data(), sidx + (size_t)q * n, x + (size_t)q * N, mode, v ? v + (size_t)q * N : nullptr,
                               grad + (size_t)q * N * 2, lap + 2 * q, ws.data(), lds.data());
    return 0;
}

extern "C" int emu_gen_param_vjp(int n, int dim, int depth, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                                 const int* sidx, const double* x, int B, const double* w_re, const double* w_im, double* g_theta /*+=*/,
                                 double* score /* B x P x 2, nullable */) {
    CgGenModel m; cg_gen_model_init(m, n, dim, depth, hs, ht, L);
    CgGenWs w = cg_gen_ws(m, true);
    std::vector<double> ws(w.total + 8), spk((size_t)M * dim);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    const int N = n * dim;
    for (int q = 0; q < B; ++q)
        CgGenK::param_vjp(b, m, w, theta, spk.data(), sidx + (size_t)q * n, x + (size_t)q * N, w_re ? w_re[q] : 1.0, w_im ? w_im[q] : 0.0,
                          g_theta, score ? score + (size_t)q * m.nparam * 2 : nullptr, ws.data());
    return m.nparam;
}
