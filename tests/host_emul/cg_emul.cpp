// tests/host_emul/cg_emul.cpp -- TEST INFRASTRUCTURE ONLY.
// Compiles the *same* per-walker device code as the HIP library (coulombgas_amd/csrc/*.hpp) for
// the host with a 1-thread workgroup shim, so that the kernel arithmetic can be compared with
// the oracle in the GPU-less build container (pytest -m "not gpu").  It is never loaded by the
// coulombgas_amd package: the product path is the HIP library only.
#include <vector>
#include <cstring>
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

template <int D, int HS, int HT>
static void emu_gradlap_t(int n, double L, const double* theta, const double* sp_indices, int M, const int* sidx,
                          const double* x, int B, int mode, const double* v, double* grad, double* lap) {
    using G = CgDerivs<D, HS, HT>;
    std::vector<double> ws(G::ws_doubles(n) + 8), lds(G::lds_doubles(n, 1) + 8), spk((size_t)M * D);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    for (int w = 0; w < B; ++w)
        G::grad_laplacian(b, theta, x + (size_t)w * n * D, spk.data(), sidx + (size_t)w * n, n, L, mode,
                          v ? v + (size_t)w * n * D : nullptr, grad + (size_t)w * n * D * 2, lap + 2 * w, ws.data(), lds.data());
}
extern "C" int emu_grad_laplacian(int n, int dim, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                                  const int* sidx, const double* x, int B, int mode, const double* v, double* grad, double* lap) {
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) { emu_gradlap_t<D, HS, HT>(n, L, theta, sp_indices, M, sidx, x, B, mode, v, grad, lap); return 0; }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return -1;
}

template <int D, int HS, int HT>
static void emu_vjp_t(int n, double L, const double* theta, const double* sp_indices, int M, const int* sidx,
                      const double* x, int B, const double* w_re, const double* w_im, double* g, double* score) {
    using G = CgDerivs<D, HS, HT>;
    std::vector<double> ws(G::ws_doubles(n) + 8), lds(G::lds_doubles(n, 1) + 8), spk((size_t)M * D);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);
    CgBlk b{0, 1};
    if (g) for (int e = 0; e < G::NP; ++e) g[e] = 0.0;
    for (int w = 0; w < B; ++w)
        G::param_vjp(b, theta, x + (size_t)w * n * D, spk.data(), sidx + (size_t)w * n, n, L, w_re ? w_re[w] : 1.0,
                     w_im ? w_im[w] : 0.0, g, score ? score + (size_t)w * G::NP * 2 : nullptr, ws.data(), lds.data());
}
extern "C" int emu_param_vjp(int n, int dim, int hs, int ht, double L, const double* theta, const double* sp_indices, int M,
                             const int* sidx, const double* x, int B, const double* w_re, const double* w_im, double* g, double* score) {
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) { emu_vjp_t<D, HS, HT>(n, L, theta, sp_indices, M, sidx, x, B, w_re, w_im, g, score); return 0; }
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return -1;
}
