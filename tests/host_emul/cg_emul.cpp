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
