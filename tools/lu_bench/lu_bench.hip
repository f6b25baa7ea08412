// lu_bench.hip -- standalone timing / correctness harness for the sampler's determinant pair (development tool, not product):
// every workgroup factors the same pair (real N x N Jacobian-like matrix, complex n x n Slater-like matrix) out of LDS, exactly as
// CgFast::logpsi calls it, REPS times; prints cycles per call (s_memtime, workgroup wall) and the error against a long-double LU.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 [-DCG_STAMPS] [-DLU_VARIANT=k] -o lu_bench lu_bench.hip
//   ./lu_bench n nthreads reps kind       kind 0: J = I + small (no exchanges), 1: dense random J (exchanges)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <complex>
#include <algorithm>
#if defined(LU_TRACE)
__device__ unsigned long long lu_trace[2][32];
#define CG_LU_TRACE(chain, k) if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) lu_trace[chain][k] = __builtin_readcyclecounter();
#endif
#include "../../coulombgas_amd/csrc/cg_common.hpp"
#include "../../coulombgas_amd/csrc/cg_linalg.hpp"

#ifndef LU_MAXT
#define LU_MAXT 512
#endif
#ifndef LU_VARIANT
#define LU_VARIANT 0
#endif

__global__ void __launch_bounds__(LU_MAXT, 1) k_lu(const double* __restrict__ tab, const double* __restrict__ Ag, const double* __restrict__ Cg,
                                                int N, int n, int reps, double* out, unsigned long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    CG_STAMP_INIT
    double* A = lds;
    double* C = A + ((N * N + 1) & ~1);
    double* res = C + 2 * n * n;
    unsigned long long tot = 0;
    double lr = 0, la = 0, ar = 0;
    for (int r = 0; r < reps; ++r) {
        for (int e = b.tid; e < N * N; e += b.nthr) A[e] = Ag[e];
        for (int e = b.tid; e < 2 * n * n; e += b.nthr) C[e] = Cg[e];
        b.sync();
        const unsigned long long t0 = __builtin_readcyclecounter();
#if LU_VARIANT == 0
        cg_blocked_lu_dual(b, A, N, N, C, n, n, res, lr, la, ar);
#else
        cg_blocked_lu_dual2(b, A, N, N, C, n, n, res, lr, la, ar);
#endif
        const unsigned long long t1 = __builtin_readcyclecounter();
        tot += t1 - t0;
    }
    if (b.tid == 0) {
        out[3 * blockIdx.x] = lr; out[3 * blockIdx.x + 1] = la; out[3 * blockIdx.x + 2] = ar;
        cyc[blockIdx.x] = tot;
    }
    CG_STAMP_FLUSH
#endif
}

typedef long double ld;
static void ref_real(std::vector<double> A, int N, double& logabs) {
    std::vector<ld> a(A.begin(), A.end());
    ld s = 0;
    for (int k = 0; k < N; ++k) {
        int p = k; ld mx = fabsl(a[k * N + k]);
        for (int i = k + 1; i < N; ++i) if (fabsl(a[i * N + k]) > mx) { mx = fabsl(a[i * N + k]); p = i; }
        if (p != k) for (int j = 0; j < N; ++j) std::swap(a[k * N + j], a[p * N + j]);
        s += logl(fabsl(a[k * N + k]));
        for (int i = k + 1; i < N; ++i) {
            const ld l = a[i * N + k] / a[k * N + k];
            for (int j = k + 1; j < N; ++j) a[i * N + j] -= l * a[k * N + j];
        }
    }
    logabs = (double)s;
}
static void ref_cplx(const std::vector<double>& C, int n, double& logabs, double& arg) {
    typedef std::complex<ld> cl;
    std::vector<cl> a(n * n);
    for (int e = 0; e < n * n; ++e) a[e] = cl(C[2 * e], C[2 * e + 1]);
    ld s = 0, ph = 0;
    for (int k = 0; k < n; ++k) {
        int p = k; ld mx = std::abs(a[k * n + k]);
        for (int i = k + 1; i < n; ++i) if (std::abs(a[i * n + k]) > mx) { mx = std::abs(a[i * n + k]); p = i; }
        if (p != k) { for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]); ph += M_PIl; }
        s += logl(std::abs(a[k * n + k])); ph += std::arg(a[k * n + k]);
        for (int i = k + 1; i < n; ++i) {
            const cl l = a[i * n + k] / a[k * n + k];
            for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
        }
    }
    logabs = (double)s; arg = (double)remainderl(ph, 2 * M_PIl);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 57, nt = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int kind = argc > 4 ? atoi(argv[4]) : 0;
    const int nN = argc > 5 ? atoi(argv[5]) : n;      // size (in particles) of the real matrix if different
    const int N = 2 * nN, WG = nt == 64 ? 2048 : 256;   // one-wave workgroups: eight per CU, two per SIMD, as k_mcmc runs them at n = 13
    std::vector<double> A(N * N), C(2 * n * n), tab(CG_TAB_DOUBLES);
    cg_tab_fill(tab.data());
    srand(1234 + kind);
    auto rnd = [] { return (rand() + 0.5) / (RAND_MAX + 1.0); };
    auto gauss = [&] { return sqrt(-2 * log(rnd())) * cos(2 * M_PI * rnd()); };
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) A[i * N + j] = kind == 0 ? (i == j ? 1.0 : 0.0) + 0.3 * gauss() / sqrt((double)N) : gauss();
    // Slater-like: D_ij = exp(i k_j . z_i), k_j on a twisted integer lattice, z uniform in the box
    {
        std::vector<double> z(2 * n), k(2 * n);
        for (int i = 0; i < 2 * n; ++i) z[i] = rnd();
        int c = 0;
        for (int r = 0; c < n; ++r)
            for (int kx = -r; kx <= r && c < n; ++kx)
                for (int ky = -r; ky <= r && c < n; ++ky)
                    if (std::max(abs(kx), abs(ky)) == r) { k[2 * c] = 2 * M_PI * (kx + 0.25); k[2 * c + 1] = 2 * M_PI * (ky + 0.25); ++c; }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) { const double ph = k[2 * j] * z[2 * i] + k[2 * j + 1] * z[2 * i + 1]; C[2 * (i * n + j)] = cos(ph); C[2 * (i * n + j) + 1] = sin(ph); }
    }
    double rl, cl_, ca; ref_real(A, N, rl); ref_cplx(C, n, cl_, ca);
    double *dA, *dC, *dtab, *dout; unsigned long long* dcyc;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dC, C.size() * 8)); CK(hipMalloc(&dtab, tab.size() * 8));
    CK(hipMalloc(&dout, WG * 3 * 8)); CK(hipMalloc(&dcyc, WG * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
    const size_t lds = 8 * (size_t)(CG_TAB_DOUBLES + ((N * N + 1) & ~1) + 2 * n * n + 512);
    CK(hipFuncSetAttribute((const void*)k_lu, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1)); float ms = 0;
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(ev0, 0));
        hipLaunchKernelGGL(k_lu, dim3(WG), dim3(nt), lds, 0, dtab, dA, dC, N, n, reps, dout, dcyc);
        CK(hipEventRecord(ev1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, ev0, ev1));
    }
    std::vector<double> out(WG * 3); std::vector<unsigned long long> cyc(WG);
    CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(cyc.data(), dcyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    double e0 = 0, e1 = 0, e2 = 0;
    for (int w = 0; w < WG; ++w) {
        e0 = std::max(e0, fabs(out[3 * w] - rl)); e1 = std::max(e1, fabs(out[3 * w + 1] - cl_));
        e2 = std::max(e2, fabs(remainder(out[3 * w + 2] - ca, 2 * M_PI)));
    }
    printf("variant %d n=%d N=%d nt=%d kind=%d (%d WGs, kernel %.3f ms): %.0f cycles per dual LU (median WG; min %.0f max %.0f)   err logabsJ %.2e logabsD %.2e argD %.2e  (ref %.6f %.6f %.6f)\n",
           LU_VARIANT, n, N, nt, kind, WG, ms, (double)cyc[WG / 2] / reps, (double)cyc[0] / reps, (double)cyc[WG - 1] / reps, e0, e1, e2, rl, cl_, ca);
#if defined(LU_TRACE)
    { unsigned long long tr[2][32]; CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(lu_trace), sizeof(tr)));
      for (int c = 0; c < 2; ++c) { printf("  chain %d panel durations:", c); const int np = ((c ? n : N) + 7) / 8; for (int k = 0; k + 1 < np; ++k) printf(" %lld", (long long)(tr[c][k + 1] - tr[c][k])); printf("\n"); } }
#endif
#if defined(CG_STAMPS)
    unsigned long long st[64];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(cg_stamp_acc), sizeof(st)));
    for (int k = 0; k < 32; ++k) if (st[k]) printf("  stamp %2d: %10.0f cycles per call per WG\n", k, (double)(long long)st[k] / (3.0 * reps * WG));
#endif
    return (e0 < 1e-9 && e1 < 1e-9 && e2 < 1e-9) ? 0 : 2;
}
