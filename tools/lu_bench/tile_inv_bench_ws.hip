// tile_inv_bench.hip -- development harness: the register-tiled, panel-blocked Gauss-Jordan inverses of the derivative kernels' set-up at
// n > 16 (cg_inverse_panel_real on N = 2n, then cg_inverse_panel_complex on n, one workgroup per CU as k_grad_lap2 / k_param_vjp run them).
// -DINV_TRACE: cycles of threads 0 and 256 per phase of the real inverse (panel, barrier, pivot rows, barrier, update).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tile_inv_bench tile_inv_bench.hip;  ./tile_inv_bench n threads reps [1] [hard 0|1: no diagonal dominance in the real matrix]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
__device__ unsigned long long g_ph[2][8];
#if defined(INV_TRACE)
#define CG_INV_T_DECL unsigned long long tprev_ = 0;
#define CG_INV_T(i) { if ((b.tid == 0 || b.tid == 256) && blockIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); if (i > 0) g_ph[b.tid >> 8][i] += t_ - tprev_; tprev_ = t_; } }
#endif
#include "../../coulombgas_amd/csrc/cg_common.hpp"
#include "../../coulombgas_amd/csrc/cg_linalg.hpp"
template <int V>
__global__ void __launch_bounds__(512, 1) k_tinv(const double* __restrict__ Ag, const double* __restrict__ Cg, int N, int n, int reps,
                                                  double* outA, double* outC, double* wsg, unsigned long long* cyc) {
#if defined(__HIP_DEVICE_COMPILE__)
    double* lds = cg_dyn_lds;                       // J^-1 and the staging area in LDS, J / D / D^-1 in the workspace slot (as at n = 57)
    double* st = lds + N * N;
    double* ws = wsg + (size_t)blockIdx.x * (2 * N * N + 4 * n * n); double* Ai = ws + N * N + 4 * n * n;
    double* A = ws; double* C = A + N * N; double* Ci = C + 2 * n * n;
    const CgBlk b0{(int)threadIdx.x, (int)blockDim.x};
    for (int e = b0.tid; e < N * N; e += b0.nthr) A[e] = Ag[e];
    for (int e = b0.tid; e < 2 * n * n; e += b0.nthr) C[e] = Cg[e];
    __syncthreads();
    unsigned long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        int tid_ = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_));                 // (opaque per repetition: or the address arithmetic of every tile shape is hoisted out of the loop)
        const CgBlk b{tid_, (int)blockDim.x};
        const unsigned long long t0 = __builtin_readcyclecounter();
        cg_inverse_panel_real(b, A, N, N, Ai, N, st);
        cg_inverse_panel_complex(b, C, n, n, Ci, n, st);
        tot += __builtin_readcyclecounter() - t0;
    }
    if (blockIdx.x == 0) {
        for (int e = b0.tid; e < N * N; e += b0.nthr) outA[e] = Ai[e];
        for (int e = b0.tid; e < 2 * n * n; e += b0.nthr) outC[e] = Ci[e];
    }
    if (b0.tid == 0) cyc[blockIdx.x] = tot;
#endif
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 57, nt = argc > 2 ? atoi(argv[2]) : 512, reps = argc > 3 ? atoi(argv[3]) : 10, WG = 256,
              var = argc > 4 ? atoi(argv[4]) : 1, hard = argc > 5 ? atoi(argv[5]) : 0;
    const int N = 2 * n;
    std::vector<double> A(N * N), C(2 * n * n);
    srand(7);
    auto rnd = [] { return (rand() + 0.5) / (RAND_MAX + 1.0); };
    auto gauss = [&] { return sqrt(-2 * log(rnd())) * cos(2 * M_PI * rnd()); };
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) A[i * N + j] = hard ? gauss() : (i == j ? 1.0 : 0.0) + 0.3 * gauss() / sqrt((double)N);
    for (int e = 0; e < n * n; ++e) { const double ph = 2 * M_PI * rnd(); C[2 * e] = cos(ph); C[2 * e + 1] = sin(ph); }
    double *dA, *dC, *oA, *oC, *ws; unsigned long long* dcyc;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dC, C.size() * 8)); CK(hipMalloc(&oA, A.size() * 8)); CK(hipMalloc(&oC, C.size() * 8)); CK(hipMalloc(&dcyc, WG * 8));
    CK(hipMalloc(&ws, (size_t)WG * (2 * N * N + 4 * n * n) * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
    const size_t need = cg_inv_panel_scratch(N, n, nt);
    (void)var;
    if (!need) { printf("shape not served\n"); return 1; }
    const size_t lds = 8 * ((size_t)N * N + std::max<size_t>(9 * N + 128, need));
    auto kern = k_tinv<1>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL(kern, dim3(WG), dim3(nt), lds, 0, dA, dC, N, n, reps, oA, oC, ws, dcyc); CK(hipDeviceSynchronize()); }
    std::vector<double> Ai(N * N), Ci(2 * n * n); std::vector<unsigned long long> cyc(WG);
    CK(hipMemcpy(Ai.data(), oA, Ai.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Ci.data(), oC, Ci.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cyc.data(), dcyc, WG * 8, hipMemcpyDeviceToHost));
    double er = 0, ec = 0;
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < N; ++k) s += Ai[i * N + k] * A[k * N + j]; er = std::max(er, fabs(s - (i == j))); }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double sr = 0, si = 0; for (int k = 0; k < n; ++k) { const double ar = Ci[2 * (i * n + k)], ai = Ci[2 * (i * n + k) + 1], br = C[2 * (k * n + j)], bi = C[2 * (k * n + j) + 1]; sr += ar * br - ai * bi; si += ar * bi + ai * br; } ec = std::max(ec, std::max(fabs(sr - (i == j)), fabs(si))); }
    { unsigned long long ph[2][8]; CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof(ph)));
      for (int w = 0; w < 2; ++w) { printf("  thread %d (real, all launches): ", w * 256); for (int i = 1; i < 6; ++i) printf(" ph%d %.0f", i, (double)ph[w][i] / (3.0 * reps)); printf("\n"); } }
    std::sort(cyc.begin(), cyc.end());
    printf("%s inverses n=%d N=%d nt=%d: %.0f cycles per pair (median WG) = %.0f per column   |A^-1 A - I| %.2e  |D^-1 D - I| %.2e\n",
           "panel", n, N, nt, (double)cyc[WG / 2] / reps, (double)cyc[WG / 2] / reps / (N + n), er, ec);
    return (er < 1e-9 && ec < 1e-9) ? 0 : 2;
}
