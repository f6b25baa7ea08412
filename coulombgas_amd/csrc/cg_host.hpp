// cg_host.hpp -- host-side state shared by the translation units of libcoulombgas_hip.so (cg_hip.hip: context, Ewald,
// solver, communicator; cg_k_sampler.hip: log Psi / Metropolis kernels; cg_k_derivs.hip: grad / Laplacian, theta-VJP,
// scores; cg_k_generic.hip: general-depth kernels).  The library is split so that the units compile in parallel.
#pragma once
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
#include "cg_generic.hpp"
#include "cg_van.hpp"

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

// minimum waves per SIMD the derivative kernels are register-allocated for
#ifndef CG_DERIV_WAVES
#define CG_DERIV_WAVES 3
#endif
#define CG_DERIV_WAVES_OF(D) ((D) == 2 ? CG_DERIV_WAVES : (CG_DERIV_WAVES < 2 ? CG_DERIV_WAVES : 2))

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
extern thread_local std::string g_last_error;      // defined in cg_hip.hip

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
    double* d_rate = nullptr;             // device scalar of cg_mcmc_accept_rate
    // staging arena for host-pointer mode + internal workspaces
    std::vector<Chunk> chunks;
    size_t cur = 0, off = 0;
    // persistent workspace (derivative kernels)
    void* ws = nullptr; size_t ws_cap = 0;
    double* d_scores = nullptr; size_t scores_cap = 0; int scores_B = 0;     // resident per-sample scores (cg_scores_*)
    void* d_lay = nullptr; int lay_tag = 0;     // small device copy of a kernel's layout struct (k_grad_lap2_scores: read where used instead of
                                                // living in scalar registers through the whole kernel); lay_tag says whose
    CgVanModel van; double* d_van = nullptr; double* d_van_sp = nullptr; bool have_van = false;   // density-matrix Transformer (cg_van_*)
    double* d_van_scores = nullptr; size_t van_scores_cap = 0; int van_scores_B = 0;              // resident classical scores
    std::string err;
    hipStream_t stream2 = nullptr;                     // look-ahead stream of the blocked Cholesky (bulk trailing updates)
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
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

// workgroup size of the sampler kernels (tools/sampler_threads_sweep.py, round 4, walker-steps/s at the production batch sizes): n = 22: 128 threads
// 8.53 M against 8.37 M with 256, n = 23: 6.19 M against 7.81 M (the concurrent LU pair wants its helper waves); n = 41 ... 56: 512 threads on
// the 512-thread instantiation (256 registers per lane; these sizes used to run the 1024-thread one with 128: n = 45 1.95 -> 2.58 M, n = 53
// 1.60 -> 2.00 M; 768 threads on the latter: 2.02 / 1.63 M); n = 29, 49, 57 run kernels specialised on the size as well
static int auto_threads(int n) {
    if (n <= 16) return 64;
    if (n <= 22) return 128;
    if (n <= 40) return 256;
    if (n <= 64) return 512;
    return 1024;
}
static int threads_of(const cg_ctx* c) { return c->block_threads > 0 ? c->block_threads : auto_threads(c->n); }

static CgDev make_dev(const cg_ctx* c) {
    CgDev m; m.n = c->n; m.L = c->L; m.lay = c->lay;
    return m;
}

// launch-shape overrides for tuning runs (unset: the built-in choice)
static inline int cg_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

template <class K>
static int set_lds(cg_ctx* c, K kernel, size_t bytes) {
    if (bytes > 160 * 1024) CG_FAIL(c, CG_ERR_UNSUPPORTED, "workgroup needs %zu bytes of LDS (> 160 KiB): n too large for the LDS-resident path", bytes);
    if (bytes > 48 * 1024) CG_HIP(c, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
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

// general-depth launches (cg_k_generic.hip); every array argument is a device pointer
int cg_gen_run_logpsi(cg_ctx* c, const double* x, const int* sidx, int B, int mode, double* logphi, double* hld, double* logpsi_out,
                      double* logp_out, double* z_out, double* J_out);
int cg_gen_run_mcmc(cg_ctx* c, double* x, const int* sidx, int B, int steps, double stddev, uint64_t seed, uint64_t walker_offset,
                    const double* noise, const double* unif, double* logp_out);
int cg_gen_run_param_vjp(cg_ctx* c, int grid, const double* x, const int* sidx, int B, const double* w_re, const double* w_im,
                         double* partial, double* score);
int cg_gen_run_grad_lap(cg_ctx* c, int grid, const double* x, const int* sidx, int B, int mode, const double* v, double* grad, double* lap);
extern "C" int cg_fisher_real_nr(cg_ctx* c, const double* S_dev, int B, int P, double* F_dev);
