// cg_common.hpp -- execution-context shim + small math helpers shared by every kernel.
//
// All per-walker device code is written "workgroup-cooperatively": a CgBlk carries
// (tid, nthr), loops are `for (e = tid; e < N; e += nthr)` and phases are separated by
// blk.sync().  Under hipcc this is a real gfx950 workgroup (wave64); under a plain host
// compiler (tests/host_emul, test infrastructure only) the same source runs with
// nthr == 1 and sync() is a no-op, which lets the kernel arithmetic be checked against
// the oracle in the GPU-less build container.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CG_HD __host__ __device__ __forceinline__
#define CG_DEV __device__
#define CG_DEVI __device__ __forceinline__
#define CG_DEVN __device__ __attribute__((noinline))
#else
#define CG_HD inline
#define CG_DEV
#define CG_DEVI inline
#define CG_DEVN inline
#endif

// 1 in device code; 0 in the host builds of these headers (tests/host_emul: ONE thread walks every work item of the "workgroup", so
// per-wave partial sums live in a single row and wave-level code paths are replaced by their scalar statement).  Branches on it are
// ordinary `if` / `if constexpr`: both sides are compiled everywhere, and #if is left to the few places that name a device intrinsic.
#if defined(__HIP_DEVICE_COMPILE__)
#define CG_ON_DEVICE 1
#else
#define CG_ON_DEVICE 0
#endif
struct CgBlk {
    int tid, nthr;
    CG_DEVI void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
        __syncthreads();
#endif
    }
    CG_DEVI int waves() const { return CG_ON_DEVICE ? nthr >> 6 : 1; }      // rows of the per-wave partial arrays
};

#define CG_PI 3.14159265358979323846264338327950288

// Diagnostic builds only (-DCG_STAMPS, tools/stamps.py): lane 0 of every wave charges the s_memtime cycles between
// consecutive stamps to a per-phase device counter (CG_STAMP(k) ends phase k and starts phase k+1).  Compiles to
// nothing in the product build.
#if defined(CG_STAMPS) && defined(__HIPCC__)
static __device__ unsigned long long cg_stamp_acc[64];   // one copy per translation unit: see CG_STAMP_READER
__shared__ unsigned long long cg_stamp_lds[32];       // per-workgroup accumulation (no global contention)
// defines `int NAME(cg_ctx*, unsigned long long* out64, int clear)`: read (and clear) the counters of THIS translation unit
#define CG_STAMP_READER(NAME)                                                                                      \
    extern "C" int NAME(cg_ctx* c, unsigned long long* out64, int clear) {                                         \
        if (!c || !out64) return CG_ERR_ARG;                                                                       \
        CG_HIP(c, hipStreamSynchronize(c->stream));                                                                \
        CG_HIP(c, hipMemcpyFromSymbol(out64, HIP_SYMBOL(cg_stamp_acc), sizeof(unsigned long long) * 64));          \
        if (clear) { unsigned long long z[64] = {0}; CG_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(cg_stamp_acc), z, sizeof(z))); } \
        return CG_OK;                                                                                              \
    }
static __device__ __forceinline__ void cg_stamp_at(int end_k, int start_k) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long t = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        if (end_k >= 0) atomicAdd(&cg_stamp_lds[end_k], t);
        if (start_k >= 0) atomicAdd(&cg_stamp_lds[start_k], 0ull - t);
    }
#endif
}
static __device__ __forceinline__ void cg_stamp_init() {
    if (threadIdx.x < 32) cg_stamp_lds[threadIdx.x] = 0;
    __syncthreads();
}
static __device__ __forceinline__ void cg_stamp_flush() {
    __syncthreads();
    if (threadIdx.x < 32 && cg_stamp_lds[threadIdx.x]) atomicAdd(&cg_stamp_acc[threadIdx.x], cg_stamp_lds[threadIdx.x]);
}
#define CG_STAMP_INIT cg_stamp_init();
#define CG_STAMP_FLUSH cg_stamp_flush();
#define CG_STAMP_START(k) cg_stamp_at(-1, k);
#define CG_STAMP(k) cg_stamp_at(k, (k) + 1);
#define CG_STAMP_END(k) cg_stamp_at(k, -1);
#else
#define CG_STAMP_INIT
#define CG_STAMP_FLUSH
#define CG_STAMP_START(k)
#define CG_STAMP(k)
#define CG_STAMP_END(k)
#endif

// ---- rarely executed libm calls, kept OUT of line on the GPU --------------------------------------------------------
// log / exp / atan2 / sincos are called a handful of times per log Psi evaluation (LU epilogues, Slater phases, the
// accept test, Box-Muller).  Inlined, each drags 10-20 fp64 literal constants into the kernel body; the compiler hoists
// their materialisation out of the Metropolis loop and then spills them (they were most of the scratch traffic).
// As out-of-line leaf functions the constants live only inside the callee.
// ONLY the depth-2 sampler kernels (k_logpsi / k_mcmc) call these: inside the spill-heavy derivative kernels hipcc
// (ROCm 7.2) miscompiled the calls (d = 3 jets: the direction counter was lost across them), so every other path keeps
// the inlined libm call (`ool` arguments below default to false).
#if defined(__HIPCC__) && defined(CG_NO_OUTLINE)      /* diagnostic builds */
#define CG_OUTLINE __host__ __device__ __forceinline__
#elif defined(__HIPCC__)
#define CG_OUTLINE __host__ __device__ __attribute__((noinline))
#else
#define CG_OUTLINE inline
#endif
struct CgSinCos { double s, c; };
#if defined(__HIPCC__)
static CG_OUTLINE double cg_log_ool(double x) { return log(x); }
static CG_OUTLINE double cg_exp_ool(double x) { return exp(x); }
static CG_OUTLINE double cg_atan2_ool(double y, double x) { return atan2(y, x); }
static CG_OUTLINE CgSinCos cg_sincos_ool(double a) { CgSinCos r; sincos(a, &r.s, &r.c); return r; }
#else
static inline double cg_log_ool(double x) { return log(x); }
static inline double cg_exp_ool(double x) { return exp(x); }
static inline double cg_atan2_ool(double y, double x) { return atan2(y, x); }
static inline CgSinCos cg_sincos_ool(double a) { CgSinCos r; r.s = sin(a); r.c = cos(a); return r; }
#endif

// ---- fp64 transcendentals specialised for the flow's activations ---------------------------------------------
// gfx950 has no f64 exp/log instructions.  The generic ocml routines carry range / special-case handling these
// call sites do not need, and a plain polynomial evaluation costs ~65 VALU instructions per softplus.  The versions
// below use two small tables (1.25 KB, staged in LDS by every kernel; plain memory on the host shim):
//   exp2t[j] = 2^(j/32)                                   j = 0..31
//   logt[i]  = { 1/c_i rounded, -log(1/c_i rounded) }      c_i = 1 + (i + 1/2)/64,  i = 0..63
// e^x (x <= 0):  x = (32 m + j) ln2/32 + r, |r| <= ln2/64        -> 2^m exp2t[j] P6(r)
// log w, 1/w (w in [1,2)):  r = w/c_i - 1, |r| <= 1/128          -> logt[i][1] + log1p(r) (degree 7), (1/c_i)(1-r)(1+r^2)(1+r^4)
// Accuracy ~1-2 ulp (tests compare with libm through the oracle; the table is generated in long double).
#define CG_TAB_DOUBLES 160
static inline void cg_tab_fill(double* t) {
    for (int j = 0; j < 32; ++j) t[j] = (double)exp2l((long double)j / 32.0L);
    for (int i = 0; i < 64; ++i) {
        const long double c = 1.0L + ((long double)i + 0.5L) / 64.0L;
        const double inv = (double)(1.0L / c);
        t[32 + 2 * i] = inv;
        t[32 + 2 * i + 1] = (double)(-logl((long double)inv));
    }
}
#if defined(__HIPCC__)
extern __shared__ __attribute__((aligned(16))) double cg_dyn_lds[];   // every kernel keeps the table in its first CG_TAB_DOUBLES of LDS (16-byte aligned: b128 accesses behind it)
#define CG_TAB (cg_dyn_lds)
#else
#if !defined(__HIPCC__)
static inline const double* cg_host_tab() {
    static double t[CG_TAB_DOUBLES]; static bool init = false;
    if (!init) { cg_tab_fill(t); init = true; }
    return t;
}
#define CG_TAB (cg_host_tab())
#endif
#endif

// e^x for x <= 0
CG_DEVI double cg_exp_nonpos(double x) {
    x = fmax(x, -746.0);                                  // below this e^x underflows to 0 anyway
    const double kf = rint(x * 46.166241308446828384);    // 32 / ln 2
    double r = fma(-kf, 2.16608493924982901946e-02, x);   // ln2/32 hi
    r = fma(-kf, 7.24702129326968612006e-19, r);          // ln2/32 lo
    const int ki = (int)kf;
    const double tj = CG_TAB[ki & 31];
    double p = 1.0 / 720.0;
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(tj * p, ki >> 5);
}
// w in [1, 2]: returns log(w) and 1/w
CG_DEVI void cg_log_rcp_12(double w, double& lg, double& rc) {
    const bool two = w >= 2.0;                            // w = 1 + e^{-|u|} = 2 exactly when u = 0
    const double m = two ? 1.0 : w;
#if defined(__HIPCC__)
    const int hi = __double2hiint(m);
#else
    long long bits; memcpy(&bits, &m, 8); const int hi = (int)(bits >> 32);
#endif
    const int i = (hi >> 14) & 63;                        // top 6 mantissa bits
    const double inv = CG_TAB[32 + 2 * i], lc = CG_TAB[32 + 2 * i + 1];
    const double r = fma(m, inv, -1.0);                   // |r| <= 1/128
    double p = 1.0 / 7.0;
    p = fma(p, r, -1.0 / 6.0);
    p = fma(p, r, 0.2);
    p = fma(p, r, -0.25);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    const double l = fma(p, r, lc);                       // log c_i + log1p(r)
    const double r2 = r * r;
    double t = 1.0 - r;
    t = fma(t, r2, t);
    t = fma(t, r2 * r2, t);
    const double q = inv * t;                             // 1/m
    lg = two ? 0.693147180559945309417232121458 : l;
    rc = two ? 0.5 : q;
}
// softplus(u) = log(1 + e^u) = max(u,0) + log1p(e^{-|u|}); sigmoid from the same exponential.
// (jax.nn.softplus == logaddexp(u, 0): no large-u cut-off, reference src/flow.py:45-52)
CG_DEVI void softplus_sigmoid(double u, double& sp, double& sg) {
    const double e = cg_exp_nonpos(-fabs(u));
    const double w = 1.0 + e;
    double lg, r; cg_log_rcp_12(w, lg, r);
    // log1p(e) = log(w) + (e - (w - 1)) / w   (restores the bits of e lost in forming w)
    sp = fmax(u, 0.0) + fma(e - (w - 1.0), r, lg);
    sg = (u >= 0.0) ? r : e * r;
}
CG_DEVI double sigmoid_only(double u) {
    const double e = cg_exp_nonpos(-fabs(u));
    const double w = 1.0 + e;                             // in [1, 2]
#if defined(__HIP_DEVICE_COMPILE__)
    // 1/w from v_rcp_f64 + two Newton steps (no table look-up, no w = 2 special case): 5 instructions instead of 15
    double r = __builtin_amdgcn_rcp(w);
    r = fma(r, fma(-w, r, 1.0), r);
    r = fma(r, fma(-w, r, 1.0), r);
#else
    const double r = 1.0 / w;
#endif
    return (u >= 0.0) ? r : e * r;
}
// log(w) only, w in [1, 2]  (same table as cg_log_rcp_12, without the reciprocal)
CG_DEVI double cg_log_12(double w) {
    const bool two = w >= 2.0;
    const double m = two ? 1.0 : w;
#if defined(__HIPCC__)
    const int hi = __double2hiint(m);
#else
    long long bits; memcpy(&bits, &m, 8); const int hi = (int)(bits >> 32);
#endif
    const int i = (hi >> 14) & 63;
    const double inv = CG_TAB[32 + 2 * i], lc = CG_TAB[32 + 2 * i + 1];
    const double r = fma(m, inv, -1.0);
    double p = 1.0 / 7.0;
    p = fma(p, r, -1.0 / 6.0);
    p = fma(p, r, 0.2);
    p = fma(p, r, -0.25);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    const double l = fma(p, r, lc);
    return two ? 0.693147180559945309417232121458 : l;
}
// log(w) for w in [1, 2^1000): exponent split off, mantissa through the [1, 2) table routine
CG_DEVI double cg_log_ge1(double w) {
#if defined(__HIPCC__)
    const int hi = __double2hiint(w);
    const int ex = (hi >> 20) - 1023;
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(w));
#else
    int ex; double m = frexp(w, &ex); m *= 2.0; ex -= 1;
#endif
    return fma((double)ex, 0.693147180559945309417232121458, cg_log_12(m));
}
// log(w) for a positive normal w (any exponent), -inf for w = 0
CG_DEVI double cg_log_pos(double w) { return w > 0.0 ? cg_log_ge1(w) : -INFINITY; }
CG_DEVI double softplus_only(double u) {
    const double e = cg_exp_nonpos(-fabs(u));
    const double w = 1.0 + e;
    // log1p(e) = log(w) + (e - (w - 1))/w; the correction is <= 2^-53, so 1/w ~ 1 - e/2 is exact enough for it
    return fmax(u, 0.0) + fma(e - (w - 1.0), fma(-0.5, e, 1.0), cg_log_12(w));
}

// running product with exponent kept apart (avoids n logs per determinant and over/underflow)
struct CgScaledProd {
    double m; int e;
    CG_DEVI void init() { m = 1.0; e = 0; }
    CG_DEVI void mul(double v) {
        int ex; m = frexp(m * v, &ex); e += ex;
    }
    // ool = true (kernels): the mantissa is in [1/2, 1) after frexp, its logarithm comes from the [1, 2) table routine
    // (~15 instructions) instead of the out-of-line libm call (~100 with the call) -- once or twice per log Psi evaluation
    CG_DEVI double logabs(bool ool = false) const {
        return (ool ? cg_log_pos(fabs(m)) : log(fabs(m))) + (double)e * 0.693147180559945309417232121458;
    }
};

struct CgCplx { double re, im; };
CG_DEVI CgCplx cmul(CgCplx a, CgCplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
CG_DEVI CgCplx csub(CgCplx a, CgCplx b) { return {a.re - b.re, a.im - b.im}; }
CG_DEVI CgCplx cadd(CgCplx a, CgCplx b) { return {a.re + b.re, a.im + b.im}; }
CG_DEVI CgCplx cinv(CgCplx a) {
    // Smith's algorithm: no overflow of re^2+im^2
    if (fabs(a.re) >= fabs(a.im)) {
        double r = a.im / a.re, d = 1.0 / (a.re + a.im * r);
        return {d, -r * d};
    } else {
        double r = a.re / a.im, d = 1.0 / (a.re * r + a.im);
        return {r * d, -d};
    }
}

// e / d by multiply-shift for the index arithmetic of the work-item loops (d = n, n D: runtime values, and a 32-bit integer
// division is ~30 VALU instructions): exact for e, d < 65536.  m = cg_div_magic(d), computed on the host.
static CG_HD unsigned cg_div_magic(unsigned d) { return d <= 1 ? 0u : 0xFFFFFFFFu / d + 1u; }
static CG_HD int cg_udiv(int e, unsigned m) {
#if defined(__HIP_DEVICE_COMPILE__)
    return m ? (int)__umulhi((unsigned)e, m) : e;
#else
    return m ? (int)(((unsigned long long)(unsigned)e * m) >> 32) : e;
#endif
}

// Deterministic (fixed-order) workgroup sum through LDS scratch of nthr doubles.
CG_DEVI double cg_block_sum(const CgBlk& b, double v, double* scratch) {
    scratch[b.tid] = v;
    b.sync();
    int s = 1;
    while (s < b.nthr) s <<= 1;
    for (s >>= 1; s > 0; s >>= 1) {
        if (b.tid < s && b.tid + s < b.nthr) scratch[b.tid] += scratch[b.tid + s];
        b.sync();
    }
    const double r = scratch[0];
    b.sync();
    return r;
}

// K sums at once: wave-level butterfly (no barrier), then the per-wave partials through LDS -- 2 barriers for all K
// instead of ~10 per scalar cg_block_sum.  Fixed summation order (deterministic); every thread gets the totals.
// scratch: >= K * (nthr / 64) doubles.  nthr must be a multiple of 64 on the GPU (1 on the host shim).
template <int K>
CG_DEVI void cg_block_sum_n(const CgBlk& b, double (&v)[K], double* scratch) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = v[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
        v[k] = x;
    }
    const int nw = b.nthr >> 6;
    if (nw > 1) {
        const int wave = b.tid >> 6;
        if ((b.tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) scratch[k * nw + wave] = v[k];
        }
        b.sync();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double a = 0.0;
            for (int w = 0; w < nw; ++w) a += scratch[k * nw + w];
            v[k] = a;
        }
        b.sync();
    }
#else
    (void)b; (void)v; (void)scratch;
#endif
}

