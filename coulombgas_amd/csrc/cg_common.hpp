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

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CG_HD __host__ __device__ __forceinline__
#define CG_DEV __device__
#define CG_DEVI __device__ __forceinline__
#else
#define CG_HD inline
#define CG_DEV
#define CG_DEVI inline
#endif

struct CgBlk {
    int tid, nthr;
    CG_DEVI void sync() const {
#if defined(__HIP_DEVICE_COMPILE__)
        __syncthreads();
#endif
    }
};

#define CG_PI 3.14159265358979323846264338327950288

// softplus(u) = log(1 + e^u) = max(u,0) + log1p(e^{-|u|}); sigmoid from the same exponential.
// (jax.nn.softplus == logaddexp(u, 0): no large-u cut-off, reference src/flow.py:45-52)
CG_DEVI void softplus_sigmoid(double u, double& sp, double& sg) {
    double e = exp(-fabs(u));
    double r = 1.0 / (1.0 + e);
    sp = fmax(u, 0.0) + log1p(e);
    sg = (u >= 0.0) ? r : e * r;
}
CG_DEVI double sigmoid_only(double u) {
    double e = exp(-fabs(u));
    double r = 1.0 / (1.0 + e);
    return (u >= 0.0) ? r : e * r;
}
CG_DEVI double softplus_only(double u) {
    return fmax(u, 0.0) + log1p(exp(-fabs(u)));
}

// running product with exponent kept apart (avoids n logs per determinant and over/underflow)
struct CgScaledProd {
    double m; int e;
    CG_DEVI void init() { m = 1.0; e = 0; }
    CG_DEVI void mul(double v) {
        int ex; m = frexp(m * v, &ex); e += ex;
    }
    CG_DEVI double logabs() const { return log(fabs(m)) + (double)e * 0.693147180559945309417232121458; }
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
