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

// ---- fp64 transcendentals specialised for the flow's activations ---------------------------------------------
// gfx950 has no f64 exp/log instructions; the generic ocml routines carry range/special-case handling these
// call sites do not need.  Accuracy of the pieces below is ~1 ulp (tests compare with libm through the oracle).

// e^x for x <= 0 (x = -|u|).  k = rint(x log2 e), r = x - k ln2 (two-part), degree-13 Taylor, scale by 2^k.
CG_DEVI double cg_exp_nonpos(double x) {
    x = fmax(x, -746.0);                                  // below this e^x underflows to 0 anyway
    const double kf = rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821614599e-10;                 // 1/13!
    p = fma(p, r, 2.0876756987868098979e-09);             // 1/12!
    p = fma(p, r, 2.5052108385441718775e-08);             // 1/11!
    p = fma(p, r, 2.7557319223985890653e-07);             // 1/10!
    p = fma(p, r, 2.7557319223985892511e-06);             // 1/9!
    p = fma(p, r, 2.4801587301587301566e-05);             // 1/8!
    p = fma(p, r, 1.9841269841269841253e-04);             // 1/7!
    p = fma(p, r, 1.3888888888888889419e-03);             // 1/6!
    p = fma(p, r, 8.3333333333333332177e-03);             // 1/5!
    p = fma(p, r, 4.1666666666666664354e-02);             // 1/4!
    p = fma(p, r, 1.6666666666666665741e-01);             // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}
// 1/w.  IEEE division on purpose: a hand-rolled v_rcp_f64 + 2 Newton steps version produced a 1e-7 relative
// deviation in one ill-conditioned parity case on gfx950 (every variant with a correctly rounded quotient in either
// call site, or a third Newton step, did not) -- correctness first, ~4 instructions per call.
CG_DEVI double cg_rcp_12(double w) { return 1.0 / w; }
// log(w) for w in [1,2]:  m = w or w/2 in [sqrt(1/2), sqrt 2], s = (m-1)/(m+1), log m = 2 s (1 + s^2/3 + ... + s^20/21)
CG_DEVI double cg_log_12(double w) {
    const bool hi = w > 1.41421356237309514547;
    const double m = hi ? 0.5 * w : w;
    const double den = m + 1.0;                           // in [1.707, 2.414]
    const double s = (m - 1.0) / den;
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    p = p * z;                                            // log m = 2s + 2s*p
    const double two_s = s + s;
    const double lg = fma(two_s, p, two_s);
    return hi ? lg + 0.693147180559945309417232121458 : lg;
}
// softplus(u) = log(1 + e^u) = max(u,0) + log1p(e^{-|u|}); sigmoid from the same exponential.
// (jax.nn.softplus == logaddexp(u, 0): no large-u cut-off, reference src/flow.py:45-52)
CG_DEVI void softplus_sigmoid(double u, double& sp, double& sg) {
    const double e = cg_exp_nonpos(-fabs(u));
    const double w = 1.0 + e;
    const double r = cg_rcp_12(w);
    // log1p(e) = log(w) + (e - (w - 1)) / w   (restores the bits of e lost in forming w)
    sp = fmax(u, 0.0) + fma(e - (w - 1.0), r, cg_log_12(w));
    sg = (u >= 0.0) ? r : e * r;
}
CG_DEVI double sigmoid_only(double u) {
    const double e = cg_exp_nonpos(-fabs(u));
    const double r = cg_rcp_12(1.0 + e);
    return (u >= 0.0) ? r : e * r;
}
CG_DEVI double softplus_only(double u) {
    const double e = cg_exp_nonpos(-fabs(u));
    const double w = 1.0 + e;
    const double c = e - (w - 1.0);                        // |c| <= 2^-53; c / w ~ c (1 - e/2) to 1e-17
    return fmax(u, 0.0) + fma(c, fma(-0.5, e, 1.0), cg_log_12(w));
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
