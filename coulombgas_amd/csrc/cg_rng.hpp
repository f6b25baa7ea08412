// cg_rng.hpp -- counter-based Philox4x32-10 stream for the production sampler.
// Stands in for jax.random.normal / jax.random.uniform of src/MCMC.py:24-29 (threefry2x32 in the
// reference; bit-parity with jax.random is not a goal -- parity runs pass the noise in).
#pragma once
#include "cg_common.hpp"

CG_DEVI void cg_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += W0; k1 += W1;
    }
}
// two uniforms: u_open in (0,1), u_half in [0,1)
CG_DEVI void cg_philox_uniform2(uint64_t seed, uint64_t walker, uint32_t step, uint32_t item, double& u_open, double& u_half) {
    uint32_t c[4] = {(uint32_t)walker, (uint32_t)(walker >> 32), step, item};
    cg_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint64_t a = (((uint64_t)c[1] << 32) | c[0]) >> 11, bb = (((uint64_t)c[3] << 32) | c[2]) >> 11;
    u_open = ((double)a + 0.5) * (1.0 / 9007199254740992.0);
    u_half = (double)bb * (1.0 / 9007199254740992.0);
}
CG_DEVI double cg_philox_normal(uint64_t seed, uint64_t walker, uint32_t step, uint32_t item) {
    double u1, u2; cg_philox_uniform2(seed, walker, step, item, u1, u2);
    double s, c; sincos(2.0 * CG_PI * u2, &s, &c);
    return sqrt(-2.0 * log(u1)) * c;
}
CG_DEVI double cg_philox_uniform(uint64_t seed, uint64_t walker, uint32_t step) {
    double u1, u2; cg_philox_uniform2(seed, walker, step, 0xFFFFFFFFu, u1, u2);
    return u2;
}
// Out-of-line versions for the depth-2 sampler kernel (one call per coordinate and Metropolis step): the Philox key
// schedule and the constants of log / sincos / sqrt stay inside the callee instead of being hoisted out of the chain
// loop and spilled (see CG_OUTLINE in cg_common.hpp for why only that kernel uses them).
#if defined(__HIPCC__)
static CG_OUTLINE double cg_philox_normal_ool(uint64_t seed, uint64_t walker, uint32_t step, uint32_t item) {
    return cg_philox_normal(seed, walker, step, item);
}
static CG_OUTLINE double cg_philox_uniform_ool(uint64_t seed, uint64_t walker, uint32_t step) {
    return cg_philox_uniform(seed, walker, step);
}
#endif
