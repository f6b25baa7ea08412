// cg_dispatch.hpp -- compile-time (dim, spsize, tpsize) instantiations of the depth-2 fast path.
#pragma once
// X(D, HS, HT)
#if defined(CG_ONLY_2_16_16)      /* diagnostic builds: one configuration, a fifth of the compile time */
#define CG_FAST_CONFIGS(X) X(2, 16, 16)
#elif defined(CG_ONLY_3_4_4)
#define CG_FAST_CONFIGS(X) X(3, 4, 4)
#else
// two groups = two translation units per kernel family (compiled in parallel): A is the shape of every shipped run
#define CG_FAST_CONFIGS_A(X) X(2, 16, 16)
#define CG_FAST_CONFIGS_B(X) X(3, 16, 16) X(2, 4, 4) X(3, 4, 4) X(2, 8, 8) X(3, 8, 8) X(2, 32, 32)
#define CG_FAST_CONFIGS(X) CG_FAST_CONFIGS_A(X) CG_FAST_CONFIGS_B(X)
#endif
#if !defined(CG_FAST_CONFIGS_A)
#define CG_FAST_CONFIGS_A(X) CG_FAST_CONFIGS(X)
#define CG_FAST_CONFIGS_B(X)
#endif

// Sampler kernels additionally specialised on the particle number and the workgroup size: X(D, HS, HT, N, THREADS).
// With n a compile-time constant every LDS offset, trip count and tile count of the chain folds away (the runtime-n
// kernel keeps ~100 scalar registers of layout state alive across the Metropolis loop and spills them).
#if defined(CG_NO_SPECIALS) || defined(CG_ONLY_3_4_4)
#define CG_MCMC_SPECIALS(X)
#else
#define CG_MCMC_SPECIALS(X) X(2, 16, 16, 13, 64) X(2, 16, 16, 29, 256) X(2, 16, 16, 49, 512) X(2, 16, 16, 57, 512)     /* the sizes of the reference's production runs (data/n_29, n_49, n_57) and of its n = 13 benchmark */
#endif

static inline bool cg_fast_supported(int depth, int dim, int hs, int ht) {
    if (depth != 2) return false;
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) return true;
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return false;
}
