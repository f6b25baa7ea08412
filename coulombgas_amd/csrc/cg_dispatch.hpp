// cg_dispatch.hpp -- compile-time (dim, spsize, tpsize) instantiations of the depth-2 fast path.
#pragma once
// X(D, HS, HT)
#if defined(CG_ONLY_2_16_16)      /* diagnostic builds: one configuration, a fifth of the compile time */
#define CG_FAST_CONFIGS(X) X(2, 16, 16)
#else
#define CG_FAST_CONFIGS(X) X(2, 16, 16) X(3, 16, 16) X(2, 4, 4) X(3, 4, 4) X(2, 8, 8) X(3, 8, 8) X(2, 32, 32)
#endif

static inline bool cg_fast_supported(int depth, int dim, int hs, int ht) {
    if (depth != 2) return false;
#define CG_X(D, HS, HT) if (dim == D && hs == HS && ht == HT) return true;
    CG_FAST_CONFIGS(CG_X)
#undef CG_X
    return false;
}
