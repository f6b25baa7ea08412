// cg_k_sampler_b.hip -- log Psi / Metropolis kernels of the remaining (dim, spsize, tpsize) instantiations.
#include "cg_host.hpp"
#include "cg_rng.hpp"

#define CG_UNIT_CONFIGS(X) CG_FAST_CONFIGS_B(X)
#define CG_UNIT_SPECIALS(X)
#define CG_UNIT_NAME(f) cg_sampler_b_##f
#include "cg_k_sampler.inc"
