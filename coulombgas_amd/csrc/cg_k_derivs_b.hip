// cg_k_derivs_b.hip -- derivative kernels of the remaining (dim, spsize, tpsize) instantiations.
#include "cg_host.hpp"
#include "cg_derivs.hpp"
#include "cg_lap.hpp"
#include "cg_score.hpp"

#define CG_UNIT_CONFIGS(X) CG_FAST_CONFIGS_B(X)
#define CG_UNIT_NAME(f) cg_derivs_b_##f
#include "cg_k_derivs.inc"
