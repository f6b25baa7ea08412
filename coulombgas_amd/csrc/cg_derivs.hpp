// placeholder, replaced below
#pragma once
#include "cg_flow_fast.hpp"
template <int D, int HS, int HT>
struct CgDerivs {
    static constexpr bool implemented = false;
    static size_t ws_doubles(int n) { return 8; }
    static size_t lds_doubles(int n, int nthr) { return 8; }
    static CG_DEVI void grad_laplacian(const CgBlk&, const double*, const double*, const double*, const int*, int, double, int,
                                       const double*, double*, double*, double*, double*) {}
    static CG_DEVI void param_vjp(const CgBlk&, const double*, const double*, const double*, const int*, int, double, double, double,
                                  double*, double*, double*, double*) {}
};
