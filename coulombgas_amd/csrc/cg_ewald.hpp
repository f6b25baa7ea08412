// cg_ewald.hpp -- Ewald Coulomb energy of one walker (reference: src/potential.py:36-77).
//
// Real space: nearest image only, r~ = r/L - rint(r/L) (src/potential.py:47-48, round-half-even),
//             sum_{i<j} erfc(kappa |r~|)/|r~|.
// Reciprocal: the reference evaluates sum_G g_G sum_{i<j} cos(2 pi G.r~_ij) on a (pairs x |G|) tensor
//             (src/potential.py:61).  For integer G this equals sum_G g_G (|S(G)|^2 - n)/2 with the
//             structure factor S(G) = sum_i exp(2 pi i G.x_i/L); S is built from per-particle power
//             tables e^{2 pi i m x/L}, m = 0..Gmax, held in LDS (no transcendental per (pair,G)).
#pragma once
#include "cg_common.hpp"

// lds: [n*D*(Gmax+1)*2 table] [nthr scratch]
template <int D>
CG_DEVI double cg_ewald_walker(const CgBlk& b, const double* x, int n, double L, double kappa, double rs,
                               const int* __restrict__ G, const double* __restrict__ gk, int nG, int Gmax, double g0,
                               double* lds) {
    const int T = Gmax + 1;
    double* tab = lds;
    double* scratch = lds + (size_t)n * D * T * 2;
    for (int e = b.tid; e < n * D; e += b.nthr) {
        double s, c; sincos(x[e] * (2.0 * CG_PI / L), &s, &c);
        double* t = tab + (size_t)e * T * 2;
        double pr = 1.0, pi = 0.0;
        t[0] = 1.0; t[1] = 0.0;
        for (int m = 1; m < T; ++m) {
            const double nr = pr * c - pi * s, ni = pr * s + pi * c;
            pr = nr; pi = ni; t[2 * m] = pr; t[2 * m + 1] = pi;
        }
    }
    b.sync();
    double acc = 0.0;
    const double rL = 1.0 / L;
    for (int e = b.tid; e < n * n; e += b.nthr) {
        const int i = e / n, j = e - i * n;
        if (j <= i) continue;
        double d2 = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            double r = (x[i * D + a] - x[j * D + a]) * rL;
            r -= rint(r);
            d2 += r * r;
        }
        const double d = sqrt(d2);
        acc += erfc(kappa * d) / d;
    }
    for (int g = b.tid; g < nG; g += b.nthr) {
        int gv[D];
#pragma unroll
        for (int a = 0; a < D; ++a) gv[a] = G[g * D + a];
        double sr = 0.0, si = 0.0;
        for (int i = 0; i < n; ++i) {
            double pr = 1.0, pi = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
                const int m = gv[a] < 0 ? -gv[a] : gv[a];
                const double* t = tab + ((size_t)(i * D + a) * T + m) * 2;
                const double tr = t[0], ti = gv[a] < 0 ? -t[1] : t[1];
                const double nr = pr * tr - pi * ti, ni = pr * ti + pi * tr;
                pr = nr; pi = ni;
            }
            sr += pr; si += pi;
        }
        acc += gk[g] * 0.5 * (sr * sr + si * si - (double)n);
    }
    const double tot = cg_block_sum(b, acc, scratch);
    return 2.0 * rs / L * (tot + g0 * (0.5 * n * (n - 1)));
}
