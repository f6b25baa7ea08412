// cg_jet.hpp -- second-order directional jets: f(x + eps v) = v + d*eps + dd*eps^2/2 + ...
// (dd is the second derivative, not the Taylor coefficient).  Pushing a jet through the flow and its
// Jacobian assembly yields, for one direction v, z' = J v, z'' = v^T (d^2 z) v, J' and J'' -- the
// quantities the reference obtains with jvp(jacrev(.)) nests (src/logpsi.py:77-103,110-164).
#pragma once
#include "cg_common.hpp"

struct Jet2 {
    double v, d, dd;
    Jet2() = default;
    CG_DEVI Jet2(double a) : v(a), d(0.0), dd(0.0) {}
    CG_DEVI Jet2(double a, double b, double c) : v(a), d(b), dd(c) {}
};
CG_DEVI Jet2 operator+(Jet2 a, Jet2 b) { return {a.v + b.v, a.d + b.d, a.dd + b.dd}; }
CG_DEVI Jet2 operator-(Jet2 a, Jet2 b) { return {a.v - b.v, a.d - b.d, a.dd - b.dd}; }
CG_DEVI Jet2 operator-(Jet2 a) { return {-a.v, -a.d, -a.dd}; }
CG_DEVI Jet2 operator*(Jet2 a, Jet2 b) { return {a.v * b.v, a.d * b.v + a.v * b.d, a.dd * b.v + 2.0 * a.d * b.d + a.v * b.dd}; }
CG_DEVI Jet2 operator*(double a, Jet2 b) { return {a * b.v, a * b.d, a * b.dd}; }
CG_DEVI Jet2 operator*(Jet2 b, double a) { return {a * b.v, a * b.d, a * b.dd}; }
CG_DEVI Jet2 operator+(Jet2 a, double b) { return {a.v + b, a.d, a.dd}; }
CG_DEVI Jet2 operator+(double b, Jet2 a) { return {a.v + b, a.d, a.dd}; }
CG_DEVI Jet2 operator-(Jet2 a, double b) { return {a.v - b, a.d, a.dd}; }
CG_DEVI Jet2 operator-(double b, Jet2 a) { return {b - a.v, -a.d, -a.dd}; }
CG_DEVI Jet2& operator+=(Jet2& a, Jet2 b) { a.v += b.v; a.d += b.d; a.dd += b.dd; return a; }
CG_DEVI Jet2& operator-=(Jet2& a, Jet2 b) { a.v -= b.v; a.d -= b.d; a.dd -= b.dd; return a; }

// f(u) with f', f'' given at u.v
CG_DEVI Jet2 jet_chain(Jet2 u, double f, double f1, double f2) { return {f, f1 * u.d, f1 * u.dd + f2 * u.d * u.d}; }

// --- scalar-generic math used by the templated flow code (T = double or Jet2) ---
CG_DEVI void cg_sincos(double a, double& s, double& c, bool ool = false) {
    if (ool) { const CgSinCos r = cg_sincos_ool(a); s = r.s; c = r.c; }
    else sincos(a, &s, &c);
}
CG_DEVI void cg_sincos(Jet2 a, Jet2& s, Jet2& c, bool = false) {
    double sv, cv; sincos(a.v, &sv, &cv);
    s = jet_chain(a, sv, cv, -sv); c = jet_chain(a, cv, -sv, -cv);
}
CG_DEVI double cg_sqrt(double a) { return sqrt(a); }
CG_DEVI double cg_rcp(double a) { return 1.0 / a; }
#if defined(__HIP_DEVICE_COMPILE__)
// jets of sqrt and 1/x from v_rsq_f64 / v_rcp_f64 + Newton steps (~1 ulp) instead of the IEEE sqrt and two divisions (~80
// instructions per pair feature of the directional passes)
CG_DEVI Jet2 cg_sqrt(Jet2 a) {
    double y = __builtin_amdgcn_rsq(a.v);
    const double hx = 0.5 * a.v;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    const double g = a.v * y;
    const double r = fma(fma(-g, g, a.v), 0.5 * y, g);           // sqrt(v), ri = y = 1 / sqrt(v)
    return jet_chain(a, r, 0.5 * y, -0.25 * y * (y * y));
}
CG_DEVI Jet2 cg_rcp(Jet2 a) {
    double r = __builtin_amdgcn_rcp(a.v);
    r = fma(r, fma(-a.v, r, 1.0), r);
    r = fma(r, fma(-a.v, r, 1.0), r);
    return jet_chain(a, r, -r * r, 2.0 * r * (r * r));
}
#else
CG_DEVI Jet2 cg_sqrt(Jet2 a) { const double r = sqrt(a.v), ri = 1.0 / r; return jet_chain(a, r, 0.5 * ri, -0.25 * ri / a.v); }
CG_DEVI Jet2 cg_rcp(Jet2 a) { const double r = 1.0 / a.v; return jet_chain(a, r, -r * r, 2.0 * r * r * r); }
#endif
CG_DEVI void cg_softplus_sigmoid(double u, double& sp, double& sg) { softplus_sigmoid(u, sp, sg); }
CG_DEVI void cg_softplus_sigmoid(Jet2 u, Jet2& sp, Jet2& sg) {
    double s, g; softplus_sigmoid(u.v, s, g);
    const double g1 = g * (1.0 - g), g2 = g1 * (1.0 - 2.0 * g);
    sp = jet_chain(u, s, g, g1); sg = jet_chain(u, g, g1, g2);
}
CG_DEVI double cg_sigmoid(double u) { return sigmoid_only(u); }
CG_DEVI Jet2 cg_sigmoid(Jet2 u) {
    const double g = sigmoid_only(u.v), g1 = g * (1.0 - g), g2 = g1 * (1.0 - 2.0 * g);
    return jet_chain(u, g, g1, g2);
}
CG_DEVI double cg_softplus(double u) { return softplus_only(u); }
CG_DEVI Jet2 cg_softplus(Jet2 u) { Jet2 s, g; cg_softplus_sigmoid(u, s, g); return s; }


// value part / "carries derivatives" trait of the scalar types the flow code is instantiated with
CG_DEVI double cg_val(double a) { return a; }
CG_DEVI double cg_val(const Jet2& a) { return a.v; }
template <class T> struct CgIsJet { static constexpr bool value = false; };
template <> struct CgIsJet<Jet2> { static constexpr bool value = true; };
