// cg_flow_generic.hpp -- FermiNet flow of ANY depth >= 2 and any layer widths (src/flow.py:5-55), value and
// Jacobian, for the configurations the depth-2 fast path (cg_flow_fast.hpp) is not instantiated for -- e.g. the
// depth-3 networks of the reference's own tests (tests/test_flow.py:42, tests/test_logpsi.py:29).
//
// Forward-mode like jax.jacfwd (src/logpsi.py:27-28) but with the pair stream's tangent kept with respect to
// r_ij only (t_ij depends on x through r_ij = x_i - x_j alone):
//     S_i[c][beta]  = d s_i[c] / d x_beta          (n x ws x N, dense)
//     Tt_ij[c][a]   = d t_ij[c] / d r_ij,a         (n x n x wt x d)
//     d mean_j t_ij / d x_(k,a) = (1/n) ( delta_ik sum_{j != i} Tt_ij[:,a]  -  (1 - delta_ik) Tt_ik[:,a] )
// Templated on the scalar type (double or Jet2) so that the same code yields z', z'', J', J'' for the Laplacian.
// One workgroup per walker; all arrays live in a per-workgroup HBM workspace (this is the slow, general path).
#pragma once
#include "cg_common.hpp"
#include "cg_jet.hpp"

#define CG_GEN_MAXDEPTH 8

struct CgGenModel {
    int n, dim, depth, hs, ht;
    double L;
    int fin_b, fin_w;
    int sp_b[CG_GEN_MAXDEPTH], sp_w[CG_GEN_MAXDEPTH], sp_in[CG_GEN_MAXDEPTH];
    int tp_b[CG_GEN_MAXDEPTH], tp_w[CG_GEN_MAXDEPTH], tp_in[CG_GEN_MAXDEPTH];
    int nparam;
    // workspace offsets in units of T
    int wmax_s, wmax_t, fmax;
    size_t o_sp, o_S, o_tp, o_Tt, o_f, o_df, o_u, o_z, o_J, o_tsum, total;
};

// host: parameter offsets in jax ravel_pytree order (sorted Haiku module names, 'b' before 'w')
static inline void cg_gen_model_init(CgGenModel& m, int n, int dim, int depth, int hs, int ht, double L) {
    m.n = n; m.dim = dim; m.depth = depth; m.hs = hs; m.ht = ht; m.L = L;
    struct Mod { char name[40]; int kind, idx; };
    Mod mods[2 * CG_GEN_MAXDEPTH];
    int k = 0;
    auto nm = [](char* out, int i) {
        int p = 0; const char* base = "fermi_net/~/linear";
        while (base[p]) { out[p] = base[p]; ++p; }
        if (i > 0) { out[p++] = '_'; char r[8]; int q = 0, v = i; while (v) { r[q++] = (char)('0' + v % 10); v /= 10; } while (q) out[p++] = r[--q]; }
        out[p] = 0;
    };
    for (int i = 0; i < depth; ++i, ++k) { nm(mods[k].name, i); mods[k].kind = 0; mods[k].idx = i; }
    for (int i = 0; i < depth - 1; ++i, ++k) { nm(mods[k].name, depth + i); mods[k].kind = 1; mods[k].idx = i; }
    { const char* f = "fermi_net/linear"; int p = 0; while (f[p]) { mods[k].name[p] = f[p]; ++p; } mods[k].name[p] = 0; mods[k].kind = 2; mods[k].idx = 0; ++k; }
    for (int a = 0; a < k; ++a)                       // insertion sort by name (strcmp order)
        for (int b = a + 1; b < k; ++b) {
            int c = 0; while (mods[a].name[c] && mods[a].name[c] == mods[b].name[c]) ++c;
            if ((unsigned char)mods[b].name[c] < (unsigned char)mods[a].name[c]) { Mod t = mods[a]; mods[a] = mods[b]; mods[b] = t; }
        }
    int off = 0;
    for (int a = 0; a < k; ++a) {
        int fin, fout;
        if (mods[a].kind == 0) { fin = mods[a].idx == 0 ? 4 * dim + 1 : 2 * hs + ht; fout = hs; m.sp_in[mods[a].idx] = fin; m.sp_b[mods[a].idx] = off; m.sp_w[mods[a].idx] = off + fout; }
        else if (mods[a].kind == 1) { fin = mods[a].idx == 0 ? 2 * dim + 1 : ht; fout = ht; m.tp_in[mods[a].idx] = fin; m.tp_b[mods[a].idx] = off; m.tp_w[mods[a].idx] = off + fout; }
        else { fin = hs; fout = dim; m.fin_b = off; m.fin_w = off + fout; }
        off += fout + fin * fout;
    }
    m.nparam = off;
    const int P = 2 * dim + 1, N = n * dim;
    m.wmax_s = hs > dim ? hs : dim; m.wmax_t = ht > P ? ht : P; m.fmax = 2 * m.wmax_s + m.wmax_t;
    size_t t = 0;
    auto take = [&](size_t c) { size_t r = t; t += c; return r; };
    m.o_sp = take((size_t)n * m.wmax_s); m.o_S = take((size_t)n * m.wmax_s * N);
    m.o_tp = take((size_t)n * n * m.wmax_t); m.o_Tt = take((size_t)n * n * m.wmax_t * dim);
    m.o_f = take((size_t)n * m.fmax); m.o_df = take((size_t)n * m.fmax * N);
    m.o_u = take((size_t)n * n * m.wmax_t);           // pre-activations (pair layer is the widest user)
    m.o_z = take(N); m.o_J = take((size_t)N * N);
    m.o_tsum = take((size_t)n * m.wmax_t * dim);
    m.total = t;
}

template <class T>
struct CgGen {
    // x: n*dim (T), ws: workspace of m.total T's.  Fills z (ws + o_z) and, if with_jac, J (ws + o_J, N x N row-major,
    // J[out][in]).
    static CG_DEVI void flow(const CgBlk& b, const CgGenModel& m, const double* __restrict__ th, const T* x, T* ws, bool with_jac) {
        const int n = m.n, d = m.dim, hs = m.hs, ht = m.ht, N = n * d, P = 2 * d + 1;
        const int WS = m.wmax_s, WT = m.wmax_t;
        T *sp = ws + m.o_sp, *S = ws + m.o_S, *tp = ws + m.o_tp, *Tt = ws + m.o_Tt, *f = ws + m.o_f, *df = ws + m.o_df,
          *u = ws + m.o_u, *z = ws + m.o_z, *J = ws + m.o_J, *tsum = ws + m.o_tsum;
        const double rn = 1.0 / (double)n;
        const double c1 = 2.0 * CG_PI / m.L, ch = CG_PI / m.L;
        // ---- initial streams (src/flow.py:16-26): s0 = 0; t0_ij = [cos, sin, |sin(pi r/L)|], diagonal [1..,0..,0]
        for (int e = b.tid; e < n * WS; e += b.nthr) sp[e] = T(0.0);
        if (with_jac) for (size_t e = b.tid; e < (size_t)n * WS * N; e += b.nthr) S[e] = T(0.0);
        for (int e = b.tid; e < n * n; e += b.nthr) {
            const int i = e / n, j = e - i * n;
            T* t = tp + (size_t)e * WT; T* tt = Tt + (size_t)e * WT * d;
            T d2 = T(0.0);
            for (int a = 0; a < d; ++a) {
                const T r = x[i * d + a] - x[j * d + a];
                T s2, c2, s1, c1v;
                cg_sincos(r * c1, s2, c2); cg_sincos(r * ch, s1, c1v);
                if (i == j) { c2 = T(1.0); s2 = T(0.0); }
                t[a] = c2; t[d + a] = s2;
                d2 += s1 * s1;
                for (int q = 0; q < d; ++q) { tt[a * d + q] = T(0.0); tt[(d + a) * d + q] = T(0.0); }
                if (i != j) { tt[a * d + a] = -c1 * s2; tt[(d + a) * d + a] = c1 * c2; }
            }
            if (i == j) { t[2 * d] = T(0.0); for (int q = 0; q < d; ++q) tt[2 * d * d + q] = T(0.0); }
            else {
                const T del = cg_sqrt(d2), rdel = cg_rcp(del);
                t[2 * d] = del;
                for (int a = 0; a < d; ++a) {
                    const T r = x[i * d + a] - x[j * d + a];
                    T s1, c1v; cg_sincos(r * ch, s1, c1v);
                    tt[2 * d * d + a] = ch * (s1 * c1v * rdel);
                }
            }
        }
        b.sync();
        int ws_cur = d, wt_cur = P;
        for (int layer = 0; layer < m.depth; ++layer) {
            const bool last = layer == m.depth - 1;
            const int fs = 2 * ws_cur + wt_cur;
            // tsum_i[c][a] = sum_{j != i} Tt_ij[c][a]
            if (with_jac)
                for (int e = b.tid; e < n * wt_cur * d; e += b.nthr) {
                    const int i = e / (wt_cur * d), r = e - i * wt_cur * d;
                    T acc = T(0.0);
                    for (int j = 0; j < n; ++j) if (j != i) acc += Tt[((size_t)(i * n + j) * WT) * d + r];
                    tsum[(size_t)i * WT * d + r] = acc;
                }
            // f_i = [s_i, mean_k s_k, mean_j t_ij]   (src/flow.py:28-37)
            for (int e = b.tid; e < n * fs; e += b.nthr) {
                const int i = e / fs, c = e - i * fs;
                T v;
                if (c < ws_cur) v = sp[i * WS + c];
                else if (c < 2 * ws_cur) { T a = T(0.0); for (int k = 0; k < n; ++k) a += sp[k * WS + (c - ws_cur)]; v = a * rn; }
                else { T a = T(0.0); for (int j = 0; j < n; ++j) a += tp[(size_t)(i * n + j) * WT + (c - 2 * ws_cur)]; v = a * rn; }
                f[i * m.fmax + c] = v;
            }
            b.sync();
            if (with_jac) {
                for (size_t e = b.tid; e < (size_t)n * fs * N; e += b.nthr) {
                    const int i = (int)(e / ((size_t)fs * N)); const int r = (int)(e - (size_t)i * fs * N); const int c = r / N, q = r - c * N;
                    T v;
                    if (c < ws_cur) v = S[((size_t)i * WS + c) * N + q];
                    else if (c < 2 * ws_cur) { T a = T(0.0); for (int k = 0; k < n; ++k) a += S[((size_t)k * WS + (c - ws_cur)) * N + q]; v = a * rn; }
                    else {
                        const int cc = c - 2 * ws_cur, k = q / d, a = q - k * d;
                        if (k == i) v = tsum[(size_t)i * WT * d + cc * d + a] * rn;
                        else v = -rn * Tt[((size_t)(i * n + k) * WT + cc) * d + a];
                    }
                    df[((size_t)i * m.fmax + c) * N + q] = v;
                }
                b.sync();
            }
            // one-particle layer: u_i = f_i W + b
            const double* W = th + m.sp_w[layer]; const double* bb = th + m.sp_b[layer];
            for (int e = b.tid; e < n * hs; e += b.nthr) {
                const int i = e / hs, h = e - i * hs;
                T a = T(bb[h]);
                for (int c = 0; c < fs; ++c) a += W[c * hs + h] * f[i * m.fmax + c];
                u[e] = a;
            }
            b.sync();
            // activation; layer 0 assigns (src/flow.py:45), later layers are residual (:48,:52)
            if (with_jac) {
                for (size_t e = b.tid; e < (size_t)n * hs * N; e += b.nthr) {
                    const int i = (int)(e / ((size_t)hs * N)); const int r = (int)(e - (size_t)i * hs * N); const int h = r / N, q = r - h * N;
                    T a = T(0.0);
                    for (int c = 0; c < fs; ++c) a += W[c * hs + h] * df[((size_t)i * m.fmax + c) * N + q];
                    const T sg = cg_sigmoid(u[i * hs + h]);
                    T* dst = S + ((size_t)i * WS + h) * N + q;
                    // NOTE: S[i][h] of layer 0 (old width d) is overwritten here; df already holds the old values
                    *dst = (layer == 0) ? sg * a : *dst + sg * a;
                }
            }
            b.sync();
            for (int e = b.tid; e < n * hs; e += b.nthr) {
                const int i = e / hs, h = e - i * hs;
                const T spv = cg_softplus(u[e]);
                sp[i * WS + h] = (layer == 0) ? spv : sp[i * WS + h] + spv;
            }
            b.sync();
            if (!last) {
                const double* Wt = th + m.tp_w[layer]; const double* bt = th + m.tp_b[layer];
                for (int e = b.tid; e < n * n * ht; e += b.nthr) {
                    const int pr = e / ht, h = e - pr * ht;
                    T a = T(bt[h]);
                    for (int c = 0; c < wt_cur; ++c) a += Wt[c * ht + h] * tp[(size_t)pr * WT + c];
                    u[e] = a;
                }
                b.sync();
                // new tangents need the OLD Tt of the same pair for every h: stage through df (free here)
                T* nt = df;
                if (with_jac) {
                    for (int e = b.tid; e < n * n * ht * d; e += b.nthr) {
                        const int pr = e / (ht * d), r = e - pr * ht * d, h = r / d, a = r - h * d;
                        T acc = T(0.0);
                        for (int c = 0; c < wt_cur; ++c) acc += Wt[c * ht + h] * Tt[((size_t)pr * WT + c) * d + a];
                        const T sg = cg_sigmoid(u[pr * ht + h]);
                        nt[e] = (layer == 0) ? sg * acc : Tt[((size_t)pr * WT + h) * d + a] + sg * acc;
                    }
                    b.sync();
                    for (int e = b.tid; e < n * n * ht * d; e += b.nthr) {
                        const int pr = e / (ht * d), r = e - pr * ht * d;
                        Tt[(size_t)pr * WT * d + r] = nt[e];
                    }
                }
                for (int e = b.tid; e < n * n * ht; e += b.nthr) {
                    const int pr = e / ht, h = e - pr * ht;
                    const T spv = cg_softplus(u[e]);
                    T* dst = tp + (size_t)pr * WT + h;
                    *dst = (layer == 0) ? spv : *dst + spv;
                }
                b.sync();
                wt_cur = ht;
            }
            ws_cur = hs;
        }
        // z = x + s Wf + bf   (src/flow.py:53-55);  J = I + S Wf
        const double* Wf = th + m.fin_w; const double* bf = th + m.fin_b;
        for (int e = b.tid; e < N; e += b.nthr) {
            const int i = e / d, a = e - i * d;
            T v = x[e] + bf[a];
            for (int h = 0; h < hs; ++h) v += Wf[h * d + a] * sp[i * WS + h];
            z[e] = v;
        }
        if (with_jac)
            for (int e = b.tid; e < N * N; e += b.nthr) {
                const int r = e / N, q = e - r * N, i = r / d, a = r - i * d;
                T v = T(r == q ? 1.0 : 0.0);
                for (int h = 0; h < hs; ++h) v += Wf[h * d + a] * S[((size_t)i * WS + h) * N + q];
                J[e] = v;
            }
        b.sync();
    }
};
