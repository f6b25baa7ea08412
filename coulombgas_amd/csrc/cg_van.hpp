// cg_van.hpp -- the autoregressive Transformer density matrix on the device: sampler and log-probability.
//
// Reference: src/autoregressive.py:50-96 (Transformer: Linear + tanh embedding, num_layers x [causal multi-head
// self-attention + residual, DenseBlock (Linear, tanh, Linear) + residual], tanh, Linear; logits shifted by one position
// with the learned x1hat in front, :93-94), src/sampler.py:6-10 (mask: strictly increasing indices that leave room for the
// remaining electrons), :30-38 (sequential sampler, jax.random.categorical = Gumbel-max), :40-44 (log_prob).
//
// One wave per sample, tokens in sequence with a key / value cache in LDS (the reference re-runs the whole network for
// every position: n full passes; here position t costs one token pass).  Lane = feature index in the dense layers, = earlier
// position in the attention (n <= 64), = orbital index (stride 64) in the logits.  Weights are staged in LDS when they fit
// next to the caches, otherwise read through L1 / L2.  Fixed summation orders: deterministic.
//
// Flat parameter order expected by cg_van_set_params (count = cg_van_num_params):
//   x1hat[M]; embedding b[ms], w[dim][ms];
//   per layer: query b[ms], w[ms][ms]; key b, w; value b, w; attention output linear b[ms], w[ms][ms];
//              mlp linear b[hs], w[ms][hs]; mlp linear_1 b[ms], w[hs][ms];
//   output b[M], w[ms][M]                                  (every w row-major (in, out), as Haiku stores it)
#pragma once
#include "cg_common.hpp"
#include "cg_rng.hpp"

#define CG_VAN_MAXLAYERS 8
struct CgVanModel {
    int M, nl, ms, nh, ks, hs, dim, n;
    int o_x1, o_eb, o_ew, o_ob, o_ow, total;
    int o_l[CG_VAN_MAXLAYERS];          // start of layer l; inside: qb qw kb kw vb vw ob ow m1b m1w m2b m2w
    int wave_doubles;                   // LDS doubles of one wave's scratch
};
static inline int cg_van_model_init(CgVanModel& m, int M, int nl, int ms, int nh, int hs, int dim, int n) {
    m.M = M; m.nl = nl; m.ms = ms; m.nh = nh; m.ks = ms / nh; m.hs = hs; m.dim = dim; m.n = n;
    int t = 0;
    m.o_x1 = t; t += M;
    m.o_eb = t; t += ms; m.o_ew = t; t += dim * ms;
    for (int l = 0; l < nl; ++l) { m.o_l[l] = t; t += 4 * (ms + ms * ms) + (hs + ms * hs) + (ms + hs * ms); }
    m.o_ob = t; t += M; m.o_ow = t; t += ms * M;
    m.total = t;
    m.wave_doubles = (6 * ms + hs + 2 * nl * n * ms + 1) & ~1;
    return t;
}
// reverse pass (cg_van_gradient): LDS doubles of one wave, and doubles of its stash of per-token activations in HBM
CG_HD int cg_van_grad_wave_doubles(const CgVanModel& m) { return (12 * m.ms + 3 * m.hs + 4 * m.nl * m.n * m.ms + 1) & ~1; }
CG_HD int cg_van_token_stash(const CgVanModel& m) { return 2 * m.ms + m.nl * (4 * m.ms + m.hs + m.nh * m.n); }

#if defined(__HIPCC__)
// wave-wide sum / maximum, the result in every lane: DPP row scans (row_shr 1, 2, 4, 8: lane 15 of a row ends with the row's total) and
// four readlanes, fixed order ((r0 + r1) + (r2 + r3)).  The xor-butterfly of __shfl_xor goes through the LDS crossbar (ds_bpermute, six
// dependent round trips of two dwords): ~700 cycles against ~150, and a token pass has ~100 of these reductions.
template <int CTRL>
__device__ __forceinline__ double cg_van_dpp(double v, double fill) {
    const long long u = __double_as_longlong(v), f = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)f, (int)u, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(f >> 32), (int)(u >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double cg_van_readlane(double v, int l) {
    const long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | lo);
}
__device__ __forceinline__ double cg_wsum(double v) {
    v += cg_van_dpp<0x111>(v, 0.0); v += cg_van_dpp<0x112>(v, 0.0); v += cg_van_dpp<0x114>(v, 0.0); v += cg_van_dpp<0x118>(v, 0.0);
    return (cg_van_readlane(v, 15) + cg_van_readlane(v, 31)) + (cg_van_readlane(v, 47) + cg_van_readlane(v, 63));
}
__device__ __forceinline__ double cg_wmax(double v) {
    v = fmax(v, cg_van_dpp<0x111>(v, -INFINITY)); v = fmax(v, cg_van_dpp<0x112>(v, -INFINITY));
    v = fmax(v, cg_van_dpp<0x114>(v, -INFINITY)); v = fmax(v, cg_van_dpp<0x118>(v, -INFINITY));
    return fmax(fmax(cg_van_readlane(v, 15), cg_van_readlane(v, 31)), fmax(cg_van_readlane(v, 47), cg_van_readlane(v, 63)));
}

// One sample on one wave.  SAMPLE: draws state_idx[0..n) (written to sidx) with Gumbel-max noise from `unif` (n x M, parity
// mode) or the Philox stream (seed, stream); otherwise reads sidx.  Returns log p(state_idx) in every lane.
// MS / HS / NL / NH > 0: the model dimensions at compile time (the 16- and 32-term products unroll, their weight loads overlap)
template <bool SAMPLE, int MS = 0, int HS = 0, int NL = 0, int NH = 0>
__device__ __forceinline__ double cg_van_sequence(const CgVanModel& m_, const double* P, const double* __restrict__ sp,
                                                  int* __restrict__ sidx, double* lw, const double* __restrict__ unif,
                                                  uint64_t seed, uint64_t stream) {
    const int lane = threadIdx.x & 63;
    CgVanModel m = m_;
    if (MS > 0) { m.ms = MS; m.hs = HS; m.nl = NL; m.nh = NH; m.ks = MS / NH; }
    const int ms = m.ms, hs = m.hs, M = m.M, n = m.n, ks = m.ks;
    double* h = lw; double* q = h + ms; double* att = q + ms; double* h1 = att + ms; double* th = h1 + ms; double* mid = th + ms;
    double* kc = mid + hs; double* vc = kc + (size_t)m.nl * n * ms;      // [layer][position][feature]
    const double rsk = 1.0 / sqrt((double)ks);
    double logp = 0.0;
    int prev = -1;
    for (int t = 0; t < n; ++t) {
        // ---- conditional of electron t over the orbitals: x1hat (t = 0) or the output layer on tanh(h_{t-1})
        double lg[4]; double best = -INFINITY; int bidx = 0x7fffffff;
        const int hi = t + M - n;                                        // src/sampler.py:7: tril(ones(n, M), k = M - n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = lane + 64 * r;
            double v = -1e50;
            if (j < M && j > prev && j <= hi) {
                if (t == 0) v = P[m.o_x1 + j];
                else { v = P[m.o_ob + j]; for (int i = 0; i < ms; ++i) v = fma(th[i], P[m.o_ow + i * M + j], v); }
            }
            lg[r] = j < M ? v : -INFINITY;
            if (SAMPLE && j < M) {
                double u, u2;
                if (unif) u = unif[(size_t)t * M + j]; else cg_philox_uniform2(seed, stream, (uint32_t)t, (uint32_t)j, u, u2);
                const double key = v - log(-log(u));                    // Gumbel-max = jax.random.categorical (src/sampler.py:36-37)
                if (key > best) { best = key; bidx = j; }
            }
        }
        int st;
        if (SAMPLE) {
            const double mx = cg_wmax(best);
            const unsigned long long mask = __ballot(best == mx);
            // ties: the smallest orbital index (np.argmax); within a lane the scan above already kept the smallest
            int cand = (best == mx) ? bidx : 0x7fffffff;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
            (void)mask;
            st = cand;
            if (lane == 0) sidx[t] = st;
        } else {
            st = sidx[t];
        }
        // log softmax at the chosen orbital (src/sampler.py:41-43)
        double mx = fmax(fmax(lg[0], lg[1]), fmax(lg[2], lg[3]));
        mx = cg_wmax(mx);
        double z = 0.0, mine = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = lane + 64 * r;
            if (j < M) { z += exp(lg[r] - mx); if (j == st) mine = lg[r]; }
        }
        z = cg_wsum(z); mine = cg_wsum(mine);
        logp += mine - mx - log(z);
        prev = st;
        if (t == n - 1) break;
        // ---- token pass of electron t (its momentum is the network input, src/sampler.py:26)
        if (lane < ms) {
            double a = P[m.o_eb + lane];
            for (int c = 0; c < m.dim; ++c) a = fma(sp[(size_t)st * m.dim + c], P[m.o_ew + c * ms + lane], a);
            h[lane] = tanh(a);
        }
        asm volatile("" ::: "memory");
        for (int l = 0; l < m.nl; ++l) {
            const double* Lp = P + m.o_l[l];
            const int blk = ms + ms * ms;
            double* kcl = kc + ((size_t)l * n + t) * ms; double* vcl = vc + ((size_t)l * n + t) * ms;
            for (int idx = lane; idx < 3 * ms; idx += 64) {               // query, key, value
                const int which = idx / ms, j = idx - which * ms;
                const double* bp = Lp + which * blk; const double* wp = bp + ms;
                double a = bp[j];
                for (int i = 0; i < ms; ++i) a = fma(h[i], wp[i * ms + j], a);
                if (which == 0) q[j] = a; else if (which == 1) kcl[j] = a; else vcl[j] = a;
            }
            asm volatile("" ::: "memory");
            const double* kl = kc + (size_t)l * n * ms; const double* vl = vc + (size_t)l * n * ms;
            for (int hd = 0; hd < m.nh; ++hd) {                           // causal attention over the positions <= t (lane = position)
                double s = -INFINITY;
                if (lane <= t) {
                    s = 0.0;
                    for (int c = 0; c < ks; ++c) s = fma(q[hd * ks + c], kl[(size_t)lane * ms + hd * ks + c], s);
                    s *= rsk;
                }
                const double smx = cg_wmax(s);
                const double e = lane <= t ? exp(s - smx) : 0.0;
                const double w = e / cg_wsum(e);
                for (int c = 0; c < ks; ++c) {
                    const double o = cg_wsum(lane <= t ? w * vl[(size_t)lane * ms + hd * ks + c] : 0.0);
                    if (lane == 0) att[hd * ks + c] = o;
                }
            }
            asm volatile("" ::: "memory");
            const double* ob = Lp + 3 * blk; const double* ow = ob + ms;
            if (lane < ms) {                                              // attention output linear + residual
                double a = ob[lane];
                for (int i = 0; i < ms; ++i) a = fma(att[i], ow[i * ms + lane], a);
                h1[lane] = h[lane] + a;
            }
            asm volatile("" ::: "memory");
            const double* b1 = Lp + 4 * blk; const double* w1 = b1 + hs;
            const double* b2 = w1 + ms * hs; const double* w2 = b2 + ms;
            for (int j = lane; j < hs; j += 64) {                         // DenseBlock (src/autoregressive.py:32-48)
                double a = b1[j];
                for (int i = 0; i < ms; ++i) a = fma(h1[i], w1[i * hs + j], a);
                mid[j] = tanh(a);
            }
            asm volatile("" ::: "memory");
            if (lane < ms) {
                double a = b2[lane];
                for (int i = 0; i < hs; ++i) a = fma(mid[i], w2[i * ms + lane], a);
                h[lane] = h1[lane] + a;
            }
            asm volatile("" ::: "memory");
        }
        if (lane < ms) th[lane] = tanh(h[lane]);
        asm volatile("" ::: "memory");
    }
    return logp;
}

// ---- per-sample gradient of log p w.r.t. the parameters: jax.grad(log_prob) of src/sampler.py:65 (the classical score of the
// SR optimizer, src/sr.py:66-67, and -- contracted with weights -- jax.jacrev(classical_lossfn), main.py:277).
// One wave per sample: a forward pass that stashes the per-token activations in HBM (keys / values in LDS), then the reverse
// pass over the tokens n-2 .. 0 and the layers top down.  Key / value adjoints of a position are complete when the reverse
// pass reaches it (every later query has contributed), so one sweep suffices.  The gradient row (cg_van_num_params doubles,
// flat parameter order) is accumulated in HBM with a fixed entry <-> lane mapping: deterministic.
// G[i][j] += a[i] d[j]; the read-modify-writes of the gradient row (HBM / L2) go four at a time so that their latencies overlap
__device__ __forceinline__ void cg_van_outer(double* __restrict__ G, const double* a, const double* d, int nin, int nout, int lane) {
    const int tot = nin * nout;
    for (int e0 = lane; e0 < tot; e0 += 256) {
        double g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = e0 + 64 * u; g[u] = e < tot ? G[e] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 64 * u;
            if (e < tot) { const int i = e / nout, j = e - i * nout; G[e] = fma(a[i], d[j], g[u]); }
        }
    }
}
__device__ __forceinline__ void cg_van_vadd(double* __restrict__ G, const double* d, int nout, int lane) {
    for (int j = lane; j < nout; j += 64) G[j] += d[j];
}
// Where the gradient row of a sample is accumulated.  CgVanAccGlobal: in the row itself (HBM / L2), any model.  CgVanAccReg: in
// registers with the same entry <-> lane mapping and the same order of operations (so the same bits), stored once at the end -- for
// workgroups of at most four waves (one wave per SIMD: 512 registers per lane); the reverse pass of a token was ~120 dependent global
// read-modify-writes per lane.
struct CgVanAccGlobal {
    static constexpr int NL_STATIC = 0, MS_STATIC = 0, HS_STATIC = 0, NH_STATIC = 0;
    double* G; const CgVanModel& m; int lane;
    __device__ __forceinline__ CgVanAccGlobal(double* G_, const CgVanModel& m_, int lane_) : G(G_), m(m_), lane(lane_) {
        for (int e = lane; e < m.total; e += 64) G[e] = 0.0;
    }
    __device__ __forceinline__ void x1(int r, int j, double v) { (void)r; G[m.o_x1 + j] = v; }
    __device__ __forceinline__ void out(int r, int j, const double* th, double dy) {
        (void)r;
        G[m.o_ob + j] += dy;
        for (int i0 = 0; i0 < m.ms; i0 += 8) {                               // eight read-modify-writes in flight
            double g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = i0 + u < m.ms ? G[m.o_ow + (i0 + u) * m.M + j] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u < m.ms) G[m.o_ow + (i0 + u) * m.M + j] = fma(th[i0 + u], dy, g[u]);
        }
    }
    // slot: 0 query, 1 key, 2 value, 3 attention output (bias + ms x ms weights each), 4 mlp linear (hs, ms x hs), 5 mlp linear_1 (ms, hs x ms)
    __device__ __forceinline__ double* base(int l, int slot) const {
        const int blk = m.ms + m.ms * m.ms;
        return G + m.o_l[l] + (slot <= 4 ? slot * blk : 4 * blk + m.hs + m.ms * m.hs);
    }
    __device__ __forceinline__ void layer(int l, int slot, const double* a, const double* d, int nin, int nout) {
        double* b = base(l, slot);
        cg_van_vadd(b, d, nout, lane); cg_van_outer(b + nout, a, d, nin, nout, lane);
    }
    __device__ __forceinline__ void emb(const double* __restrict__ spc, const double* dpre) {
        cg_van_vadd(G + m.o_eb, dpre, m.ms, lane);
        for (int e = lane; e < m.dim * m.ms; e += 64) { const int c = e / m.ms, j = e - c * m.ms; G[m.o_ew + e] = fma(spc[c], dpre[j], G[m.o_ew + e]); }
    }
    __device__ __forceinline__ void finish() {}
};
// the same accumulation in the row itself, with the model dimensions known at compile time (any number of waves per workgroup)
template <int MS, int HS, int NL, int NH>
struct CgVanAccGlobalS : CgVanAccGlobal {
    static constexpr int NL_STATIC = NL, MS_STATIC = MS, HS_STATIC = HS, NH_STATIC = NH;
    __device__ __forceinline__ CgVanAccGlobalS(double* G_, const CgVanModel& m_, int lane_) : CgVanAccGlobal(G_, m_, lane_) {}
};
template <int MS, int HS, int NL, int MR>
struct CgVanAccReg {
    static constexpr int NL_STATIC = NL, MS_STATIC = MS, HS_STATIC = HS, NH_STATIC = 4;      // (the shipped models: four heads)
    static constexpr int KW = MS * MS / 64, KH = MS * HS / 64;
    static_assert(MS * MS % 64 == 0 && MS * HS % 64 == 0 && MS <= 64 && HS <= 64, "register accumulators: whole 64-lane sweeps");
    double* G; const CgVanModel& m; int lane;
    double ow[MR][MS], ob[MR], xh[MR];
    double w[NL][4][KW], w1[NL][KH], w2[NL][KH], bs[NL][6], eb, ew;
    __device__ __forceinline__ CgVanAccReg(double* G_, const CgVanModel& m_, int lane_) : G(G_), m(m_), lane(lane_) {
#pragma unroll
        for (int r = 0; r < MR; ++r) { ob[r] = 0.0; xh[r] = 0.0;
#pragma unroll
            for (int i = 0; i < MS; ++i) ow[r][i] = 0.0; }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < KW; ++k) w[l][q][k] = 0.0;
#pragma unroll
            for (int k = 0; k < KH; ++k) { w1[l][k] = 0.0; w2[l][k] = 0.0; }
#pragma unroll
            for (int q = 0; q < 6; ++q) bs[l][q] = 0.0;
        }
        eb = 0.0; ew = 0.0;
    }
    __device__ __forceinline__ void x1(int r, int j, double v) {
        (void)j;
#pragma unroll
        for (int rr = 0; rr < MR; ++rr) if (rr == r) xh[rr] = v;
    }
    __device__ __forceinline__ void out(int r, int j, const double* th, double dy) {
        (void)j;
#pragma unroll
        for (int rr = 0; rr < MR; ++rr)
            if (rr == r) {
                ob[rr] += dy;
#pragma unroll
                for (int i = 0; i < MS; ++i) ow[rr][i] = fma(th[i], dy, ow[rr][i]);
            }
    }
    template <int CNT, int NOUT>
    __device__ __forceinline__ void outer(double (&acc)[CNT], const double* a, const double* d) {
#pragma unroll
        for (int k = 0; k < CNT; ++k) { const int e = lane + 64 * k, i = e / NOUT, j = e - i * NOUT; acc[k] = fma(a[i], d[j], acc[k]); }
    }
    // (l and slot are compile-time constants after the unrolled layer loop)
    __device__ __forceinline__ void layer(int l, int slot, const double* a, const double* d, int nin, int nout) {
        (void)nin;
#pragma unroll
        for (int ll = 0; ll < NL; ++ll)
            if (ll == l) {
#pragma unroll
                for (int q = 0; q < 6; ++q) if (q == slot && lane < nout) bs[ll][q] += d[lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (q == slot) outer<KW, MS>(w[ll][q], a, d);
                if (slot == 4) outer<KH, HS>(w1[ll], a, d);
                if (slot == 5) outer<KH, MS>(w2[ll], a, d);
            }
    }
    __device__ __forceinline__ void emb(const double* __restrict__ spc, const double* dpre) {
        if (lane < MS) eb += dpre[lane];
        if (lane < m.dim * MS) { const int c = lane / MS, j = lane - c * MS; ew = fma(spc[c], dpre[j], ew); }
    }
    __device__ __forceinline__ void finish() {
        const int M = m.M, blk = MS + MS * MS;
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const int j = lane + 64 * r;
            if (j < M) {
                G[m.o_x1 + j] = xh[r]; G[m.o_ob + j] = ob[r];
#pragma unroll
                for (int i = 0; i < MS; ++i) G[m.o_ow + i * M + j] = ow[r][i];
            }
        }
        if (lane < MS) G[m.o_eb + lane] = eb;
        if (lane < m.dim * MS) G[m.o_ew + lane] = ew;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            double* Gl = G + m.o_l[l];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (lane < MS) Gl[q * blk + lane] = bs[l][q];
#pragma unroll
                for (int k = 0; k < KW; ++k) Gl[q * blk + MS + lane + 64 * k] = w[l][q][k];
            }
            if (lane < HS) Gl[4 * blk + lane] = bs[l][4];
#pragma unroll
            for (int k = 0; k < KH; ++k) Gl[4 * blk + HS + lane + 64 * k] = w1[l][k];
            if (lane < MS) Gl[4 * blk + HS + MS * HS + lane] = bs[l][5];
#pragma unroll
            for (int k = 0; k < KH; ++k) Gl[4 * blk + HS + MS * HS + MS + lane + 64 * k] = w2[l][k];
        }
    }
};
template <class ACC>
__device__ __forceinline__ void cg_van_gradient_t(const CgVanModel& m, const double* P, const double* __restrict__ sp,
                                                  const int* __restrict__ sidx, double* lw, double* __restrict__ stash, ACC& acc);
__device__ __forceinline__ void cg_van_gradient(const CgVanModel& m, const double* P, const double* __restrict__ sp,
                                                const int* __restrict__ sidx, double* lw, double* __restrict__ stash,
                                                double* __restrict__ G) {
    CgVanAccGlobal acc(G, m, (int)(threadIdx.x & 63));
    cg_van_gradient_t(m, P, sp, sidx, lw, stash, acc);
}
template <int MS, int HS, int NL, int NH>
__device__ __forceinline__ void cg_van_gradient_static(const CgVanModel& m, const double* P, const double* __restrict__ sp,
                                                       const int* __restrict__ sidx, double* lw, double* __restrict__ stash,
                                                       double* __restrict__ G) {
    CgVanAccGlobalS<MS, HS, NL, NH> acc(G, m, (int)(threadIdx.x & 63));
    cg_van_gradient_t(m, P, sp, sidx, lw, stash, acc);
}
template <int MS, int HS, int NL, int MR>
__device__ __forceinline__ void cg_van_gradient_reg(const CgVanModel& m, const double* P, const double* __restrict__ sp,
                                                    const int* __restrict__ sidx, double* lw, double* __restrict__ stash,
                                                    double* __restrict__ G) {
    CgVanAccReg<MS, HS, NL, MR> acc(G, m, (int)(threadIdx.x & 63));
    cg_van_gradient_t(m, P, sp, sidx, lw, stash, acc);
}
template <class ACC>
__device__ __forceinline__ void cg_van_gradient_t(const CgVanModel& m, const double* P, const double* __restrict__ sp,
                                                  const int* __restrict__ sidx, double* lw, double* __restrict__ stash, ACC& acc) {
    const int lane = threadIdx.x & 63;
    // (compile-time model dimensions in the register variant: the 16- and 32-term products below unroll, their weight loads overlap)
    const int ms = ACC::MS_STATIC > 0 ? ACC::MS_STATIC : m.ms, hs = ACC::HS_STATIC > 0 ? ACC::HS_STATIC : m.hs, M = m.M, n = m.n;
    const int nl = ACC::NL_STATIC > 0 ? ACC::NL_STATIC : m.nl, nh = ACC::NH_STATIC > 0 ? ACC::NH_STATIC : m.nh, ks = ms / nh;
    const int TS = cg_van_token_stash(m), LS = 4 * ms + hs + nh * n;       // stash per token / per layer inside it
    double* h = lw; double* q = h + ms; double* att = q + ms; double* h1 = att + ms; double* th = h1 + ms; double* mid = th + ms;
    double* dh = mid + hs; double* dh1 = dh + ms; double* dov = dh1 + ms; double* dq = dov + ms; double* dkt = dq + ms; double* dvt = dkt + ms;
    double* hin = dvt + ms; double* dpre = hin + ms; double* dyb = dpre + hs;      // dyb: hs doubles of scratch
    double* kc = dyb + hs; double* vc = kc + (size_t)nl * n * ms;
    double* dkc = vc + (size_t)nl * n * ms; double* dvc = dkc + (size_t)nl * n * ms;
    const double rsk = 1.0 / sqrt((double)ks);
    for (int e = lane; e < 2 * nl * n * ms; e += 64) dkc[e] = 0.0;                 // dkc and dvc are contiguous
    // ---- forward over the tokens 0 .. n-2, stashing what the reverse pass needs
    for (int t = 0; t + 1 < n; ++t) {
        double* st = stash + (size_t)t * TS;
        const int cur = sidx[t];
        if (lane < ms) {
            double a = P[m.o_eb + lane];
            for (int c = 0; c < m.dim; ++c) a = fma(sp[(size_t)cur * m.dim + c], P[m.o_ew + c * ms + lane], a);
            const double v = tanh(a); h[lane] = v; st[lane] = v;                   // h0
        }
        asm volatile("" ::: "memory");
        for (int l = 0; l < nl; ++l) {
            const double* Lp = P + m.o_l[l];
            const int blk = ms + ms * ms;
            double* sl = st + 2 * ms + (size_t)l * LS;                             // hin q o h1 mid A
            double* kcl = kc + ((size_t)l * n + t) * ms; double* vcl = vc + ((size_t)l * n + t) * ms;
            if (lane < ms) sl[lane] = h[lane];
            for (int idx = lane; idx < 3 * ms; idx += 64) {
                const int which = idx / ms, j = idx - which * ms;
                const double* bp = Lp + which * blk; const double* wp = bp + ms;
                double a = bp[j];
                for (int i = 0; i < ms; ++i) a = fma(h[i], wp[i * ms + j], a);
                if (which == 0) { q[j] = a; sl[ms + j] = a; } else if (which == 1) kcl[j] = a; else vcl[j] = a;
            }
            asm volatile("" ::: "memory");
            const double* kl = kc + (size_t)l * n * ms; const double* vl = vc + (size_t)l * n * ms;
            for (int hd = 0; hd < nh; ++hd) {
                double s = -INFINITY;
                if (lane <= t) {
                    s = 0.0;
                    for (int c = 0; c < ks; ++c) s = fma(q[hd * ks + c], kl[(size_t)lane * ms + hd * ks + c], s);
                    s *= rsk;
                }
                const double smx = cg_wmax(s);
                const double e = lane <= t ? exp(s - smx) : 0.0;
                const double w = e / cg_wsum(e);
                if (lane < n) sl[4 * ms + hs + hd * n + lane] = w;                 // A[hd][position]
                for (int c = 0; c < ks; ++c) {
                    const double o = cg_wsum(lane <= t ? w * vl[(size_t)lane * ms + hd * ks + c] : 0.0);
                    if (lane == 0) att[hd * ks + c] = o;
                }
            }
            asm volatile("" ::: "memory");
            const double* ob = Lp + 3 * blk; const double* ow = ob + ms;
            if (lane < ms) {
                sl[2 * ms + lane] = att[lane];
                double a = ob[lane];
                for (int i = 0; i < ms; ++i) a = fma(att[i], ow[i * ms + lane], a);
                const double v = h[lane] + a; h1[lane] = v; sl[3 * ms + lane] = v;
            }
            asm volatile("" ::: "memory");
            const double* b1 = Lp + 4 * blk; const double* w1 = b1 + hs;
            const double* b2 = w1 + ms * hs; const double* w2 = b2 + ms;
            for (int j = lane; j < hs; j += 64) {
                double a = b1[j];
                for (int i = 0; i < ms; ++i) a = fma(h1[i], w1[i * hs + j], a);
                const double v = tanh(a); mid[j] = v; sl[4 * ms + j] = v;
            }
            asm volatile("" ::: "memory");
            if (lane < ms) {
                double a = b2[lane];
                for (int i = 0; i < hs; ++i) a = fma(mid[i], w2[i * ms + lane], a);
                h[lane] = h1[lane] + a;
            }
            asm volatile("" ::: "memory");
        }
        if (lane < ms) st[ms + lane] = tanh(h[lane]);                              // th
        asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0);                                                  // the stash stores have left the wave
    // ---- d log p / d logits of position 0: one-hot minus softmax over the allowed orbitals of x1hat
    {
        const int s0 = sidx[0], hi = M - n;
        double lg[4]; double mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; lg[r] = (j < M && j <= hi) ? P[m.o_x1 + j] : -INFINITY; mx = fmax(mx, lg[r]); }
        mx = cg_wmax(mx);
        double z = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) z += (lg[r] > -INFINITY) ? exp(lg[r] - mx) : 0.0;
        z = cg_wsum(z);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; if (j < M && j <= hi) acc.x1(r, j, (j == s0 ? 1.0 : 0.0) - exp(lg[r] - mx) / z); }
    }
    // ---- reverse pass
    for (int t = n - 2; t >= 0; --t) {
        const double* st = stash + (size_t)t * TS;
        const int cur = sidx[t], nxt = sidx[t + 1], hi = t + 1 + M - n;
        if (lane < ms) { th[lane] = st[ms + lane]; }
        asm volatile("" ::: "memory");
        // dy = d log p / d logits of position t+1 (the output of token t); output layer
        double dy[4], lg[4]; double mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = lane + 64 * r;
            double v = -INFINITY;
            if (j < M && j > cur && j <= hi) { v = P[m.o_ob + j]; for (int i = 0; i < ms; ++i) v = fma(th[i], P[m.o_ow + i * M + j], v); }
            lg[r] = v; mx = fmax(mx, v);
        }
        mx = cg_wmax(mx);
        double z = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) z += (lg[r] > -INFINITY) ? exp(lg[r] - mx) : 0.0;
        z = cg_wsum(z);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = lane + 64 * r;
            dy[r] = (lg[r] > -INFINITY) ? (j == nxt ? 1.0 : 0.0) - exp(lg[r] - mx) / z : 0.0;
            if (j < M && dy[r] != 0.0) acc.out(r, j, th, dy[r]);
        }
        for (int i = 0; i < ms; ++i) {                                             // dh = (Wout dy) (1 - th^2)
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; if (j < M) a = fma(P[m.o_ow + i * M + j], dy[r], a); }
            a = cg_wsum(a);
            if (lane == i) dh[i] = a * (1.0 - th[i] * th[i]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int li = 0; li < (ACC::NL_STATIC > 0 ? ACC::NL_STATIC : nl); ++li) {
            const int l = (ACC::NL_STATIC > 0 ? ACC::NL_STATIC : nl) - 1 - li;
            const double* Lp = P + m.o_l[l];
            const int blk = ms + ms * ms;
            const double* sl = st + 2 * ms + (size_t)l * LS;
            if (lane < ms) { hin[lane] = sl[lane]; q[lane] = sl[ms + lane]; att[lane] = sl[2 * ms + lane]; h1[lane] = sl[3 * ms + lane]; }
            for (int j = lane; j < hs; j += 64) mid[j] = sl[4 * ms + j];
            asm volatile("" ::: "memory");
            const double* b1 = Lp + 4 * blk; const double* w1 = b1 + hs;
            const double* b2 = w1 + ms * hs; const double* w2 = b2 + ms;
            // DenseBlock: h = h1 + W2^T tanh(W1^T h1 + b1) + b2
            acc.layer(l, 5, mid, dh, hs, ms);
            for (int i = lane; i < hs; i += 64) {
                double a = 0.0;
                for (int j = 0; j < ms; ++j) a = fma(w2[i * ms + j], dh[j], a);
                dpre[i] = a * (1.0 - mid[i] * mid[i]);
            }
            asm volatile("" ::: "memory");
            acc.layer(l, 4, h1, dpre, ms, hs);
            if (lane < ms) {
                double a = dh[lane];
                for (int i = 0; i < hs; ++i) a = fma(w1[lane * hs + i], dpre[i], a);
                dh1[lane] = a;
            }
            asm volatile("" ::: "memory");
            // attention output linear
            const double* ow = Lp + 3 * blk + ms;
            acc.layer(l, 3, att, dh1, ms, ms);
            if (lane < ms) {
                double a = 0.0;
                for (int j = 0; j < ms; ++j) a = fma(ow[lane * ms + j], dh1[j], a);
                dov[lane] = a;
            }
            asm volatile("" ::: "memory");
            // attention of query t over the positions <= t (lane = position)
            const double* kl = kc + (size_t)l * n * ms; const double* vl = vc + (size_t)l * n * ms;
            double* dkl = dkc + (size_t)l * n * ms; double* dvl = dvc + (size_t)l * n * ms;
            for (int hd = 0; hd < nh; ++hd) {
                const bool on = lane <= t;
                const double a = on ? sl[4 * ms + hs + hd * n + lane] : 0.0;
                double dA = 0.0;
                if (on) for (int c = 0; c < ks; ++c) dA = fma(dov[hd * ks + c], vl[(size_t)lane * ms + hd * ks + c], dA);
                const double sada = cg_wsum(a * dA);
                const double dS = a * (dA - sada) * rsk;
                for (int c = 0; c < ks; ++c) {
                    const double dqc = cg_wsum(on ? dS * kl[(size_t)lane * ms + hd * ks + c] : 0.0);
                    if (lane == 0) dq[hd * ks + c] = dqc;
                    if (on) {
                        dvl[(size_t)lane * ms + hd * ks + c] = fma(a, dov[hd * ks + c], dvl[(size_t)lane * ms + hd * ks + c]);
                        dkl[(size_t)lane * ms + hd * ks + c] = fma(dS, q[hd * ks + c], dkl[(size_t)lane * ms + hd * ks + c]);
                    }
                }
            }
            asm volatile("" ::: "memory");
            if (lane < ms) { dkt[lane] = dkl[(size_t)t * ms + lane]; dvt[lane] = dvl[(size_t)t * ms + lane]; }
            asm volatile("" ::: "memory");
            // query / key / value linears of token t
            acc.layer(l, 0, hin, dq, ms, ms);
            acc.layer(l, 1, hin, dkt, ms, ms);
            acc.layer(l, 2, hin, dvt, ms, ms);
            if (lane < ms) {
                const double* wq = Lp + ms; const double* wk = Lp + blk + ms; const double* wv = Lp + 2 * blk + ms;
                double a = dh1[lane];
                for (int j = 0; j < ms; ++j) a = fma(wq[lane * ms + j], dq[j], fma(wk[lane * ms + j], dkt[j], fma(wv[lane * ms + j], dvt[j], a)));
                dyb[lane] = a;
            }
            asm volatile("" ::: "memory");
            if (lane < ms) dh[lane] = dyb[lane];
            asm volatile("" ::: "memory");
        }
        // embedding
        if (lane < ms) { const double h0 = st[lane]; dpre[lane] = dh[lane] * (1.0 - h0 * h0); }
        asm volatile("" ::: "memory");
        acc.emb(sp + (size_t)cur * m.dim, dpre);
        asm volatile("" ::: "memory");
    }
    acc.finish();
}
#endif
