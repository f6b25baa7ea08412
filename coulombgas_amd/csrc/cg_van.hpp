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

#if defined(__HIPCC__)
__device__ __forceinline__ double cg_wsum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ double cg_wmax(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// One sample on one wave.  SAMPLE: draws state_idx[0..n) (written to sidx) with Gumbel-max noise from `unif` (n x M, parity
// mode) or the Philox stream (seed, stream); otherwise reads sidx.  Returns log p(state_idx) in every lane.
template <bool SAMPLE>
__device__ __forceinline__ double cg_van_sequence(const CgVanModel& m, const double* P, const double* __restrict__ sp,
                                                  int* __restrict__ sidx, double* lw, const double* __restrict__ unif,
                                                  uint64_t seed, uint64_t stream) {
    const int lane = threadIdx.x & 63;
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
#endif
