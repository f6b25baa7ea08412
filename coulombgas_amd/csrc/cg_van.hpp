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
__device__ __forceinline__ void cg_van_gradient(const CgVanModel& m, const double* P, const double* __restrict__ sp,
                                                const int* __restrict__ sidx, double* lw, double* __restrict__ stash,
                                                double* __restrict__ G) {
    const int lane = threadIdx.x & 63;
    const int ms = m.ms, hs = m.hs, M = m.M, n = m.n, ks = m.ks, nl = m.nl, nh = m.nh;
    const int TS = cg_van_token_stash(m), LS = 4 * ms + hs + nh * n;       // stash per token / per layer inside it
    double* h = lw; double* q = h + ms; double* att = q + ms; double* h1 = att + ms; double* th = h1 + ms; double* mid = th + ms;
    double* dh = mid + hs; double* dh1 = dh + ms; double* dov = dh1 + ms; double* dq = dov + ms; double* dkt = dq + ms; double* dvt = dkt + ms;
    double* hin = dvt + ms; double* dpre = hin + ms; double* dyb = dpre + hs;      // dyb: hs doubles of scratch
    double* kc = dyb + hs; double* vc = kc + (size_t)nl * n * ms;
    double* dkc = vc + (size_t)nl * n * ms; double* dvc = dkc + (size_t)nl * n * ms;
    const double rsk = 1.0 / sqrt((double)ks);
    for (int e = lane; e < m.total; e += 64) G[e] = 0.0;
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
        for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; if (j < M && j <= hi) G[m.o_x1 + j] = (j == s0 ? 1.0 : 0.0) - exp(lg[r] - mx) / z; }
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
            if (j < M && dy[r] != 0.0) {
                G[m.o_ob + j] += dy[r];
                for (int i0 = 0; i0 < ms; i0 += 8) {                               // eight read-modify-writes in flight
                    double g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = i0 + u < ms ? G[m.o_ow + (i0 + u) * M + j] : 0.0;
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (i0 + u < ms) G[m.o_ow + (i0 + u) * M + j] = fma(th[i0 + u], dy[r], g[u]);
                }
            }
        }
        for (int i = 0; i < ms; ++i) {                                             // dh = (Wout dy) (1 - th^2)
            double a = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; if (j < M) a = fma(P[m.o_ow + i * M + j], dy[r], a); }
            a = cg_wsum(a);
            if (lane == i) dh[i] = a * (1.0 - th[i] * th[i]);
        }
        asm volatile("" ::: "memory");
        for (int l = nl - 1; l >= 0; --l) {
            const double* Lp = P + m.o_l[l];
            double* Gl = G + m.o_l[l];
            const int blk = ms + ms * ms;
            const double* sl = st + 2 * ms + (size_t)l * LS;
            if (lane < ms) { hin[lane] = sl[lane]; q[lane] = sl[ms + lane]; att[lane] = sl[2 * ms + lane]; h1[lane] = sl[3 * ms + lane]; }
            for (int j = lane; j < hs; j += 64) mid[j] = sl[4 * ms + j];
            asm volatile("" ::: "memory");
            const double* b1 = Lp + 4 * blk; const double* w1 = b1 + hs;
            const double* b2 = w1 + ms * hs; const double* w2 = b2 + ms;
            double* Gb1 = Gl + 4 * blk; double* Gw1 = Gb1 + hs; double* Gb2 = Gw1 + ms * hs; double* Gw2 = Gb2 + ms;
            // DenseBlock: h = h1 + W2^T tanh(W1^T h1 + b1) + b2
            cg_van_vadd(Gb2, dh, ms, lane); cg_van_outer(Gw2, mid, dh, hs, ms, lane);
            for (int i = lane; i < hs; i += 64) {
                double a = 0.0;
                for (int j = 0; j < ms; ++j) a = fma(w2[i * ms + j], dh[j], a);
                dpre[i] = a * (1.0 - mid[i] * mid[i]);
            }
            asm volatile("" ::: "memory");
            cg_van_vadd(Gb1, dpre, hs, lane); cg_van_outer(Gw1, h1, dpre, ms, hs, lane);
            if (lane < ms) {
                double a = dh[lane];
                for (int i = 0; i < hs; ++i) a = fma(w1[lane * hs + i], dpre[i], a);
                dh1[lane] = a;
            }
            asm volatile("" ::: "memory");
            // attention output linear
            const double* ow = Lp + 3 * blk + ms;
            double* Gob = Gl + 3 * blk; double* Gow = Gob + ms;
            cg_van_vadd(Gob, dh1, ms, lane); cg_van_outer(Gow, att, dh1, ms, ms, lane);
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
            cg_van_vadd(Gl, dq, ms, lane); cg_van_outer(Gl + ms, hin, dq, ms, ms, lane);
            cg_van_vadd(Gl + blk, dkt, ms, lane); cg_van_outer(Gl + blk + ms, hin, dkt, ms, ms, lane);
            cg_van_vadd(Gl + 2 * blk, dvt, ms, lane); cg_van_outer(Gl + 2 * blk + ms, hin, dvt, ms, ms, lane);
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
        cg_van_vadd(G + m.o_eb, dpre, ms, lane);
        for (int e = lane; e < m.dim * ms; e += 64) { const int c = e / ms, j = e - c * ms; G[m.o_ew + e] = fma(sp[(size_t)cur * m.dim + c], dpre[j], G[m.o_ew + e]); }
        asm volatile("" ::: "memory");
    }
}
#endif
