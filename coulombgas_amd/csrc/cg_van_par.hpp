// cg_van_par.hpp -- per-sample gradient of log p of the autoregressive Transformer density matrix with the POSITIONS IN PARALLEL.
//
// Reference: jax.grad(log_prob) of src/sampler.py:40-46, 65 (the classical score of src/sr.py:66-67 and, contracted with weights,
// jax.jacrev(classical_lossfn), main.py:277); network: src/autoregressive.py:50-96.
//
// cg_van.hpp runs one wave per sample with the tokens in sequence (the shape the SAMPLER needs: token t depends on the draw of t - 1),
// forward and reverse: 2 (n - 1) dependent token passes, each a chain of LDS round trips on a quarter of the wave.  For the gradient the
// sample is given (teacher forcing), so nothing orders the positions: here lane t of the wave owns token t for the whole pass,
//   * dense layers: per-lane 16- / 32-wide vectors in registers against weights read through the scalar cache (wave-uniform addresses);
//   * attention: keys / values of all positions in LDS, lane t walks the positions j <= t with broadcast reads; in the reverse pass the
//     query-side lanes publish a_tj and dS_tj through an n x (n + 1) LDS matrix that the key-side lanes read transposed (lane j sums over
//     the queries t >= j): no atomics, fixed order;
//   * weight gradients dW = sum_t a_t (x) d_t are products over the position axis: f64 MFMA with K = position (the operands pass through
//     LDS once to get from "lane = position" to the MFMA layout), each 16 x 16 tile of the score row written exactly once;
//   * activations the reverse pass needs are stashed in HBM position-minor ([feature][lane]: 512-byte coalesced rows).
// The shipped architecture only (model size 16, hidden 32, two layers, four heads, dim 2; n <= 64, M <= 256): the sizes of every run the
// reference published.  Deterministic (fixed summation orders).
#pragma once
#include "cg_van.hpp"

#if defined(__HIPCC__)
struct CgVanPar {
    static constexpr int MS = 16, HS = 32, NL = 2, NH = 4, KS = 4, DIM = 2;
    // stash rows (64 doubles each, one per lane): h0, th, per layer [hin q att h1 (16 each) mid (32) | e[NH][n] | zinv[NH]], then lg / e [M]
    static __host__ __device__ int layer_rows(int n) { return 4 * MS + HS + NH * n + NH; }
    static __host__ __device__ int rows(int n, int M) { return 2 * MS + NL * layer_rows(n) + M; }
    static __host__ __device__ size_t stash_doubles(int n, int M) { return (size_t)rows(n, M) * 64; }
    // LDS doubles of one wave: K, V caches [NL][n][MS] x 2 | transpose / operand region 64 (n - 1) | broadcast buffers 36 (n - 1)
    static __host__ __device__ int wave_doubles(int n) { return ((2 * NL * n * MS + 64 * (n - 1) + 36 * (n - 1)) + 1) & ~1; }
    // packed variant (short sequences): S = 64 / (n - 1) samples share a wave, lane = g (n - 1) + t.  LDS doubles of one wave:
    // K, V caches [L][MS] x 2 of ONE layer (the reverse pass forms them again from the stashed layer input: 2 x 256 multiply-adds per lane
    // against 31 KB of LDS, i.e. four waves per CU instead of two) | region: MFMA operand rows XA [L][32], XB [L][16], aliased by the transpose matrices [L][n] and
    // the query / cotangent rows [L][MS] x 2 of the attention's reverse pass      (L = S (n - 1) <= 64)
    static __host__ __device__ int packed_samples(int n) { return n >= 3 && n <= 33 ? 64 / (n - 1) : 1; }
    static __host__ __device__ int packed_region(int n) { const int nt = n - 1, L = packed_samples(n) * nt; const int a = 48 * L, c = L * (nt + 1 + 2 * MS); return ((a > c ? a : c) + 1) & ~1; }
    static __host__ __device__ int packed_wave_doubles(int n) { const int L = packed_samples(n) * (n - 1); return 2 * L * MS + packed_region(n); }
    static __host__ bool serves(const CgVanModel& m) {
        return m.ms == MS && m.hs == HS && m.nl == NL && m.nh == NH && m.dim == DIM && m.n >= 2 && m.n <= 64 && m.M <= 256 && m.M >= m.n;
    }
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef double cg_vp_d4 __attribute__((ext_vector_type(4)));
// out[(i0 + row) * ld + jo + col] = sum_(t < nt) XA[t * WA + i0 + row] * XB[t * WB + j0 + col]   (one 16 x 16 tile, K = position, nt <= 64;
// columns jo + col < jmax are stored)
__device__ __forceinline__ void cg_vp_tile(const double* XA, int WA, int i0, const double* XB, int WB, int j0, int nt, double* __restrict__ out, int ld, int jo, int jmax) {
    const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
    cg_vp_d4 acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
        const int tt = 4 * ks + kq; const bool ok = tt < nt; const int tc = ok ? tt : 0;
        const double a = ok ? XA[tc * WA + i0 + col] : 0.0, bv = ok ? XB[tc * WB + j0 + col] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
    }
    if (jo + col < jmax) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(i0 + kq + 4 * r) * ld + jo + col] = acc[r];
    }
}
// lane t publishes its vector v[0..W) as row t of X (row width WX)
template <int W>
__device__ __forceinline__ void cg_vp_put(double* X, int WX, int t, bool act, const double (&v)[W], int off = 0) {
    if (act) {
#pragma unroll
        for (int i = 0; i < W; ++i) X[t * WX + off + i] = v[i];
    }
}
// sum over the positions of one per-lane value; every lane gets it
__device__ __forceinline__ double cg_vp_sum(double v, bool act) { return cg_wsum(act ? v : 0.0); }

// One sample on one wave.  lw: CgVanPar::wave_doubles(n) doubles of LDS; st: CgVanPar::stash_doubles doubles of HBM; G: the score row
__device__ __forceinline__ void cg_van_grad_par(const CgVanModel& m, const double* __restrict__ P, const double* __restrict__ sp,
                                                const int* __restrict__ sidx, double* lw, double* __restrict__ st, double* __restrict__ G) {
    constexpr int MS = CgVanPar::MS, HS = CgVanPar::HS, NL = CgVanPar::NL, NH = CgVanPar::NH, KS = CgVanPar::KS;
    constexpr int blk = MS + MS * MS;
    const int lane = threadIdx.x & 63, n = m.n, M = m.M, nt = n - 1;      // tokens 0 .. nt - 1 feed the conditionals of positions 1 .. n - 1
    const int t = lane; const bool act = t < nt;
    const int cur = sidx[act ? t : 0], nxt = sidx[act ? t + 1 : 0], hi = t + 1 + M - n;
    const double rsk = 0.5;                                                 // 1 / sqrt(KS)
    double* Kc = lw; double* Vc = Kc + NL * n * MS; double* Pm = Vc + NL * n * MS; double* Qb = Pm + 64 * nt; double* Db = Qb + MS * nt; double* Sd = Db + MS * nt;
    const int PW = nt + 1;                                                  // row stride of the transpose matrix (odd multiples of 8 bytes apart)
    double* XA = Pm; double* XB = Pm + 32 * nt;                             // MFMA operand rows [t][32] (the transpose matrix is idle then)
    const int LR = CgVanPar::layer_rows(n);
    auto srow = [&](int r) -> double* { return st + (size_t)r * 64 + lane; };
    // ------------------------------------------------------------------ forward
    double h[MS];
    {
        const double x0 = sp[(size_t)cur * 2], x1 = sp[(size_t)cur * 2 + 1];
#pragma unroll
        for (int j = 0; j < MS; ++j) { h[j] = tanh(fma(x1, P[m.o_ew + MS + j], fma(x0, P[m.o_ew + j], P[m.o_eb + j]))); *srow(j) = h[j]; }
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const double* Lp = P + m.o_l[l];
        const int R0 = 2 * MS + l * LR;
        double q[MS], kk[MS], vv[MS];
#pragma unroll
        for (int j = 0; j < MS; ++j) { *srow(R0 + j) = h[j]; q[j] = Lp[j]; kk[j] = Lp[blk + j]; vv[j] = Lp[2 * blk + j]; }
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int j = 0; j < MS; ++j) {
                q[j] = fma(h[i], Lp[MS + i * MS + j], q[j]); kk[j] = fma(h[i], Lp[blk + MS + i * MS + j], kk[j]); vv[j] = fma(h[i], Lp[2 * blk + MS + i * MS + j], vv[j]);
            }
        double* Kl = Kc + l * n * MS; double* Vl = Vc + l * n * MS;
        if (act) {
#pragma unroll
            for (int j = 0; j < MS; ++j) { Kl[t * MS + j] = kk[j]; Vl[t * MS + j] = vv[j]; }
        }
#pragma unroll
        for (int j = 0; j < MS; ++j) *srow(R0 + MS + j) = q[j];
        asm volatile("" ::: "memory");
        double att[MS];
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {                                   // causal attention of query t over the positions j <= t
            double mx = -INFINITY;
            for (int j = 0; j < nt; ++j) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) s = fma(q[hd * KS + c], Kl[j * MS + hd * KS + c], s);
                mx = j <= t ? fmax(mx, s * rsk) : mx;
            }
            double z = 0.0, o[KS];
#pragma unroll
            for (int c = 0; c < KS; ++c) o[c] = 0.0;
            for (int j = 0; j < nt; ++j) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) s = fma(q[hd * KS + c], Kl[j * MS + hd * KS + c], s);
                const double e = j <= t ? cg_exp_nonpos(s * rsk - mx) : 0.0;
                z += e;
#pragma unroll
                for (int c = 0; c < KS; ++c) o[c] = fma(e, Vl[j * MS + hd * KS + c], o[c]);
                *srow(R0 + 4 * MS + HS + hd * n + j) = e;
            }
            const double zi = 1.0 / z;
            *srow(R0 + 4 * MS + HS + NH * n + hd) = zi;
#pragma unroll
            for (int c = 0; c < KS; ++c) att[hd * KS + c] = o[c] * zi;
        }
        double h1[MS];
        {
            const double* ob = Lp + 3 * blk; const double* ow = ob + MS;
#pragma unroll
            for (int j = 0; j < MS; ++j) { *srow(R0 + 2 * MS + j) = att[j]; h1[j] = h[j] + ob[j]; }
#pragma unroll
            for (int i = 0; i < MS; ++i)
#pragma unroll
                for (int j = 0; j < MS; ++j) h1[j] = fma(att[i], ow[i * MS + j], h1[j]);
#pragma unroll
            for (int j = 0; j < MS; ++j) *srow(R0 + 3 * MS + j) = h1[j];
        }
        const double* b1 = Lp + 4 * blk; const double* w1 = b1 + HS; const double* b2 = w1 + MS * HS; const double* w2 = b2 + MS;
        double mid[HS];
#pragma unroll
        for (int j = 0; j < HS; ++j) mid[j] = b1[j];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int j = 0; j < HS; ++j) mid[j] = fma(h1[i], w1[i * HS + j], mid[j]);
#pragma unroll
        for (int j = 0; j < HS; ++j) { mid[j] = tanh(mid[j]); *srow(R0 + 4 * MS + j) = mid[j]; }
#pragma unroll
        for (int j = 0; j < MS; ++j) h[j] = h1[j] + b2[j];
#pragma unroll
        for (int i = 0; i < HS; ++i)
#pragma unroll
            for (int j = 0; j < MS; ++j) h[j] = fma(mid[i], w2[i * MS + j], h[j]);
    }
    double th[MS];
#pragma unroll
    for (int j = 0; j < MS; ++j) { th[j] = tanh(h[j]); *srow(MS + j) = th[j]; }
    // ------------------------------------------------------------------ conditionals of positions 1 .. n - 1: logits, softmax
    const int RL = 2 * MS + NL * LR;
    double mx = -INFINITY;
    for (int j = 0; j < M; ++j) {
        double v = P[m.o_ob + j];
#pragma unroll
        for (int i = 0; i < MS; ++i) v = fma(th[i], P[m.o_ow + i * M + j], v);
        const bool ok = j > cur && j <= hi;
        *srow(RL + j) = v;
        mx = ok ? fmax(mx, v) : mx;
    }
    double z = 0.0;
    for (int j = 0; j < M; ++j) {
        const bool ok = j > cur && j <= hi;
        const double e = ok ? cg_exp_nonpos(*srow(RL + j) - mx) : 0.0;
        z += e;
        *srow(RL + j) = e;
    }
    const double zinv = act ? 1.0 / z : 0.0;                               // (lanes beyond the last token contribute nothing anywhere)
    __builtin_amdgcn_s_waitcnt(0);
    // ------------------------------------------------------------------ reverse
    // position 0: one-hot minus softmax over the allowed orbitals of x1hat (lane = orbital)
    {
        const int s0 = sidx[0], h0i = M - n;
        double lg[4]; double m0 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; lg[r] = (j < M && j <= h0i) ? P[m.o_x1 + j] : -INFINITY; m0 = fmax(m0, lg[r]); }
        m0 = cg_wmax(m0);
        double z0 = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) z0 += (lg[r] > -INFINITY) ? exp(lg[r] - m0) : 0.0;
        z0 = cg_wsum(z0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; if (j < M) G[m.o_x1 + j] = (j <= h0i) ? (j == s0 ? 1.0 : 0.0) - exp(lg[r] - m0) / z0 : 0.0; }
    }
    // output layer: dy_t[j] = [j = next] - softmax, in chunks of 16 orbitals; d ow = sum_t th_t (x) dy_t on the matrix cores
    double dh[MS];
#pragma unroll
    for (int i = 0; i < MS; ++i) dh[i] = 0.0;
    cg_vp_put<MS>(XA, 32, t, act, th);
    for (int j0 = 0; j0 < M; j0 += 16) {
        double dy[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = j0 + jj; const bool ok = act && j < M && j > cur && j <= hi;
            dy[jj] = ok ? (j == nxt ? 1.0 : 0.0) - *srow(RL + (j < M ? j : 0)) * zinv : 0.0;
            if (j < M) {
#pragma unroll
                for (int i = 0; i < MS; ++i) dh[i] = fma(P[m.o_ow + i * M + j], dy[jj], dh[i]);
            }
        }
        cg_vp_put<16>(XB, 32, t, act, dy);
        asm volatile("" ::: "memory");
        cg_vp_tile(XA, 32, 0, XB, 32, 0, nt, G + m.o_ow + j0, M, 0, M - j0);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) { const double s = cg_wsum(dy[jj]); if (lane == 0 && j0 + jj < M) G[m.o_ob + j0 + jj] = s; }
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < MS; ++i) dh[i] *= 1.0 - th[i] * th[i];
#pragma unroll
    for (int li = 0; li < NL; ++li) {
        const int l = NL - 1 - li;
        const double* Lp = P + m.o_l[l];
        double* Gl = G + m.o_l[l];
        const int R0 = 2 * MS + l * LR;
        const double* b1 = Lp + 4 * blk; const double* w1 = b1 + HS; const double* w2 = w1 + MS * HS + MS;
        double* Kl = Kc + l * n * MS; double* Vl = Vc + l * n * MS;
        // ---- DenseBlock: h = h1 + W2^T tanh(W1^T h1 + b1) + b2
        double dh1[MS];
        {
            double mid[HS], dpre[HS], h1[MS];
#pragma unroll
            for (int j = 0; j < HS; ++j) mid[j] = *srow(R0 + 4 * MS + j);
#pragma unroll
            for (int j = 0; j < MS; ++j) h1[j] = *srow(R0 + 3 * MS + j);
            cg_vp_put<HS>(XA, 32, t, act, mid); cg_vp_put<MS>(XB, 32, t, act, dh);
            asm volatile("" ::: "memory");
            double* g2 = Gl + 4 * blk + HS + MS * HS;                      // m2b[MS], m2w[HS][MS]
            cg_vp_tile(XA, 32, 0, XB, 32, 0, nt, g2 + MS, MS, 0, MS); cg_vp_tile(XA, 32, 16, XB, 32, 0, nt, g2 + MS, MS, 0, MS);
#pragma unroll
            for (int j = 0; j < MS; ++j) { const double s = cg_vp_sum(dh[j], act); if (lane == 0) g2[j] = s; }
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < MS; ++j) a = fma(w2[i * MS + j], dh[j], a);
                dpre[i] = a * (1.0 - mid[i] * mid[i]);
            }
            asm volatile("" ::: "memory");
            cg_vp_put<MS>(XA, 32, t, act, h1); cg_vp_put<HS>(XB, 32, t, act, dpre);
            asm volatile("" ::: "memory");
            double* g1 = Gl + 4 * blk;                                       // m1b[HS], m1w[MS][HS]
            cg_vp_tile(XA, 32, 0, XB, 32, 0, nt, g1 + HS, HS, 0, HS); cg_vp_tile(XA, 32, 0, XB, 32, 16, nt, g1 + HS, HS, 16, HS);
#pragma unroll
            for (int j = 0; j < HS; ++j) { const double s = cg_vp_sum(dpre[j], act); if (lane == 0) g1[j] = s; }
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                double a = dh[i];
#pragma unroll
                for (int j = 0; j < HS; ++j) a = fma(w1[i * HS + j], dpre[j], a);
                dh1[i] = a;
            }
            asm volatile("" ::: "memory");
        }
        // ---- attention output linear
        double dov[MS], q[MS];
        {
            double att[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) { att[j] = *srow(R0 + 2 * MS + j); q[j] = *srow(R0 + MS + j); }
            cg_vp_put<MS>(XA, 32, t, act, att); cg_vp_put<MS>(XB, 32, t, act, dh1);
            asm volatile("" ::: "memory");
            double* go = Gl + 3 * blk;
            cg_vp_tile(XA, 32, 0, XB, 32, 0, nt, go + MS, MS, 0, MS);
#pragma unroll
            for (int j = 0; j < MS; ++j) { const double s = cg_vp_sum(dh1[j], act); if (lane == 0) go[j] = s; }
            const double* ow = Lp + 3 * blk + MS;
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < MS; ++j) a = fma(ow[i * MS + j], dh1[j], a);
                dov[i] = a;
            }
            asm volatile("" ::: "memory");
        }
        // ---- attention: query side (lane t over j <= t) -> dq;  key side (lane j over the queries t >= j) -> dk, dv
        double dq[MS], dk[MS], dv[MS];
#pragma unroll
        for (int i = 0; i < MS; ++i) { dq[i] = 0.0; dk[i] = 0.0; dv[i] = 0.0; }
        cg_vp_put<MS>(Qb, MS, t, act, q); cg_vp_put<MS>(Db, MS, t, act, dov);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {
            const double zi = *srow(R0 + 4 * MS + HS + NH * n + hd);
            const int RA = R0 + 4 * MS + HS + hd * n;
            double sada = 0.0;
            for (int j = 0; j < nt; ++j) {                                  // (a_tj = 0 beyond j = t: stashed as e = 0)
                const double a = *srow(RA + j) * zi;
                double dA = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) dA = fma(dov[hd * KS + c], Vl[j * MS + hd * KS + c], dA);
                sada = fma(a, dA, sada);
                if (act) Pm[t * PW + j] = a;
            }
            asm volatile("" ::: "memory");
            if (act) {                                                      // key side, values: dv_j += sum_t a_tj dov_t
                for (int tq = 0; tq < nt; ++tq) {
                    const double a = Pm[tq * PW + t];
#pragma unroll
                    for (int c = 0; c < KS; ++c) dv[hd * KS + c] = fma(a, Db[tq * MS + hd * KS + c], dv[hd * KS + c]);
                }
            }
            asm volatile("" ::: "memory");
            for (int j = 0; j < nt; ++j) {
                const double a = *srow(RA + j) * zi;
                double dA = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) dA = fma(dov[hd * KS + c], Vl[j * MS + hd * KS + c], dA);
                const double dS = a * (dA - sada) * rsk;
#pragma unroll
                for (int c = 0; c < KS; ++c) dq[hd * KS + c] = fma(dS, Kl[j * MS + hd * KS + c], dq[hd * KS + c]);
                if (act) Pm[t * PW + j] = dS;
            }
            asm volatile("" ::: "memory");
            if (act) {                                                      // key side, keys: dk_j += sum_t dS_tj q_t
                for (int tq = 0; tq < nt; ++tq) {
                    const double dS = Pm[tq * PW + t];
#pragma unroll
                    for (int c = 0; c < KS; ++c) dk[hd * KS + c] = fma(dS, Qb[tq * MS + hd * KS + c], dk[hd * KS + c]);
                }
            }
            asm volatile("" ::: "memory");
        }
        // ---- query / key / value linears of token t
        {
            double hin[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) hin[j] = *srow(R0 + j);
            cg_vp_put<MS>(XA, 32, t, act, hin); cg_vp_put<MS>(XB, 32, t, act, dq); cg_vp_put<MS>(XB, 32, t, act, dk, 16);
            asm volatile("" ::: "memory");
            cg_vp_tile(XA, 32, 0, XB, 32, 0, nt, Gl + MS, MS, 0, MS); cg_vp_tile(XA, 32, 0, XB, 32, 16, nt, Gl + blk + MS, MS, 0, MS);
            asm volatile("" ::: "memory");
            cg_vp_put<MS>(XB, 32, t, act, dv);
            asm volatile("" ::: "memory");
            cg_vp_tile(XA, 32, 0, XB, 32, 0, nt, Gl + 2 * blk + MS, MS, 0, MS);
#pragma unroll
            for (int j = 0; j < MS; ++j) {
                const double s0 = cg_vp_sum(dq[j], act), s1 = cg_vp_sum(dk[j], act), s2 = cg_vp_sum(dv[j], act);
                if (lane == 0) { Gl[j] = s0; Gl[blk + j] = s1; Gl[2 * blk + j] = s2; }
            }
            const double* wq = Lp + MS; const double* wk = Lp + blk + MS; const double* wv = Lp + 2 * blk + MS;
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                double a = dh1[i];
#pragma unroll
                for (int j = 0; j < MS; ++j) a = fma(wq[i * MS + j], dq[j], fma(wk[i * MS + j], dk[j], fma(wv[i * MS + j], dv[j], a)));
                dh[i] = a;
            }
            asm volatile("" ::: "memory");
        }
    }
    // ---- embedding: h0 = tanh(eb + sp[cur] ew)
    {
        const double x0 = sp[(size_t)cur * 2], x1 = sp[(size_t)cur * 2 + 1];
#pragma unroll
        for (int j = 0; j < MS; ++j) {
            const double h0 = *srow(j), dp = dh[j] * (1.0 - h0 * h0);
            const double s0 = cg_vp_sum(dp, act), s1 = cg_vp_sum(x0 * dp, act), s2 = cg_vp_sum(x1 * dp, act);
            if (lane == 0) { G[m.o_eb + j] = s0; G[m.o_ew + j] = s1; G[m.o_ew + MS + j] = s2; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Packed variant for short sequences (n - 1 <= 32 tokens): S = 64 / (n - 1) samples on one wave, lane = g (n - 1) + t owns token t of
// sample g.  Per-lane work (dense layers, logits) is that of cg_van_grad_par; everything that runs over the positions of ONE sample is
// confined to the lanes of its group: attention over the keys / values of the group's rows of the caches, the transpose matrices of
// the reverse attention per group, the weight gradients as MFMA products whose K runs over the group's rows (ceil((n - 1) / 4) steps
// instead of 16), bias gradients as column sums over the group's rows of the operand buffer (fixed order).
// ---------------------------------------------------------------------------------------------------------------------------------
// out[(i0 + row) * ld + jo + col] = sum_(t < nt) XA[(r0 + t) * WA + i0 + row] * XB[(r0 + t) * WB + j0 + col]
__device__ __forceinline__ void cg_vp_tile_g(const double* XA, int WA, int i0, const double* XB, int WB, int j0, int r0, int nt, double* __restrict__ out, int ld, int jo, int jmax) {
    const int lane = threadIdx.x & 63, col = lane & 15, kq = lane >> 4;
    cg_vp_d4 acc = {0, 0, 0, 0};
    const int nks = (nt + 3) >> 2;
    for (int ks = 0; ks < nks; ++ks) {
        const int tt = 4 * ks + kq; const bool ok = tt < nt; const int tc = r0 + (ok ? tt : 0);
        const double a = ok ? XA[tc * WA + i0 + col] : 0.0, bv = ok ? XB[tc * WB + j0 + col] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc, 0, 0, 0);
    }
    if (jo + col < jmax) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(i0 + kq + 4 * r) * ld + jo + col] = acc[r];
    }
}
// out[c] = sum_(t < nt) X[(r0 + t) * WX + c0 + c], c < cmax <= 16 (lanes 0 .. 15, fixed order)
__device__ __forceinline__ void cg_vp_colsum(const double* X, int WX, int c0, int r0, int nt, double* __restrict__ out, int cmax) {
    const int lane = threadIdx.x & 63;
    if (lane < cmax) {
        double acc = 0.0;
        for (int t = 0; t < nt; ++t) acc += X[(r0 + t) * WX + c0 + lane];
        out[lane] = acc;
    }
}

// Sv <= S samples (consecutive rows of sidx0 / G0) on one wave.  lw: CgVanPar::packed_wave_doubles(n) doubles of LDS; st: the wave's HBM stash
__device__ __forceinline__ void cg_van_grad_packed(const CgVanModel& m, const double* __restrict__ P, const double* __restrict__ sp,
                                                   const int* __restrict__ sidx0, int Sv, double* lw, double* __restrict__ st, double* __restrict__ G0) {
    constexpr int MS = CgVanPar::MS, HS = CgVanPar::HS, NL = CgVanPar::NL, NH = CgVanPar::NH, KS = CgVanPar::KS;
    constexpr int blk = MS + MS * MS;
    const int lane = threadIdx.x & 63, n = m.n, M = m.M, nt = n - 1, S = CgVanPar::packed_samples(n), L = S * nt;
    const int g = lane / nt, t = lane - g * nt, gb = g * nt;            // group, token, first row of the group
    const bool act = g < Sv;
    const int* sidx = sidx0 + (size_t)(act ? g : 0) * n;
    const int cur = sidx[act ? t : 0], nxt = sidx[act ? t + 1 : 0], hi = t + 1 + M - n;
    const double rsk = 0.5;                                                 // 1 / sqrt(KS)
    double* Kc = lw; double* Vc = Kc + L * MS; double* R = Vc + L * MS;      // keys / values of the layer in hand
    double* XA = R; double* XB = R + 32 * L;                                // MFMA operand rows [lane][32], [lane][16]
    const int PW = nt + 1;
    double* Pm = R; double* Qb = R + ((L * PW + 1) & ~1); double* Db = Qb + MS * L;      // reverse attention (the operand rows are idle then)
    const int LR = CgVanPar::layer_rows(n);
    auto srow = [&](int r) -> double* { return st + (size_t)r * 64 + lane; };
    auto Gof = [&](int gg) -> double* { return G0 + (size_t)gg * m.total; };
    // ------------------------------------------------------------------ forward (per lane as in cg_van_grad_par; keys / values of the own group)
    double h[MS];
    {
        const double x0 = sp[(size_t)cur * 2], x1 = sp[(size_t)cur * 2 + 1];
#pragma unroll
        for (int j = 0; j < MS; ++j) { h[j] = tanh(fma(x1, P[m.o_ew + MS + j], fma(x0, P[m.o_ew + j], P[m.o_eb + j]))); *srow(j) = h[j]; }
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const double* Lp = P + m.o_l[l];
        const int R0 = 2 * MS + l * LR;
        double q[MS], kk[MS], vv[MS];
#pragma unroll
        for (int j = 0; j < MS; ++j) { *srow(R0 + j) = h[j]; q[j] = Lp[j]; kk[j] = Lp[blk + j]; vv[j] = Lp[2 * blk + j]; }
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int j = 0; j < MS; ++j) {
                q[j] = fma(h[i], Lp[MS + i * MS + j], q[j]); kk[j] = fma(h[i], Lp[blk + MS + i * MS + j], kk[j]); vv[j] = fma(h[i], Lp[2 * blk + MS + i * MS + j], vv[j]);
            }
        double* Kl = Kc; double* Vl = Vc;
        asm volatile("" ::: "memory");                                       // (the previous layer's reads are done: in-order LDS)
        if (act) {
#pragma unroll
            for (int j = 0; j < MS; ++j) { Kl[lane * MS + j] = kk[j]; Vl[lane * MS + j] = vv[j]; }
        }
#pragma unroll
        for (int j = 0; j < MS; ++j) *srow(R0 + MS + j) = q[j];
        asm volatile("" ::: "memory");
        const double* Kg = Kl + (act ? gb : 0) * MS; const double* Vg = Vl + (act ? gb : 0) * MS;
        double att[MS];
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {
            double mx = -INFINITY;
            for (int j = 0; j < nt; ++j) {
                double sc = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) sc = fma(q[hd * KS + c], Kg[j * MS + hd * KS + c], sc);
                mx = j <= t ? fmax(mx, sc * rsk) : mx;
            }
            double z = 0.0, o[KS];
#pragma unroll
            for (int c = 0; c < KS; ++c) o[c] = 0.0;
            for (int j = 0; j < nt; ++j) {
                double sc = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) sc = fma(q[hd * KS + c], Kg[j * MS + hd * KS + c], sc);
                const double e = j <= t ? cg_exp_nonpos(sc * rsk - mx) : 0.0;
                z += e;
#pragma unroll
                for (int c = 0; c < KS; ++c) o[c] = fma(e, Vg[j * MS + hd * KS + c], o[c]);
                *srow(R0 + 4 * MS + HS + hd * n + j) = e;
            }
            const double zi = 1.0 / z;
            *srow(R0 + 4 * MS + HS + NH * n + hd) = zi;
#pragma unroll
            for (int c = 0; c < KS; ++c) att[hd * KS + c] = o[c] * zi;
        }
        double h1[MS];
        {
            const double* ob = Lp + 3 * blk; const double* ow = ob + MS;
#pragma unroll
            for (int j = 0; j < MS; ++j) { *srow(R0 + 2 * MS + j) = att[j]; h1[j] = h[j] + ob[j]; }
#pragma unroll
            for (int i = 0; i < MS; ++i)
#pragma unroll
                for (int j = 0; j < MS; ++j) h1[j] = fma(att[i], ow[i * MS + j], h1[j]);
#pragma unroll
            for (int j = 0; j < MS; ++j) *srow(R0 + 3 * MS + j) = h1[j];
        }
        const double* b1 = Lp + 4 * blk; const double* w1 = b1 + HS; const double* b2 = w1 + MS * HS; const double* w2 = b2 + MS;
        double mid[HS];
#pragma unroll
        for (int j = 0; j < HS; ++j) mid[j] = b1[j];
#pragma unroll
        for (int i = 0; i < MS; ++i)
#pragma unroll
            for (int j = 0; j < HS; ++j) mid[j] = fma(h1[i], w1[i * HS + j], mid[j]);
#pragma unroll
        for (int j = 0; j < HS; ++j) { mid[j] = tanh(mid[j]); *srow(R0 + 4 * MS + j) = mid[j]; }
#pragma unroll
        for (int j = 0; j < MS; ++j) h[j] = h1[j] + b2[j];
#pragma unroll
        for (int i = 0; i < HS; ++i)
#pragma unroll
            for (int j = 0; j < MS; ++j) h[j] = fma(mid[i], w2[i * MS + j], h[j]);
    }
    double th[MS];
#pragma unroll
    for (int j = 0; j < MS; ++j) { th[j] = tanh(h[j]); *srow(MS + j) = th[j]; }
    // ------------------------------------------------------------------ conditionals of positions 1 .. n - 1: logits, softmax
    const int RL = 2 * MS + NL * LR;
    double mx = -INFINITY;
    for (int j = 0; j < M; ++j) {
        double v = P[m.o_ob + j];
#pragma unroll
        for (int i = 0; i < MS; ++i) v = fma(th[i], P[m.o_ow + i * M + j], v);
        const bool ok = j > cur && j <= hi;
        *srow(RL + j) = v;
        mx = ok ? fmax(mx, v) : mx;
    }
    double z = 0.0;
    for (int j = 0; j < M; ++j) {
        const bool ok = j > cur && j <= hi;
        const double e = ok ? cg_exp_nonpos(*srow(RL + j) - mx) : 0.0;
        z += e;
        *srow(RL + j) = e;
    }
    const double zinv = act ? 1.0 / z : 0.0;
    __builtin_amdgcn_s_waitcnt(0);
    // ------------------------------------------------------------------ reverse
    for (int gg = 0; gg < Sv; ++gg) {      // position 0 of every sample: one-hot minus softmax over the allowed orbitals of x1hat (lane = orbital)
        const int s0 = sidx0[(size_t)gg * n], h0i = M - n;
        double* G = Gof(gg);
        double lg[4]; double m0 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; lg[r] = (j < M && j <= h0i) ? P[m.o_x1 + j] : -INFINITY; m0 = fmax(m0, lg[r]); }
        m0 = cg_wmax(m0);
        double z0 = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) z0 += (lg[r] > -INFINITY) ? exp(lg[r] - m0) : 0.0;
        z0 = cg_wsum(z0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = lane + 64 * r; if (j < M) G[m.o_x1 + j] = (j <= h0i) ? (j == s0 ? 1.0 : 0.0) - exp(lg[r] - m0) / z0 : 0.0; }
    }
    // output layer: dy_t[j] = [j = next] - softmax, in chunks of 16 orbitals; d ow = sum_t th_t (x) dy_t per sample on the matrix cores
    double dh[MS];
#pragma unroll
    for (int i = 0; i < MS; ++i) dh[i] = 0.0;
    cg_vp_put<MS>(XA, 32, lane, act, th);
    for (int j0 = 0; j0 < M; j0 += 16) {
        double dy[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = j0 + jj; const bool ok = act && j < M && j > cur && j <= hi;
            dy[jj] = ok ? (j == nxt ? 1.0 : 0.0) - *srow(RL + (j < M ? j : 0)) * zinv : 0.0;
            if (j < M) {
#pragma unroll
                for (int i = 0; i < MS; ++i) dh[i] = fma(P[m.o_ow + i * M + j], dy[jj], dh[i]);
            }
        }
        cg_vp_put<16>(XB, 16, lane, act, dy);
        asm volatile("" ::: "memory");
        for (int gg = 0; gg < Sv; ++gg) {
            double* G = Gof(gg);
            cg_vp_tile_g(XA, 32, 0, XB, 16, 0, gg * nt, nt, G + m.o_ow + j0, M, 0, M - j0);
            cg_vp_colsum(XB, 16, 0, gg * nt, nt, G + m.o_ob + j0, M - j0 < 16 ? M - j0 : 16);
        }
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < MS; ++i) dh[i] *= 1.0 - th[i] * th[i];
#pragma unroll
    for (int li = 0; li < NL; ++li) {
        const int l = NL - 1 - li;
        const double* Lp = P + m.o_l[l];
        const int R0 = 2 * MS + l * LR;
        const double* b1 = Lp + 4 * blk; const double* w1 = b1 + HS; const double* w2 = w1 + MS * HS + MS;
        double* Kl = Kc; double* Vl = Vc;
        const double* Kg = Kl + (act ? gb : 0) * MS; const double* Vg = Vl + (act ? gb : 0) * MS;
        {   // keys / values of this layer again, from the stashed layer input (the caches hold one layer)
            double kk[MS], vv[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) { kk[j] = Lp[blk + j]; vv[j] = Lp[2 * blk + j]; }
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                const double hi_ = *srow(R0 + i);
#pragma unroll
                for (int j = 0; j < MS; ++j) { kk[j] = fma(hi_, Lp[blk + MS + i * MS + j], kk[j]); vv[j] = fma(hi_, Lp[2 * blk + MS + i * MS + j], vv[j]); }
            }
            asm volatile("" ::: "memory");
            if (act) {
#pragma unroll
                for (int j = 0; j < MS; ++j) { Kl[lane * MS + j] = kk[j]; Vl[lane * MS + j] = vv[j]; }
            }
            asm volatile("" ::: "memory");
        }
        // ---- DenseBlock: h = h1 + W2^T tanh(W1^T h1 + b1) + b2
        double dh1[MS];
        {
            double mid[HS], dpre[HS], h1[MS];
#pragma unroll
            for (int j = 0; j < HS; ++j) mid[j] = *srow(R0 + 4 * MS + j);
#pragma unroll
            for (int j = 0; j < MS; ++j) h1[j] = *srow(R0 + 3 * MS + j);
            cg_vp_put<HS>(XA, 32, lane, act, mid); cg_vp_put<MS>(XB, 16, lane, act, dh);
            asm volatile("" ::: "memory");
            for (int gg = 0; gg < Sv; ++gg) {
                double* g2 = Gof(gg) + m.o_l[l] + 4 * blk + HS + MS * HS;                      // m2b[MS], m2w[HS][MS]
                cg_vp_tile_g(XA, 32, 0, XB, 16, 0, gg * nt, nt, g2 + MS, MS, 0, MS); cg_vp_tile_g(XA, 32, 16, XB, 16, 0, gg * nt, nt, g2 + MS, MS, 0, MS);
                cg_vp_colsum(XB, 16, 0, gg * nt, nt, g2, MS);
            }
#pragma unroll
            for (int i = 0; i < HS; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < MS; ++j) a = fma(w2[i * MS + j], dh[j], a);
                dpre[i] = a * (1.0 - mid[i] * mid[i]);
            }
            asm volatile("" ::: "memory");
            cg_vp_put<MS>(XA, 32, lane, act, h1);
#pragma unroll
            for (int half = 0; half < 2; ++half) {                          // m1b[HS], m1w[MS][HS]: 16 columns of dpre at a time
                double dp[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) dp[j] = dpre[16 * half + j];
                cg_vp_put<16>(XB, 16, lane, act, dp);
                asm volatile("" ::: "memory");
                for (int gg = 0; gg < Sv; ++gg) {
                    double* g1 = Gof(gg) + m.o_l[l] + 4 * blk;
                    cg_vp_tile_g(XA, 32, 0, XB, 16, 0, gg * nt, nt, g1 + HS, HS, 16 * half, HS);
                    cg_vp_colsum(XB, 16, 0, gg * nt, nt, g1 + 16 * half, 16);
                }
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                double a = dh[i];
#pragma unroll
                for (int j = 0; j < HS; ++j) a = fma(w1[i * HS + j], dpre[j], a);
                dh1[i] = a;
            }
            asm volatile("" ::: "memory");
        }
        // ---- attention output linear
        double dov[MS], q[MS];
        {
            double att[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) { att[j] = *srow(R0 + 2 * MS + j); q[j] = *srow(R0 + MS + j); }
            cg_vp_put<MS>(XA, 32, lane, act, att); cg_vp_put<MS>(XB, 16, lane, act, dh1);
            asm volatile("" ::: "memory");
            for (int gg = 0; gg < Sv; ++gg) {
                double* go = Gof(gg) + m.o_l[l] + 3 * blk;
                cg_vp_tile_g(XA, 32, 0, XB, 16, 0, gg * nt, nt, go + MS, MS, 0, MS);
                cg_vp_colsum(XB, 16, 0, gg * nt, nt, go, MS);
            }
            const double* ow = Lp + 3 * blk + MS;
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                double a = 0.0;
#pragma unroll
                for (int j = 0; j < MS; ++j) a = fma(ow[i * MS + j], dh1[j], a);
                dov[i] = a;
            }
            asm volatile("" ::: "memory");
        }
        // ---- attention: query side (lane (g, t) over j <= t) -> dq;  key side (lane (g, j) over the queries t >= j of ITS group) -> dk, dv
        double dq[MS], dk[MS], dv[MS];
#pragma unroll
        for (int i = 0; i < MS; ++i) { dq[i] = 0.0; dk[i] = 0.0; dv[i] = 0.0; }
        cg_vp_put<MS>(Qb, MS, lane, act, q); cg_vp_put<MS>(Db, MS, lane, act, dov);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {
            const double zi = *srow(R0 + 4 * MS + HS + NH * n + hd);
            const int RA = R0 + 4 * MS + HS + hd * n;
            double sada = 0.0;
            for (int j = 0; j < nt; ++j) {
                const double a = *srow(RA + j) * zi;
                double dA = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) dA = fma(dov[hd * KS + c], Vg[j * MS + hd * KS + c], dA);
                sada = fma(a, dA, sada);
                if (act) Pm[lane * PW + j] = a;
            }
            asm volatile("" ::: "memory");
            if (act) {                                                      // key side, values: dv_j += sum_t a_tj dov_t
                for (int tq = 0; tq < nt; ++tq) {
                    const double a = Pm[(gb + tq) * PW + t];
#pragma unroll
                    for (int c = 0; c < KS; ++c) dv[hd * KS + c] = fma(a, Db[(gb + tq) * MS + hd * KS + c], dv[hd * KS + c]);
                }
            }
            asm volatile("" ::: "memory");
            for (int j = 0; j < nt; ++j) {
                const double a = *srow(RA + j) * zi;
                double dA = 0.0;
#pragma unroll
                for (int c = 0; c < KS; ++c) dA = fma(dov[hd * KS + c], Vg[j * MS + hd * KS + c], dA);
                const double dS = a * (dA - sada) * rsk;
#pragma unroll
                for (int c = 0; c < KS; ++c) dq[hd * KS + c] = fma(dS, Kg[j * MS + hd * KS + c], dq[hd * KS + c]);
                if (act) Pm[lane * PW + j] = dS;
            }
            asm volatile("" ::: "memory");
            if (act) {                                                      // key side, keys: dk_j += sum_t dS_tj q_t
                for (int tq = 0; tq < nt; ++tq) {
                    const double dS = Pm[(gb + tq) * PW + t];
#pragma unroll
                    for (int c = 0; c < KS; ++c) dk[hd * KS + c] = fma(dS, Qb[(gb + tq) * MS + hd * KS + c], dk[hd * KS + c]);
                }
            }
            asm volatile("" ::: "memory");
        }
        // ---- query / key / value linears of token t
        {
            double hin[MS];
#pragma unroll
            for (int j = 0; j < MS; ++j) hin[j] = *srow(R0 + j);
            cg_vp_put<MS>(XA, 32, lane, act, hin);
            auto lin = [&](const double (&dvec)[MS], int which) {           // bias and weights of the query (0) / key (1) / value (2) linear
                cg_vp_put<MS>(XB, 16, lane, act, dvec);
                asm volatile("" ::: "memory");
                for (int gg = 0; gg < Sv; ++gg) {
                    double* Gl = Gof(gg) + m.o_l[l] + which * blk;
                    cg_vp_tile_g(XA, 32, 0, XB, 16, 0, gg * nt, nt, Gl + MS, MS, 0, MS);
                    cg_vp_colsum(XB, 16, 0, gg * nt, nt, Gl, MS);
                }
                asm volatile("" ::: "memory");
            };
            lin(dq, 0); lin(dk, 1); lin(dv, 2);
            const double* wq = Lp + MS; const double* wk = Lp + blk + MS; const double* wv = Lp + 2 * blk + MS;
#pragma unroll
            for (int i = 0; i < MS; ++i) {
                double a = dh1[i];
#pragma unroll
                for (int j = 0; j < MS; ++j) a = fma(wq[i * MS + j], dq[j], fma(wk[i * MS + j], dk[j], fma(wv[i * MS + j], dv[j], a)));
                dh[i] = a;
            }
            asm volatile("" ::: "memory");
        }
    }
    // ---- embedding: h0 = tanh(eb + sp[cur] ew)
    {
        const double x0 = sp[(size_t)cur * 2], x1 = sp[(size_t)cur * 2 + 1];
#pragma unroll
        for (int j = 0; j < MS; ++j) { const double h0 = *srow(j); dh[j] *= 1.0 - h0 * h0; }
        for (int which = 0; which < 3; ++which) {                           // d eb, d ew[0][:], d ew[1][:]
            const double f = which == 0 ? 1.0 : which == 1 ? x0 : x1;
            if (act) {
#pragma unroll
                for (int j = 0; j < MS; ++j) XB[lane * 16 + j] = f * dh[j];
            }
            asm volatile("" ::: "memory");
            const int off = which == 0 ? m.o_eb : which == 1 ? m.o_ew : m.o_ew + MS;
            for (int gg = 0; gg < Sv; ++gg) cg_vp_colsum(XB, 16, 0, gg * nt, nt, Gof(gg) + off, MS);
            asm volatile("" ::: "memory");
        }
    }
}
#endif      // __HIP_DEVICE_COMPILE__
#endif
