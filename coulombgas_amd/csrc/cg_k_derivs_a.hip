// cg_k_derivs_a.hip -- derivative kernels of the (2, 16, 16) flow (every shipped run), the score reductions, the quantum Fisher
// matrix and the entry points of the family.  The other instantiations are compiled in cg_k_derivs_b.hip.
#include "cg_host.hpp"
#include "cg_derivs.hpp"
#include "cg_lap.hpp"
#include "cg_score.hpp"

#define CG_UNIT_CONFIGS(X) CG_FAST_CONFIGS_A(X)
#define CG_UNIT_NAME(f) cg_derivs_a_##f
#include "cg_k_derivs.inc"

int cg_derivs_b_grad_lap(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, int mode, const double* v,
                         double* grad, double* lap);
int cg_derivs_b_param_vjp(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, double* score);
int cg_derivs_b_scores(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, double* score);

// deterministic second-stage reduction of per-workgroup partial gradients: out[p] = sum_g partial[g][p]
__global__ void k_reduce_rows(const double* __restrict__ partial, int rows, int P, double* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double a = 0.0;
    for (int r = 0; r < rows; ++r) a += partial[(size_t)r * P + p];
    out[p] = a;
}

// Quantum Fisher matrix of stochastic reconfiguration (src/sr.py:74-76):  F[p][q] = (1/B) sum_b Re( conj(S[b][p]) S[b][q] )
// = (1/B) sum_b ( Sr[b][p] Sr[b][q] + Si[b][p] Si[b][q] ),  S = per-sample scores (B x P, complex interleaved).
// One wave per 16 x 16 tile of the upper triangle (mirrored on store); the batch axis is the K of v_mfma_f64_16x16x4.
#define CG_MFMA_4X4(acc, av, bv)                                                                                          \
    _Pragma("unroll") for (int a_ = 0; a_ < 4; ++a_)                                                                      \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_)(acc)[a_][c_] = __builtin_amdgcn_mfma_f64_16x16x4f64((av)[a_], (bv)[c_], (acc)[a_][c_], 0, 0, 0)
// Same blocking as k_fisher_real (csrc/cg_solve.inc): one wave owns a 64 x 64 block of F as 4 x 4 MFMA tiles, MFMA row i of tile
// a is parameter 4 i + a (a lane's eight operands -- four complex entries -- are 64 contiguous bytes of a score row), the operands
// of step t + 1 are in flight while step t multiplies, upper block-triangle only, the batch sliced over blockIdx.z (partial
// matrices summed in fixed order by k_rows_sum_d) because P = 1074 alone gives only 45 workgroups.
__global__ void __launch_bounds__(256, 2) k_fisher(const double* __restrict__ S, int B, int P, int chunk, double scale, double* __restrict__ F) {
    typedef double d4_t __attribute__((ext_vector_type(4)));
    if (blockIdx.x < blockIdx.y) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p0 = 128 * blockIdx.y + 64 * (wave >> 1), q0 = 128 * blockIdx.x + 64 * (wave & 1);
    if (p0 >= P || q0 >= P || q0 + 64 <= p0) return;
    const int col = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.z * chunk, b1 = min(B, b0 + chunk);
    F += (size_t)blockIdx.z * P * P;
    d4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = d4_t{0, 0, 0, 0};
    const int pl = p0 + 4 * col, ql = q0 + 4 * col;
    int bb = b0;
    if (p0 + 64 <= P && q0 + 64 <= P && b1 - b0 >= 8) {                   // interior block (wave-uniform)
        const double* pa = S + ((size_t)(b0 + kq) * P + pl) * 2;
        const double* pb = S + ((size_t)(b0 + kq) * P + ql) * 2;
        const size_t step = (size_t)8 * P;
        double ar[4], ai[4], br[4], bi[4], anr[4], ani[4], bnr[4], bni[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { ar[a] = pa[2 * a]; ai[a] = pa[2 * a + 1]; br[a] = pb[2 * a]; bi[a] = pb[2 * a + 1]; }
        const int steps = (b1 - b0) >> 2;
        for (int t = 1; t < steps; ++t) {
            pa += step; pb += step;
#pragma unroll
            for (int a = 0; a < 4; ++a) { anr[a] = pa[2 * a]; ani[a] = pa[2 * a + 1]; bnr[a] = pb[2 * a]; bni[a] = pb[2 * a + 1]; }
            CG_MFMA_4X4(acc, ar, br);
            CG_MFMA_4X4(acc, ai, bi);
#pragma unroll
            for (int a = 0; a < 4; ++a) { ar[a] = anr[a]; ai[a] = ani[a]; br[a] = bnr[a]; bi[a] = bni[a]; }
        }
        CG_MFMA_4X4(acc, ar, br);
        CG_MFMA_4X4(acc, ai, bi);
        bb = b0 + 4 * steps;
    }
    for (; bb < b1; bb += 4) {                                            // ragged blocks and the last rows of the batch
        const int b = bb + kq;
        double ar[4], ai[4], br[4], bi[4];
        const double* rowp = S + (size_t)min(b, B - 1) * P * 2;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const bool pa_ok = b < b1 && pl + a < P, pb_ok = b < b1 && ql + a < P;
            ar[a] = pa_ok ? rowp[2 * (pl + a)] : 0.0; ai[a] = pa_ok ? rowp[2 * (pl + a) + 1] : 0.0;
            br[a] = pb_ok ? rowp[2 * (ql + a)] : 0.0; bi[a] = pb_ok ? rowp[2 * (ql + a) + 1] : 0.0;
        }
        CG_MFMA_4X4(acc, ar, br);
        CG_MFMA_4X4(acc, ai, bi);
    }
    const bool mirror = q0 != p0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pr = p0 + 4 * (kq + 4 * r) + a;
            if (pr >= P) continue;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int q = ql + c;
                if (q < P) {
                    const double v = acc[a][c][r] * scale;
                    F[(size_t)pr * P + q] = v;
                    if (mirror) F[(size_t)q * P + pr] = v;
                }
            }
        }
}
// out[p] = scale * sum_r partial[r][p], rows summed in fixed order
__global__ void k_rows_sum_d(const double* __restrict__ partial, int rows, int P, double scale, double* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double a = 0.0;
    for (int r = 0; r < rows; ++r) a += partial[(size_t)r * P + p];
    out[p] = a * scale;
}
// Re(S^H S) / B of the resident complex scores into F (device pointer); partial matrices from the arena when the batch is sliced
static int fisher_complex_launch(cg_ctx* c, const double* S, int B, int P, double* F) {
    const int nbt = (P + 127) / 128, blocks = nbt * (nbt + 1) / 2;
    int nsl = 1;
    while (nsl < 16 && blocks * nsl < c->cu_count && B / (2 * nsl) >= 64) nsl *= 2;
    const int chunk = (((B + nsl - 1) / nsl) + 3) & ~3;
    if (nsl == 1) {
        hipLaunchKernelGGL(k_fisher, dim3(nbt, nbt, 1), dim3(256), 0, c->stream, S, B, P, chunk, 1.0 / (double)B, F);
    } else {
        double* part = (double*)arena_take(c, sizeof(double) * (size_t)nsl * P * P);
        if (!part) CG_FAIL(c, CG_ERR_HIP, "quantum Fisher matrix: workspace allocation failed");
        hipLaunchKernelGGL(k_fisher, dim3(nbt, nbt, nsl), dim3(256), 0, c->stream, S, B, P, chunk, 1.0, part);
        hipLaunchKernelGGL(k_rows_sum_d, dim3((P * P + 255) / 256), dim3(256), 0, c->stream, (const double*)part, nsl, P * P, 1.0 / (double)B, F);
    }
    CG_HIP(c, hipGetLastError());
    return CG_OK;
}
// mean over the batch of the complex scores (src/sr.py:70): out[2 p + c] = (1/B) sum_b S[b][p][c]; fixed summation order
// Column sums of the resident score matrix over one slice of the batch (blockIdx.y): out[slice][c] = sum_{b in slice} S[b][c].
// The slices are summed in fixed order by k_reduce_rows (deterministic), the 1/B of the mean is applied afterwards.
__global__ void __launch_bounds__(256) k_score_mean(const double* __restrict__ S, int B, int P2 /* 2 P */, int chunk, double* __restrict__ out) {
    __shared__ double part[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
    double a = 0.0;
    if (c < P2) for (int b = b0 + rg; b < b1; b += 4) a += S[(size_t)b * P2 + c];
    part[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0 && c < P2) out[(size_t)blockIdx.y * P2 + c] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
}

// out[slice][p] = sum_{b in slice} ( w_re[b] Sre[b][p] + w_im[b] Sim[b][p] ): the theta-VJP from resident scores
__global__ void __launch_bounds__(256) k_score_gemv(const double* __restrict__ S, const double* __restrict__ w_re,
                                                    const double* __restrict__ w_im, int B, int P, int chunk, double* __restrict__ out) {
    __shared__ double part[256];
    const int p = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
    double a = 0.0;
    if (p < P)
        for (int b = b0 + rg; b < b1; b += 4) {
            const double* s = S + ((size_t)b * P + p) * 2;
            a += w_re[b] * s[0] + w_im[b] * s[1];
        }
    part[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0 && p < P) out[(size_t)blockIdx.y * P + p] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
}



// Small systems (n <= 16, everything in LDS): k_grad_lap2 and k_scores in one kernel.  The set-up of the two is the same computation (61 % of
// k_scores at n = 13): CgLap's set-up parks what the score sweep reads in the walker's workspace slot (CgLap::Stash: 39 KB at n = 13, written and
// read back through the L2 of the workgroup's XCD, laid out like the score kernel's LDS image so that reading it is three flat copies), the grad / Laplacian runs on, then the sweep of cg_score.hpp runs in its own LDS layout.
template <int D, int HS, int HT>
__global__ void __launch_bounds__(256, 2) k_grad_lap2_scores(CgDev m, const double* __restrict__ theta, const double* __restrict__ spk, const double* __restrict__ tab,
                           const double* __restrict__ x, const int* __restrict__ sidx, int B, int w0, int mode, const double* __restrict__ v,
                           double* __restrict__ grad, double* __restrict__ lap, double* __restrict__ score, double* ws,
                           typename CgLap<D, HS, HT>::Lay lg, const typename CgScore<D, HS, HT>::Lay* __restrict__ lsp /* device memory */,
                           const typename CgLap<D, HS, HT>::Stash* __restrict__ stp /* device memory */, unsigned stash_doubles) {
    double* lds = cg_dyn_lds + CG_TAB_DOUBLES;
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    constexpr int NP = CgFast<D, HS, HT>::NPARAM;
    const int n = m.n, N = n * D, w = w0 + blockIdx.x;        // (the batch goes in launches of gridDim.x walkers: one stash slot per workgroup)
    const double* thg = theta;
    if (lg.th_lds) {                     // (always, on this path; the branch keeps the pointer generic for the compiler: flat loads of the weights
        double* th_l = lds + lg.th;      // overlap the ds_ queue -- 8 % faster than ds_read in k_grad_lap2)
        for (int e = b.tid; e < NP; e += b.nthr) th_l[e] = theta[e];
        thg = th_l;
    }
    __syncthreads();
    CG_STAMP_INIT
    if (w < B) {
        double* stash = ws + (size_t)blockIdx.x * stash_doubles;
        CgLap<D, HS, HT>::template grad_laplacian<true>(b, thg, x + (size_t)w * N, spk, sidx + (size_t)w * n, n, m.L, mode,
                                                        v + (size_t)w * N, grad + (size_t)w * N * 2, lap + 2 * w, lds, ws, lg, stash, stp);
        b.sync();
        const typename CgScore<D, HS, HT>::Lay ls = *lsp;          // (read here: ~60 scalar registers that the grad / Laplacian part does not carry)
        double* th_l = lds + ls.th;
        for (int e = b.tid; e < NP; e += b.nthr) th_l[e] = theta[e];
        const double* th = ls.th >= 0 ? (const double*)th_l : theta;       // (as in k_scores: a pointer the compiler treats as generic)
        CgScore<D, HS, HT>::unstash(b, n, lds, ls, stash);
        CG_STAMP_START(21)
        CgScore<D, HS, HT>::sweep(b, th, n, m.L, lds, ls, score + (size_t)w * NP * 2);
        CG_STAMP_END(21)
    }
    CG_STAMP_FLUSH
}

// 1 launched, 0 not served, < 0 error
static int small_grad_lap_scores(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, int mode, const double* v, double* grad, double* lap,
                                 double* score) {
    constexpr int D = 2, HS = 16, HT = 16;
    if (c->dim != D || c->hs != HS || c->ht != HT || (mode != 1 && mode != 2) || !v || cg_env_int("CG_SMALL_FUSED", 1) == 0) return 0;
    int rc;
    const int n = c->n;
    const auto lg = CgLap<D, HS, HT>::layout(n, 256, mode, (size_t)CG_LAP_LDS_BYTES / sizeof(double) - CG_TAB_DOUBLES);
    const auto ls = CgScore<D, HS, HT>::layout(n, 256, (size_t)CG_SCORE_LDS_BYTES / sizeof(double) - CG_TAB_DOUBLES);
    if (!(lg.all_lds && lg.th_lds) || !ls.ok) return 0;
    const auto st = CgScore<D, HS, HT>::stash_of(ls);
    const size_t lds = sizeof(double) * (CG_TAB_DOUBLES + (size_t)std::max(lg.lds_total, ls.total));
    const int chunk = std::min(B, cg_env_int("CG_SMALL_FUSED_CHUNK", 16384));        // (77 KB of stash per walker in flight: 1.2 GB at most)
    if ((rc = ensure_ws(c, sizeof(double) * ((size_t)st.total * chunk + 64)))) return rc;
    if (c->lay_tag != 1) {                     // the score layout, once per context, where the kernel reads it
        if (!c->d_lay && hipMalloc(&c->d_lay, 4096) != hipSuccess) CG_FAIL(c, CG_ERR_HIP, "cg_grad_laplacian_scores: device allocation failed");
        static_assert(sizeof(ls) + sizeof(st) <= 2048, "layout buffer");
        CG_HIP(c, hipMemcpyAsync(c->d_lay, &ls, sizeof(ls), hipMemcpyHostToDevice, c->stream));
        CG_HIP(c, hipMemcpyAsync((char*)c->d_lay + 2048, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
        CG_HIP(c, hipStreamSynchronize(c->stream));
        c->lay_tag = 1;
    }
    if ((rc = set_lds(c, k_grad_lap2_scores<D, HS, HT>, lds))) return rc;
    for (int w0 = 0; w0 < B; w0 += chunk)
        hipLaunchKernelGGL((k_grad_lap2_scores<D, HS, HT>), dim3(std::min(chunk, B - w0)), dim3(256), lds, c->stream, m, (const double*)c->d_theta, (const double*)c->d_spk,
                       (const double*)c->d_tab, x, sidx, B, w0, mode, v, grad, lap, score, (double*)c->ws, lg, (const CgScore<D, HS, HT>::Lay*)c->d_lay,
                       (const CgLap<D, HS, HT>::Stash*)((const char*)c->d_lay + 2048), st.total);
    return 1;
}

int cg_big_grad_lap_scores(cg_ctx* c, const CgDev& m, const double* x, const int* sidx, int B, int mode, const double* v, double* grad, double* lap,
                           double* score);      // cg_k_big.hip

extern "C" {

int cg_grad_laplacian(cg_ctx* c, const double* x, const int32_t* sidx, int B, int mode, const double* v,
                      double* grad, double* lap) {
    int rc = check_ready(c, "cg_grad_laplacian", B); if (rc) return rc;
    if (mode < 0 || mode > 2) CG_FAIL(c, CG_ERR_ARG, "cg_grad_laplacian: mode %d", mode);
    if (mode != CG_LAP_EXACT && !v) CG_FAIL(c, CG_ERR_ARG, "cg_grad_laplacian: Hutchinson modes need v");
    if (B == 0) return CG_OK;
    if (!x || !sidx || !grad || !lap) CG_FAIL(c, CG_ERR_ARG, "cg_grad_laplacian: NULL argument");
    const int n = c->n, N = n * c->dim;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_grad_laplacian: arena");
    Arg ax{(void*)x, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg as{(void*)sidx, nullptr, sizeof(int32_t) * (size_t)B * n, true, false};
    Arg av{(void*)v, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg ag{grad, nullptr, sizeof(double) * (size_t)B * N * 2, false, true};
    Arg al{lap, nullptr, sizeof(double) * (size_t)B * 2, false, true};
    Arg* all[] = {&ax, &as, &av, &ag, &al};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    if (!c->fast) {
        const int grid = std::min(B, c->cu_count * 2 * CG_DERIV_WAVES);
        if ((rc = cg_gen_run_grad_lap(c, grid, (const double*)ax.dev, (const int*)as.dev, B, mode, (const double*)av.dev, (double*)ag.dev,
                                      (double*)al.dev))) return rc;
        for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
        return finish(c);
    }
    const CgDev m = make_dev(c);
    bool launched = false;
    if ((rc = cg_derivs_a_grad_lap(c, m, (const double*)ax.dev, (const int*)as.dev, B, mode, (const double*)av.dev, (double*)ag.dev,
                                   (double*)al.dev)) < 0) return rc;
    if (rc == 0 && (rc = cg_derivs_b_grad_lap(c, m, (const double*)ax.dev, (const int*)as.dev, B, mode, (const double*)av.dev,
                                              (double*)ag.dev, (double*)al.dev)) < 0) return rc;
    launched = rc == 1;
    if (!launched) CG_FAIL(c, CG_ERR_UNSUPPORTED, "cg_grad_laplacian: configuration not instantiated");
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}

/* cg_grad_laplacian AND cg_scores_compute of the same walkers in one call: what an optimisation step needs (src/VMC.py:35 for the local
 * energies, then jax.jacrev(quantum_lossfn) of main.py:278 on the same x).  Where a fused kernel serves the configuration (device pointers,
 * (dim 2, 16, 16) flow, Hutchinson modes: k_gradlap_scores_big at n > 16, k_grad_lap2_scores below it) the set-up of the two -- flow, Jacobian,
 * both inverses -- runs once;
 * otherwise the two calls run one after the other.  Results are those of the separate calls bit for bit. */
int cg_grad_laplacian_scores(cg_ctx* c, const double* x, const int32_t* sidx, int B, int mode, const double* v, double* grad, double* lap) {
    int rc = check_ready(c, "cg_grad_laplacian_scores", B); if (rc) return rc;
    if (B <= 0) CG_FAIL(c, CG_ERR_ARG, "cg_grad_laplacian_scores: empty batch");
    if (c->fast && c->ptr_mode == CG_PTR_DEVICE && (mode == 1 || mode == 2) && x && sidx && v && grad && lap) {
        const size_t bytes = sizeof(double) * (size_t)B * c->P * 2;
        if (c->scores_cap < bytes) {
            if (c->d_scores) { CG_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_scores); c->d_scores = nullptr; c->scores_cap = 0; }
            if (hipMalloc((void**)&c->d_scores, bytes) != hipSuccess)
                CG_FAIL(c, CG_ERR_HIP, "cg_grad_laplacian_scores: %zu bytes for the per-sample scores could not be allocated", bytes);
            c->scores_cap = bytes;
        }
        if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_grad_laplacian_scores: arena");
        c->scores_B = 0;                       // (not valid until the launch below has gone out)
        rc = cg_big_grad_lap_scores(c, make_dev(c), x, (const int*)sidx, B, mode, v, grad, lap, (double*)c->d_scores);
        if (rc == 0) rc = small_grad_lap_scores(c, make_dev(c), x, (const int*)sidx, B, mode, v, grad, lap, (double*)c->d_scores);
        if (rc < 0) return rc;
        if (rc == 1) { c->scores_B = B; return finish(c); }
    }
    if ((rc = cg_grad_laplacian(c, x, sidx, B, mode, v, grad, lap))) return rc;
    return cg_scores_compute(c, x, sidx, B);
}

// Batch reductions over the resident score matrix, sliced over the batch so that the whole chip streams it (one slice
// per ~64 walkers, at most 64 slices), then summed in fixed order: mean over b (w_re == nullptr, count = 2P, scaled by 1/B)
// or the weighted sum of cg_scores_vjp (count = P).
static int score_reduce(cg_ctx* c, const double* S, const double* w_re, const double* w_im, int B, int count, double* out) {
    const int nsl = std::max(1, std::min(64, (B + 63) / 64)), chunk = (B + nsl - 1) / nsl;
    double* partial = (double*)arena_take(c, sizeof(double) * (size_t)nsl * count);
    if (!partial) CG_FAIL(c, CG_ERR_HIP, "score reduction: workspace allocation failed");
    if (w_re) hipLaunchKernelGGL(k_score_gemv, dim3((count + 63) / 64, nsl), dim3(256), 0, c->stream, S, w_re, w_im, B, count, chunk, partial);
    else hipLaunchKernelGGL(k_score_mean, dim3((count + 63) / 64, nsl), dim3(256), 0, c->stream, S, B, count, chunk, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3((count + 127) / 128), dim3(128), 0, c->stream, (const double*)partial, nsl, count, out);
    if (!w_re) return cg_scale_dev(c, out, (size_t)count, 1.0 / (double)B);
    return CG_OK;
}

static int run_vjp(cg_ctx* c, const char* fn, const double* x, const int32_t* sidx, int B, const double* w_re,
                   const double* w_im, double* g_theta, double* score, double* fisher = nullptr, double* smean = nullptr,
                   bool keep_scores = false) {
    int rc = check_ready(c, fn, B); if (rc) return rc;
    const int n = c->n, N = n * c->dim, P = c->P;
    if (B == 0) {
        if (g_theta && c->ptr_mode == CG_PTR_HOST) memset(g_theta, 0, sizeof(double) * P);
        else if (g_theta) CG_HIP(c, hipMemsetAsync(g_theta, 0, sizeof(double) * P, c->stream));
        return CG_OK;
    }
    if (!x || !sidx) CG_FAIL(c, CG_ERR_ARG, "%s: NULL argument", fn);
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "%s: arena", fn);
    Arg ax{(void*)x, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg as{(void*)sidx, nullptr, sizeof(int32_t) * (size_t)B * n, true, false};
    Arg awr{(void*)w_re, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg awi{(void*)w_im, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg ag{g_theta, nullptr, sizeof(double) * (size_t)P, false, true};
    Arg asc{score, nullptr, sizeof(double) * (size_t)B * P * 2, false, true};
    Arg afi{fisher, nullptr, sizeof(double) * (size_t)P * P, false, true};
    Arg asm_{smean, nullptr, sizeof(double) * (size_t)P * 2, false, true};
    Arg* all[] = {&ax, &as, &awr, &awi, &ag, &asc, &afi, &asm_};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    if ((fisher || keep_scores) && !asc.dev) {        // the scores stay on the device, in the context's resident buffer
        if (c->scores_cap < asc.bytes) {
            if (c->d_scores) { CG_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(c->d_scores); c->d_scores = nullptr; c->scores_cap = 0; }
            if (hipMalloc((void**)&c->d_scores, asc.bytes) != hipSuccess)
                CG_FAIL(c, CG_ERR_HIP, "%s: %zu bytes for the per-sample scores could not be allocated", fn, asc.bytes);
            c->scores_cap = asc.bytes;
        }
        asc.dev = c->d_scores; c->scores_B = B;
    }
    if (c->fast) {
        // depth-2 flow: per-sample scores first -- into the caller's buffer, the resident buffer or (plain theta-VJP) a temporary -- by
        // the second-generation kernel (cg_score.hpp) where the system fits its LDS plan, otherwise by the first-generation one
        // (cg_derivs.hpp); the weighted sum over the batch is then the sliced GEMV over the score matrix
        const CgDev m = make_dev(c);
        auto scores_of = [&](const double* xs, const int* ss, int nb, double* out) -> int {
            int r;
            if ((r = cg_derivs_a_scores(c, m, xs, ss, nb, out)) < 0) return r;
            if (r == 0 && (r = cg_derivs_b_scores(c, m, xs, ss, nb, out)) < 0) return r;
            if (r == 0 && (r = cg_derivs_a_param_vjp(c, m, xs, ss, nb, out)) < 0) return r;
            if (r == 0 && (r = cg_derivs_b_param_vjp(c, m, xs, ss, nb, out)) < 0) return r;
            if (r != 1) CG_FAIL(c, CG_ERR_UNSUPPORTED, "%s: configuration not instantiated", fn);
            return CG_OK;
        };
        double* sc = (double*)asc.dev;
        if (sc) {
            if ((rc = scores_of((const double*)ax.dev, (const int*)as.dev, B, sc))) return rc;
            if (g_theta && (rc = score_reduce(c, sc, (const double*)awr.dev, (const double*)awi.dev, B, P, (double*)ag.dev))) return rc;
        } else {
            // plain theta-VJP (no score output, no resident scores wanted): the batch in slices through a bounded temporary -- the
            // arena only ever grows, and a (B, P, 2) score matrix (140 MB at n = 13, B = 8192) would stay in the context's footprint
            const int Bc = std::min(B, 1024);
            sc = (double*)arena_take(c, sizeof(double) * (size_t)Bc * P * 2);
            double* gt = (double*)arena_take(c, sizeof(double) * (size_t)P);
            if (!sc || !gt) CG_FAIL(c, CG_ERR_HIP, "%s: workspace allocation failed", fn);
            for (int b0 = 0; b0 < B; b0 += Bc) {
                const int nb = std::min(Bc, B - b0);
                if ((rc = scores_of((const double*)ax.dev + (size_t)b0 * N, (const int*)as.dev + (size_t)b0 * n, nb, sc))) return rc;
                if ((rc = score_reduce(c, sc, (const double*)awr.dev + b0, (const double*)awi.dev + b0, nb, P, b0 ? gt : (double*)ag.dev))) return rc;
                if (b0 && (rc = cg_axpby(c, 1.0, gt, 1.0, (double*)ag.dev, (size_t)P))) return rc;
            }
        }
    } else {              // any depth / widths: dual-number reverse passes of the primal flow (cg_generic.hpp)
        const int grid = std::min(B, c->cu_count * 2 * CG_DERIV_WAVES);
        double* partial = g_theta ? (double*)arena_take(c, sizeof(double) * (size_t)grid * P) : nullptr;
        if (g_theta && !partial) CG_FAIL(c, CG_ERR_HIP, "%s: workspace allocation failed", fn);
        if ((rc = cg_gen_run_param_vjp(c, grid, (const double*)ax.dev, (const int*)as.dev, B, (const double*)awr.dev, (const double*)awi.dev,
                                       partial, (double*)asc.dev))) return rc;
        if (g_theta)
            hipLaunchKernelGGL(k_reduce_rows, dim3((P + 127) / 128), dim3(128), 0, c->stream, (const double*)partial, grid, P, (double*)ag.dev);
    }
    if (fisher) {
        if ((rc = fisher_complex_launch(c, (const double*)asc.dev, B, P, (double*)afi.dev))) return rc;
        if (smean && (rc = score_reduce(c, (const double*)asc.dev, nullptr, nullptr, B, 2 * P, (double*)asm_.dev))) return rc;
    }
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}

int cg_param_vjp(cg_ctx* c, const double* x, const int32_t* sidx, int B, const double* w_re, const double* w_im, double* g_theta) {
    if (c && !g_theta) CG_FAIL(c, CG_ERR_ARG, "cg_param_vjp: g_theta is NULL");
    if (c && (!w_re || !w_im) && B > 0) CG_FAIL(c, CG_ERR_ARG, "cg_param_vjp: weights are NULL");
    return run_vjp(c, "cg_param_vjp", x, sidx, B, w_re, w_im, g_theta, nullptr);
}
int cg_quantum_score(cg_ctx* c, const double* x, const int32_t* sidx, int B, double* score) {
    if (c && !score) CG_FAIL(c, CG_ERR_ARG, "cg_quantum_score: score is NULL");
    return run_vjp(c, "cg_quantum_score", x, sidx, B, nullptr, nullptr, nullptr, score);
}
/* Resident per-sample scores: computed once per (x, state_idx, theta), then reused for the theta-VJPs of the loss
 * (weights 2 Re/Im E_clip / B and 2 / B, main.py:278) and for the Fisher matrix -- 2 reverse sweeps instead of 6. */
int cg_scores_compute(cg_ctx* c, const double* x, const int32_t* sidx, int B) {
    if (c && B <= 0) CG_FAIL(c, CG_ERR_ARG, "cg_scores_compute: empty batch");
    return run_vjp(c, "cg_scores_compute", x, sidx, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true);
}
int cg_scores_vjp(cg_ctx* c, const double* w_re, const double* w_im, double* g_theta) {
    if (!c) return CG_ERR_ARG;
    if (!c->d_scores || c->scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_scores_vjp: cg_scores_compute has not been called");
    if (!w_re || !w_im || !g_theta) CG_FAIL(c, CG_ERR_ARG, "cg_scores_vjp: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    int rc; const int B = c->scores_B, P = c->P;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_scores_vjp: arena");
    Arg awr{(void*)w_re, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg awi{(void*)w_im, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg ag{g_theta, nullptr, sizeof(double) * (size_t)P, false, true};
    Arg* all[] = {&awr, &awi, &ag};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    if ((rc = score_reduce(c, (const double*)c->d_scores, (const double*)awr.dev, (const double*)awi.dev, B, P, (double*)ag.dev))) return rc;
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}
int cg_scores_fisher(cg_ctx* c, double* fisher, double* score_mean) {
    if (!c) return CG_ERR_ARG;
    if (!c->d_scores || c->scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_scores_fisher: cg_scores_compute has not been called");
    if (!fisher || !score_mean) CG_FAIL(c, CG_ERR_ARG, "cg_scores_fisher: NULL output");
    CG_HIP(c, hipSetDevice(c->device));
    int rc; const int B = c->scores_B, P = c->P;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_scores_fisher: arena");
    Arg afi{fisher, nullptr, sizeof(double) * (size_t)P * P, false, true};
    Arg asm_{score_mean, nullptr, sizeof(double) * (size_t)P * 2, false, true};
    Arg* all[] = {&afi, &asm_};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    if ((rc = fisher_complex_launch(c, (const double*)c->d_scores, B, P, (double*)afi.dev))) return rc;
    if ((rc = score_reduce(c, (const double*)c->d_scores, nullptr, nullptr, B, 2 * P, (double*)asm_.dev))) return rc;
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}
int cg_scores_mean(cg_ctx* c, double* score_mean) {
    if (!c) return CG_ERR_ARG;
    if (!c->d_scores || c->scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_scores_mean: cg_scores_compute has not been called");
    if (!score_mean) CG_FAIL(c, CG_ERR_ARG, "cg_scores_mean: NULL output");
    CG_HIP(c, hipSetDevice(c->device));
    int rc; const int B = c->scores_B, P = c->P;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_scores_mean: arena");
    Arg asm_{score_mean, nullptr, sizeof(double) * (size_t)P * 2, false, true};
    if ((rc = stage(c, asm_))) return rc;
    if ((rc = score_reduce(c, (const double*)c->d_scores, nullptr, nullptr, B, 2 * P, (double*)asm_.dev))) return rc;
    if ((rc = unstage(c, asm_))) return rc;
    return finish(c);
}
int cg_quantum_fisher(cg_ctx* c, const double* x, const int32_t* sidx, int B, double* fisher, double* score_mean) {
    if (c && (!fisher || !score_mean)) CG_FAIL(c, CG_ERR_ARG, "cg_quantum_fisher: NULL output");
    if (c && B <= 0) CG_FAIL(c, CG_ERR_ARG, "cg_quantum_fisher: empty batch");
    return run_vjp(c, "cg_quantum_fisher", x, sidx, B, nullptr, nullptr, nullptr, nullptr, fisher, score_mean);
}

}  // extern "C"

#if defined(CG_STAMPS)
CG_STAMP_READER(cg_debug_stamps_derivs)      /* diagnostic builds only (tools/stamps*.py): the per-phase cycle counters of this unit's kernels */
#endif
