// cg_hip.hip -- context, memory helpers, Ewald / wrap kernels, fp64 peak micro-benchmarks, RCCL communicator and the SR
// solver of libcoulombgas_hip.so (C-ABI: include/coulombgas.h).  The walker kernels live in cg_k_*.hip.
#include "cg_host.hpp"
#include "cg_ewald.hpp"
#include "cg_rng.hpp"

thread_local std::string g_last_error;

template <int D>
__global__ void k_ewald(const double* __restrict__ x, int B, int n, double L, double kappa, double rs,
                        const int* __restrict__ G, const double* __restrict__ gk, int nG, int Gmax, double g0,
                        double* __restrict__ V) {
    extern __shared__ double lds[];
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    const int N = n * D;
    double* xs = lds;
    double* rest = lds + ((N + 1) & ~1);
    for (int w = blockIdx.x; w < B; w += gridDim.x) {
        for (int e = b.tid; e < N; e += b.nthr) xs[e] = x[(size_t)w * N + e];
        b.sync();
        const double v = cg_ewald_walker<D>(b, xs, n, L, kappa, rs, G, gk, nG, Gmax, g0, rest);
        if (b.tid == 0) V[w] = v;
        b.sync();
    }
}

__global__ void k_wrap(double* __restrict__ x, size_t count, double L) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { const double v = x[i]; x[i] = v - L * floor(v / L); }
}

__global__ void k_scale(double* __restrict__ buf, size_t count, double s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) buf[i] *= s;
}

// ---- optimisation-step kernels that keep the per-walker energies on the device --------------------------------------
// K8: local energies and their moments, src/VMC.py:39-58 before the pmean.
//   kinetic = -lap - sum_ia grad_ia^2 (complex square, :39);  E_loc = kinetic + V + Vconst (:40-41);
//   F_loc = logp_states / beta + Re E_loc (:42);  sums of K, K^2, V, V^2, E, E^2, F, F^2, -logp, logp^2 (:44-53).
// 16 lanes (one DPP row) per walker: the lanes stride over the n*d complex gradient entries (coalesced), the row is
// reduced by xor-shuffles, lane 0 of the row assembles the energies; the ten sums of a block's 16 walkers go through
// one wave-butterfly + LDS reduction; one row of partial sums per block, summed in fixed order by k_rows_sum.
__global__ void __launch_bounds__(256) k_local_energy(const double* __restrict__ grad, const double* __restrict__ lap,
                                                      const double* __restrict__ V, const double* __restrict__ logp_states,
                                                      int B, int N, double Vconst, double rbeta, double* __restrict__ eloc,
                                                      double* __restrict__ floc, double* __restrict__ partial) {
    __shared__ double scratch[10 * 4];
    const int sub = threadIdx.x & 15, w = blockIdx.x * 16 + (threadIdx.x >> 4);
    double sr = 0.0, si = 0.0;
    if (w < B) {
        const double* g = grad + (size_t)w * N * 2;
        for (int e = sub; e < N; e += 16) { const double a = g[2 * e], b = g[2 * e + 1]; sr += a * a - b * b; si += 2.0 * a * b; }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
    double v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = 0.0;
    if (w < B && sub == 0) {
        const double kr = -lap[2 * w] - sr, ki = -lap[2 * w + 1] - si;
        const double pot = V[w] + Vconst;
        const double er = kr + pot;
        const double lps = logp_states ? logp_states[w] : 0.0;
        const double fl = lps * rbeta + er;
        eloc[2 * w] = er; eloc[2 * w + 1] = ki;
        if (floc) floc[w] = fl;
        v[0] = kr; v[1] = kr * kr; v[2] = pot; v[3] = pot * pot; v[4] = er; v[5] = er * er; v[6] = fl; v[7] = fl * fl;
        v[8] = -lps; v[9] = lps * lps;
    }
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    cg_block_sum_n<10>(b, v, scratch);
    if (threadIdx.x < 10) partial[(size_t)blockIdx.x * 10 + threadIdx.x] = v[threadIdx.x];
}
// mean absolute deviation, local part: sum_b |e_b - c|  (src/VMC.py:63 real F_loc, :72 complex E_loc; c = the pmean'd mean,
// read from device memory so that no host round trip sits between the all-reduce and this kernel)
__global__ void __launch_bounds__(256) k_abs_dev(const double* __restrict__ e, int B, int cplx, const double* __restrict__ center,
                                                 double* __restrict__ partial) {
    __shared__ double scratch[4];
    const int w = blockIdx.x * 256 + threadIdx.x;
    const double c = center[0];
    double v[1] = {0.0};
    if (w < B) {
        if (cplx) { const double dr = e[2 * w] - c, di = e[2 * w + 1]; v[0] = sqrt(dr * dr + di * di); }
        else v[0] = fabs(e[w] - c);
    }
    const CgBlk b{(int)threadIdx.x, (int)blockDim.x};
    cg_block_sum_n<1>(b, v, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = v[0];
}
// out[p] = scale * sum_r partial[r][p], rows summed in fixed order (deterministic)
__global__ void k_rows_sum(const double* __restrict__ partial, int rows, int P, double scale, double* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    double a = 0.0;
    for (int r = 0; r < rows; ++r) a += partial[(size_t)r * P + p];
    out[p] = a * scale;
}
// Weights of the theta-VJP behind jax.jacrev(quantum_lossfn) (src/VMC.py:72-76, main.py:278):
//   E_c = clip(E_loc, <E> - 5 tv, <E> + 5 tv) with the lexicographic complex order of the JAX generation the reference
//   targets;  d/dtheta 2 mean Re(logPsi conj(E_c)) = sum_b (2/B) [Re E_c,b dRe logPsi_b + Im E_c,b dIm logPsi_b].
// cplx = 0: real clip of F_loc (src/VMC.py:64), w_re = scale * F_c, w_im untouched.
__global__ void __launch_bounds__(256) k_clip_weights(const double* __restrict__ e, int B, int cplx, const double* __restrict__ center,
                                                      const double* __restrict__ tv, double scale, double* __restrict__ w_re,
                                                      double* __restrict__ w_im) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    if (w >= B) return;
    const double lo = center[0] - 5.0 * tv[0], hi = center[0] + 5.0 * tv[0];
    if (!cplx) { const double f = e[w]; w_re[w] = scale * fmin(fmax(f, lo), hi); return; }
    double re = e[2 * w], im = e[2 * w + 1];
    if (re < lo || (re == lo && im < 0.0)) { re = lo; im = 0.0; }          // maximum(a, lo)
    if (hi < re || (hi == re && 0.0 < im)) { re = hi; im = 0.0; }          // minimum(., hi)
    w_re[w] = scale * re; w_im[w] = scale * im;
}
// standard normals from the Philox stream (seed, offset + i): the Hutchinson probe jax.random.normal(key, x.shape) of
// src/logpsi.py:110 drawn on the device (bit-parity with jax.random is not a goal; parity runs pass v in)
__global__ void __launch_bounds__(256) k_randn(double* __restrict__ out, size_t count, uint64_t seed, uint64_t offset) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = cg_philox_normal(seed, offset + i, 0x5eedu, 0u);
}
// Autoregressive Transformer density matrix (cg_van.hpp): one wave per sample, `waves` samples per workgroup.
// LDS: [parameters (if plds)] [per-wave scratch: activations + key / value cache].
template <bool SAMPLE, bool SHIPPED>       // SHIPPED: model size 16, hidden 32, two layers, four heads at compile time
__global__ void __launch_bounds__(256) k_van(CgVanModel m, const double* __restrict__ Pg, const double* __restrict__ sp, int B,
                                             int* __restrict__ sidx, const double* __restrict__ unif, uint64_t seed, uint64_t offset,
                                             double* __restrict__ logp, int plds) {
    extern __shared__ double van_lds[];
    const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
    const double* P = Pg;
    double* scratch = van_lds;
    if (plds) {
        for (int e = threadIdx.x; e < m.total; e += blockDim.x) van_lds[e] = Pg[e];
        P = van_lds; scratch = van_lds + ((m.total + 1) & ~1);
        __syncthreads();
    }
    double* lw = scratch + (size_t)wave * m.wave_doubles;
    for (int s = blockIdx.x * waves + wave; s < B; s += gridDim.x * waves) {
        const double lp = SHIPPED ? cg_van_sequence<SAMPLE, 16, 32, 2, 4>(m, P, sp, sidx + (size_t)s * m.n, lw, unif ? unif + (size_t)s * m.n * m.M : nullptr, seed, offset + (uint64_t)s)
                                  : cg_van_sequence<SAMPLE>(m, P, sp, sidx + (size_t)s * m.n, lw, unif ? unif + (size_t)s * m.n * m.M : nullptr, seed, offset + (uint64_t)s);
        if ((threadIdx.x & 63) == 0 && logp) logp[s] = lp;
    }
}

// per-sample gradients of log p (cg_van_gradient): one wave per sample; stash: (n-1) * token-stash doubles per wave in HBM
__global__ void __launch_bounds__(512) k_van_grad(CgVanModel m, const double* __restrict__ Pg, const double* __restrict__ sp, int B,
                                                  const int* __restrict__ sidx, double* __restrict__ S, double* __restrict__ stash_all, int plds) {
    extern __shared__ double van_lds[];
    const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
    const double* P = Pg;
    double* scratch = van_lds;
    if (plds) {
        for (int e = threadIdx.x; e < m.total; e += blockDim.x) van_lds[e] = Pg[e];
        P = van_lds; scratch = van_lds + ((m.total + 1) & ~1);
        __syncthreads();
    }
    double* lw = scratch + (size_t)wave * cg_van_grad_wave_doubles(m);
    double* stash = stash_all + (size_t)(blockIdx.x * waves + wave) * (size_t)(m.n > 1 ? m.n - 1 : 1) * cg_van_token_stash(m);
    for (int s = blockIdx.x * waves + wave; s < B; s += gridDim.x * waves)
        cg_van_gradient(m, P, sp, sidx + (size_t)s * m.n, lw, stash, S + (size_t)s * m.total);
}
// the shipped model dimensions at compile time (model size 16, hidden 32, two layers, four heads), gradient row accumulated in HBM / L2
__global__ void __launch_bounds__(512) k_van_grad_s(CgVanModel m, const double* __restrict__ Pg, const double* __restrict__ sp, int B,
                                                    const int* __restrict__ sidx, double* __restrict__ S, double* __restrict__ stash_all, int plds) {
    extern __shared__ double van_lds[];
    const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
    const double* P = Pg;
    double* scratch = van_lds;
    if (plds) {
        for (int e = threadIdx.x; e < m.total; e += blockDim.x) van_lds[e] = Pg[e];
        P = van_lds; scratch = van_lds + ((m.total + 1) & ~1);
        __syncthreads();
    }
    double* lw = scratch + (size_t)wave * cg_van_grad_wave_doubles(m);
    double* stash = stash_all + (size_t)(blockIdx.x * waves + wave) * (size_t)(m.n > 1 ? m.n - 1 : 1) * cg_van_token_stash(m);
    for (int s = blockIdx.x * waves + wave; s < B; s += gridDim.x * waves)
        cg_van_gradient_static<16, 32, 2, 4>(m, P, sp, sidx + (size_t)s * m.n, lw, stash, S + (size_t)s * m.total);
}
// the same with the gradient row accumulated in registers (CgVanAccReg): at most four waves per workgroup = one per SIMD, the shipped
// model dimensions (model size 16, hidden 32, two layers), M <= 64 MR orbitals
template <int MS, int HS, int NL, int MR>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_van_grad_reg(CgVanModel m, const double* __restrict__ Pg, const double* __restrict__ sp, int B,
               const int* __restrict__ sidx, double* __restrict__ S, double* __restrict__ stash_all, int plds) {
    extern __shared__ double van_lds[];
    const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
    const double* P = Pg;
    double* scratch = van_lds;
    if (plds) {
        for (int e = threadIdx.x; e < m.total; e += blockDim.x) van_lds[e] = Pg[e];
        P = van_lds; scratch = van_lds + ((m.total + 1) & ~1);
        __syncthreads();
    }
    double* lw = scratch + (size_t)wave * cg_van_grad_wave_doubles(m);
    double* stash = stash_all + (size_t)(blockIdx.x * waves + wave) * (size_t)(m.n > 1 ? m.n - 1 : 1) * cg_van_token_stash(m);
    for (int s = blockIdx.x * waves + wave; s < B; s += gridDim.x * waves)
        cg_van_gradient_reg<MS, HS, NL, MR>(m, P, sp, sidx + (size_t)s * m.n, lw, stash, S + (size_t)s * m.total);
}
// out[slice][p] = sum_{b in slice} w[b] S[b][p] for a real (B x P) matrix (the classical theta-VJP from resident scores)
__global__ void __launch_bounds__(256) k_gemv_t(const double* __restrict__ S, const double* __restrict__ w, int B, int P, int chunk,
                                                double* __restrict__ out) {
    __shared__ double part[256];
    const int p = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const int b0 = blockIdx.y * chunk, b1 = min(B, b0 + chunk);
    double a = 0.0;
    if (p < P) for (int b = b0 + rg; b < b1; b += 4) a = fma(w[b], S[(size_t)b * P + p], a);
    part[threadIdx.x] = a;
    __syncthreads();
    if (rg == 0 && p < P) out[(size_t)blockIdx.y * P + p] = part[threadIdx.x] + part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
}

// out[i][j] = in[perm[i]][perm[j]]: the classical Fisher matrix from the device's flat parameter order to the caller's
__global__ void __launch_bounds__(256) k_permute_sym(const double* __restrict__ in, const int* __restrict__ perm, int P, double* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j < P) out[(size_t)i * P + j] = in[(size_t)perm[i] * P + perm[j]];
}
__global__ void __launch_bounds__(256) k_axpby(double a, const double* __restrict__ x, double bcoef, double* __restrict__ y, size_t count) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) y[i] = a * x[i] + (bcoef != 0.0 ? bcoef * y[i] : 0.0);
}

// fp64 peak micro-benchmarks (roofline denominators for bench.py; /opt/skills/guides has no f64 row)
__global__ void __launch_bounds__(256) k_peak_fma64(double* out, int iters, double a, double b) {
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (double)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(v[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 12345.678) out[0] = s;
}
typedef double cg_d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_peak_mfma64(double* out, int iters, double a, double b) {
    cg_d4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1}, c2 = {2, 2, 2, 2}, c3 = {3, 3, 3, 3};
    const double av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c3, 0, 0, 0);
    }
    double s = c0[0] + c1[1] + c2[2] + c3[3];
    if (s == 12345.678) out[0] = s;
}

int cg_van_grad_par_launch(cg_ctx* c, const int* sidx_dev, int B, double* S);      // cg_k_van.hip: positions-in-parallel reverse pass of the Transformer

extern "C" {

const char* cg_last_error(const cg_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int cg_create(cg_ctx** out, int device, int n, int dim, int depth, int spsize, int tpsize, double L,
              const double* sp_indices, int M) {
    if (!out) CG_FAIL((cg_ctx*)nullptr, CG_ERR_ARG, "cg_create: out is NULL");
    *out = nullptr;
    if (n < 1 || (dim != 2 && dim != 3) || depth < 2 || spsize < 1 || tpsize < 1 || !(L > 0) || !sp_indices || M < n)
        CG_FAIL((cg_ctx*)nullptr, CG_ERR_ARG, "cg_create: bad argument (n=%d dim=%d depth=%d spsize=%d tpsize=%d L=%g M=%d); depth >= 2 "
                "(src/flow.py:52 is ill-formed for depth 1), dim in {2,3}, M >= n", n, dim, depth, spsize, tpsize, L, M);
    const bool fast_ok = cg_fast_supported(depth, dim, spsize, tpsize);
    if (!fast_ok && (depth > CG_GEN_MAXDEPTH || spsize > 256 || tpsize > 256))
        CG_FAIL((cg_ctx*)nullptr, CG_ERR_UNSUPPORTED, "cg_create: depth=%d spsize=%d tpsize=%d exceeds the general path's limits "
                "(depth <= %d, widths <= 256)", depth, spsize, tpsize, CG_GEN_MAXDEPTH);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) CG_FAIL((cg_ctx*)nullptr, CG_ERR_HIP, "cg_create: no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) CG_FAIL((cg_ctx*)nullptr, CG_ERR_ARG, "cg_create: device %d out of range [0,%d)", device, ndev);
    cg_ctx* c = new cg_ctx();
    c->device = device; c->n = n; c->dim = dim; c->depth = depth; c->hs = spsize; c->ht = tpsize; c->M = M; c->L = L;
    c->fast = fast_ok;
    cg_gen_model_init(c->gm, n, dim, depth, spsize, tpsize, L);
    c->gw = cg_gen_ws(c->gm);
    c->gwv = cg_gen_ws(c->gm, true);
    c->P = c->gm.nparam;
    memset(&c->lay, 0, sizeof(c->lay));
    if (fast_ok) {
#define CG_X(D, HS, HT) if (dim == D && spsize == HS && tpsize == HT) c->P = CgFast<D, HS, HT>::NPARAM;
        CG_FAST_CONFIGS(CG_X)
#undef CG_X
        c->lay = cg_fast_layout(n, dim, spsize, tpsize, true, spsize == 16 && tpsize == 16);
        // maximum size of the LDS-resident path: J (n d)^2 + the per-particle factors must fit 160 KiB.  Beyond it (n > ~64
        // at d = 2) the same model runs on the general path (HBM workspace), which provides every entry point.
        const size_t NN = (size_t)n * dim;
        if (sizeof(double) * (CG_TAB_DOUBLES + (size_t)c->lay.total + 2 * ((NN + 1) & ~(size_t)1) + 2) > 160 * 1024) {
            c->fast = false;
            c->P = c->gm.nparam;
        }
    }
    auto fail = [&](const char* what, hipError_t err) {
        g_last_error = std::string("cg_create: ") + what + ": " + hipGetErrorString(err);
        cg_destroy(c); return CG_ERR_HIP;          // frees whatever was created so far (every member is null-checked)
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->cu_count = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return fail("hipEventCreate", e);
    if ((e = hipMalloc((void**)&c->d_theta, sizeof(double) * c->P)) != hipSuccess) return fail("hipMalloc theta", e);
    if ((e = hipMalloc((void**)&c->d_spk, sizeof(double) * (size_t)M * dim)) != hipSuccess) return fail("hipMalloc orbitals", e);
    if ((e = hipMalloc((void**)&c->d_accept, sizeof(unsigned long long))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMalloc((void**)&c->d_tab, sizeof(double) * CG_TAB_DOUBLES)) != hipSuccess) return fail("hipMalloc tables", e);
    { double tabh[CG_TAB_DOUBLES]; cg_tab_fill(tabh);
      if ((e = hipMemcpy(c->d_tab, tabh, sizeof(tabh), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy tables", e); }
    std::vector<double> spk((size_t)M * dim);
    for (size_t i = 0; i < spk.size(); ++i) spk[i] = sp_indices[i] * (2.0 * CG_PI / L);     // src/slater.py:14
    if ((e = hipMemcpy(c->d_spk, spk.data(), sizeof(double) * spk.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy orbitals", e);
    if ((e = hipMalloc((void**)&c->d_rate, sizeof(double))) != hipSuccess) return fail("hipMalloc", e);
    if ((e = hipMemset(c->d_accept, 0, sizeof(unsigned long long))) != hipSuccess) return fail("hipMemset", e);
    *out = c;
    return CG_OK;
}

void cg_destroy(cg_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& ch : c->chunks) (void)hipFree(ch.p);
    if (c->ws) (void)hipFree(c->ws);
    if (c->bounce) (void)hipHostFree(c->bounce);
    if (c->d_scores) (void)hipFree(c->d_scores);
    if (c->d_lay) (void)hipFree(c->d_lay);
    if (c->d_van) (void)hipFree(c->d_van);
    if (c->d_van_scores) (void)hipFree(c->d_van_scores);
    if (c->d_van_sp) (void)hipFree(c->d_van_sp);
    if (c->d_theta) (void)hipFree(c->d_theta);
    if (c->d_spk) (void)hipFree(c->d_spk);
    if (c->d_tab) (void)hipFree(c->d_tab);
    if (c->d_G) (void)hipFree(c->d_G);
    if (c->d_gk) (void)hipFree(c->d_gk);
    if (c->d_accept) (void)hipFree(c->d_accept);
    if (c->d_rate) (void)hipFree(c->d_rate);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int cg_set_pointer_mode(cg_ctx* c, int mode) {
    if (!c) return CG_ERR_ARG;
    if (mode != CG_PTR_HOST && mode != CG_PTR_DEVICE) CG_FAIL(c, CG_ERR_ARG, "cg_set_pointer_mode: mode %d", mode);
    c->ptr_mode = mode; return CG_OK;
}
int cg_sync(cg_ctx* c) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    return CG_OK;
}
int cg_num_params(const cg_ctx* c) { return c ? c->P : CG_ERR_ARG; }

int cg_set_flow_params(cg_ctx* c, const double* theta) {
    if (!c || !theta) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemcpyAsync(c->d_theta, theta, sizeof(double) * c->P, hipMemcpyHostToDevice, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    c->have_theta = true;
    return CG_OK;
}

int cg_set_ewald(cg_ctx* c, double kappa, const int64_t* G, int nG, double rs) {
    if (!c || !G || nG < 1 || !(kappa > 0)) CG_FAIL(c, CG_ERR_ARG, "cg_set_ewald: bad argument");
    CG_HIP(c, hipSetDevice(c->device));
    const int D = c->dim;
    std::vector<int> g32((size_t)nG * D);
    std::vector<double> gk(nG);
    int gmax = 0;
    for (int g = 0; g < nG; ++g) {
        double g2 = 0;
        for (int a = 0; a < D; ++a) {
            int64_t v = G[(size_t)g * D + a];
            if (v > 4096 || v < -4096) CG_FAIL(c, CG_ERR_ARG, "cg_set_ewald: |G| component %lld too large", (long long)v);
            g32[(size_t)g * D + a] = (int)v; gmax = std::max(gmax, (int)std::llabs(v)); g2 += (double)v * (double)v;
        }
        if (g2 == 0) CG_FAIL(c, CG_ERR_ARG, "cg_set_ewald: G = 0 must not be in the list (src/potential.py:16)");
        const double gn = std::sqrt(g2);
        // src/potential.py:54-59
        gk[g] = (D == 3) ? std::exp(-CG_PI * CG_PI * g2 / (kappa * kappa)) / (CG_PI * g2) : std::erfc(CG_PI * gn / kappa) / gn;
    }
    c->g0 = (D == 3) ? -CG_PI / (kappa * kappa) : -2.0 * std::sqrt(CG_PI) / kappa;
    if (c->d_G) { (void)hipFree(c->d_G); c->d_G = nullptr; }
    if (c->d_gk) { (void)hipFree(c->d_gk); c->d_gk = nullptr; }
    CG_HIP(c, hipMalloc((void**)&c->d_G, sizeof(int) * g32.size()));
    CG_HIP(c, hipMalloc((void**)&c->d_gk, sizeof(double) * nG));
    CG_HIP(c, hipMemcpy(c->d_G, g32.data(), sizeof(int) * g32.size(), hipMemcpyHostToDevice));
    CG_HIP(c, hipMemcpy(c->d_gk, gk.data(), sizeof(double) * nG, hipMemcpyHostToDevice));
    c->kappa = kappa; c->rs = rs; c->nG = nG; c->Gmax = gmax; c->have_ewald = true;
    return CG_OK;
}

int cg_dev_alloc(cg_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMalloc(dptr, bytes ? bytes : 8));
    return CG_OK;
}
int cg_dev_free(cg_ctx* c, void* dptr) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    CG_HIP(c, hipFree(dptr));
    return CG_OK;
}
int cg_memcpy_h2d(cg_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    return CG_OK;
}
int cg_memcpy_d2h(cg_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    return CG_OK;
}
int cg_memset(cg_ctx* c, void* dst, int value, size_t bytes) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipMemsetAsync(dst, value, bytes, c->stream));
    return CG_OK;
}
int cg_timer_start(cg_ctx* c) {
    if (!c) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipEventRecord(c->ev0, c->stream));
    return CG_OK;
}
int cg_timer_stop(cg_ctx* c, float* ms) {
    if (!c || !ms) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    CG_HIP(c, hipEventRecord(c->ev1, c->stream));
    CG_HIP(c, hipEventSynchronize(c->ev1));
    CG_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev1));
    return CG_OK;
}
int cg_set_block_threads(cg_ctx* c, int threads) {
    if (!c) return CG_ERR_ARG;
    if (threads != 0 && (threads < 64 || threads > 1024 || threads % 64)) CG_FAIL(c, CG_ERR_ARG, "cg_set_block_threads: %d is not 0 or a multiple of 64 in [64,1024]", threads);
    c->block_threads = threads; return CG_OK;
}
int cg_get_launch_info(cg_ctx* c, int64_t* info) {
    if (!c || !info) return CG_ERR_ARG;
    const int N = c->n * c->dim;
    info[0] = threads_of(c);
    info[1] = (int64_t)sizeof(double) * (CG_TAB_DOUBLES + c->lay.total + 3 * ((N + 1) & ~1) + 2);
    info[2] = c->cu_count; info[3] = c->P; info[4] = c->fast ? 1 : 0; info[5] = info[6] = info[7] = 0;
    return CG_OK;
}

int cg_mcmc_accepts(cg_ctx* c, int64_t* n_accept) {
    if (!c || !n_accept) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    unsigned long long v = 0;
    CG_HIP(c, hipMemcpyAsync(&v, c->d_accept, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    *n_accept = (int64_t)v;
    return CG_OK;
}

int cg_wrap(cg_ctx* c, double* x, int B) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (B == 0) return CG_OK;
    if (!x) CG_FAIL(c, CG_ERR_ARG, "cg_wrap: x is NULL");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_wrap: arena");
    const size_t cnt = (size_t)B * c->n * c->dim;
    Arg ax{x, nullptr, sizeof(double) * cnt, true, true};
    if ((rc = stage(c, ax))) return rc;
    hipLaunchKernelGGL(k_wrap, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, (double*)ax.dev, cnt, c->L);
    if ((rc = unstage(c, ax))) return rc;
    return finish(c);
}

int cg_ewald(cg_ctx* c, const double* x, int B, double* V) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (!c->have_ewald) CG_FAIL(c, CG_ERR_STATE, "cg_ewald: cg_set_ewald has not been called");
    if (B == 0) return CG_OK;
    if (!x || !V) CG_FAIL(c, CG_ERR_ARG, "cg_ewald: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_ewald: arena");
    const int n = c->n, D = c->dim, N = n * D;
    Arg ax{(void*)x, nullptr, sizeof(double) * (size_t)B * N, true, false};
    Arg av{V, nullptr, sizeof(double) * (size_t)B, false, true};
    if ((rc = stage(c, ax)) || (rc = stage(c, av))) return rc;
    const int nt = 256;
    const size_t lds = sizeof(double) * (((N + 1) & ~1) + (size_t)N * (c->Gmax + 1) * 2 + nt);
    if (D == 2) {
        if ((rc = set_lds(c, k_ewald<2>, lds))) return rc;
        hipLaunchKernelGGL((k_ewald<2>), dim3(B), dim3(nt), lds, c->stream, (const double*)ax.dev, B, n, c->L, c->kappa,
                           c->rs, (const int*)c->d_G, (const double*)c->d_gk, c->nG, c->Gmax, c->g0, (double*)av.dev);
    } else {
        if ((rc = set_lds(c, k_ewald<3>, lds))) return rc;
        hipLaunchKernelGGL((k_ewald<3>), dim3(B), dim3(nt), lds, c->stream, (const double*)ax.dev, B, n, c->L, c->kappa,
                           c->rs, (const int*)c->d_G, (const double*)c->d_gk, c->nG, c->Gmax, c->g0, (double*)av.dev);
    }
    if ((rc = unstage(c, av))) return rc;
    return finish(c);
}

/* which: 0 = v_fma_f64 (VALU), 1 = v_mfma_f64_16x16x4_f64.  Returns achieved TFLOP/s (HIP-event timed). */
int cg_microbench_fp64(cg_ctx* c, int which, double* tflops) {
    if (!c || !tflops) return CG_ERR_ARG;
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_microbench_fp64: arena");
    double* out = (double*)arena_take(c, 64);
    const int blocks = c->cu_count * 8, iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        CG_HIP(c, hipEventRecord(c->ev0, c->stream));
        if (which == 0) hipLaunchKernelGGL(k_peak_fma64, dim3(blocks), dim3(256), 0, c->stream, out, iters, 0.999999, 1e-9);
        else hipLaunchKernelGGL(k_peak_mfma64, dim3(blocks), dim3(256), 0, c->stream, out, iters, 0.5, 0.25);
        CG_HIP(c, hipEventRecord(c->ev1, c->stream));
        CG_HIP(c, hipEventSynchronize(c->ev1));
    }
    float ms = 0; CG_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    const double flops = which == 0 ? (double)blocks * 256 * iters * 16 * 2 : (double)blocks * 4 /*waves*/ * iters * 4 * (16.0 * 16 * 4 * 2);
    *tflops = flops / (ms * 1e-3) / 1e12;
    return CG_OK;
}


/* ---- autoregressive Transformer density matrix on the device (src/autoregressive.py, src/sampler.py) ---- */
int cg_van_num_params(int M, int num_layers, int model_size, int num_heads, int hidden_size, int dim) {
    if (num_layers < 1 || num_layers > CG_VAN_MAXLAYERS || num_heads < 1 || model_size % num_heads) return CG_ERR_ARG;
    CgVanModel m; return cg_van_model_init(m, M, num_layers, model_size, num_heads, hidden_size, dim, 1);
}
int cg_van_set_params(cg_ctx* c, int M, int num_layers, int model_size, int num_heads, int hidden_size,
                      const double* sp_indices, const double* params) {
    if (!c || !sp_indices || !params) return CG_ERR_ARG;
    if (num_layers < 1 || num_layers > CG_VAN_MAXLAYERS || num_heads < 1 || model_size % num_heads || model_size < 1 || model_size > 64 ||
        hidden_size < 1 || hidden_size > 256 || M < c->n || M > 256 || c->n > 64)
        CG_FAIL(c, CG_ERR_UNSUPPORTED, "cg_van_set_params: supported: 1..%d layers, model_size <= 64 divisible by num_heads, hidden_size <= 256, "
                "n <= %d <= M <= 256 (got layers=%d model=%d heads=%d hidden=%d M=%d n=%d)", CG_VAN_MAXLAYERS, 64, num_layers, model_size, num_heads, hidden_size, M, c->n);
    CG_HIP(c, hipSetDevice(c->device));
    CgVanModel m; const int total = cg_van_model_init(m, M, num_layers, model_size, num_heads, hidden_size, c->dim, c->n);
    if (!c->have_van || c->van.total != total || c->van.M != M) {
        CG_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_van) { (void)hipFree(c->d_van); c->d_van = nullptr; }
        if (c->d_van_sp) { (void)hipFree(c->d_van_sp); c->d_van_sp = nullptr; }
        CG_HIP(c, hipMalloc((void**)&c->d_van, sizeof(double) * total));
        CG_HIP(c, hipMalloc((void**)&c->d_van_sp, sizeof(double) * (size_t)M * c->dim));
    }
    CG_HIP(c, hipMemcpyAsync(c->d_van, params, sizeof(double) * total, hipMemcpyHostToDevice, c->stream));
    CG_HIP(c, hipMemcpyAsync(c->d_van_sp, sp_indices, sizeof(double) * (size_t)M * c->dim, hipMemcpyHostToDevice, c->stream));
    CG_HIP(c, hipStreamSynchronize(c->stream));
    c->van = m; c->have_van = true;
    return CG_OK;
}
static int van_launch(cg_ctx* c, bool sample, int B, int* sidx_dev, const double* unif_dev, uint64_t seed, uint64_t offset, double* logp_dev) {
    const CgVanModel& m = c->van;
    const size_t pbytes = sizeof(double) * (size_t)((m.total + 1) & ~1), wbytes = sizeof(double) * (size_t)m.wave_doubles;
    int waves = 4, plds = 1;
    while (waves > 1 && pbytes + waves * wbytes > 160 * 1024) --waves;
    if (pbytes + waves * wbytes > 160 * 1024) { plds = 0; waves = 4; while (waves > 1 && waves * wbytes > 160 * 1024) --waves; }
    const size_t lds = (plds ? pbytes : 0) + waves * wbytes;
    if (lds > 160 * 1024) CG_FAIL(c, CG_ERR_UNSUPPORTED, "cg_van: the key / value cache of one sample needs %zu bytes of LDS", wbytes);
    const int grid = std::min((B + waves - 1) / waves, c->cu_count * 8);
    int rc;
    // compile-time dimensions pay when a CU holds few waves (latency-bound: n = 57, B = 512: 1.29 -> 1.05 ms; n = 29, B = 2048: 1.22 -> 1.02);
    // with every SIMD full (n = 13, B = 8192) the unrolled products cost more registers than they hide: 1.27 against 1.72 ms
    const bool shipped = m.ms == 16 && m.hs == 32 && m.nl == 2 && m.nh == 4 && cg_env_int("CG_VAN_STATIC", B <= 16 * c->cu_count ? 1 : 0);
#define CG_VAN_LAUNCH(S, H)                                                                                                              \
    { if ((rc = set_lds(c, k_van<S, H>, lds))) return rc;                                                                                \
      hipLaunchKernelGGL((k_van<S, H>), dim3(grid), dim3(64 * waves), lds, c->stream, m, (const double*)c->d_van, (const double*)c->d_van_sp, B, \
                         sidx_dev, unif_dev, seed, offset, logp_dev, plds); }
    if (sample) { if (shipped) CG_VAN_LAUNCH(true, true) else CG_VAN_LAUNCH(true, false) }
    else { if (shipped) CG_VAN_LAUNCH(false, true) else CG_VAN_LAUNCH(false, false) }
#undef CG_VAN_LAUNCH
    return CG_OK;
}
int cg_van_log_prob(cg_ctx* c, const int32_t* state_idx, int B, double* logp) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (!c->have_van) CG_FAIL(c, CG_ERR_STATE, "cg_van_log_prob: cg_van_set_params has not been called");
    if (B == 0) return CG_OK;
    if (!state_idx || !logp) CG_FAIL(c, CG_ERR_ARG, "cg_van_log_prob: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_van_log_prob: arena");
    Arg as{(void*)state_idx, nullptr, sizeof(int32_t) * (size_t)B * c->n, true, false};
    Arg al{logp, nullptr, sizeof(double) * (size_t)B, false, true};
    if ((rc = stage(c, as)) || (rc = stage(c, al))) return rc;
    if ((rc = van_launch(c, false, B, (int*)as.dev, nullptr, 0, 0, (double*)al.dev))) return rc;
    if ((rc = unstage(c, al))) return rc;
    return finish(c);
}
int cg_van_sample(cg_ctx* c, int B, uint64_t seed, uint64_t offset, const double* unif, int32_t* state_idx, double* logp) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (!c->have_van) CG_FAIL(c, CG_ERR_STATE, "cg_van_sample: cg_van_set_params has not been called");
    if (B == 0) return CG_OK;
    if (!state_idx) CG_FAIL(c, CG_ERR_ARG, "cg_van_sample: state_idx is NULL");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_van_sample: arena");
    Arg au{(void*)unif, nullptr, sizeof(double) * (size_t)B * c->n * c->van.M, true, false};
    Arg as{state_idx, nullptr, sizeof(int32_t) * (size_t)B * c->n, false, true};
    Arg al{logp, nullptr, sizeof(double) * (size_t)B, false, true};
    if ((rc = stage(c, au)) || (rc = stage(c, as)) || (rc = stage(c, al))) return rc;
    if ((rc = van_launch(c, true, B, (int*)as.dev, (const double*)au.dev, seed, offset, (double*)al.dev))) return rc;
    if ((rc = unstage(c, as)) || (rc = unstage(c, al))) return rc;
    return finish(c);
}

/* classical scores d log p_b / d params (flat parameter order), resident on the device like the quantum scores */
int cg_van_scores_compute(cg_ctx* c, const int32_t* state_idx, int B) {
    if (!c || B <= 0) return CG_ERR_ARG;
    if (!c->have_van) CG_FAIL(c, CG_ERR_STATE, "cg_van_scores_compute: cg_van_set_params has not been called");
    if (!state_idx) CG_FAIL(c, CG_ERR_ARG, "cg_van_scores_compute: state_idx is NULL");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_van_scores_compute: arena");
    const CgVanModel& m = c->van;
    Arg as{(void*)state_idx, nullptr, sizeof(int32_t) * (size_t)B * c->n, true, false};
    if ((rc = stage(c, as))) return rc;
    const size_t need = sizeof(double) * (size_t)B * m.total;
    if (c->van_scores_cap < need) {
        CG_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_van_scores) { (void)hipFree(c->d_van_scores); c->d_van_scores = nullptr; c->van_scores_cap = 0; }
        if (hipMalloc((void**)&c->d_van_scores, need) != hipSuccess) CG_FAIL(c, CG_ERR_HIP, "cg_van_scores_compute: %zu bytes for the classical scores could not be allocated", need);
        c->van_scores_cap = need;
    }
    // the shipped architecture: positions in parallel (cg_van_par.hpp, cg_k_van.hip); every other model: tokens in sequence, below
    if ((rc = cg_van_grad_par_launch(c, (const int*)as.dev, B, c->d_van_scores)) < 0) return rc;
    if (rc == 1) { c->van_scores_B = B; return finish(c); }
    const size_t pbytes = sizeof(double) * (size_t)((m.total + 1) & ~1), wbytes = sizeof(double) * (size_t)cg_van_grad_wave_doubles(m);
    // workgroup shape: as many waves per CU as the 160 KB of LDS allow with the weights staged once per workgroup (the kernel is
    // latency-bound: one wave per SIMD left it at 8.0 ms for B = 8192, n = 13)
    int waves = 0, plds = 1;
    // (workgroups of <= 4 waves of the shipped architecture run the register-accumulating kernel, which is pinned to one wave per SIMD:
    // never more than 4 waves per CU whatever the LDS would allow -- the estimates below count that)
    const bool reg_possible = cg_env_int("CG_VAN_GRAD_REG", 1) > 0 && m.ms == 16 && m.hs == 32 && m.nl == 2 && m.nh == 4 && m.dim * m.ms <= 64 && m.M <= 256;
    auto cap = [&](int w, int per_cu) { return (reg_possible && w <= 4) ? std::min(per_cu, 4) : per_cu; };
    { int best = 0;
      for (int w = 1; w <= 8; ++w) {
          const size_t need = pbytes + w * wbytes;
          if (need > 160 * 1024) break;
          const int per_cu = cap(w, w * (int)((160 * 1024) / need));
          if (per_cu > best) { best = per_cu; waves = w; }
      } }
    if (waves == 0) { plds = 0; waves = 4; while (waves > 1 && waves * wbytes > 160 * 1024) --waves; }
    // small batches of long sequences (n = 57, B = 512: the staged weights leave LDS for ONE wave per CU, two rounds of samples): without the
    // staged copy (weights through L1 / L2) more waves fit a CU; taken when it saves >= 1.4x in rounds (8.5 -> 5.8 ms there; at n = 13,
    // B = 8192 the staged weights win, 5.8 against 6.3 ms)
    if (plds) {
        int w0 = 1; while (w0 < 8 && (size_t)(w0 + 1) * wbytes <= 160 * 1024) ++w0;
        int best0 = 0, wsel = 1;
        for (int w = 1; w <= w0; ++w) { const int per_cu = cap(w, w * (int)((160 * 1024) / (w * wbytes))); if (per_cu > best0 || (per_cu == best0 && w <= 2)) { best0 = per_cu; wsel = w; } }
        const int per_cu1 = cap(waves, waves * (int)((160 * 1024) / (pbytes + waves * wbytes)));
        const int rounds1 = (B + c->cu_count * per_cu1 - 1) / (c->cu_count * per_cu1), rounds0 = (B + c->cu_count * best0 - 1) / (c->cu_count * best0);
        if (rounds0 * 7 <= rounds1 * 5) { plds = 0; waves = wsel; }
    }
    if (cg_env_int("CG_VAN_GRAD_REG", 1) == 2 && waves > 4) waves = 4;
    const size_t lds = (plds ? pbytes : 0) + waves * wbytes;
    if (lds > 160 * 1024) CG_FAIL(c, CG_ERR_UNSUPPORTED, "cg_van_scores_compute: one sample needs %zu bytes of LDS", wbytes);
    const int grid = std::min((B + waves - 1) / waves, c->cu_count * 4);
    const size_t stash_per_wave = (size_t)(m.n > 1 ? m.n - 1 : 1) * cg_van_token_stash(m);
    if ((rc = ensure_ws(c, sizeof(double) * stash_per_wave * (size_t)grid * waves))) return rc;
    const int regmode = cg_env_int("CG_VAN_GRAD_REG", 1);
    const bool shipped = m.ms == 16 && m.hs == 32 && m.nl == 2 && m.nh == 4;
    const bool reg_ok = regmode > 0 && waves <= 4 && shipped && m.dim * m.ms <= 64 && m.M <= 256;
#define CG_VAN_REG_LAUNCH(MR)                                                                                                         \
    { if ((rc = set_lds(c, k_van_grad_reg<16, 32, 2, MR>, lds))) return rc;                                                           \
      hipLaunchKernelGGL((k_van_grad_reg<16, 32, 2, MR>), dim3(grid), dim3(64 * waves), lds, c->stream, m, (const double*)c->d_van,   \
                         (const double*)c->d_van_sp, B, (const int*)as.dev, c->d_van_scores, (double*)c->ws, plds); }
    if (reg_ok && m.M <= 64) CG_VAN_REG_LAUNCH(1)
    else if (reg_ok && m.M <= 128) CG_VAN_REG_LAUNCH(2)
    else if (reg_ok && m.M <= 192) CG_VAN_REG_LAUNCH(3)
    else if (reg_ok) CG_VAN_REG_LAUNCH(4)
    else if (shipped && regmode >= 0) {
        if ((rc = set_lds(c, k_van_grad_s, lds))) return rc;
        hipLaunchKernelGGL(k_van_grad_s, dim3(grid), dim3(64 * waves), lds, c->stream, m, (const double*)c->d_van, (const double*)c->d_van_sp, B,
                           (const int*)as.dev, c->d_van_scores, (double*)c->ws, plds);
    } else {
        if ((rc = set_lds(c, k_van_grad, lds))) return rc;
        hipLaunchKernelGGL(k_van_grad, dim3(grid), dim3(64 * waves), lds, c->stream, m, (const double*)c->d_van, (const double*)c->d_van_sp, B,
                           (const int*)as.dev, c->d_van_scores, (double*)c->ws, plds);
    }
#undef CG_VAN_REG_LAUNCH
    c->van_scores_B = B;
    return finish(c);
}
int cg_van_scores_get(cg_ctx* c, double* scores) {
    if (!c || !scores) return CG_ERR_ARG;
    if (!c->d_van_scores || c->van_scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_van_scores_get: cg_van_scores_compute has not been called");
    CG_HIP(c, hipSetDevice(c->device));
    const size_t bytes = sizeof(double) * (size_t)c->van_scores_B * c->van.total;
    CG_HIP(c, hipMemcpyAsync(scores, c->d_van_scores, bytes, c->ptr_mode == CG_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    return finish(c);
}
int cg_van_scores_vjp(cg_ctx* c, const double* w, double* g) {
    if (!c || !w || !g) return CG_ERR_ARG;
    if (!c->d_van_scores || c->van_scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_van_scores_vjp: cg_van_scores_compute has not been called");
    CG_HIP(c, hipSetDevice(c->device));
    int rc; const int B = c->van_scores_B, P = c->van.total;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_van_scores_vjp: arena");
    Arg aw{(void*)w, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg ag{g, nullptr, sizeof(double) * (size_t)P, false, true};
    if ((rc = stage(c, aw)) || (rc = stage(c, ag))) return rc;
    const int nsl = std::max(1, std::min(64, (B + 63) / 64)), chunk = (B + nsl - 1) / nsl;
    double* partial = (double*)arena_take(c, sizeof(double) * (size_t)nsl * P);
    if (!partial) CG_FAIL(c, CG_ERR_HIP, "cg_van_scores_vjp: workspace allocation failed");
    hipLaunchKernelGGL(k_gemv_t, dim3((P + 63) / 64, nsl), dim3(256), 0, c->stream, (const double*)c->d_van_scores, (const double*)aw.dev, B, P, chunk, partial);
    hipLaunchKernelGGL(k_rows_sum, dim3((P + 127) / 128), dim3(128), 0, c->stream, (const double*)partial, nsl, P, 1.0, (double*)ag.dev);
    if ((rc = unstage(c, ag))) return rc;
    return finish(c);
}
int cg_van_scores_fisher(cg_ctx* c, const int32_t* perm, double* fisher) {
    if (!c || !fisher) return CG_ERR_ARG;
    if (!c->d_van_scores || c->van_scores_B <= 0) CG_FAIL(c, CG_ERR_STATE, "cg_van_scores_fisher: cg_van_scores_compute has not been called");
    CG_HIP(c, hipSetDevice(c->device));
    int rc; const int P = c->van.total;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_van_scores_fisher: arena");
    Arg af{fisher, nullptr, sizeof(double) * (size_t)P * P, false, true};
    if ((rc = stage(c, af))) return rc;
    double* raw = (double*)af.dev;
    Arg ap{(void*)perm, nullptr, sizeof(int32_t) * (size_t)P, true, false};
    if (perm) {
        if ((rc = stage(c, ap))) return rc;
        raw = (double*)arena_take(c, sizeof(double) * (size_t)P * P);
        if (!raw) CG_FAIL(c, CG_ERR_HIP, "cg_van_scores_fisher: workspace allocation failed");
    }
    if ((rc = cg_fisher_real_nr(c, c->d_van_scores, c->van_scores_B, P, raw))) return rc;
    if (perm) hipLaunchKernelGGL(k_permute_sym, dim3((P + 255) / 256, P), dim3(256), 0, c->stream, (const double*)raw, (const int*)ap.dev, P, (double*)af.dev);
    if ((rc = unstage(c, af))) return rc;
    return finish(c);
}

/* ---- device-resident optimisation step (src/VMC.py:39-76, main.py:277-289) ---- */
int cg_local_energy(cg_ctx* c, const double* grad, const double* lap, const double* V, const double* logp_states, int B,
                    double Vconst, double beta, double* eloc, double* floc, double* moments) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (B == 0) return CG_OK;
    if (!grad || !lap || !V || !eloc || !moments || !(beta > 0)) CG_FAIL(c, CG_ERR_ARG, "cg_local_energy: NULL argument or beta <= 0");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_local_energy: arena");
    const int N = c->n * c->dim;
    Arg ag{(void*)grad, nullptr, sizeof(double) * (size_t)B * N * 2, true, false};
    Arg al{(void*)lap, nullptr, sizeof(double) * (size_t)B * 2, true, false};
    Arg av{(void*)V, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg as{(void*)logp_states, nullptr, sizeof(double) * (size_t)B, true, false};
    Arg ae{eloc, nullptr, sizeof(double) * (size_t)B * 2, false, true};
    Arg af{floc, nullptr, sizeof(double) * (size_t)B, false, true};
    Arg am{moments, nullptr, sizeof(double) * 10, false, true};
    Arg* all[] = {&ag, &al, &av, &as, &ae, &af, &am};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    const int grid = (B + 15) / 16;
    double* partial = (double*)arena_take(c, sizeof(double) * (size_t)grid * 10);
    if (!partial) CG_FAIL(c, CG_ERR_HIP, "cg_local_energy: workspace allocation failed");
    hipLaunchKernelGGL(k_local_energy, dim3(grid), dim3(256), 0, c->stream, (const double*)ag.dev, (const double*)al.dev, (const double*)av.dev,
                       (const double*)as.dev, B, N, Vconst, 1.0 / beta, (double*)ae.dev, (double*)af.dev, partial);
    hipLaunchKernelGGL(k_rows_sum, dim3(1), dim3(64), 0, c->stream, (const double*)partial, grid, 10, 1.0 / (double)B, (double*)am.dev);
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}
int cg_abs_dev(cg_ctx* c, const double* e, int B, int is_complex, const double* center, double* out) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (B == 0) return CG_OK;
    if (!e || !center || !out) CG_FAIL(c, CG_ERR_ARG, "cg_abs_dev: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_abs_dev: arena");
    Arg ae{(void*)e, nullptr, sizeof(double) * (size_t)B * (is_complex ? 2 : 1), true, false};
    Arg ac{(void*)center, nullptr, sizeof(double), true, false};
    Arg ao{out, nullptr, sizeof(double), false, true};
    Arg* all[] = {&ae, &ac, &ao};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    const int grid = (B + 255) / 256;
    double* partial = (double*)arena_take(c, sizeof(double) * (size_t)grid);
    if (!partial) CG_FAIL(c, CG_ERR_HIP, "cg_abs_dev: workspace allocation failed");
    hipLaunchKernelGGL(k_abs_dev, dim3(grid), dim3(256), 0, c->stream, (const double*)ae.dev, B, is_complex ? 1 : 0, (const double*)ac.dev, partial);
    hipLaunchKernelGGL(k_rows_sum, dim3(1), dim3(64), 0, c->stream, (const double*)partial, grid, 1, 1.0 / (double)B, (double*)ao.dev);
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}
int cg_clip_weights(cg_ctx* c, const double* e, int B, int is_complex, const double* center, const double* tv, double scale,
                    double* w_re, double* w_im) {
    if (!c || B < 0) return CG_ERR_ARG;
    if (B == 0) return CG_OK;
    if (!e || !center || !tv || !w_re || (is_complex && !w_im)) CG_FAIL(c, CG_ERR_ARG, "cg_clip_weights: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_clip_weights: arena");
    Arg ae{(void*)e, nullptr, sizeof(double) * (size_t)B * (is_complex ? 2 : 1), true, false};
    Arg ac{(void*)center, nullptr, sizeof(double), true, false};
    Arg at{(void*)tv, nullptr, sizeof(double), true, false};
    Arg awr{w_re, nullptr, sizeof(double) * (size_t)B, false, true};
    Arg awi{is_complex ? w_im : nullptr, nullptr, sizeof(double) * (size_t)B, false, true};
    Arg* all[] = {&ae, &ac, &at, &awr, &awi};
    for (Arg* a : all) if ((rc = stage(c, *a))) return rc;
    hipLaunchKernelGGL(k_clip_weights, dim3((B + 255) / 256), dim3(256), 0, c->stream, (const double*)ae.dev, B, is_complex ? 1 : 0,
                       (const double*)ac.dev, (const double*)at.dev, scale, (double*)awr.dev, (double*)awi.dev);
    for (Arg* a : all) if ((rc = unstage(c, *a))) return rc;
    return finish(c);
}
int cg_randn(cg_ctx* c, double* out, size_t count, uint64_t seed, uint64_t offset) {
    if (!c) return CG_ERR_ARG;
    if (count == 0) return CG_OK;
    if (!out) CG_FAIL(c, CG_ERR_ARG, "cg_randn: out is NULL");
    CG_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = arena_reset(c))) CG_FAIL(c, rc, "cg_randn: arena");
    Arg ao{out, nullptr, sizeof(double) * count, false, true};
    if ((rc = stage(c, ao))) return rc;
    hipLaunchKernelGGL(k_randn, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, (double*)ao.dev, count, seed, offset);
    if ((rc = unstage(c, ao))) return rc;
    return finish(c);
}
/* y = a x + b y on DEVICE pointers (both pointer modes): the accumulators of main.py:281-289 kept in HBM */
int cg_axpby(cg_ctx* c, double a, const double* x_dev, double b, double* y_dev, size_t count) {
    if (!c) return CG_ERR_ARG;
    if (count == 0) return CG_OK;
    if (!x_dev || !y_dev) CG_FAIL(c, CG_ERR_ARG, "cg_axpby: NULL argument");
    CG_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, a, x_dev, b, y_dev, count);
    CG_HIP(c, hipGetLastError());
    return CG_OK;
}

int cg_scale_dev(cg_ctx* c, double* buf, size_t count, double s) {
    if (!c) return CG_ERR_ARG;
    hipLaunchKernelGGL(k_scale, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, buf, count, s);
    CG_HIP(c, hipGetLastError());
    return CG_OK;
}

}  // extern "C"

#include "cg_comm.inc"
#include "cg_solve.inc"
