// cg_k_van.hip -- the Transformer density matrix's per-sample gradient with the positions in parallel (cg_van_par.hpp): kernels + launch.
// The sampler / log-probability kernels and the sequential reverse pass (every other architecture) stay in cg_hip.hip.
#include "cg_host.hpp"
#include "cg_van_par.hpp"

// one wave per sample, at most four waves per workgroup (one per SIMD: the per-lane vectors want the whole register file)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_van_grad_par(CgVanModel m, const double* __restrict__ P, const double* __restrict__ sp, const double* __restrict__ tab, int B,
               const int* __restrict__ sidx, double* __restrict__ S, double* __restrict__ stash_all, int wave_doubles, size_t stash_doubles) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6;
    double* lw = cg_dyn_lds + CG_TAB_DOUBLES + (size_t)wave * wave_doubles;
    double* st = stash_all + (size_t)(blockIdx.x * waves + wave) * stash_doubles;
    for (int s = blockIdx.x * waves + wave; s < B; s += gridDim.x * waves)
        cg_van_grad_par(m, P, sp, sidx + (size_t)s * m.n, lw, st, S + (size_t)s * m.total);
#endif
}

// short sequences: several samples per wave (cg_van_grad_packed)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_van_grad_packed(CgVanModel m, const double* __restrict__ P, const double* __restrict__ sp, const double* __restrict__ tab, int B,
                  const int* __restrict__ sidx, double* __restrict__ S, double* __restrict__ stash_all, int wave_doubles, size_t stash_doubles) {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int e = threadIdx.x; e < CG_TAB_DOUBLES; e += blockDim.x) cg_dyn_lds[e] = tab[e];
    __syncthreads();
    const int waves = blockDim.x >> 6, wave = threadIdx.x >> 6, per = CgVanPar::packed_samples(m.n);
    double* lw = cg_dyn_lds + CG_TAB_DOUBLES + (size_t)wave * wave_doubles;
    double* st = stash_all + (size_t)(blockIdx.x * waves + wave) * stash_doubles;
    for (int w = blockIdx.x * waves + wave; w * per < B; w += gridDim.x * waves) {
        const int s0 = w * per, sv = B - s0 < per ? B - s0 : per;
        double* stw = st;
        asm volatile("" : "+v"(stw));          // (opaque per pass: otherwise the ~200 stash-row addresses are hoisted out of this loop and spilled)
        cg_van_grad_packed(m, P, sp, sidx + (size_t)s0 * m.n, sv, lw, stw, S + (size_t)s0 * m.total);
    }
#endif
}

// scores of B samples (device pointers) into S (B x total): 1 launched, 0 this model is not served, < 0 error
int cg_van_grad_par_launch(cg_ctx* c, const int* sidx_dev, int B, double* S) {
    const CgVanModel& m = c->van;
    // (short sequences with ONE sample per wave leave most lanes without a position -- n = 13, B = 8192: 3.9 ms against 3.7 ms sequential --
    // so up to n = 33 several samples share a wave: 1.36 ms at n = 13, 1.35 ms against 1.40 ms at n = 29, B = 2048)
    if (!CgVanPar::serves(m)) return 0;
    if (m.n <= cg_env_int("CG_VAN_PACKED_MAXN", 33) && cg_env_int("CG_VAN_PACKED", 1) != 0 && cg_env_int("CG_VAN_PAR", 1) != 0) {
        // short sequences: 64 / (n - 1) samples per wave
        const int per = CgVanPar::packed_samples(m.n);
        const size_t wbp = sizeof(double) * (size_t)CgVanPar::packed_wave_doubles(m.n), tbp = sizeof(double) * CG_TAB_DOUBLES;
        int wv = cg_env_int("CG_VAN_PAR_WAVES", 0);
        if (wv <= 0) { const int per_cu = (int)std::min<size_t>(4, (160 * 1024 - 2 * tbp) / wbp); wv = per_cu >= 4 ? 4 : per_cu >= 2 ? 2 : per_cu; }
        if (per >= 2 && wv >= 1 && tbp + wv * wbp <= 160 * 1024) {
            int rc;
            const int nw = (B + per - 1) / per;
            const int grid = std::min((nw + wv - 1) / wv, c->cu_count * 4);
            const size_t sd = CgVanPar::stash_doubles(m.n, m.M);
            if ((rc = ensure_ws(c, sizeof(double) * sd * (size_t)grid * wv))) return rc;
            if ((rc = set_lds(c, k_van_grad_packed, tbp + wv * wbp))) return rc;
            hipLaunchKernelGGL(k_van_grad_packed, dim3(grid), dim3(64 * wv), tbp + wv * wbp, c->stream, m, (const double*)c->d_van, (const double*)c->d_van_sp,
                               (const double*)c->d_tab, B, sidx_dev, S, (double*)c->ws, CgVanPar::packed_wave_doubles(m.n), sd);
            return 1;
        }
    }
    if (cg_env_int("CG_VAN_PAR", m.n >= 20 ? 1 : 0) == 0) return 0;
    const size_t wb = sizeof(double) * (size_t)CgVanPar::wave_doubles(m.n), tb = sizeof(double) * CG_TAB_DOUBLES;
    int waves = cg_env_int("CG_VAN_PAR_WAVES", 0);
    if (waves <= 0) {                       // as many waves per CU as LDS allows (<= 4: one per SIMD), in workgroups that tile the CU
        const int per_cu = (int)std::min<size_t>(4, (160 * 1024 - 2 * tb) / wb);
        if (per_cu < 1) return 0;
        waves = per_cu >= 4 ? 4 : per_cu >= 2 ? 2 : 1;
    }
    const size_t lds = tb + waves * wb;
    if (lds > 160 * 1024) return 0;
    int rc;
    const int grid = std::min((B + waves - 1) / waves, c->cu_count * 4);
    const size_t sd = CgVanPar::stash_doubles(m.n, m.M);
    if ((rc = ensure_ws(c, sizeof(double) * sd * (size_t)grid * waves))) return rc;
    if ((rc = set_lds(c, k_van_grad_par, lds))) return rc;
    hipLaunchKernelGGL(k_van_grad_par, dim3(grid), dim3(64 * waves), lds, c->stream, m, (const double*)c->d_van, (const double*)c->d_van_sp,
                       (const double*)c->d_tab, B, sidx_dev, S, (double*)c->ws, CgVanPar::wave_doubles(m.n), sd);
    return 1;
}
