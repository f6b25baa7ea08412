/*
 * coulombgas.h -- C-ABI of libcoulombgas_hip.so: the MI355X (gfx950) implementation of the
 * data-parallel VMC inner loop of fermiflow/CoulombGas.
 *
 * The reference has no FFI layer: its boundary for this path is the set of Python factory
 * functions re-exported by src/__init__.py:1-13 and called from main.py.  Each entry point
 * below names the reference function it replaces (file:line relative to the reference tree);
 * coulombgas_amd/ binds them with ctypes (see INTEGRATION.md for the stub a maintainer of the
 * reference would add).
 *
 * Conventions
 *   - plain C types only; every array is C-contiguous; complex = trailing (re, im) pair.
 *   - all floating point is fp64; state indices are int32.
 *   - return 0 = CG_OK, negative = error; cg_last_error() gives the message.  No exception,
 *     longjmp or abort crosses the ABI.  NaN in -> NaN out.
 *   - pointer mode (cg_set_pointer_mode): CG_PTR_HOST (default) = array arguments are host
 *     pointers, the call stages them through device workspace and returns after the result
 *     is back on the host; CG_PTR_DEVICE = array arguments are device pointers (cg_dev_alloc
 *     or any hipMalloc'ed memory of the ctx's device), the call only enqueues work on the
 *     ctx's stream and returns; cg_sync() waits.
 *   - the caller owns every buffer it passes; the library never frees caller memory.
 *   - a cg_ctx is not thread-safe: one host thread per ctx, or serialise calls.
 *   - one cg_ctx per GPU and per (n, dim, flow architecture, orbital table).
 */
#ifndef COULOMBGAS_H
#define COULOMBGAS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cg_ctx cg_ctx;
typedef struct cg_comm cg_comm;

enum { CG_OK = 0, CG_ERR_ARG = -1, CG_ERR_HIP = -2, CG_ERR_UNSUPPORTED = -3, CG_ERR_STATE = -4, CG_ERR_RCCL = -5 };
enum { CG_PTR_HOST = 0, CG_PTR_DEVICE = 1 };
enum { CG_LAP_EXACT = 0, CG_LAP_HUTCHINSON = 1, CG_LAP_HUTCHINSON_SPLIT = 2 };

/* ---- context ------------------------------------------------------------------------- */

/* Builds the model the closures of main.py:152-164 capture: FermiNet(depth, spsize, tpsize, L)
 * (src/flow.py:5-14), the twisted+sorted+reversed orbital table `sp_indices_twist`
 * (main.py:79-90; M x dim doubles) and the box.  device = HIP device ordinal. */
int cg_create(cg_ctx** out, int device, int n, int dim, int depth, int spsize, int tpsize, double L,
              const double* sp_indices, int M);
void cg_destroy(cg_ctx* ctx);
/* message of the last failing call on ctx (ctx == NULL: last failing cg_create / cg_comm_*). */
const char* cg_last_error(const cg_ctx* ctx);
int cg_set_pointer_mode(cg_ctx* ctx, int mode);
int cg_sync(cg_ctx* ctx);
/* number of flow parameters P = size of jax.flatten_util.ravel_pytree(params_flow) (main.py:159). */
int cg_num_params(const cg_ctx* ctx);
/* theta: HOST pointer, P doubles in ravel_pytree order of the Haiku tree (sorted module names,
 * 'b' before 'w', w row-major (in,out)); replaces passing `params_flow` into every closure. */
int cg_set_flow_params(cg_ctx* ctx, const double* theta);
/* kappa, G (nG x dim integers, as returned by kpoints(), src/potential.py:7-17), rs.
 * HOST pointers.  Replaces the (kappa, G, L, rs) arguments of make_loss (src/VMC.py:31). */
int cg_set_ewald(cg_ctx* ctx, double kappa, const int64_t* G, int nG, double rs);

/* device memory helpers so that host code needs no other GPU runtime */
int cg_dev_alloc(cg_ctx* ctx, size_t bytes, void** dptr);
int cg_dev_free(cg_ctx* ctx, void* dptr);
int cg_memcpy_h2d(cg_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int cg_memcpy_d2h(cg_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int cg_memset(cg_ctx* ctx, void* dst_dev, int value, size_t bytes);
/* HIP-event stopwatch on the ctx's stream (bench.py's kernel timing) */
int cg_timer_start(cg_ctx* ctx);
int cg_timer_stop(cg_ctx* ctx, float* ms);   /* records, synchronises, returns elapsed ms */
/* tuning knob: threads per walker workgroup (0 = automatic from n) */
int cg_set_block_threads(cg_ctx* ctx, int threads);
/* fills info[0..7]: {threads per workgroup, LDS bytes per workgroup, device CU count, P, fast path (0/1), 0,0,0} */
int cg_get_launch_info(cg_ctx* ctx, int64_t* info);

/* diagnostics: measured fp64 peak of this GPU, which = 0: v_fma_f64 (VALU), 1: v_mfma_f64_16x16x4_f64;
 * result in TFLOP/s.  bench.py uses it as the roofline denominator. */
int cg_microbench_fp64(cg_ctx* ctx, int which, double* tflops);

/* ---- wavefunction ---------------------------------------------------------------------- */

/* z = flow.apply(params, None, x) batched: x, z (B,n,dim).   src/flow.py:39-55 */
int cg_flow_forward(cg_ctx* ctx, const double* x, int B, double* z);
/* Jacobian dz/dx per walker, J (B, n*dim, n*dim) row-major = jax.jacfwd(flow_flatten) of src/logpsi.py:27-28 */
int cg_flow_jacobian(cg_ctx* ctx, const double* x, int B, double* J);
/* out (B,2) = vmap(logpsi)(x, params, state_idx):  [Re log phi + 1/2 log|det J|, Im log phi].  src/logpsi.py:9-31 */
int cg_logpsi(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B, double* out);
/* logphi (B,2), half_logdetJ (B): the two closures of make_logphi_logjacdet.  src/logpsi.py:35-53 */
int cg_logphi_logjacdet(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B, double* logphi, double* half_logdetJ);
/* logp (B) = 2 Re log Psi.  src/logpsi.py:174-181 */
int cg_logp(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B, double* logp);

/* ---- sampler --------------------------------------------------------------------------- */

/* mcmc(logp_fn, x_init, key, mc_steps, mc_stddev) of src/MCMC.py:7-40 for logp = 2 Re logpsi(., params, state_idx).
 *   x         (B,n,dim) in/out (NOT wrapped into the box; src/VMC.py:24 does that: cg_wrap)
 *   noise     nullable (mc_steps,B,n,dim) standard normals; unif nullable (mc_steps,B) uniforms in [0,1):
 *             when both are given they replace jax.random.normal / uniform of src/MCMC.py:26,29 (parity mode);
 *             when NULL an in-kernel Philox4x32-10 stream keyed by (seed, walker_offset + walker, step) is used.
 *   logp_out  nullable (B): final log-probabilities
 *   n_accept  HOST pointer (both pointer modes), nullable: total accepted moves (the float accumulator of
 *             src/MCMC.py:33,37 before the division at :39).  In CG_PTR_DEVICE mode a non-NULL n_accept
 *             forces a stream synchronisation; use cg_mcmc_accepts() to read it later instead. */
int cg_mcmc(cg_ctx* ctx, double* x, const int32_t* state_idx, int B, int mc_steps, double mc_stddev,
            uint64_t seed, uint64_t walker_offset, const double* noise, const double* unif,
            double* logp_out, int64_t* n_accept);
/* accepted-move count of the most recent cg_mcmc on this ctx (synchronises the stream) */
int cg_mcmc_accepts(cg_ctx* ctx, int64_t* n_accept);
/* x -= L * floor(x / L).   src/VMC.py:24 */
int cg_wrap(cg_ctx* ctx, double* x, int B);

/* ---- potential --------------------------------------------------------------------------- */

/* V (B) = potential_energy(x, kappa, G, L, rs) = 2 rs / L * psi(x / L, kappa, G)  (no Madelung term).
 * src/potential.py:36-77 */
int cg_ewald(cg_ctx* ctx, const double* x, int B, double* V);

/* ---- local energy ingredients ------------------------------------------------------------ */

/* grad (B,n,dim,2) complex, lap (B,2) complex of log Psi w.r.t. x:
 *   mode CG_LAP_EXACT            make_logpsi_grad_laplacian(logpsi)            src/logpsi.py:63-106
 *   mode CG_LAP_HUTCHINSON       ...(hutchinson=True)  v^T H v, v (B,n,dim)    src/logpsi.py:112-132
 *   mode CG_LAP_HUTCHINSON_SPLIT ...(hutchinson=True, logphi, logjacdet)       src/logpsi.py:134-164
 * v: nullable for CG_LAP_EXACT; replaces jax.random.normal(key, x.shape) of src/logpsi.py:110. */
int cg_grad_laplacian(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B, int mode,
                      const double* v, double* grad, double* lap);

/* g_theta (P) = sum_b [ w_re[b] * d/dtheta Re log Psi_b + w_im[b] * d/dtheta Im log Psi_b ]:
 * the vector-Jacobian product jax.jacrev(quantum_lossfn) needs (src/VMC.py:69-76, main.py:278).
 * Workspace: per-sample scores of at most 1024 walkers at a time (16 P bytes each) plus the kernels' own slots; the
 * (B, P, 2) score matrix is only materialised by cg_quantum_score / cg_scores_compute. */
int cg_param_vjp(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B,
                 const double* w_re, const double* w_im, double* g_theta);
/* per-sample scores S (B,P,2) complex = make_quantum_score(logpsi)  (src/logpsi.py:183-203) */
int cg_quantum_score(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B, double* score);
/* Quantum Fisher matrix and mean score of the hybrid SR optimizer, fishers_fn of src/sr.py:62-80 for ONE device (before
 * its pmean): fisher (P,P) = Re(S^H S) / B, score_mean (P,2) = mean_b S[b].  The scores stay on the device. */
int cg_quantum_fisher(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B, double* fisher, double* score_mean);
/* The same in pieces, with the (B,P) score matrix resident on the device between the calls: one score computation (two
 * reverse sweeps) then serves both theta-VJPs of jax.jacrev(quantum_lossfn) (main.py:278) and the Fisher matrix.
 * cg_scores_vjp: g_theta (P) = sum_b w_re[b] Re S[b] + w_im[b] Im S[b]  (== cg_param_vjp on the same inputs). */
int cg_scores_compute(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B);
int cg_scores_vjp(cg_ctx* ctx, const double* w_re, const double* w_im, double* g_theta);
int cg_scores_fisher(cg_ctx* ctx, double* fisher, double* score_mean);
/* cg_grad_laplacian and cg_scores_compute on the same walkers in ONE call -- the pair an optimisation step makes (src/VMC.py:35 inside
 * make_loss, then jax.jacrev(quantum_lossfn), main.py:278; XLA compiles the reference's two passes into one program as well).  With device
 * pointers, the (dim 2, 16, 16) flow and a Hutchinson mode one fused kernel runs the common set-up (flow, Jacobian, the two
 * inverses) once; every other case runs the two calls one after the other.  grad, lap and the resident scores are those of the separate
 * calls bit for bit. */
int cg_grad_laplacian_scores(cg_ctx* ctx, const double* x, const int32_t* state_idx, int B, int mode,
                             const double* v, double* grad, double* lap);

/* mean over the batch of the resident scores: score_mean (P,2) = mean_b S[b]  (src/sr.py:70 before its pmean; also
 * 1/2 of d/dtheta of quantum_score = 2 mean Re logPsi, src/VMC.py:75, main.py:278) */
int cg_scores_mean(cg_ctx* ctx, double* score_mean);

/* ---- density matrix: autoregressive Transformer over momentum occupations, on the device ----------------- */

/* Transformer(output_size = M, num_layers, model_size, num_heads, hidden_size) of src/autoregressive.py:50-96 with the sampler
 * and log-probability of src/sampler.py:4-50 for the ctx's n electrons in M orbitals (n <= 64, M <= 256).
 * sp_indices (M x dim): the table make_autoregressive_sampler receives (main.py:107: sp_indices_twist), HOST pointer.
 * params: HOST pointer, cg_van_num_params(...) doubles in the order
 *   x1hat[M]; embedding_mlp b[ms], w[dim][ms]; per layer: query b, w[ms][ms]; key b, w; value b, w; attention linear b, w;
 *   layer_mlp/linear b[hs], w[ms][hs]; layer_mlp/linear_1 b[ms], w[hs][ms]; output_mlp b[M], w[ms][M]   (w row-major (in, out)). */
int cg_van_num_params(int M, int num_layers, int model_size, int num_heads, int hidden_size, int dim);
int cg_van_set_params(cg_ctx* ctx, int M, int num_layers, int model_size, int num_heads, int hidden_size,
                      const double* sp_indices, const double* params);
/* logp (B) = vmap(log_prob)(params, state_idx (B,n))   src/sampler.py:40-44 */
int cg_van_log_prob(cg_ctx* ctx, const int32_t* state_idx, int B, double* logp);
/* state_idx (B,n) = sampler(params, key, B)   src/sampler.py:30-38: n sequential conditionals, jax.random.categorical as
 * Gumbel-max.  unif: nullable (B,n,M) uniforms in (0,1) replacing the random draws (parity mode); NULL: the in-library
 * Philox stream (seed, offset + sample).  logp: nullable (B): log-probabilities of the drawn samples (saves the
 * log_prob(params_van, state_indices) pass of src/VMC.py:34). */
int cg_van_sample(cg_ctx* ctx, int B, uint64_t seed, uint64_t offset, const double* unif, int32_t* state_idx, double* logp);

/* Per-sample classical scores S_b = d log p(state_idx_b) / d params (flat parameter order): jax.vmap(jax.grad(log_prob)) of
 * src/sampler.py:52-65, computed by a hand-written reverse pass on the device and kept resident there (B x count doubles):
 *   cg_van_scores_vjp    g (count) = sum_b w[b] S_b: jax.jacrev(classical_lossfn) of main.py:277 with w = F_clipped / B (and 1 / B)
 *   cg_van_scores_fisher F (count, count) = S^T S / B: the classical Fisher matrix of src/sr.py:77-79 before its pmean; perm (count
 *                        int32, or NULL) reorders it to the caller's parameter order, F[i][j] = <S[perm[i]] S[perm[j]]> (jax's
 *                        ravel_pytree order differs from the flat order above)
 *   cg_van_scores_get    the matrix itself. */
int cg_van_scores_compute(cg_ctx* ctx, const int32_t* state_idx, int B);
int cg_van_scores_vjp(cg_ctx* ctx, const double* w, double* g);
int cg_van_scores_fisher(cg_ctx* ctx, const int32_t* perm, double* fisher);
int cg_van_scores_get(cg_ctx* ctx, double* scores);

/* ---- local energy and loss weights on the device (K8) ----------------------------------------------------- */

/* src/VMC.py:39-58 for ONE device, before the pmean: from grad (B,n,dim,2), lap (B,2) of cg_grad_laplacian and V (B) of
 * cg_ewald:  kinetic = -lap - sum grad^2 (complex square), E_loc = kinetic + V + Vconst, F_loc = logp_states / beta + Re E_loc.
 *   logp_states  nullable (B): log-probabilities of the sampled occupations (NULL = zeros)
 *   eloc (B,2) complex, floc nullable (B)
 *   moments (10) = local means of [K, K^2, V, V^2, E, E^2, F, F^2, -logp_states, logp_states^2] (real parts), the order of
 *   src/VMC.py:46-53; the caller all-reduces them (cg_allreduce_mean). */
int cg_local_energy(cg_ctx* ctx, const double* grad, const double* lap, const double* V, const double* logp_states, int B,
                    double Vconst, double beta, double* eloc, double* floc, double* moments);
/* out (1) = mean_b |e_b - center[0]|: the clip width tv before its pmean (src/VMC.py:63 real F_loc: is_complex 0, e (B);
 * src/VMC.py:72 complex E_loc: is_complex 1, e (B,2)).  center is read from memory when the kernel runs (device-pointer
 * mode: the all-reduced mean never visits the host). */
int cg_abs_dev(cg_ctx* ctx, const double* e, int B, int is_complex, const double* center, double* out);
/* Loss weights behind jax.jacrev of the loss closures (src/VMC.py:64-66, 73-75; main.py:277-278):
 *   clipped = clip(e, center - 5 tv, center + 5 tv)   (complex: lexicographic order of the reference's JAX generation)
 *   w_re = scale * Re clipped, w_im = scale * Im clipped (complex only); quantum loss: scale = 2 / B, then
 *   cg_scores_vjp / cg_param_vjp with (w_re, w_im) give d gradF_theta / d theta. */
int cg_clip_weights(cg_ctx* ctx, const double* e, int B, int is_complex, const double* center, const double* tv, double scale,
                    double* w_re, double* w_im);
/* count standard normals from the in-library Philox4x32-10 stream (seed, offset + i): the Hutchinson probe
 * jax.random.normal(key, x.shape) of src/logpsi.py:110 without a host round trip. */
int cg_randn(cg_ctx* ctx, double* out, size_t count, uint64_t seed, uint64_t offset);
/* y = a x + b y for DEVICE pointers in both pointer modes: the accumulators of main.py:281-289 kept in HBM. */
int cg_axpby(cg_ctx* ctx, double a, const double* x_dev, double b, double* y_dev, size_t count);
/* buf *= s for a DEVICE pointer (both pointer modes) */
int cg_scale_dev(cg_ctx* ctx, double* buf_dev, size_t count, double s);

/* Classical Fisher matrix of the SR optimizer (src/sr.py:36, 74) on the device: F (P,P) = S^T S / B for a real (B,P) score
 * matrix, f64 MFMA. */
int cg_fisher_real(cg_ctx* ctx, const double* S, int B, int P, double* F);
/* In-place blocked Cholesky (f64 MFMA trailing updates) for the damped SR solve (src/sr.py:38-45, 102-117): on return the
 * lower triangle of A (P,P) holds L, A = L L^T; the strict upper triangle is untouched.  CG_ERR_STATE if not positive definite. */
int cg_cholesky(cg_ctx* ctx, double* A, int P);
/* The whole damped solve of src/sr.py:38-41 / 88, 102-112 on the device: x = (A - Re(conj(m) m^T) + damping I)^-1 b for a
 * symmetric A (P,P) whose shifted form is positive definite: optional centring with m = center_re + i center_im (both NULL:
 * none), diagonal shift, blocked Cholesky, forward and transposed substitution in 64-row blocks.  Host-pointer mode leaves
 * the caller's A intact; in device-pointer mode A is overwritten by its factor.  b and x may be the same array. */
int cg_spd_solve(cg_ctx* ctx, double* A, int P, double damping, const double* center_re, const double* center_im,
                 const double* b, double* x);

/* ---- multi-GPU (one process per GPU) ------------------------------------------------------ */

/* RCCL communicator over the ranks of one node; replaces jax.lax.pmean(axis_name="p")
 * (src/MCMC.py:39, src/VMC.py:46-53,63,72, main.py:280).  unique_id: 128 bytes. */
int cg_comm_unique_id(void* unique_id_128);
int cg_comm_create(cg_comm** out, cg_ctx* ctx, int rank, int world, const void* unique_id_128);
void cg_comm_destroy(cg_comm* comm);
/* in-place mean over ranks of `count` doubles (DEVICE pointer on the ctx's device) */
int cg_allreduce_mean(cg_comm* comm, double* buf_dev, size_t count);
/* in-place SUM over ranks (no 1/world): what the gathers of the checkpoint path are built on -- a rank's walkers in its own slot of
 * a zero-filled buffer come back bit for bit at any world size (main.py:374-381 collects x over the device axis) */
int cg_allreduce_sum(cg_comm* comm, double* buf_dev, size_t count);
/* acceptance rate of the most recent cg_mcmc on ctx: accepted moves / denom (= mc_steps x batch), formed on the device and, with a
 * communicator, averaged over its ranks there -- the lax.pmean of src/MCMC.py:39 without staging the operand through the host.
 * comm may be NULL (this rank alone).  rate: HOST pointer.  Synchronises the stream. */
int cg_mcmc_accept_rate(cg_ctx* ctx, cg_comm* comm, double denom, double* rate);

#ifdef __cplusplus
}
#endif
#endif
