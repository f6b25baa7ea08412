/*
 * oracle/cg_oracle.c -- TEST / BENCH INFRASTRUCTURE ONLY (never linked or loaded by coulombgas_amd/).
 *
 * Plain-C CPU restatement of the sampling path of fermiflow/CoulombGas @ v1, written to do the arithmetic
 * the reference's XLA program does, not the structured algorithm of the HIP kernels:
 *   - FermiNet flow, any depth >= 2                                   src/flow.py:16-55
 *   - its Jacobian by DENSE forward mode with all n*d tangents carried through every layer, zeros
 *     included (what jax.jacfwd does)                                 src/logpsi.py:26-28
 *   - real LU log|det| of the n*d x n*d Jacobian                      src/logpsi.py:29
 *   - plane-wave Slater matrix, complex LU, log|det| + i arg          src/slater.py:14-19
 *   - logp = 2 Re log Psi                                             src/logpsi.py:174-181
 *   - Metropolis chain with supplied normal / uniform draws           src/MCMC.py:22-39
 *   - pair-form Ewald sum                                             src/potential.py:36-77
 * OpenMP over walkers, SIMD over the n*d tangents.  Used (a) as bench.py's cpu_baseline ("port": the reference's JAX path cannot run on
 * this image), (b) as a second, AD-free checker for the GPU results at sizes torch.func is too slow for.
 *
 * PARITY PINNING: this file is validated against oracle/cg_ref.py (tests/test_oracle_kat.py), which in turn is
 * pinned by the reference's own analytic KATs and shipped data; see the header of cg_ref.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <complex.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846264338327950288

int cgo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* the container's CPU share may be far below the machine's core count (cgroup quota): the caller sets the team size */
void cgo_set_num_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

typedef struct {
    int n, dim, depth, hs, ht, M;
    double L;
    const double* theta;
    const double* sp;      /* M x dim orbital table (twisted indices) */
    /* parameter offsets, ravel_pytree order (sorted Haiku names, b before w) */
    int fin_b, fin_w;
    int sp_b[16], sp_w[16], sp_in[16];
    int tp_b[16], tp_w[16], tp_in[16];
} Model;

static int name_cmp(const void* a, const void* b) { return strcmp(*(const char* const*)a, *(const char* const*)b); }

/* returns number of parameters */
static int model_init(Model* m, int n, int dim, int depth, int hs, int ht, double L, const double* theta, const double* sp, int M) {
    m->n = n; m->dim = dim; m->depth = depth; m->hs = hs; m->ht = ht; m->L = L; m->theta = theta; m->sp = sp; m->M = M;
    /* module list: sp layers '~/linear', '~/linear_1'.. (depth), tp layers next (depth-1), final 'fermi_net/linear' */
    int nmod = 2 * depth;               /* depth sp + (depth-1) tp + final */
    char names[40][48]; const char* ptr[40]; int kind[40], idx[40];
    int k = 0;
    for (int i = 0; i < depth; ++i, ++k) { if (i == 0) strcpy(names[k], "fermi_net/~/linear"); else { strcpy(names[k], "fermi_net/~/linear_"); char t[8]; int v = i, p = 0; char r[8]; while (v) { r[p++] = '0' + v % 10; v /= 10; } for (int q = 0; q < p; ++q) t[q] = r[p - 1 - q]; t[p] = 0; strcat(names[k], t); } kind[k] = 0; idx[k] = i; }
    for (int i = 0; i < depth - 1; ++i, ++k) { strcpy(names[k], "fermi_net/~/linear_"); char t[8]; int v = depth + i, p = 0; char r[8]; while (v) { r[p++] = '0' + v % 10; v /= 10; } for (int q = 0; q < p; ++q) t[q] = r[p - 1 - q]; t[p] = 0; strcat(names[k], t); kind[k] = 1; idx[k] = i; }
    strcpy(names[k], "fermi_net/linear"); kind[k] = 2; idx[k] = 0; ++k;
    for (int i = 0; i < nmod; ++i) ptr[i] = names[i];
    qsort(ptr, nmod, sizeof(char*), name_cmp);
    int off = 0;
    for (int s = 0; s < nmod; ++s) {
        int j = (int)((ptr[s] - names[0]) / 48);
        int fin, fout;
        if (kind[j] == 0) { fin = idx[j] == 0 ? 4 * dim + 1 : 2 * hs + ht; fout = hs; m->sp_in[idx[j]] = fin; m->sp_b[idx[j]] = off; m->sp_w[idx[j]] = off + fout; }
        else if (kind[j] == 1) { fin = idx[j] == 0 ? 2 * dim + 1 : ht; fout = ht; m->tp_in[idx[j]] = fin; m->tp_b[idx[j]] = off; m->tp_w[idx[j]] = off + fout; }
        else { fin = hs; fout = dim; m->fin_b = off; m->fin_w = off + fout; }
        off += fout + fin * fout;
    }
    return off;
}

int cgo_num_params(int dim, int depth, int hs, int ht) {
    Model m; return model_init(&m, 1, dim, depth, hs, ht, 1.0, NULL, NULL, 0);
}

static inline double softplus(double u) { return fmax(u, 0.0) + log1p(exp(-fabs(u))); }
static inline double sigmoid(double u) { double e = exp(-fabs(u)); return u >= 0 ? 1.0 / (1.0 + e) : e / (1.0 + e); }

/* Flow value z (n*dim) and, if J != NULL, dense forward-mode Jacobian J (N x N row-major, J[out][in]).
 * The tangent loops run over the contiguous q axis so that the compiler vectorises them; target_clones builds AVX-512,
 * AVX2 and baseline versions and the loader picks one for the CPU it runs on (the .so is built in one container and
 * timed on another machine). */
#if defined(__GNUC__) && defined(__x86_64__) && !defined(CGO_NO_CLONES)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
static void flow_forward(const Model* m, const double* x, double* z, double* J) {
    const int n = m->n, d = m->dim, hs = m->hs, ht = m->ht, N = n * d, P = 2 * d + 1;
    const int NT = J ? N : 0;
    const int wmax_s = hs > d ? hs : d, wmax_t = ht > P ? ht : P;
    const int fmax_ = 2 * wmax_s + wmax_t;
    double* sp = calloc((size_t)n * wmax_s, sizeof(double));
    double* dsp = calloc((size_t)n * wmax_s * (NT + 1), sizeof(double));
    double* tp = calloc((size_t)n * n * wmax_t, sizeof(double));
    double* dtp = calloc((size_t)n * n * wmax_t * (NT + 1), sizeof(double));
    double* f = calloc((size_t)n * fmax_, sizeof(double));
    double* df = calloc((size_t)n * fmax_ * (NT + 1), sizeof(double));
    double* u = calloc((size_t)(hs > ht ? hs : ht), sizeof(double));
    double* du = calloc((size_t)(hs > ht ? hs : ht) * (NT + 1), sizeof(double));
    int ws = d, wt = P;     /* current widths of the two streams */
    /* initial streams: src/flow.py:16-26 */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double* t = tp + ((size_t)i * n + j) * wmax_t;
            double* dt = dtp + ((size_t)i * n + j) * wmax_t * NT;
            double nrm2 = 0;
            for (int a = 0; a < d; ++a) {
                double r = x[i * d + a] - x[j * d + a];
                t[a] = cos(2 * PI / m->L * r); t[d + a] = sin(2 * PI / m->L * r);
                double s = sin(PI / m->L * r) + (i == j ? 1.0 : 0.0);
                nrm2 += s * s;
            }
            double nrm = sqrt(nrm2);
            t[2 * d] = nrm * (i == j ? 0.0 : 1.0);
            if (NT)
                for (int a = 0; a < d; ++a) {
                    double r = x[i * d + a] - x[j * d + a];
                    double dc = -2 * PI / m->L * sin(2 * PI / m->L * r), ds = 2 * PI / m->L * cos(2 * PI / m->L * r);
                    double s = sin(PI / m->L * r) + (i == j ? 1.0 : 0.0);
                    double dn = (i == j ? 0.0 : 1.0) * s / nrm * (PI / m->L) * cos(PI / m->L * r);
                    /* d r / d x_{i,a} = +1, d r / d x_{j,a} = -1 */
                    dt[(size_t)a * NT + i * d + a] += dc; dt[(size_t)a * NT + j * d + a] -= dc;
                    dt[(size_t)(d + a) * NT + i * d + a] += ds; dt[(size_t)(d + a) * NT + j * d + a] -= ds;
                    dt[(size_t)(2 * d) * NT + i * d + a] += dn; dt[(size_t)(2 * d) * NT + j * d + a] -= dn;
                }
        }
    for (int layer = 0; layer < m->depth; ++layer) {
        const int last = (layer == m->depth - 1);
        const int fs = 2 * ws + wt;
        /* f = [sp, mean_k sp, mean_j tp]  src/flow.py:28-37 */
        memset(f, 0, sizeof(double) * n * fmax_);
        if (NT) memset(df, 0, sizeof(double) * (size_t)n * fmax_ * NT);
        /* mean_k sp (the same for every i: computed once, broadcast) */
        for (int c = 0; c < ws; ++c) {
            double mean = 0; for (int k = 0; k < n; ++k) mean += sp[k * wmax_s + c];
            f[0 * fs + ws + c] = mean / n;
            if (NT) {
                double* restrict o = df + ((size_t)0 * fs + ws + c) * NT;
                for (int k = 0; k < n; ++k) {
                    const double* restrict src = dsp + ((size_t)k * wmax_s + c) * NT;
                    for (int q = 0; q < NT; ++q) o[q] += src[q];
                }
                for (int q = 0; q < NT; ++q) o[q] /= n;
            }
        }
        for (int i = 0; i < n; ++i) {
            for (int c = 0; c < ws; ++c) {
                f[i * fs + c] = sp[i * wmax_s + c];
                f[i * fs + ws + c] = f[0 * fs + ws + c];
                if (NT) {
                    memcpy(df + ((size_t)i * fs + c) * NT, dsp + ((size_t)i * wmax_s + c) * NT, sizeof(double) * NT);
                    if (i) memcpy(df + ((size_t)i * fs + ws + c) * NT, df + ((size_t)0 * fs + ws + c) * NT, sizeof(double) * NT);
                }
            }
            for (int c = 0; c < wt; ++c) {
                double mean = 0; for (int j = 0; j < n; ++j) mean += tp[((size_t)i * n + j) * wmax_t + c];
                f[i * fs + 2 * ws + c] = mean / n;
                if (NT) {
                    double* restrict o = df + ((size_t)i * fs + 2 * ws + c) * NT;
                    for (int j = 0; j < n; ++j) {
                        const double* restrict src = dtp + (((size_t)i * n + j) * wmax_t + c) * NT;
                        for (int q = 0; q < NT; ++q) o[q] += src[q];
                    }
                    for (int q = 0; q < NT; ++q) o[q] /= n;
                }
            }
        }
        /* one-particle layer */
        const double* W = m->theta + m->sp_w[layer]; const double* bb = m->theta + m->sp_b[layer];
        for (int i = 0; i < n; ++i) {
            for (int h = 0; h < hs; ++h) {
                double a = bb[h];
                for (int c = 0; c < fs; ++c) a += f[i * fs + c] * W[c * hs + h];
                u[h] = a;
            }
            if (NT) {      /* tangents: the q axis innermost and contiguous (SIMD), every tangent carried, zeros included */
                memset(du, 0, sizeof(double) * (size_t)hs * NT);
                for (int c = 0; c < fs; ++c) {
                    const double* restrict dfc = df + ((size_t)i * fs + c) * NT;
                    for (int h = 0; h < hs; ++h) {
                        const double w = W[c * hs + h];
                        double* restrict duh = du + (size_t)h * NT;
                        for (int q = 0; q < NT; ++q) duh[q] += dfc[q] * w;
                    }
                }
            }
            for (int h = 0; h < hs; ++h) {
                double spv = softplus(u[h]), sg = sigmoid(u[h]);
                if (layer == 0) {
                    /* assignment (src/flow.py:45); old width ws=d is dropped */
                    sp[i * wmax_s + h] = spv;
                    for (int q = 0; q < NT; ++q) dsp[((size_t)i * wmax_s + h) * NT + q] = sg * du[(size_t)h * NT + q];
                } else {
                    sp[i * wmax_s + h] += spv;
                    for (int q = 0; q < NT; ++q) dsp[((size_t)i * wmax_s + h) * NT + q] += sg * du[(size_t)h * NT + q];
                }
            }
        }
        /* NOTE: for layer 0 the loop above overwrote sp[i][h] for h < d while later i still needed the OLD sp via f: f was
           built before the loop, so this is safe. */
        if (!last) {
            const double* Wt = m->theta + m->tp_w[layer]; const double* bt = m->theta + m->tp_b[layer];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double* t = tp + ((size_t)i * n + j) * wmax_t;
                    double* dt = dtp + ((size_t)i * n + j) * wmax_t * NT;
                    for (int h = 0; h < ht; ++h) {
                        double a = bt[h];
                        for (int c = 0; c < wt; ++c) a += t[c] * Wt[c * ht + h];
                        u[h] = a;
                    }
                    if (NT) {
                        memset(du, 0, sizeof(double) * (size_t)ht * NT);
                        for (int c = 0; c < wt; ++c) {
                            const double* restrict dtc = dt + (size_t)c * NT;
                            for (int h = 0; h < ht; ++h) {
                                const double w = Wt[c * ht + h];
                                double* restrict duh = du + (size_t)h * NT;
                                for (int q = 0; q < NT; ++q) duh[q] += dtc[q] * w;
                            }
                        }
                    }
                    for (int h = 0; h < ht; ++h) {
                        double spv = softplus(u[h]), sg = sigmoid(u[h]);
                        if (layer == 0) { t[h] = spv; for (int q = 0; q < NT; ++q) dt[(size_t)h * NT + q] = sg * du[(size_t)h * NT + q]; }
                        else { t[h] += spv; for (int q = 0; q < NT; ++q) dt[(size_t)h * NT + q] += sg * du[(size_t)h * NT + q]; }
                    }
                }
            wt = ht;
        }
        ws = hs;
    }
    /* z = x + final(sp)  src/flow.py:53-55 */
    const double* Wf = m->theta + m->fin_w; const double* bf = m->theta + m->fin_b;
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < d; ++a) {
            double v = x[i * d + a] + bf[a];
            for (int h = 0; h < hs; ++h) v += sp[i * wmax_s + h] * Wf[h * d + a];
            z[i * d + a] = v;
            if (J)
                for (int q = 0; q < N; ++q) {
                    double dv = (q == i * d + a) ? 1.0 : 0.0;
                    for (int h = 0; h < hs; ++h) dv += dsp[((size_t)i * wmax_s + h) * NT + q] * Wf[h * d + a];
                    J[(size_t)(i * d + a) * N + q] = dv;
                }
        }
    free(sp); free(dsp); free(tp); free(dtp); free(f); free(df); free(u); free(du);
}

/* getrf-style partial-pivot LU log|det| (real) */
static double lu_logabsdet(double* A, int N) {
    double s = 0;
    for (int k = 0; k < N; ++k) {
        int p = k; double best = fabs(A[k * N + k]);
        for (int i = k + 1; i < N; ++i) if (fabs(A[i * N + k]) > best) { best = fabs(A[i * N + k]); p = i; }
        if (p != k) for (int j = 0; j < N; ++j) { double t = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = t; }
        double piv = A[k * N + k];
        s += log(fabs(piv));
        for (int i = k + 1; i < N; ++i) {
            double l = A[i * N + k] / piv;
            for (int j = k + 1; j < N; ++j) A[i * N + j] -= l * A[k * N + j];
        }
    }
    return s;
}

static double complex lu_logdet_c(double complex* A, int N) {
    double la = 0; double complex phase = 1.0;
    for (int k = 0; k < N; ++k) {
        int p = k; double best = cabs(A[k * N + k]);
        for (int i = k + 1; i < N; ++i) if (cabs(A[i * N + k]) > best) { best = cabs(A[i * N + k]); p = i; }
        if (p != k) { for (int j = 0; j < N; ++j) { double complex t = A[k * N + j]; A[k * N + j] = A[p * N + j]; A[p * N + j] = t; } phase = -phase; }
        double complex piv = A[k * N + k];
        la += log(cabs(piv)); phase *= piv / cabs(piv);
        for (int i = k + 1; i < N; ++i) {
            double complex l = A[i * N + k] / piv;
            for (int j = k + 1; j < N; ++j) A[i * N + j] -= l * A[k * N + j];
        }
    }
    return la + I * carg(phase);
}

/* [Re log phi, Im log phi, 1/2 log|det J|] of one walker */
static void logpsi_one(const Model* m, const double* x, const int* sidx, double* out3) {
    const int n = m->n, d = m->dim, N = n * d;
    double* z = malloc(sizeof(double) * N);
    double* J = malloc(sizeof(double) * N * N);
    flow_forward(m, x, z, J);
    double complex* D = malloc(sizeof(double complex) * n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double ph = 0;
            for (int a = 0; a < d; ++a) ph += 2 * PI / m->L * m->sp[(size_t)sidx[j] * d + a] * z[i * d + a];
            D[i * n + j] = 1.0 / pow(m->L, d / 2.0) * cexp(I * ph);
        }
    double complex lp = lu_logdet_c(D, n);
    out3[0] = creal(lp); out3[1] = cimag(lp); out3[2] = 0.5 * lu_logabsdet(J, N);
    free(z); free(J); free(D);
}

void cgo_flow(int n, int dim, int depth, int hs, int ht, double L, const double* theta, const double* x, int B, double* z, double* J) {
    Model m; model_init(&m, n, dim, depth, hs, ht, L, theta, NULL, 0);
    const int N = n * dim;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) flow_forward(&m, x + (size_t)b * N, z + (size_t)b * N, J ? J + (size_t)b * N * N : NULL);
}

/* out (B,3): Re log phi, Im log phi, 1/2 log|det J| */
void cgo_logpsi(int n, int dim, int depth, int hs, int ht, double L, const double* theta, const double* sp, int M,
                const int* sidx, const double* x, int B, double* out) {
    Model m; model_init(&m, n, dim, depth, hs, ht, L, theta, sp, M);
    const int N = n * dim;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) logpsi_one(&m, x + (size_t)b * N, sidx + (size_t)b * n, out + 3 * b);
}

/* src/MCMC.py:22-39 with supplied draws; x in/out; returns accepted / (steps*B) */
double cgo_mcmc(int n, int dim, int depth, int hs, int ht, double L, const double* theta, const double* sp, int M,
                const int* sidx, double* x, int B, int steps, double stddev, const double* noise, const double* unif,
                double* logp_out) {
    Model m; model_init(&m, n, dim, depth, hs, ht, L, theta, sp, M);
    const int N = n * dim;
    long total = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : total)
    for (int b = 0; b < B; ++b) {
        double* xc = x + (size_t)b * N;
        double* xp = malloc(sizeof(double) * N);
        double o[3];
        logpsi_one(&m, xc, sidx + (size_t)b * n, o);
        double logp = 2 * (o[0] + o[2]);
        for (int s = 0; s < steps; ++s) {
            for (int e = 0; e < N; ++e) xp[e] = xc[e] + stddev * noise[((size_t)s * B + b) * N + e];
            logpsi_one(&m, xp, sidx + (size_t)b * n, o);
            double lp = 2 * (o[0] + o[2]);
            double ratio = exp(lp - logp);
            if (unif[(size_t)s * B + b] < ratio) { memcpy(xc, xp, sizeof(double) * N); logp = lp; ++total; }
        }
        if (logp_out) logp_out[b] = logp;
        free(xp);
    }
    return (steps > 0 && B > 0) ? (double)total / ((double)steps * B) : 0.0;
}

/* src/potential.py:36-77 (pair form, as the reference evaluates it); V (B) without the Madelung term */
void cgo_ewald(int n, int dim, double L, double kappa, double rs, const long* G, int nG, const double* x, int B, double* V) {
    double* gk = malloc(sizeof(double) * nG);
    for (int g = 0; g < nG; ++g) {
        double g2 = 0; for (int a = 0; a < dim; ++a) g2 += (double)G[g * dim + a] * (double)G[g * dim + a];
        double gn = sqrt(g2);
        gk[g] = dim == 3 ? exp(-PI * PI * g2 / (kappa * kappa)) / (PI * g2) : erfc(PI * gn / kappa) / gn;
    }
    const double g0 = dim == 3 ? -PI / (kappa * kappa) : -2 * sqrt(PI) / kappa;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        const double* xb = x + (size_t)b * n * dim;
        double vs = 0, vl = 0; int npair = 0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                double r[3], d2 = 0;
                for (int a = 0; a < dim; ++a) { r[a] = (xb[i * dim + a] - xb[j * dim + a]) / L; r[a] -= rint(r[a]); d2 += r[a] * r[a]; }
                double dd = sqrt(d2);
                vs += erfc(kappa * dd) / dd;
                for (int g = 0; g < nG; ++g) {
                    double ph = 0; for (int a = 0; a < dim; ++a) ph += (double)G[g * dim + a] * r[a];
                    vl += gk[g] * cos(2 * PI * ph);
                }
                ++npair;
            }
        V[b] = 2 * rs / L * (vs + vl + g0 * npair);
    }
    free(gk);
}
