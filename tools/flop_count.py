#!/usr/bin/env python3
"""Structured fp64 flop counts per walker of the derivative kernels, in the manner of SURVEY.md 8(d) for the sampler: the arithmetic
the ALGORITHM needs as implemented in csrc/cg_lap.hpp (k_grad_lap2) and csrc/cg_score.hpp (k_scores), loop nest by loop nest
(FMA = 2 flops; index arithmetic, loads / stores and the transcendental evaluations themselves -- exp, log, reciprocal, sqrt,
sincos -- are not counted: `transcendentals()` lists those separately).  h = spsize = tpsize, p = 2d + 1 pair features,
N = n d, Pp = n (n - 1) ordered pairs.  bench.py divides these by the HIP-event kernel time for the `update_path.roofline` extras.

   python tools/flop_count.py            # table for n = 13, 29, 49, 57
"""


def primal_flops(n, d=2, h=16):
    """pair features, two-particle layer, means, two one-particle layers, final layer"""
    p, Pp = 2 * d + 1, n * (n - 1)
    return 5 * d * Pp + (Pp + 1) * (2 * p * h + h) + n * n * (h + p) + n * (2 * p * h + h) + n * (4 * h * h + h) + 2 * h * h + n * (2 * h * d + d)


def jacobian_flops(n, d=2, h=16):
    """structured block Jacobian (per-particle factors, G pass, pair blocks, diagonal blocks)"""
    return n * n * (4 * h * h * d + 2 * h * d * d + 20 * h * d) + 2 * n * h * h * d


def setup_flops(n, d=2, h=16, full_T=True):
    """z, J, Slater matrix, both Gauss-Jordan inverses (2 N^3 real, 8 n^3 complex), g = diag T^a (and T^a, diag K^ab when full_T)"""
    N = n * d
    w = primal_flops(n, d, h) + jacobian_flops(n, d, h) + n * n * (2 * d + 2) + 2 * N ** 3 + 8 * n ** 3
    w += (10 * d * n ** 3 + 10 * d * d * n * n) if full_T else 10 * N * n
    return w


def slater_part_flops(n, d=2):
    """grad += J^T g (complex x real), C = J J^T (symmetric half), tr(J^T H J) contractions with diag K^ab and T^a T^b"""
    N = n * d
    return 4 * N * N + N ** 3 + 4 * n * d * d + 10 * n * n * d * d


def reverse_x_flops(n, d=2, h=16):
    """CgLap::reverse_x: Jhat, the adjoints that are sums over k (U'bar, Bbar, Gbar, Vbar with its sigmoid pass), the dense chain
    (sg1bar ... m0bar), the pair pass down to rbar_ik, xbar"""
    p, N, Pp = 2 * d + 1, n * d, n * (n - 1)
    w = 2 * N * N + 2 * N * p * n + 4 * N * N * h
    w += n * h * (n - 1) * (2 * p + 10 * d + 2 * d * d)                       # Vbar pair pass
    w += n * h * (n - 1) * 8 * d + 3 * n * h * d * p                          # sg1bar
    w += N * h * (2 * p + 2) + 6 * h * h * N                                  # Ubar, Rbar
    w += n * h * (2 * d + 3) + h * n + 2 * h * h + n * h * (2 * h + 5) + 2 * n * h * h + 2 * n * p * h   # u2bar ... m0bar
    w += Pp * ((3 * d + 1) + p + 6 * d * d + 9 * d * h + h * (2 * p + 18 * d + 2 * d * d + 7) + 18 * d)  # pair pass
    return w + 2 * N * n


def forward_laplacian_flops(n, d=2, h=16):
    """CgLap::forward_laplacian: (value, |grad|^2, lap) carried through the hidden units; |grad u2|^2 from the dense x-gradient
    E_ik (one (n x h)(h x h) product per particle and direction on the matrix cores)"""
    p, N = 2 * d + 1, n * d
    w = n * h * (n - 1) * (2 * p + 12 * d + 9) + 2 * n * p * (n - 1) + 8 * d * n * h * (n - 1)
    w += 2 * n * h * h * p + 2 * N * h * h + n * h * (2 * p + 4) + h * n
    w += n * n * h * (2 * p + 16 * d + 2 * h * d)                             # E_ik: C operand, A operand, product, norms
    return w + n * h * (6 * h + 5) + N * (2 * h + 4)


def jet_pass_flops(n, d=2, h=16):
    """one second-order directional jet (value, d, dd) through the flow and the Jacobian assembly + the two traces.
    jet x jet product = 10 flops, jet x double = 3, jet + jet = 3; activations: 12 per hidden unit (two chain rules)."""
    p, N, Pp = 2 * d + 1, n * d, n * (n - 1)
    JJ, JC, JA, ACT = 10, 3, 3, 12
    # primal: half-angle products and features per pair and direction (5 products), two-particle layer (weights are doubles), means,
    # one-particle layers, final layer
    w = Pp * d * 5 * JJ + (Pp + 1) * h * (p * (JC + JA) + ACT) + n * n * (h + p) * JA
    w += n * h * (p * (JC + JA) + ACT) + n * h * (3 * h * (JC + JA) + ACT) + 2 * h * h * JA + n * d * h * (JC + JA)
    # Jacobian assembly: factors R W^T (3 products N x h x h), U' (N x p x h, jet x jet), G pass, pair blocks, diagonal blocks
    w += 3 * N * h * h * (JC + JA) + N * h * JJ + N * p * h * (JJ + JA)
    w += n * h * (n - 1) * d * (4 * JJ + 2 * JA)                              # G pass
    w += Pp * (h * (p * (JC + JA) + d * 3 * JC + ACT + d * JJ + d * d * (JJ + JA)) + h * d * d * (JJ + JA) + 3 * d * d * (JJ + JA))
    w += N * n * d * JA
    # traces: t2 = tr(J^-1 J''), M = J^-1 J', t3 = tr(M^2)
    return w + 2 * N * N + 2 * N ** 3 + 2 * N * N


def grad_lap_flops(n, d=2, h=16, mode=2):
    """cg_grad_laplacian: modes 2 (Hutchinson-split, every shipped run) and 1 (one jet pass); mode 0: N basis-direction passes"""
    passes = 1 if mode else n * d
    return (setup_flops(n, d, h) + slater_part_flops(n, d) + reverse_x_flops(n, d, h) + (forward_laplacian_flops(n, d, h) if mode != 1 else 0)
            + passes * jet_pass_flops(n, d, h))


def scores_flops(n, d=2, h=16, nw=4):
    """cg_scores_compute (k_scores): set-up + one reverse sweep for both parts + assembly of the (P, 2) score row"""
    p, N = 2 * d + 1, n * d
    w = setup_flops(n, d, h, full_T=False)
    w += 2 * N * N + 2 * N * p * n + 4 * N * N * h
    w += n * h * (n - 1) * (2 * p + 16 * d + 6 * d * d + 5)                   # (J5) pair pass: Vbar + weight-gradient partials
    w += n * h * (n - 1) * 14 * d + 3 * n * h * d * p                         # (J4)
    w += N * h * (2 * p + 2) + 6 * h * h * N                                  # Ubar, Rbar
    w += n * h * (6 * d + 5) + 2 * h * n + 4 * h * h + n * h * (4 * h + 4 * d + 8) + 4 * n * h * h    # both parts of the chain
    w += n * h * n * (2 * p + 2 * (p + 1)) + 4 * n * h * (p + 1)              # (F4/F5) pair pass, one sigmoid for both parts
    w += h * d * n * 7 + 2 * h * n + p * h * (n * (4 + 3 * d) + nw) + h * h * n * (4 + 3 * d) + h * h * (2 + 3 * n * d) + h * h * n * (4 + 4 * d) + 3 * (p + 1) * h * nw
    return w


def transcendentals(n, d=2, h=16):
    """evaluations per walker (each an exp + reciprocal or log): sampler evaluation, k_grad_lap2 (mode 2), k_scores"""
    Pp = n * (n - 1)
    primal = 3 * h * (Pp + 1 + 2 * n) + d * Pp // 2 + n * n + Pp // 2
    sig_pass = h * Pp
    return {"logp evaluation": primal, "k_grad_lap2 (mode 2)": primal + 6 * sig_pass + 3 * h * 2 * n, "k_scores": primal + 2 * sig_pass}


if __name__ == "__main__":
    print("%4s %14s %14s %14s | %12s %12s %12s %12s %12s" % ("n", "logp eval", "k_grad_lap2", "k_scores", "set-up", "Slater part", "reverse", "fwd Laplace", "jet pass"))
    for n in (13, 29, 49, 57):
        print("%4d %14.4g %14.4g %14.4g | %12.4g %12.4g %12.4g %12.4g %12.4g" % (
            n, primal_flops(n) + jacobian_flops(n) + 2 / 3 * (2 * n) ** 3 + n * n * 6 + 8 / 3 * n ** 3, grad_lap_flops(n), scores_flops(n),
            setup_flops(n), slater_part_flops(n), reverse_x_flops(n), forward_laplacian_flops(n), jet_pass_flops(n)))
