"""Synthetic inputs of the hot path: what bench.py, the smoke test, the tools and the parity tests feed the kernels.

The single-particle orbital tables are DATA shipped with the package (coulombgas_amd/data/orbitals_*.npy): the twisted,
sorted and reversed `sp_indices_twist` of main.py:79-90 for Emax = 25 / 36 / 49 (twist 1/4, 1/4) and the untwisted 3-D table
of the reference's tests.  Their row order is part of the shipped models (it comes from an unstable sort over
degenerate levels and is not re-derivable); tests/golden/make_reference_data_fixtures.py is the script that produced them."""
import os
import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def orbitals(dim=2, Emax=25):
    if dim == 2:
        return np.load(os.path.join(DATA, "orbitals_dim2_Emax%d.npy" % Emax))
    return np.load(os.path.join(DATA, "orbitals_dim3_Emax60.npy"))


def box_length(n, dim):
    """main.py:63-69"""
    return float((4 / 3 * np.pi * n) ** (1 / 3)) if dim == 3 else float(np.sqrt(np.pi * n))


def flow_theta(rng, depth, spsize, tpsize, dim, w_std=0.01, b_std=0.0):
    """ravel_pytree-ordered parameter vector; weights N(0,w_std^2), biases N(0,b_std^2)
    (the reference initialises with w_std = 0.01, b = 0: src/flow.py:6-14)."""
    from .flow import ravel_order
    th = []
    for _, leaf, shp in ravel_order(depth, spsize, tpsize, dim):
        std = b_std if leaf == "b" else w_std
        th.append(std * rng.standard_normal(int(np.prod(shp))))
    return np.concatenate(th)


def state_indices(rng, B, n, M, excitations=3):
    """ground state of the reversed table (last n rows) with 0..excitations random
    single excitations; strictly increasing int32."""
    out = np.empty((B, n), dtype=np.int32)
    for b in range(B):
        occ = list(range(M - n, M))
        for _ in range(rng.integers(0, excitations + 1)):
            free = [i for i in range(M) if i not in occ]
            occ[rng.integers(0, n)] = free[rng.integers(0, len(free))]
        out[b] = np.sort(np.array(occ))
    return out


def walkers(rng, B, n, dim, L):
    return rng.uniform(0.0, L, (B, n, dim))       # main.py:236


def bench_inputs(n, dim, B, Emax, rank):
    """box, orbital table, init-like flow parameters N(0, 0.01^2) (PCG64(1)), state indices (PCG64(rank)) and
    uniform walkers (PCG64(1000 + rank))."""
    L = box_length(n, dim)
    sp = orbitals(dim, Emax)
    from .flow import ravel_order
    rng_p = np.random.default_rng(np.random.PCG64(1))
    theta = np.concatenate([(np.zeros(int(np.prod(s))) if leaf == "b" else 0.01 * rng_p.standard_normal(int(np.prod(s))))
                            for _, leaf, s in ravel_order(2, 16, 16, dim)])
    sidx = state_indices(np.random.default_rng(np.random.PCG64(rank)), B, n, sp.shape[0])
    x = np.random.default_rng(np.random.PCG64(1000 + rank)).uniform(0.0, L, (B, n, dim))
    return L, sp, theta, sidx, x
