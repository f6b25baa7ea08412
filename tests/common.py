"""Shared helpers for the parity tests: seeded synthetic inputs."""
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def orbitals(dim=2, Emax=25):
    if dim == 2:
        return np.load(os.path.join(GOLDEN, "orbitals_dim2_Emax%d.npy" % Emax))
    return np.load(os.path.join(GOLDEN, "orbitals_dim3_Emax60.npy"))


def box_length(n, dim):
    """main.py:63-69"""
    return float((4 / 3 * np.pi * n) ** (1 / 3)) if dim == 3 else float(np.sqrt(np.pi * n))


def flow_theta(rng, depth, spsize, tpsize, dim, w_std=0.01, b_std=0.0):
    """ravel_pytree-ordered parameter vector; weights N(0,w_std^2), biases N(0,b_std^2)
    (the reference initialises with w_std = 0.01, b = 0: src/flow.py:6-14)."""
    from coulombgas_amd.flow import ravel_order
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
