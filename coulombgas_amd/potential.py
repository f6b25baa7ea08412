"""src/potential.py: kpoints / Madelung are host-side one-off set-up (numpy); potential_energy runs
the Ewald kernel (cg_ewald)."""
import math
import numpy as np


def kpoints(dim, Gmax):
    """src/potential.py:7-17"""
    n = np.arange(-Gmax, Gmax + 1)
    nis = np.meshgrid(*([n] * dim))
    G = np.array([ni.flatten() for ni in nis]).T
    G2 = (G ** 2).sum(axis=-1)
    return G[(G2 <= Gmax ** 2) * (G2 > 0)]


def Madelung(dim, kappa, G):
    """src/potential.py:19-34"""
    Gnorm = np.linalg.norm(np.asarray(G, dtype=np.float64), axis=-1)
    if dim == 3:
        g_k = np.exp(-math.pi ** 2 * Gnorm ** 2 / kappa ** 2) / (math.pi * Gnorm ** 2)
        g_0 = -math.pi / kappa ** 2
    elif dim == 2:
        g_k = np.array([math.erfc(math.pi * g / kappa) / g for g in Gnorm])
        g_0 = -2 * math.sqrt(math.pi) / kappa
    else:
        raise ValueError("dim must be 2 or 3")
    return float(g_k.sum() + g_0 - 2 * kappa / math.sqrt(math.pi))


def potential_energy(x, kappa, G, L, rs, engine=None):
    """src/potential.py:69-77 (vmapped over the batch): x (B,n,dim) -> (B,).  No Madelung term."""
    from .flow import get_engine
    x = np.asarray(x, dtype=np.float64)
    n, dim = x.shape[-2:]
    eng = engine if engine is not None else get_engine(n, dim, 2, 16, 16, L)
    eng.set_ewald(kappa, G, rs)
    return eng.ewald(x)
