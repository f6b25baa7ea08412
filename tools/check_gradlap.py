#!/usr/bin/env python3
"""Diagnostic: GPU grad/Laplacian vs the host build of the same device code, per direction."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_parity import _setup, DCASES
from tests.emul_engine import EmulEngine
np.set_printoptions(linewidth=200, precision=2)
for case in DCASES[1:2]:
    n, dim, hs, ht = case[:4]
    for B in (2,):
        s = _setup(case, B, seed=11)
        v = s["rng"].standard_normal(s["x"].shape)
        eng = s["flow"].engine(n, dim, s["sp"]); eng.set_params(s["theta"])
        em = EmulEngine(n, dim, 2, hs, ht, s["L"], s["sp"]); em.set_params(s["theta"])
        for mode in (0, 1, 2):
            for rep in range(1):
                g, l = eng.grad_laplacian(s["x"], s["sidx"], mode, v); ge, le = em.grad_laplacian(s["x"], s["sidx"], mode, v)
                err = np.abs(g - ge).reshape(B, -1)
                print(case[:4], "B", B, "mode", mode, "rep", rep, "lap err %.1e" % np.abs(l - le).max(), "grad err per dir (walker 0):", err[0], flush=True)
