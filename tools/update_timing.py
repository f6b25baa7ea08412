#!/usr/bin/env python3
"""Diagnostic: wall time of the pieces of one optimisation step (main.py:335-346) at BASELINE config 3 sizes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine
import coulombgas_amd as cg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
Emax = {13: 25, 29: 25, 49: 36, 57: 49}[n]
L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
eng.set_ewald(10, cg.kpoints(2, 15), 10.0)
v = np.random.default_rng(0).standard_normal(x.shape)
def t(name, fn, reps=2):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    print("%-28s %.1f ms" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True)
t("mcmc 50 steps (host ptrs)", lambda: eng.mcmc(x, sidx, 50, 0.1, seed=1))
t("ewald", lambda: eng.ewald(x))
t("grad_lap hutchinson-split", lambda: eng.grad_laplacian(x, sidx, 2, v))
t("grad_lap exact", lambda: eng.grad_laplacian(x[:B // 4], sidx[:B // 4], 0), reps=1)
w = np.ones(B)
t("param_vjp", lambda: eng.param_vjp(x, sidx, w, 0.5 * w))
t("quantum_fisher (scores+SYRK)", lambda: eng.quantum_fisher(x, sidx), reps=1)
