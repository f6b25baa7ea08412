#!/usr/bin/env python3
"""Diagnostic: device blocked Cholesky (cg_cholesky) vs numpy, sizes up to the classical Fisher matrix of the shipped models."""
import sys, os, time, faulthandler
faulthandler.dump_traceback_later(150, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coulombgas_amd.engine import Engine
eng = Engine(5, 2, 2, 16, 16, 4.0, np.zeros((9, 2)))
rng = np.random.default_rng(0)
for P, B in ((53, 37), (300, 500), (1074, 2000), (5826, 1024)):
    A = rng.standard_normal((B, P)); F = eng.fisher_real(A) + 1e-3 * np.eye(P)
    for rep in range(2):
        t0 = time.perf_counter(); L = eng.cholesky(F); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); Lr = np.linalg.cholesky(F); th = time.perf_counter() - t0
    print("P=%d: device %.3f s (incl. copies), numpy %.3f s, max|L - Lref| %.1e, max|LL^T - F| %.1e"
          % (P, dt, th, np.abs(L - Lr).max(), np.abs(L @ L.T - F).max()), flush=True)
