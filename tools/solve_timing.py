#!/usr/bin/env python3
"""Diagnostic: the damped SPD solve on a device-resident matrix (cg_spd_solve: shift, blocked Cholesky, both substitutions) at the
sizes of the shipped models' Fisher matrices (flow P = 1074, Transformer P = 5907), HIP-event time and the residual against numpy.
   python tools/solve_timing.py [P ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coulombgas_amd.engine import Engine, DeviceArray

eng = Engine(5, 2, 2, 16, 16, 4.0, np.zeros((9, 2)))
rng = np.random.default_rng(0)
for P in [int(a) for a in sys.argv[1:]] or [333, 1074, 5907]:
    S = rng.standard_normal((2 * P, P)) / np.sqrt(2 * P)
    F = S.T @ S + 1e-3 * np.eye(P)
    F_d = DeviceArray.from_numpy(eng, F)
    b = rng.standard_normal(P)
    x = eng.spd_solve_d(F_d, b, 1e-3)
    ts = []
    for _ in range(5):
        eng.timer_start(); x = eng.spd_solve_d(F_d, b, 1e-3); ts.append(eng.timer_stop())
    xr = np.linalg.solve(F + 1e-3 * np.eye(P), b)
    print("P=%d: solve %.3f ms (min of 5; median %.3f)   max|x - x_ref| / max|x_ref| = %.2e" % (P, min(ts), sorted(ts)[2], np.abs(x - xr).max() / np.abs(xr).max()), flush=True)
    F_d.free()
