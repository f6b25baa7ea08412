#!/usr/bin/env python3
"""Diagnostic: time the phases of one logp evaluation (flow primal | + Jacobian | + both LUs) on the GPU."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine
from coulombgas_amd._lib import lib, check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
Emax = {13: 25, 29: 25, 57: 49}[n]
L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta); eng.device_mode(True)
N = n * 2
d_x = eng.alloc((B, n, 2)).upload(x); d_s = eng.alloc((B, n), np.int32).upload(sidx)
d_z = eng.alloc((B, n, 2)); d_J = eng.alloc((B, N, N)); d_lp = eng.alloc((B,))
def t(fn, reps=5):
    fn(); eng.sync(); eng.timer_start()
    for _ in range(reps): fn()
    return eng.timer_stop() / reps
for thr in ([64, 128, 256] if n <= 16 else [256, 512]):
    eng.set_block_threads(thr)
    a = t(lambda: check(lib().cg_flow_forward(eng._ctx, d_x.ptr, B, d_z.ptr), eng._ctx))
    b = t(lambda: check(lib().cg_flow_jacobian(eng._ctx, d_x.ptr, B, d_J.ptr), eng._ctx))
    c = t(lambda: check(lib().cg_logp(eng._ctx, d_x.ptr, d_s.ptr, B, d_lp.ptr), eng._ctx))
    print("n=%d B=%d threads=%d: primal %.3f ms | +jacobian %.3f ms (jac %.3f) | full logp %.3f ms (LU+slater %.3f)" % (n, B, thr, a, b, b - a, c, c - b), flush=True)
