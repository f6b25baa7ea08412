#!/usr/bin/env python3
"""Diagnostic: HIP-event time of the two derivative kernels of an optimisation step on device-resident inputs --
cg_grad_laplacian (Hutchinson-split, src/logpsi.py:134-164) and cg_scores_compute (per-sample scores, src/logpsi.py:183-203).
   python tools/deriv_timing.py [n] [B] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine, DeviceArray

n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
Emax = {13: 25, 29: 25, 49: 36, 57: 49}.get(n, 25 if n <= 40 else 49)
L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
x_d = DeviceArray.from_numpy(eng, x); s_d = DeviceArray.from_numpy(eng, sidx, np.int32)
v_d = DeviceArray.from_numpy(eng, np.random.default_rng(0).standard_normal(x.shape))


def timed(name, fn):
    fn(); eng.sync()
    ts = []
    for _ in range(reps):
        eng.timer_start(); fn(); ts.append(eng.timer_stop())
    print("n=%d B=%d %-34s median %.3f ms  (min %.3f, max %.3f)" % (n, B, name, sorted(ts)[len(ts) // 2], min(ts), max(ts)), flush=True)


def scores():
    x_d.version += 1                      # defeat the engine's score cache: time the kernel, not the look-up
    eng.scores_compute_d(x_d, s_d)


timed("grad_laplacian hutchinson-split", lambda: eng.grad_laplacian_d(x_d, s_d, 2, v_d))
g_d, l_d = eng.grad_laplacian_d(x_d, s_d, 2, v_d)[:2]
g, lp = np.asarray(g_d), np.asarray(l_d)
print("   checksums: |grad| %.15e  lap re %.15e im %.15e" % (np.abs(g).sum(), lp.real.sum() if np.iscomplexobj(lp) else lp[..., 0].sum(), lp.imag.sum() if np.iscomplexobj(lp) else lp[..., -1].sum()))
timed("scores_compute (k_param_vjp)", scores)


def fused():
    x_d.version += 1
    eng.grad_laplacian_d(x_d, s_d, 2, v_d, with_scores=True)


timed("grad_laplacian + scores, one call", fused)
