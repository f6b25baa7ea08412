#!/usr/bin/env python3
"""Tuning run: k_mcmc (50 Metropolis steps) against the workgroup size (cg_set_block_threads) at the larger systems.
   python tools/sampler_threads_sweep.py [n] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine, DeviceArray

n = int(sys.argv[1]) if len(sys.argv) > 1 else 29
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
Emax = {13: 25, 29: 25, 49: 36, 57: 49}.get(n, 25 if n <= 40 else 36 if n <= 52 else 49)
L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
s_d = DeviceArray.from_numpy(eng, sidx, np.int32)
for t in [int(a) for a in sys.argv[3:]] or [0, 128, 192, 256, 320, 384, 512, 768, 1024]:
    try:
        eng.set_block_threads(t)
        x_d = DeviceArray.from_numpy(eng, x)
        eng.mcmc_d(x_d, s_d, 50, 0.1, seed=1); eng.sync()
        ts = []
        for r in range(3):
            eng.timer_start(); out = eng.mcmc_d(x_d, s_d, 50, 0.1, seed=2 + r); ts.append(eng.timer_stop())
        print("n=%d B=%d threads %4d: %.3f ms  -> %.3f M walker-steps/s" % (n, B, t, min(ts), B * 50 / min(ts) / 1e3), flush=True)
    except Exception as e:                                  # noqa: BLE001 -- a shape the kernel refuses is a data point
        print("n=%d B=%d threads %4d: %s" % (n, B, t, str(e)[:120]), flush=True)
