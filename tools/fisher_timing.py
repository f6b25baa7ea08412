#!/usr/bin/env python3
"""Diagnostic: the classical Fisher matrix S^T S / B (k_fisher_real, src/sr.py:36 / :74) on a device-resident score matrix at the
shipped Transformer's size (B = 8192, P = 5907): HIP-event time, TFLOP/s of the upper block triangle, error against numpy on a corner.
   python tools/fisher_timing.py [B] [P]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from coulombgas_amd.engine import Engine, DeviceArray
from coulombgas_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = int(sys.argv[2]) if len(sys.argv) > 2 else 5907
eng = Engine(5, 2, 2, 16, 16, 4.0, np.zeros((9, 2)))
S = np.random.default_rng(0).standard_normal((B, P))
S_d = DeviceArray.from_numpy(eng, S)
F_d = eng.scratch("fisher_timing", (P, P))


def run():
    eng._dev_call(lib().cg_fisher_real, S_d.ptr, B, P, F_d.ptr)


run(); eng.sync()
ts = []
for _ in range(5):
    eng.timer_start(); run(); ts.append(eng.timer_stop())
F = np.asarray(F_d)
ref = S[:, :200].T @ S[:, :300] / B
nb = (P + 127) // 128
print("B=%d P=%d: k_fisher_real %.3f ms (min of 5; median %.3f) = %.1f TFLOP/s over the %d upper 128-blocks;  max|F - ref| / max|ref| = %.1e  symmetric %s"
      % (B, P, min(ts), sorted(ts)[2], nb * (nb + 1) / 2 * 128 * 128 * 2.0 * B / min(ts) / 1e9, nb * (nb + 1) // 2,
         np.abs(F[:200, :300] - ref).max() / np.abs(ref).max(), np.array_equal(F, F.T)), flush=True)
