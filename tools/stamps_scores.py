#!/usr/bin/env python3
"""Diagnostic: cycles per phase of k_scores (s_memtime stamps, -DCG_STAMPS build; python -m coulombgas_amd.build --diag).
   COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python tools/stamps_scores.py [n] [B]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine
from coulombgas_amd import _lib
NAMES = {20: "set-up (primal, J, pair table, J^-1, D^-1, g)", 21: "reverse sweep (both parts) + score row"}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
L, sp, theta, sidx, x = synthetic(n, 2, B, {13: 25, 29: 25, 57: 49}[n], 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
fn = C.CDLL(_lib.LIB_PATH).cg_debug_stamps_derivs
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros(64, dtype=np.uint64)
eng.quantum_fisher(x, sidx)
fn(eng._ctx, buf.ctypes.data, 1)
eng._score_key = None
eng.quantum_fisher(x, sidx)
fn(eng._ctx, buf.ctypes.data, 1)
cyc = buf.astype(np.int64).astype(np.float64)
tot = sum(cyc[k] for k in NAMES)
print("n=%d B=%d: wave-cycles per walker by phase (4 waves per workgroup):" % (n, B))
for k, nm in NAMES.items():
    print("  %2d %-48s %10.0f  %5.1f %%" % (k, nm, cyc[k] / B / 4, 100 * cyc[k] / tot))
print("  total %.0f cycles per walker per wave" % (tot / B / 4))
