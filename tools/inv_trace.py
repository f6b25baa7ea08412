#!/usr/bin/env python3
"""Diagnostic: cycles of thread 0 per phase of the real panel inverse inside k_scores_big (-DCG_INV_TRACE build of cg_k_big.hip).
   tools/devbuild_variant.sh invtrace "-DCG_INV_TRACE" cg_k_big
   COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libinvtrace.so python tools/inv_trace.py [n] [B]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine
from coulombgas_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 57
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L, sp, theta, sidx, x = synthetic(n, 2, B, {29: 25, 49: 36, 57: 49}.get(n, 25 if n <= 40 else 49), 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
fn = C.CDLL(_lib.LIB_PATH).cg_debug_inv_trace
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros(8, dtype=np.uint64)
eng.quantum_score(x, sidx); fn(eng._ctx, buf.ctypes.data, 1)
eng.quantum_score(x, sidx); fn(eng._ctx, buf.ctypes.data, 1)
names = {1: "panel (owners)", 2: "barrier 1", 3: "pivot rows", 4: "barrier 2", 5: "update"}
print("real panel inverse inside k_scores_big, n=%d B=%d: cycles of thread 0 per walker" % (n, B))
for k, nm in names.items():
    print("  ph%d %-16s %10.0f" % (k, nm, buf[k] / B))
print("  sum %.0f" % (buf[1:6].sum() / B))
