#!/usr/bin/env python3
"""Diagnostic: cycles per phase of k_grad_lap2 (s_memtime stamps, -DCG_STAMPS build).
   python -m coulombgas_amd.build --diag cg_stamps -DCG_STAMPS -DCG_ONLY_2_16_16
   COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python tools/stamps_gradlap.py [n] [B] [mode]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine
from coulombgas_amd import _lib
NAMES = {20: "set-up (primal, J, J^-1, D^-1, T, K)", 21: "Slater part (J^T g, tr J^T H J)", 22: "reverse sweep (xbar)",
         23: "forward Laplacian", 24: "jet pass(es)"}
SUB = {25: "set-up: x, k_occ, primal", 26: "set-up: pair table + Jacobian assembly", 27: "set-up: Slater matrix + both inverses",
       28: "set-up: T^a, diag K^ab, g", 15: "reverse: Jhat, U'bar, Bbar, Gbar, Vbar", 16: "reverse: dense chain (sg1bar ... m0bar)",
       17: "reverse: pair pass + xbar", 29: "jet: primal", 30: "jet: Jacobian assembly", 31: "jet: traces"}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L, sp, theta, sidx, x = synthetic(n, 2, B, {13: 25, 29: 25, 57: 49}[n], 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
v = np.random.default_rng(0).standard_normal(x.shape)
fn = C.CDLL(_lib.LIB_PATH).cg_debug_stamps_derivs
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros(64, dtype=np.uint64)
eng.grad_laplacian(x, sidx, mode, v)
fn(eng._ctx, buf.ctypes.data, 1)
eng.grad_laplacian(x, sidx, mode, v)
fn(eng._ctx, buf.ctypes.data, 1)
cyc = buf.astype(np.int64).astype(np.float64)
tot = sum(cyc[k] for k in NAMES)
W = 4 if n <= 16 else 8                # waves per workgroup of the launch (256 threads all-LDS / 512 threads)
print("n=%d B=%d mode=%d: wave-cycles per walker by phase (%d waves per workgroup):" % (n, B, mode, W))
for k, nm in NAMES.items():
    print("  %2d %-36s %10.0f  %5.1f %%" % (k, nm, cyc[k] / B / W, 100 * cyc[k] / tot))
print("  total %.0f cycles per walker per wave" % (tot / B / W))
for k, nm in SUB.items():
    if cyc[k]:
        print("  %2d %-44s %10.0f  %5.1f %% of the kernel" % (k, nm, cyc[k] / B / W, 100 * cyc[k] / tot))
