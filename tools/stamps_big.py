#!/usr/bin/env python3
"""Diagnostic: cycles per phase of the planned large-n derivative kernels (csrc/cg_big.hpp; s_memtime stamps, -DCG_STAMPS build).
   python -m coulombgas_amd.build --diag cg_stamps -DCG_STAMPS -DCG_ONLY_2_16_16
   COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python tools/stamps_big.py [n] [B]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine
from coulombgas_amd import _lib
GL = {20: "set-up (flow, J, J^-T, D^-1, g, K, T^a)", 21: "Slater part (J^T g, tr J^T H J)", 23: "forward Laplacian", 22: "reverse sweep (xbar)", 24: "jet pass"}
GLS = {25: "set-up: x, k_occ, primal", 26: "set-up: Jacobian assembly", 27: "set-up: Slater matrix + both inverses", 28: "set-up: g, diag K", 18: "set-up: T^a (MFMA)",
       0: "set-up:   of which the real inverse (J^-T)",
       13: "forward Laplacian: row pass (pair sums)", 14: "forward Laplacian: |grad u2|^2 (MFMA)",
       15: "reverse: pass A, Gbar, pass B", 16: "reverse: Rbar + dense chain", 17: "reverse: pair pass + xbar", 29: "jet: half-angle jets, pair sums, dense tangents",
       30: "jet: factor tangents, G pass", 31: "jet: pair pass (J', t2)", 19: "jet: traces (M = J^-1 J', t3)"}
SC = {20: "set-up (flow, J, J^-T, D^-1, g)", 13: "pass A (U'bar, Bbar, Vbar, Wt partials)", 14: "Gbar (MFMA)", 19: "pass B (sg1bar, Ubar, W0 partials)",
      21: "Rbar + dense chain", 22: "pass C (primal pair stream)", 23: "score row"}
SCS = {25: "set-up: x, k_occ, primal", 26: "set-up: Jacobian assembly", 27: "set-up: Slater matrix + both inverses", 0: "set-up:   of which the real inverse (J^-T)", 28: "set-up: g"}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 57
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
L, sp, theta, sidx, x = synthetic(n, 2, B, {29: 25, 49: 36, 57: 49}.get(n, 25 if n <= 40 else 49), 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
v = np.random.default_rng(0).standard_normal(x.shape)
fn = C.CDLL(_lib.LIB_PATH).cg_debug_stamps_big
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
W = 4 if n * 2 <= 64 else 8


def show(title, names, subs, run):
    buf = np.zeros(64, dtype=np.uint64)
    run(); fn(eng._ctx, buf.ctypes.data, 1)
    run(); fn(eng._ctx, buf.ctypes.data, 1)
    cyc = buf.astype(np.int64).astype(np.float64)
    tot = sum(cyc[k] for k in names)
    print("%s n=%d B=%d: wave-cycles per walker by phase (%d waves per workgroup):" % (title, n, B, W))
    for k, nm in names.items():
        print("  %2d %-48s %10.0f  %5.1f %%" % (k, nm, cyc[k] / B / W, 100 * cyc[k] / tot))
    print("  total %.0f cycles per walker per wave" % (tot / B / W))
    for k, nm in subs.items():
        if cyc[k]:
            print("     %2d %-45s %10.0f  %5.1f %% of the kernel" % (k, nm, cyc[k] / B / W, 100 * cyc[k] / tot))


show("k_gradlap_big (Hutchinson-split)", GL, GLS, lambda: eng.grad_laplacian(x, sidx, 2, v))
show("k_scores_big", SC, SCS, lambda: eng.quantum_score(x, sidx))
