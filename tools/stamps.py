#!/usr/bin/env python3
"""Diagnostic: where the cycles of one Metropolis step go (s_memtime stamps, -DCG_STAMPS build).
   python -m coulombgas_amd.build --diag cg_stamps -DCG_STAMPS -DCG_ONLY_2_16_16
   COULOMBGAS_HIP_LIB=coulombgas_amd/lib/diag/libcg_stamps.so python tools/stamps.py [n] [B]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine
from coulombgas_amd import _lib

NAMES = ["proposal+rng", "sincos", "primal pairs", "primal dense", "wt staging", "factors U,Bm", "G pass", "Up", "B.G",
         "factors V", "jacobian pairs", "diag blocks", "slater matrix", "real LU", "complex LU", "accept",
         "dual LU: real chain (wave 0)", "dual LU: complex chain (wave 1)", "-", "dual LU: helper tasks (all waves summed)",
         "real chain: column steps + E", "real chain: publish", "real chain: waiting", "real chain: fetch + rank-8 update", "-",
         "complex chain: column steps + E", "complex chain: publish", "complex chain: waiting", "complex chain: fetch + rank-8 update"]
# (n > 16, both matrices in LDS: 12 = Slater matrix, 13 = both LUs (concurrent), 14 = 0; 16-28: inside the dual LU (cg_blocked_lu_dual2), already counted in 13;
#  16 / 17 and 20-28 are one wave each)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
Emax = {13: 25, 29: 25, 49: 36, 57: 49}[n]
L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta); eng.device_mode(True)
d_x = eng.alloc((B, n, 2)).upload(x); d_s = eng.alloc((B, n), np.int32).upload(sidx); d_lp = eng.alloc((B,))
fn = C.CDLL(_lib.LIB_PATH).cg_debug_stamps
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros(64, dtype=np.uint64)
steps = 50
eng.mcmc_dev(d_x, d_s, B, steps, 0.1, seed=1, walker_offset=0, logp_buf=d_lp)
fn(eng._ctx, buf.ctypes.data, 1)
eng.timer_start()
eng.mcmc_dev(d_x, d_s, B, steps, 0.1, seed=2, walker_offset=0, logp_buf=d_lp)
ms = eng.timer_stop()
fn(eng._ctx, buf.ctypes.data, 1)
cyc = buf.astype(np.int64)[:len(NAMES)].astype(np.float64)
tot = cyc[:16].sum()
print("n=%d B=%d: kernel %.2f ms (stamped build); cycles per wave per evaluation:" % (n, B, ms))
for k, nm in enumerate(NAMES):
    if nm == "-" or (k >= 16 and n <= 16): continue
    print("  %2d %-16s %9.0f  %5.1f %%" % (k, nm, cyc[k] / (B * (steps + 1)), 100 * cyc[k] / tot))
print("  total %.0f cycles per evaluation" % (tot / (B * (steps + 1))))
