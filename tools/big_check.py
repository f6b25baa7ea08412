#!/usr/bin/env python3
"""Development check (GPU box): the planned large-n derivative kernels (csrc/cg_big.hpp) against the first generation on the same
walkers -- per-sample scores and grad / Laplacian -- and their HIP-event times.
   python tools/big_check.py [n ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import synthetic
from coulombgas_amd.engine import Engine, DeviceArray

sizes = [int(a) for a in sys.argv[1:]] or [29, 57, 49, 20, 33, 45, 64]
rel = lambda a, b: float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
worst = 0.0
for n in sizes:
    Emax = 25 if n <= 40 else 49
    B = {29: 2048, 57: 512, 49: 512}.get(n, 300)
    L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 0)
    eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
    v = np.random.default_rng(1).standard_normal(x.shape)
    out = {}
    for big in ("0", "1"):
        os.environ["CG_BIG"] = big
        out["s" + big] = eng.quantum_score(x, sidx)
        for mode in (2, 1):
            out["g%d%s" % (mode, big)], out["l%d%s" % (mode, big)] = eng.grad_laplacian(x, sidx, mode, v)
    es = rel(out["s1"], out["s0"])
    eg = max(rel(out["g21"], out["g20"]), rel(out["g11"], out["g10"]))
    el = max(rel(out["l21"], out["l20"]), rel(out["l11"], out["l10"]))
    worst = max(worst, es, eg, el)
    print("n=%d B=%d: scores rel diff %.2e   grad %.2e   laplacian %.2e   (finite: %s)" % (n, B, es, eg, el, bool(np.isfinite(out["s1"]).all())), flush=True)
    x_d = DeviceArray.from_numpy(eng, x); s_d = DeviceArray.from_numpy(eng, sidx, np.int32); v_d = DeviceArray.from_numpy(eng, v)
    for big in ("0", "1"):
        os.environ["CG_BIG"] = big
        ts, tg = [], []
        for r in range(4):
            x_d.version += 1
            eng.timer_start(); eng.scores_compute_d(x_d, s_d); ts.append(eng.timer_stop())
            eng.timer_start(); eng.grad_laplacian_d(x_d, s_d, 2, v_d); tg.append(eng.timer_stop())
        print("   CG_BIG=%s: scores %.3f ms   grad/laplacian (split) %.3f ms" % (big, sorted(ts[1:])[1], sorted(tg[1:])[1]), flush=True)
    eng.close()
print("worst relative difference %.2e" % worst)
sys.exit(0 if worst < 1e-9 else 1)
