#!/usr/bin/env python3
"""Diagnostic: GPU log phi / half log|det J| vs the oracle for the 2-D spsize=tpsize=16 test cases."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_parity import _setup, CASES
import coulombgas_amd as cg
from oracle import cg_ref as R
for case in [CASES[0], CASES[1], (16, 2, 16, 16, None, 0.2, 0.1), (9, 2, 16, 16, None, 0.2, 0.1)]:
    s = _setup(case, 4)
    logphi, logjacdet = cg.make_logphi_logjacdet(s["flow"], s["sp"], s["L"])
    lphi = logphi(s["x"], s["theta"], s["sidx"]); ljd = logjacdet(s["x"], s["theta"])
    r_logphi, r_logjacdet = R.make_logphi_logjacdet(s["rflow"], s["sp"], s["L"])
    for b in range(4):
        xb, sb = R.T(s["x"][b]), torch.as_tensor(s["sidx"][b].astype(np.int64))
        rphi = r_logphi(xb, s["rparams"], sb).numpy(); rj = float(r_logjacdet(xb, s["rparams"]))
        print(case[:2], b, "logphi", lphi[b], rphi, "d=%.2e %.2e" % (lphi[b, 0] - rphi[0], np.angle(np.exp(1j * (lphi[b, 1] - rphi[1])))),
              "| hlogdetJ %.12f %.12f d=%.2e" % (ljd[b], rj, ljd[b] - rj))
