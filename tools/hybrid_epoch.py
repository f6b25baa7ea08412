#!/usr/bin/env python3
"""Diagnostic: a few hybrid SR epochs (Transformer density matrix and flow both on the GPU) at finite temperature."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coulombgas_amd as cg
from coulombgas_amd.synthetic import orbitals, box_length
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = box_length(n, 2); sp = orbitals(2, 25)
flow = cg.FermiNet(2, 16, 16, L); pf = flow.init(1, np.zeros((n, 2)))
van = cg.Transformer(sp.shape[0], 2, 16, 4, 32); pv = van.init(2, sp[-n:])
samp, logp = cg.make_autoregressive_sampler(van, sp, n, sp.shape[0])        # train() attaches its engine
t = [time.perf_counter()]
def log(row):
    t.append(time.perf_counter()); print(row, " | %.0f ms" % ((t[-1] - t[-2]) * 1e3), flush=True)
cg.train(flow, pf, sp, n, 2, L, rs=10.0, beta=1 / (4 * 0.15), batch=B, epochs=4, sampler=samp, log_prob=logp, params_van=pv,
         sr=(1e-3, 1e-3), mc_therm=2, mc_steps=50, seed=3, log=log)
