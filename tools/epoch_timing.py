#!/usr/bin/env python3
"""Diagnostic: wall time per epoch of the main.py:316-384 mirror at a BASELINE size (GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coulombgas_amd as cg
from coulombgas_amd.synthetic import orbitals, box_length
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
L = box_length(n, 2); sp = orbitals(2, {13: 25, 29: 25, 57: 49}.get(n, 25))
flow = cg.FermiNet(2, 16, 16, L)
p0 = flow.init(1, np.zeros((n, 2)))
samp = cg.GroundStateSampler(n, sp.shape[0])
log_prob, pv = samp.log_prob, None
if "--van" in sys.argv:                      # finite temperature: the shipped Transformer density matrix sampled and trained on the GPU too
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                             {13: "pretrained_van_n13.npz", 29: "shipped_n29_rs10_van.npz", 57: "shipped_n57_rs10_van.npz"}[n]))
    pv = {}
    for k in z.files:
        if "|" in k:
            m, l = k.split("|"); pv.setdefault(m, {})[l] = z[k]
    van = cg.Transformer(sp.shape[0], 2, 16, 4, 32)
    samp, log_prob = cg.make_autoregressive_sampler(van, sp, n, sp.shape[0], engine=flow.engine(n, 2, sp))
t = [time.perf_counter()]
def log(row):
    t.append(time.perf_counter()); print(row, " | %.1f ms" % ((t[-1] - t[-2]) * 1e3), flush=True)
cg.train(flow, p0, sp, n, 2, L, rs=10.0, beta=1 / (4 * 0.15), batch=B, epochs=6, sampler=samp, log_prob=log_prob, params_van=pv,
         sr=(1e-3, 1e-3), mc_therm=2, mc_steps=50, seed=3, log=log)
