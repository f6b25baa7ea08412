#!/usr/bin/env python3
"""Diagnostic: the density-matrix Transformer kernels on the device (shipped / pretrained models of tests/golden): sampling
(k_van<true>), log-probability (k_van<false>) and per-sample scores (k_van_grad*), HIP-event time per call.
   python tools/van_timing.py [n] [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coulombgas_amd as cg
from coulombgas_amd.engine import Engine
from tests.common import orbitals, box_length, GOLDEN

n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
Emax = {13: 25, 29: 25, 49: 36, 57: 49}[n]
z = np.load(os.path.join(GOLDEN, {13: "pretrained_van_n13.npz", 29: "shipped_n29_rs10_van.npz", 49: "pretrained_van_n49.npz", 57: "shipped_n57_rs10_van.npz"}[n]))
pv = {}
for k in z.files:
    if "|" in k:
        m, l = k.split("|"); pv.setdefault(m, {})[l] = z[k]
sp = orbitals(2, Emax)
eng = Engine(n, 2, 2, 16, 16, box_length(n, 2), sp)
van = cg.Transformer(sp.shape[0], 2, 16, 4, 32)
sampler, log_prob = cg.make_autoregressive_sampler(van, sp, n, sp.shape[0], engine=eng)


def timed(name, fn, reps=5):
    fn(); eng.sync()
    ts = []
    for _ in range(reps):
        eng.timer_start(); fn(); ts.append(eng.timer_stop())
    print("n=%d B=%d %-28s %.3f ms (min of %d; median %.3f)" % (n, B, name, min(ts), reps, sorted(ts)[reps // 2]), flush=True)


seed = [0]
def samp():
    seed[0] += 1
    return sampler(pv, seed[0], B)
s_d = samp()
timed("sample (k_van<true>)", samp)
timed("log_prob (k_van<false>)", lambda: log_prob(pv, s_d))
def scores():
    s_d.version += 1
    eng.van_scores_compute_d(s_d)
timed("scores (k_van_grad)", scores)
