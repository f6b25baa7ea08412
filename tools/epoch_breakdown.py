#!/usr/bin/env python3
"""Diagnostic: host wall time of each call of one SR epoch (main.py:316-346 mirror) at a BASELINE size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import coulombgas_amd as cg
from coulombgas_amd import sr as SR
from coulombgas_amd.synthetic import orbitals, box_length
n = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
L = box_length(n, 2); sp = orbitals(2, {13: 25, 29: 25, 49: 36, 57: 49}.get(n, 25)); rs = 10.0
flow = cg.FermiNet(2, 16, 16, L); p0 = flow.init(1, np.zeros((n, 2)))
samp = cg.GroundStateSampler(n, sp.shape[0])
pv = None
if "--van" in sys.argv:                      # finite temperature: the shipped Transformer density matrix, sampled on the GPU
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                             {13: "pretrained_van_n13.npz", 29: "shipped_n29_rs10_van.npz", 57: "shipped_n57_rs10_van.npz"}[n]))
    pv = {}
    for k in z.files:
        if "|" in k:
            m, l = k.split("|"); pv.setdefault(m, {})[l] = z[k]
    van = cg.Transformer(sp.shape[0], 2, 16, 4, 32)
    s_fn, lp_fn = cg.make_autoregressive_sampler(van, sp, n, sp.shape[0], engine=flow.engine(n, 2, sp))
    samp = s_fn; samp.log_prob = lp_fn
G = cg.kpoints(2, 15); Vconst = n * rs / L * cg.Madelung(2, 10, G)
lp0 = cg.make_logpsi(flow, sp, L); logphi, logjac = cg.make_logphi_logjacdet(flow, sp, L); logp = cg.make_logp(lp0)
logpsi, lgl = cg.make_logpsi_grad_laplacian(lp0, hutchinson=True, logphi=logphi, logjacdet=logjac)
loss = cg.make_loss(samp.log_prob, logpsi, lgl, 10, G, L, rs, Vconst, 1 / 0.6)
cscore = cg.make_classical_score(samp.log_prob) if pv is not None and "--frozen-van" not in sys.argv else None
fishers_fn, opt = cg.hybrid_fisher_sr(cscore, cg.make_quantum_score(lp0), 1e-3, 1e-3)
x = np.random.default_rng(0).uniform(0, L, (B, n, 2)); key = np.random.SeedSequence(1)
if "--host-arrays" not in sys.argv:          # default: the walkers live in HBM (DeviceArray), as in coulombgas_amd.train
    from coulombgas_amd.engine import DeviceArray
    x = DeviceArray.from_numpy(flow.engine(n, 2, sp), x)
T = {}
EP = 8                                       # epochs 1 .. EP - 1 are reported (median per phase: a host hiccup in one epoch does not move it)
def tm(name, fn):
    t0 = time.perf_counter(); r = fn(); T.setdefault(name, {}); T[name][ep] = T[name].get(ep, 0.0) + time.perf_counter() - t0; return r
for ep in range(EP):
    if ep == 1: T.clear()
    key, sidx, x, acc = tm("sample", lambda: cg.sample_stateindices_and_x(key, samp, pv, logp, x, p0, 50, 0.1, L))
    data, closs, qloss = tm("observable (grad_lap+ewald)", lambda: loss(pv, p0, sidx, x, key))
    g, s = tm("quantum grad (scores + VJP)", lambda: qloss.grad(p0, reduce=True))
    gvan = None
    if cscore is not None:                   # main.py:277: jax.jacrev(classical_lossfn)
        def cgrad():
            closs(pv)
            gv = samp.log_prob.vjp(pv, sidx, closs.weights); sv = samp.log_prob.vjp(pv, sidx, closs.score_weights)
            return {m: {l: gv[m][l] - data["F_mean"] * sv[m][l] for l in gv[m]} for m in gv}
        gvan = tm("classical grad (scores + VJPs)", cgrad)
    f = tm("fishers_fn", lambda: fishers_fn(pv, p0, sidx, x))
    gf = {k: {l: g[k][l] - data["E_mean"] * s[k][l] for l in g[k]} for k in g}
    (uv, uf), _ = tm("SR solve+clip", lambda: opt.update((gvan, gf), None, f))
    p0 = tm("apply", lambda: cg.apply_updates(p0, uf))
    if uv is not None:
        pv = tm("apply", lambda: cg.apply_updates(pv, uv))
med = lambda d: sorted(d.values())[len(d) // 2]
for k, v in T.items():
    print("%-30s %7.1f ms" % (k, med(v) * 1e3))
tot = {e: sum(v.get(e, 0.0) for v in T.values()) for e in range(1, EP)}
print("%-30s %7.1f ms   (median of %d epochs; fastest %.1f, slowest %.1f)" % ("total", med(tot) * 1e3, EP - 1, min(tot.values()) * 1e3, max(tot.values()) * 1e3))
