#!/usr/bin/env python3
"""Generates tests/golden/golden_*.npz with the oracle (oracle/cg_ref.py, torch.func restatement of the
reference): inputs + expected outputs in fp64.  Run once in the build container; the .npz are committed.

  golden_n7_d3.npz    the reference tests' shape (tests/test_logpsi.py:30-31; depth 2, spsize = tpsize = 16), L = 1.234
  golden_n13_d2.npz   BASELINE configs 1-3 shape, init-like N(0,0.01^2) weights
  golden_n29_d2.npz   shipped trained flow parameters (data/n_29_..._rs_10.0, epoch 3000) on 2 shipped walkers
  golden_n49_d2.npz   (--n49) shipped parameters + walkers of data/n_49_..._rs_10.0 (Emax 36); Hutchinson variants only
  golden_n57_d2.npz, golden_n29_d2_rs1.npz   (--large) BASELINE configs 5 and 4: shipped parameters + walkers of
                      data/n_57_..._rs_10.0 (Emax 49) and data/n_29_..._rs_1.0; Hutchinson variants only
each: x, state_idx, theta, sp_indices -> z, J, half_logdetJ, logphi, grad/lap (exact), lap (Hutchinson-split and full for the
stored probe v), Ewald V, theta-VJP for stored weights, a 5-step Metropolis trajectory for stored noise.
"""
import os, sys, time
import numpy as np
import torch
from torch.func import jacfwd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import cg_ref as R
from tests.common import orbitals, box_length, flow_theta, state_indices, walkers


def make(name, n, dim, hs, ht, L, theta, x, sidx, sp, rs, seed, exact=True):
    t0 = time.time()
    rng = np.random.default_rng(seed)
    B = x.shape[0]
    flow = R.FermiNet(2, hs, ht, L)
    params = R.flow_unravel(R.T(theta), 2, hs, ht, dim)
    logpsi = R.make_logpsi(flow, sp, L)
    logphi, logjacdet = R.make_logphi_logjacdet(flow, sp, L)
    sb = torch.as_tensor(sidx.astype(np.int64))
    out = dict(n=n, dim=dim, depth=2, spsize=hs, tpsize=ht, L=L, theta=theta, x=x, state_idx=sidx, sp_indices=sp, rs=rs, kappa=10.0)
    z, J, hld, lphi = [], [], [], []
    for b in range(B):
        xb = R.T(x[b])
        z.append(flow.apply(params, xb).numpy())
        J.append(jacfwd(lambda xf: flow.apply(params, xf.reshape(n, dim)).reshape(-1))(xb.reshape(-1)).numpy())
        hld.append(float(logjacdet(xb, params)))
        lphi.append(logphi(xb, params, sb[b]).numpy())
    out.update(z=np.array(z), J=np.array(J), half_logdetJ=np.array(hld), logphi=np.array(lphi))
    v = rng.standard_normal(x.shape)
    out["v"] = v
    if exact:
        _, fn = R.make_logpsi_grad_laplacian(logpsi)
        g, l = fn(R.T(x), params, sb)
        out.update(grad=g.numpy(), lap_exact=l.numpy())
    _, fn = R.make_logpsi_grad_laplacian(logpsi, hutchinson=True)
    g1, l1 = fn(R.T(x), params, sb, R.T(v))
    _, fn = R.make_logpsi_grad_laplacian(logpsi, hutchinson=True, logphi=logphi, logjacdet=logjacdet)
    g2, l2 = fn(R.T(x), params, sb, R.T(v))
    out.update(grad_hutch=g1.numpy(), lap_hutch=l1.numpy(), grad_split=g2.numpy(), lap_split=l2.numpy())
    G = R.kpoints(dim, 15 if dim == 2 else 7)
    out.update(G=G, V=R.potential_energy(R.T(x), 10.0, G, L, rs).numpy(), madelung=R.Madelung(dim, 10.0, G))
    w_re, w_im = rng.standard_normal(B), rng.standard_normal(B)
    lpt = lambda xb, th, sbb: logpsi(xb, R.flow_unravel(th, 2, hs, ht, dim), sbb)

    def S(th):
        o = torch.stack([lpt(R.T(x[b]), th, sb[b]) for b in range(B)])
        return (R.T(w_re) * o[:, 0] + R.T(w_im) * o[:, 1]).sum()
    out.update(w_re=w_re, w_im=w_im, vjp=torch.func.grad(S)(R.T(theta)).numpy())
    steps = 5
    noise = rng.standard_normal((steps,) + x.shape); unif = rng.uniform(size=(steps, B))
    logp = R.make_logp(logpsi)
    xm, lpm, rate = R.mcmc(lambda xx: logp(xx, params, sb), R.T(x), R.T(noise), R.T(unif), steps, 0.1)
    out.update(mc_noise=noise, mc_unif=unif, mc_x=xm.numpy(), mc_logp=lpm.numpy(), mc_rate=rate, mc_stddev=0.1)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "%.1fs" % (time.time() - t0), "rate", rate)


def make_large():
    """BASELINE configs 4 and 5: shipped trained parameters + shipped walkers at n = 57 (rs = 10, Emax = 49) and n = 29 at
    rs = 1.  The exact AD Laplacian is minutes per walker there: Hutchinson variants only."""
    rng = np.random.default_rng(20261004)
    d = np.load(os.path.join(HERE, "shipped_n57_rs10.npz"))
    n = 57; L = box_length(n, 2); sp = orbitals(2, 49)
    make("golden_n57_d2.npz", n, 2, 16, 16, L, d["theta"], d["x"][:2], state_indices(rng, 2, n, sp.shape[0]), sp, 10.0, 4, exact=False)
    d = np.load(os.path.join(HERE, "shipped_n29_rs1.npz"))
    n = 29; L = box_length(n, 2); sp = orbitals(2, 25)
    make("golden_n29_d2_rs1.npz", n, 2, 16, 16, L, d["theta"], d["x"][:2], state_indices(rng, 2, n, sp.shape[0]), sp, 1.0, 5, exact=False)


def make_n49():
    """n = 49 / Emax = 36 (six shipped production runs, no BASELINE config): shipped parameters + walkers of data/n_49_..._rs_10.0"""
    rng = np.random.default_rng(20261049)
    d = np.load(os.path.join(HERE, "shipped_runs", "n49_rs10.0.npz"))
    n = 49; L = box_length(n, 2); sp = orbitals(2, 36)
    make("golden_n49_d2.npz", n, 2, 16, 16, L, d["theta"], d["x"][:2], state_indices(rng, 2, n, sp.shape[0]), sp, 10.0, 6, exact=False)


def add_exact_to_large():
    """(--exact-large, round 5) the reference's DEFAULT Laplacian mode (src/logpsi.py:63-106, jacrev + n d jvp's) at the production
    sizes: adds `grad_exact1`, `lap_exact1` -- walker 0 only, the AD nest is minutes per walker there -- to golden_n57_d2.npz,
    golden_n49_d2.npz and golden_n29_d2_rs1.npz.  Every other array of the files is kept as it is."""
    for name in ("golden_n29_d2_rs1.npz", "golden_n49_d2.npz", "golden_n57_d2.npz"):
        t0 = time.time()
        path = os.path.join(HERE, name)
        g = dict(np.load(path))
        n, dim, hs, ht, L = int(g["n"]), int(g["dim"]), int(g["spsize"]), int(g["tpsize"]), float(g["L"])
        flow = R.FermiNet(2, hs, ht, L)
        params = R.flow_unravel(R.T(g["theta"]), 2, hs, ht, dim)
        logpsi = R.make_logpsi(flow, g["sp_indices"], L)
        _, fn = R.make_logpsi_grad_laplacian(logpsi)
        gr, lp = fn(R.T(g["x"][:1]), params, torch.as_tensor(g["state_idx"][:1].astype(np.int64)))
        g["grad_exact1"] = gr.numpy(); g["lap_exact1"] = lp.numpy()
        # the exact gradient is the Hutchinson variants' gradient (tests/test_logpsi.py:151)
        assert np.abs(g["grad_exact1"][0] - g["grad_split"][0]).max() < 1e-9 * max(1.0, np.abs(g["grad_split"][0]).max())
        np.savez_compressed(path, **g)
        print(name, "exact mode, walker 0: %.1fs" % (time.time() - t0), "lap", g["lap_exact1"], flush=True)


if __name__ == "__main__":
    if "--exact-large" in sys.argv:
        add_exact_to_large(); sys.exit(0)
    if "--n49" in sys.argv:
        make_n49(); sys.exit(0)
    if "--large" in sys.argv:
        make_large(); sys.exit(0)
    rng = np.random.default_rng(20261003)
    # n=7, d=3 (reference test shape)
    n, dim, L = 7, 3, 1.234
    sp = orbitals(3)
    make("golden_n7_d3.npz", n, dim, 16, 16, L, flow_theta(rng, 2, 16, 16, dim, 0.2, 0.1), walkers(rng, 2, n, dim, L),
         state_indices(rng, 2, n, sp.shape[0]), sp, 1.0, 1)
    # n=13, d=2
    n, dim = 13, 2
    L = box_length(n, dim); sp = orbitals(2, 25)
    make("golden_n13_d2.npz", n, dim, 16, 16, L, flow_theta(rng, 2, 16, 16, dim, 0.01, 0.0), walkers(rng, 3, n, dim, L),
         state_indices(rng, 3, n, sp.shape[0]), sp, 10.0, 2)
    # n=29 shipped parameters + shipped walkers (Hutchinson variants only: exact AD Laplacian is ~15 s/walker)
    d = np.load(os.path.join(HERE, "shipped_n29_rs10.npz"))
    n = 29; L = box_length(n, 2)
    make("golden_n29_d2.npz", n, 2, 16, 16, L, d["theta"], d["x"][:2], state_indices(rng, 2, n, sp.shape[0]), sp, 10.0, 3, exact=True)
