"""CPU checks of the KERNEL ARITHMETIC: coulombgas_amd/csrc/*.hpp compiled for the host (1-thread workgroup shim,
tests/host_emul) against the committed golden vectors and the oracle.  The same comparisons run on the real
GPU build in tests/test_gpu_golden.py."""
import numpy as np
import pytest

from tests.common import GOLDEN
from tests.emul_engine import EmulEngine


def check_against_golden(eng_factory, name, tol=1e-10, exact=True):
    g = np.load(GOLDEN + "/" + name)
    n, dim, hs, ht, L = int(g["n"]), int(g["dim"]), int(g["spsize"]), int(g["tpsize"]), float(g["L"])
    eng = eng_factory(n, dim, 2, hs, ht, L, g["sp_indices"])
    eng.set_params(g["theta"])
    x, s, v = g["x"], g["state_idx"], g["v"]
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert rel(eng.flow_forward(x), g["z"]) < tol
    assert rel(eng.flow_jacobian(x), g["J"]) < tol
    lphi, hld = eng.logphi_logjacdet(x, s)
    assert rel(hld, g["half_logdetJ"]) < tol and rel(lphi[:, 0], g["logphi"][:, 0]) < tol
    assert np.abs(np.angle(np.exp(1j * (lphi[:, 1] - g["logphi"][:, 1])))).max() < tol
    if exact and "grad" in g:
        gr, lp = eng.grad_laplacian(x, s, 0)
        assert rel(gr, g["grad"]) < tol and rel(lp, g["lap_exact"]) < 10 * tol
    elif exact:
        # production sizes (n = 49, 57, n = 29 at rs = 1): the exact mode -- the reference's default, src/logpsi.py:63-106 -- on walker 0
        # (the oracle's AD nest is minutes per walker there; tests/golden/make_golden_vectors.py --exact-large)
        assert "lap_exact1" in g, "%s carries no exact-mode Laplacian" % name
        gr, lp = eng.grad_laplacian(x[:1], s[:1], 0)
        assert rel(gr, g["grad_exact1"]) < tol and rel(lp, g["lap_exact1"]) < 10 * tol
    gr, lp = eng.grad_laplacian(x, s, 1, v)
    assert rel(gr, g["grad_hutch"]) < tol and rel(lp, g["lap_hutch"]) < 10 * tol
    gr, lp = eng.grad_laplacian(x, s, 2, v)
    assert rel(gr, g["grad_split"]) < tol and rel(lp, g["lap_split"]) < 10 * tol
    eng.set_ewald(float(g["kappa"]), g["G"], float(g["rs"]))
    assert rel(eng.ewald(x), g["V"]) < tol
    assert rel(eng.param_vjp(x, s, g["w_re"], g["w_im"]), g["vjp"]) < tol
    xm, lpm, nacc = eng.mcmc(x, s, 5, float(g["mc_stddev"]), noise=g["mc_noise"], unif=g["mc_unif"])
    assert nacc / (5 * x.shape[0]) == pytest.approx(float(g["mc_rate"]), abs=1e-15)
    assert np.abs(xm - g["mc_x"]).max() < 1e-12 and rel(lpm, g["mc_logp"]) < tol


@pytest.mark.parametrize("name", ["golden_n7_d3.npz", "golden_n13_d2.npz", "golden_n29_d2.npz", "golden_n29_d2_rs1.npz",
                                  "golden_n49_d2.npz", "golden_n57_d2.npz"])
def test_device_code_on_host_vs_golden(name):
    check_against_golden(EmulEngine, name)


def test_symmetries_through_device_code():
    """tests/test_flow.py:25,32,38 and tests/test_logpsi.py:45,54,72,77 restated on the device code (depth 2)."""
    from tests.common import orbitals, flow_theta, state_indices
    n, dim, L = 7, 3, 1.234
    rng = np.random.default_rng(0)
    sp = orbitals(3)
    eng = EmulEngine(n, dim, 2, 16, 16, L, sp)
    eng.set_params(flow_theta(rng, 2, 16, 16, dim, 0.3, 0.2))
    x = rng.uniform(0, L, (n, dim))
    s = state_indices(rng, 1, n, sp.shape[0])[0]
    z = eng.flow_forward(x)
    image = rng.integers(-5, 6, size=(n, dim)) * L
    shift = rng.standard_normal(dim)
    P = rng.permutation(n)
    assert np.allclose(eng.flow_forward(x + image), z + image, atol=1e-10)
    assert np.allclose(eng.flow_forward(x + shift), z + shift, atol=1e-10)
    assert np.allclose(eng.flow_forward(x[P]), z[P], atol=1e-12)
    a = eng.logpsi(x, s)
    assert np.allclose(eng.logpsi(x + image, s)[0], a[0], atol=1e-9)
    bP = eng.logpsi(x[P], s)
    pa, pb = np.exp(a[0] + 1j * a[1]), np.exp(bP[0] + 1j * bP[1])
    assert np.allclose(pb, pa) or np.allclose(pb, -pa)
    assert np.allclose(eng.logp(x + shift, s), eng.logp(x, s), atol=1e-9)


@pytest.mark.parametrize("n,dim,depth,hs,ht,L", [(7, 3, 3, 16, 16, 1.234), (5, 2, 4, 8, 4, 2.0)])
def test_general_depth_device_code_vs_c_oracle(n, dim, depth, hs, ht, L):
    """cg_flow_generic.hpp / cg_generic.hpp (any depth) on the host shim against oracle/cg_oracle.c."""
    import ctypes as C
    from tests.common import orbitals, flow_theta, state_indices, walkers
    from tests.emul_engine import lib as emul_lib
    from coulombgas_amd.build import build_oracle
    olib = C.CDLL(build_oracle())
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(3)
    sp = orbitals(dim)
    B = 2
    theta = flow_theta(rng, depth, hs, ht, dim, 0.3, 0.1)
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, sp.shape[0])
    out = np.zeros((B, 3)); z = np.zeros((B, n, dim)); J = np.zeros((B, n * dim, n * dim))
    npar = emul_lib().emu_gen_logpsi(n, dim, depth, hs, ht, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(x), B, p(out), p(z), p(J))
    assert npar == theta.size == olib.cgo_num_params(dim, depth, hs, ht)
    ref = np.zeros((B, 3)); zr = np.zeros_like(z); Jr = np.zeros_like(J)
    olib.cgo_logpsi(n, dim, depth, hs, ht, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(x), B, p(ref))
    olib.cgo_flow(n, dim, depth, hs, ht, C.c_double(L), p(theta), p(x), B, p(zr), p(Jr))
    assert np.abs(z - zr).max() < 1e-13 and np.abs(J - Jr).max() < 1e-13
    assert np.abs(out[:, 0] - ref[:, 0]).max() < 1e-11 * np.abs(ref[:, 0]).max()
    assert np.abs(np.angle(np.exp(1j * (out[:, 1] - ref[:, 1])))).max() < 1e-11
    assert np.abs(out[:, 2] - ref[:, 2]).max() < 1e-12


@pytest.mark.parametrize("n,dim,depth,hs,ht,L", [(4, 2, 3, 8, 4, 2.0), (3, 3, 4, 4, 6, 1.234), (5, 2, 2, 6, 6, 2.5)])
def test_general_depth_param_vjp_vs_oracle(n, dim, depth, hs, ht, L):
    """theta-VJP / per-sample scores of the general-depth path (cg_generic.hpp: dual-number reverse passes of the primal
    flow) on the host shim against jacrev of the oracle's log Psi (main.py:277-278, src/logpsi.py:183-203)."""
    import ctypes as C
    import torch
    from oracle import cg_ref as R
    from tests.common import orbitals, flow_theta, state_indices, walkers
    from tests.emul_engine import lib as emul_lib
    p = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    rng = np.random.default_rng(10 + depth)
    sp = orbitals(dim)
    B = 2
    theta = flow_theta(rng, depth, hs, ht, dim, 0.4, 0.2)
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, sp.shape[0])
    w_re, w_im = rng.standard_normal(B), rng.standard_normal(B)
    g = np.zeros(theta.size)
    npar = emul_lib().emu_gen_param_vjp(n, dim, depth, hs, ht, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(x), B,
                                        p(w_re), p(w_im), p(g), None)
    assert npar == theta.size
    rflow = R.FermiNet(depth, hs, ht, L)
    r_logpsi = R.make_logpsi(rflow, sp, L)
    sb = torch.as_tensor(sidx.astype(np.int64))
    lpt = lambda xb, th, sbb: r_logpsi(xb, R.flow_unravel(th, depth, hs, ht, dim), sbb)

    def S(th):
        out = torch.stack([lpt(R.T(x[b]), th, sb[b]) for b in range(B)])
        return (R.T(w_re) * out[:, 0] + R.T(w_im) * out[:, 1]).sum()
    gr = torch.func.grad(S)(R.T(theta)).numpy()
    assert np.abs(g - gr).max() < 1e-10 * max(1.0, np.abs(gr).max())
    sc = np.zeros((B, theta.size, 2))
    emul_lib().emu_gen_param_vjp(n, dim, depth, hs, ht, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(x), B, None, None, None, p(sc))
    qr = R.make_quantum_score(lpt)(R.T(x), R.T(theta), sb).numpy()
    assert np.abs(sc[..., 0] + 1j * sc[..., 1] - qr).max() < 1e-10 * max(1.0, np.abs(qr).max())


@pytest.mark.parametrize("budget", [0, 3000, 6000, 1 << 20])
def test_grad_laplacian_memory_placements(budget):
    """cg_lap.hpp places its three memory blocks in LDS or in the HBM workspace depending on the LDS budget; every
    placement must give the same numbers (n = 13: all three modes against the golden vectors)."""
    g = np.load(GOLDEN + "/golden_n13_d2.npz")
    eng = EmulEngine(int(g["n"]), int(g["dim"]), 2, int(g["spsize"]), int(g["tpsize"]), float(g["L"]), g["sp_indices"])
    eng.lds_budget = budget
    eng.set_params(g["theta"])
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    for mode, gk, lk in ((0, "grad", "lap_exact"), (1, "grad_hutch", "lap_hutch"), (2, "grad_split", "lap_split")):
        gr, lp = eng.grad_laplacian(g["x"], g["state_idx"], mode, g["v"])
        assert rel(gr, g[gk]) < 1e-11 and rel(lp, g[lk]) < 1e-10


@pytest.mark.parametrize("name", ["golden_n7_d3.npz", "golden_n13_d2.npz"])
def test_second_generation_scores_vs_golden(name):
    """csrc/cg_score.hpp (the score kernel of the small systems: both parts in one sweep, pair table, planned LDS lifetimes) on the
    host shim: its per-sample scores reproduce the oracle's theta-VJP of the golden vectors and the first-generation sweep."""
    g = np.load(GOLDEN + "/" + name)
    n, dim = int(g["n"]), int(g["dim"])
    eng = EmulEngine(n, dim, 2, 16, 16, float(g["L"]), g["sp_indices"]); eng.set_params(g["theta"])
    x, s = g["x"], g["state_idx"]
    S2 = eng.quantum_score2(x, s)
    vjp = (g["w_re"][:, None] * S2.real + g["w_im"][:, None] * S2.imag).sum(axis=0)
    assert np.abs(vjp - g["vjp"]).max() < 1e-10 * np.abs(g["vjp"]).max()
    S1 = eng.quantum_score(x, s)
    assert np.abs(S2 - S1).max() < 1e-11 * np.abs(S1).max()
