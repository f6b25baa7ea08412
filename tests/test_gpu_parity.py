"""GPU parity tests (run with -m gpu on an MI355X): libcoulombgas_hip.so through the C-ABI vs the oracle
(oracle/cg_ref.py, torch.func restatement of the reference) on the same seeded inputs.
fp64 tolerances are stated per assertion; BASELINE.json asks for 1e-8 relative on energies."""
import os

import numpy as np
import pytest
import torch

from tests.common import orbitals, box_length, flow_theta, state_indices, walkers, GOLDEN as GOLDEN_DIR
from tests.host_transformer import make_host_sampler

pytestmark = pytest.mark.gpu

CASES = [  # n, dim, spsize, tpsize, L (None = main.py's box), weight std, bias std
    (13, 2, 16, 16, None, 0.01, 0.0),
    (13, 2, 16, 16, None, 0.3, 0.2),
    (7, 3, 16, 16, 1.234, 0.3, 0.2),      # the reference tests' shape (tests/test_logpsi.py:30-31), depth 2
    (7, 3, 4, 4, 1.234, 0.5, 0.3),        # tests/test_logpsi.py:110
    (29, 2, 16, 16, None, 0.2, 0.1),
]


def _setup(case, B, seed=0):
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    n, dim, hs, ht, L, ws, bs = case
    L = box_length(n, dim) if L is None else L
    rng = np.random.default_rng(seed)
    sp = orbitals(dim)
    theta = flow_theta(rng, 2, hs, ht, dim, ws, bs)
    x = walkers(rng, B, n, dim, L)
    sidx = state_indices(rng, B, n, sp.shape[0])
    flow = cg.FermiNet(2, hs, ht, L)
    rflow = R.FermiNet(2, hs, ht, L)
    rparams = R.flow_unravel(R.T(theta), 2, hs, ht, dim)
    return dict(n=n, dim=dim, hs=hs, ht=ht, L=L, sp=sp, theta=theta, x=x, sidx=sidx, flow=flow, rflow=rflow,
                rparams=rparams, rng=rng)


@pytest.mark.parametrize("case", CASES)
def test_flow_and_jacobian(case):
    from oracle import cg_ref as R
    from torch.func import jacfwd
    s = _setup(case, 3)
    n, dim = s["n"], s["dim"]
    z = s["flow"].apply(s["theta"], None, s["x"])
    eng = s["flow"].engine(n, dim)
    J = eng.flow_jacobian(s["x"])
    for b in range(3):
        xb = R.T(s["x"][b])
        zr = s["rflow"].apply(s["rparams"], xb).numpy()
        Jr = jacfwd(lambda xf: s["rflow"].apply(s["rparams"], xf.reshape(n, dim)).reshape(-1))(xb.reshape(-1)).numpy()
        assert np.abs(z[b] - zr).max() < 1e-12 * max(1.0, np.abs(zr).max())
        assert np.abs(J[b] - Jr).max() < 1e-12


@pytest.mark.parametrize("case", CASES)
def test_logpsi(case):
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    s = _setup(case, 4)
    logpsi = cg.make_logpsi(s["flow"], s["sp"], s["L"])
    logphi, logjacdet = cg.make_logphi_logjacdet(s["flow"], s["sp"], s["L"])
    logp = cg.make_logp(logpsi)
    out = logpsi(s["x"], s["theta"], s["sidx"])
    lphi = logphi(s["x"], s["theta"], s["sidx"])
    ljd = logjacdet(s["x"], s["theta"])
    lp = logp(s["x"], s["theta"], s["sidx"])
    r_logpsi = R.make_logpsi(s["rflow"], s["sp"], s["L"])
    r_logphi, r_logjacdet = R.make_logphi_logjacdet(s["rflow"], s["sp"], s["L"])
    for b in range(4):
        xb, sb = R.T(s["x"][b]), torch.as_tensor(s["sidx"][b].astype(np.int64))
        ref = r_logpsi(xb, s["rparams"], sb).numpy()
        assert abs(out[b, 0] - ref[0]) < 1e-11 * max(1.0, abs(ref[0]))
        assert abs(np.angle(np.exp(1j * (out[b, 1] - ref[1])))) < 1e-11         # Im log is defined mod 2 pi
        assert abs(lp[b] - 2 * ref[0]) < 2e-11 * max(1.0, abs(ref[0]))
        rphi = r_logphi(xb, s["rparams"], sb).numpy()
        assert abs(lphi[b, 0] - rphi[0]) < 1e-11 * max(1.0, abs(rphi[0]))
        assert abs(ljd[b] - float(r_logjacdet(xb, s["rparams"]))) < 1e-12
    # single-walker (un-vmapped) call signature of the reference: x (n,dim), state_idx (n,) -> (2,)
    one = logpsi(s["x"][0], s["theta"], s["sidx"][0])
    assert one.shape == (2,) and np.array_equal(one, out[0])


@pytest.mark.parametrize("case", CASES[:3])
def test_mcmc_supplied_noise(case):
    """src/MCMC.py:22-39 with identical proposal / acceptance draws: the trajectory must be identical
    (same accept decisions), positions to 1e-12."""
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    s = _setup(case, 6, seed=3)
    steps, std = 8, 0.1
    rng = s["rng"]
    noise = rng.standard_normal((steps,) + s["x"].shape)
    unif = rng.uniform(size=(steps, s["x"].shape[0]))
    logp = cg.make_logp(cg.make_logpsi(s["flow"], s["sp"], s["L"]))
    x_new, rate = cg.mcmc(logp.bind(s["theta"], s["sidx"]), s["x"], 0, steps, std, noise=noise, unif=unif)
    r_logp = R.make_logp(R.make_logpsi(s["rflow"], s["sp"], s["L"]))
    sb = torch.as_tensor(s["sidx"].astype(np.int64))
    xr, logpr, rate_r = R.mcmc(lambda xx: r_logp(xx, s["rparams"], sb), R.T(s["x"]), R.T(noise), R.T(unif), steps, std)
    assert rate == pytest.approx(rate_r, abs=1e-15)
    assert np.abs(x_new - xr.numpy()).max() < 1e-12
    assert 0.0 < rate < 1.0


def test_mcmc_philox_statistics():
    """production sampler (in-kernel Philox): deterministic in the seed, different seeds differ,
    acceptance in the range the reference logs for identity-like flows (data.txt epoch 1: 0.537 at n=29)."""
    import coulombgas_amd as cg
    s = _setup(CASES[0], 256, seed=5)
    logp = cg.make_logp(cg.make_logpsi(s["flow"], s["sp"], s["L"]))
    bound = logp.bind(s["theta"], s["sidx"])
    x1, r1 = cg.mcmc(bound, s["x"], 42, 20, 0.1)
    x2, r2 = cg.mcmc(bound, s["x"], 42, 20, 0.1)
    x3, r3 = cg.mcmc(bound, s["x"], 43, 20, 0.1)
    assert np.array_equal(x1, x2) and r1 == r2
    assert not np.array_equal(x1, x3)
    assert 0.3 < r1 < 0.9
    # the chain's bookkeeping is consistent: final logp equals logp of the final x
    eng = s["flow"].engine(s["n"], s["dim"], s["sp"])
    xf, lpf, _ = eng.mcmc(s["x"], s["sidx"], 20, 0.1, seed=42)
    assert np.abs(lpf - eng.logp(xf, s["sidx"])).max() < 1e-10


@pytest.mark.parametrize("n,threads", [(24, 256), (45, 512), (23, 256), (41, 512)])
def test_mcmc_at_sizes_without_a_specialised_kernel(n, threads):
    """The size-generic instantiations of k_mcmc behind the automatic workgroup size (csrc/cg_host.hpp auto_threads: 256 threads from
    n = 23, the 512-thread instantiation from n = 41 -- round 4): the trajectory with supplied draws against the oracle, and the chain's
    own log-probability against the separate log Psi kernel."""
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    s = _setup((n, 2, 16, 16, None, 0.2, 0.1), 3, seed=11)
    eng = s["flow"].engine(n, 2, s["sp"])
    assert eng.launch_info()["threads"] == threads
    steps, std = 3, 0.1
    noise = s["rng"].standard_normal((steps,) + s["x"].shape)
    unif = s["rng"].uniform(size=(steps, s["x"].shape[0]))
    logp = cg.make_logp(cg.make_logpsi(s["flow"], s["sp"], s["L"]))
    x_new, rate = cg.mcmc(logp.bind(s["theta"], s["sidx"]), s["x"], 0, steps, std, noise=noise, unif=unif)
    r_logp = R.make_logp(R.make_logpsi(s["rflow"], s["sp"], s["L"]))
    sb = torch.as_tensor(s["sidx"].astype(np.int64))
    xr, logpr, rate_r = R.mcmc(lambda xx: r_logp(xx, s["rparams"], sb), R.T(s["x"]), R.T(noise), R.T(unif), steps, std)
    assert rate == pytest.approx(rate_r, abs=1e-15)
    assert np.abs(x_new - xr.numpy()).max() < 1e-12
    xf, lpf, _ = eng.mcmc(s["x"], s["sidx"], 10, 0.1, seed=42)
    assert np.abs(lpf - eng.logp(xf, s["sidx"])).max() < 1e-9 * np.abs(lpf).max()


@pytest.mark.parametrize("n,dim", [(13, 2), (29, 2), (19, 3)])
def test_ewald(n, dim):
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    rng = np.random.default_rng(1)
    L = box_length(n, dim)
    x = rng.uniform(-L, 2 * L, (5, n, dim))               # unwrapped positions, as inside the chain
    G = cg.kpoints(dim, 15 if dim == 2 else 7)
    V = cg.potential_energy(x, 10, G, L, 2.5, engine=cg.flow.get_engine(n, dim, 2, 16, 16, L))
    Vr = R.potential_energy(R.T(x), 10, G, L, 2.5).numpy()
    assert np.abs(V - Vr).max() < 1e-10 * np.abs(Vr).max()
    assert cg.Madelung(dim, 10, G) == pytest.approx(R.Madelung(dim, 10, G), rel=1e-14)


def test_wrap_and_errors():
    import coulombgas_amd as cg
    from coulombgas_amd._lib import CoulombGasError
    s = _setup(CASES[0], 2)
    eng = s["flow"].engine(13, 2, s["sp"])
    x = s["x"] + 3.7 * s["L"]
    w = eng.wrap(x)
    assert np.allclose(w, x - s["L"] * np.floor(x / s["L"]), atol=1e-12) and (w >= 0).all() and (w < s["L"]).all()
    with pytest.raises(CoulombGasError):
        cg.Engine(13, 2, 9, 16, 16, s["L"])                 # depth 9: beyond the general path's limit -> loud error
    with pytest.raises(IndexError):
        eng.set_params(s["theta"]); eng.logpsi(s["x"], np.full((2, 13), 999))
    # empty batch
    eng.set_params(s["theta"])
    assert eng.logp(np.zeros((0, 13, 2)), np.zeros((0, 13), dtype=np.int32)).shape == (0,)


# ---------------------------------------------------------------------------------------------
# derivatives: grad / Laplacian (src/logpsi.py:55-172), theta-VJP and scores, make_loss
# ---------------------------------------------------------------------------------------------
DCASES = [(5, 2, 4, 4, 2.0, 0.4, 0.2), (7, 3, 4, 4, 1.234, 0.4, 0.2), (13, 2, 16, 16, None, 0.2, 0.1)]


@pytest.mark.parametrize("case", DCASES)
def test_grad_laplacian_all_modes(case):
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    B = 2
    s = _setup(case, B, seed=11)
    v = s["rng"].standard_normal(s["x"].shape)
    logpsi = cg.make_logpsi(s["flow"], s["sp"], s["L"])
    logphi, logjacdet = cg.make_logphi_logjacdet(s["flow"], s["sp"], s["L"])
    r_logpsi = R.make_logpsi(s["rflow"], s["sp"], s["L"])
    r_logphi, r_logjacdet = R.make_logphi_logjacdet(s["rflow"], s["sp"], s["L"])
    sb = torch.as_tensor(s["sidx"].astype(np.int64))
    variants = [(dict(), dict()),
                (dict(hutchinson=True), dict(hutchinson=True)),
                (dict(hutchinson=True, logphi=logphi, logjacdet=logjacdet), dict(hutchinson=True, logphi=r_logphi, logjacdet=r_logjacdet))]
    grads = []
    for kw, rkw in variants:
        _, fn = cg.make_logpsi_grad_laplacian(logpsi, **kw)
        _, rfn = R.make_logpsi_grad_laplacian(r_logpsi, **rkw)
        g, l = fn(s["x"], s["theta"], s["sidx"], v)                # key = explicit probe array
        gr, lr = rfn(R.T(s["x"]), s["rparams"], sb, R.T(v))
        gr, lr = gr.numpy(), lr.numpy()
        assert g.shape == s["x"].shape and l.shape == (B,)
        assert np.abs(g - gr).max() < 1e-10 * max(1.0, np.abs(gr).max())
        assert np.abs(l - lr).max() < 1e-9 * max(1.0, np.abs(lr).max())
        grads.append(g)
    # tests/test_logpsi.py:151: the Hutchinson variants return the exact gradient
    assert np.abs(grads[1] - grads[0]).max() < 1e-10 * max(1.0, np.abs(grads[0]).max())
    assert np.abs(grads[2] - grads[0]).max() < 1e-10 * max(1.0, np.abs(grads[0]).max())


@pytest.mark.parametrize("n,dim,L", [(7, 3, 1.234), (13, 2, None)])
def test_kinetic_energy_identity_flow(n, dim, L):
    """Analytic KAT of the reference (tests/test_logpsi.py:79-106, tests/test_slater.py:81-112): with an identity flow
    (all flow parameters zero) -lap - sum grad^2 = (2 pi / L)^2 sum |n|^2, through the full GPU path."""
    import coulombgas_amd as cg
    L = box_length(n, dim) if L is None else L
    rng = np.random.default_rng(2)
    sp = orbitals(dim)
    flow = cg.FermiNet(2, 16, 16, L)
    theta = np.zeros(flow.engine(n, dim, sp).P)
    x = walkers(rng, 3, n, dim, L)
    sidx = state_indices(rng, 3, n, sp.shape[0])
    logpsi = cg.make_logpsi(flow, sp, L)
    for kw in (dict(), dict(hutchinson=True, logphi=1, logjacdet=1)):
        _, fn = cg.make_logpsi_grad_laplacian(logpsi, **kw)
        g, l = fn(x, theta, sidx, 7)
        kin = -l - (g ** 2).sum(axis=(-2, -1))
        ref = (2 * np.pi / L) ** 2 * (sp[sidx] ** 2).sum(axis=(-2, -1))
        assert np.abs(kin - ref).max() < 1e-9 * np.abs(ref).max()


@pytest.mark.parametrize("case", DCASES)
def test_param_vjp_and_scores(case):
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    n, dim, hs, ht = case[:4]
    B = 3
    s = _setup(case, B, seed=13)
    rng = s["rng"]
    w_re, w_im = rng.standard_normal(B), rng.standard_normal(B)
    eng = s["flow"].engine(n, dim, s["sp"])
    eng.set_params(s["theta"])
    g = eng.param_vjp(s["x"], s["sidx"], w_re, w_im, use_scores=False)      # cg_param_vjp: one weighted reverse sweep
    g2 = eng.param_vjp(s["x"], s["sidx"], w_re, w_im)                        # cg_scores_compute + cg_scores_vjp (resident scores)
    assert np.abs(g2 - g).max() < 1e-11 * max(1.0, np.abs(g).max())
    r_logpsi = R.make_logpsi(s["rflow"], s["sp"], s["L"])
    sb = torch.as_tensor(s["sidx"].astype(np.int64))
    lpt = lambda xb, th, sbb: r_logpsi(xb, R.flow_unravel(th, 2, hs, ht, dim), sbb)

    def S(th):
        out = torch.stack([lpt(R.T(s["x"][b]), th, sb[b]) for b in range(B)])
        return (R.T(w_re) * out[:, 0] + R.T(w_im) * out[:, 1]).sum()
    gr = torch.func.grad(S)(R.T(s["theta"])).numpy()
    assert np.abs(g - gr).max() < 1e-10 * max(1.0, np.abs(gr).max())
    qs = cg.make_quantum_score(cg.make_logpsi(s["flow"], s["sp"], s["L"]))(s["x"], s["theta"], s["sidx"])
    qr = R.make_quantum_score(lpt)(R.T(s["x"]), R.T(s["theta"]), sb).numpy()
    flat = np.concatenate([qs[nm][lf].reshape(B, -1) for nm, lf, _ in cg.flow.ravel_order(2, hs, ht, dim)], axis=1)
    assert np.abs(flat - qr).max() < 1e-10 * max(1.0, np.abs(qr).max())


def test_make_loss_against_oracle():
    """src/VMC.py:31-80 + main.py:277-278 on one device: observables, complex clip, loss values and theta-gradients."""
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    case = (13, 2, 16, 16, None, 0.1, 0.05)
    n, dim, hs, ht = case[:4]
    B = 6
    s = _setup(case, B, seed=17)
    rng = s["rng"]
    L, rs, kappa, beta = s["L"], 10.0, 10, 1 / (4 * 0.15)
    G = cg.kpoints(dim, 15)
    Vconst = n * rs / L * cg.Madelung(dim, kappa, G)
    logp_states = -5.0 + rng.standard_normal(B)
    log_prob = lambda params_van, state_indices: logp_states
    v = rng.standard_normal(s["x"].shape)
    logpsi_novmap = cg.make_logpsi(s["flow"], s["sp"], L)
    logphi, logjacdet = cg.make_logphi_logjacdet(s["flow"], s["sp"], L)
    logpsi, lgl = cg.make_logpsi_grad_laplacian(logpsi_novmap, hutchinson=True, logphi=logphi, logjacdet=logjacdet)
    obs_fn = cg.make_loss(log_prob, logpsi, lgl, kappa, G, L, rs, Vconst, beta)
    obs, closs, qloss = obs_fn(None, s["theta"], s["sidx"], s["x"], v)
    qv = qloss(s["theta"])
    cv = closs(None)
    g_grad, g_score = qloss.grad(s["theta"], as_pytree=False)
    # oracle
    r_logpsi = R.make_logpsi(s["rflow"], s["sp"], L)
    r_logphi, r_logjacdet = R.make_logphi_logjacdet(s["rflow"], s["sp"], L)
    _, rfn = R.make_logpsi_grad_laplacian(r_logpsi, hutchinson=True, logphi=r_logphi, logjacdet=r_logjacdet)
    sb = torch.as_tensor(s["sidx"].astype(np.int64))
    gr, lr = rfn(R.T(s["x"]), s["rparams"], sb, R.T(v))
    pot = R.potential_energy(R.T(s["x"]), kappa, G, L, rs)
    robs, Eloc, Floc, Fc, Ec = R.observables_and_weights(R.T(logp_states), gr, lr, pot, Vconst, beta)
    for k in robs:
        assert obs[k] == pytest.approx(float(robs[k]), rel=1e-9, abs=1e-9), k
    lpt = lambda xb, th, sbb: r_logpsi(xb, R.flow_unravel(th, 2, hs, ht, dim), sbb)
    v0, v1, dg, ds = R.quantum_loss_and_grads(lpt, R.T(s["theta"]), R.T(s["x"]), sb, Ec)
    assert qv[0] == pytest.approx(float(v0), rel=1e-9, abs=1e-9) and qv[1] == pytest.approx(float(v1), rel=1e-10)
    # loss gradient = sum_b E_b * d log Psi_b (measured: 6e-14 of max(1, |dg|)); bar of the north star: 1e-8
    print("loss-gradient error %.2e (relative to max(1, |dg|_max) = %.3g)" % (np.abs(g_grad - dg.numpy()).max() / max(1.0, np.abs(dg.numpy()).max()), max(1.0, np.abs(dg.numpy()).max())))
    assert np.abs(g_grad - dg.numpy()).max() < 1e-10 * max(1.0, np.abs(dg.numpy()).max())
    assert np.abs(g_score - ds.numpy()).max() < 1e-10 * max(1.0, np.abs(ds.numpy()).max())
    assert cv[0] == pytest.approx(float((R.T(logp_states) * Fc).mean()), rel=1e-10)
    assert cv[1] == pytest.approx(float(logp_states.mean()), rel=1e-12)
    # main.py:295-298 final-step combination stays host algebra: grad - <E> score
    final = g_grad - obs["E_mean"] * g_score
    assert np.isfinite(final).all()


def test_rccl_allreduce_world1():
    """RCCL plumbing (dlopen, communicator on the library's stream, sum + 1/N scale) with a 1-rank communicator."""
    from coulombgas_amd.comm import RcclComm
    s = _setup(CASES[0], 1)
    eng = s["flow"].engine(13, 2, s["sp"])
    comm = RcclComm(eng, 0, 1)
    a = np.arange(37, dtype=np.float64) * 0.25
    assert np.array_equal(comm.pmean(a), a)
    assert comm.pmean(3.5) == 3.5
    comm.close()


# ---------------------------------------------------------------------------------------------
# largest published sizes (BASELINE configs 4 and 5): checked against the C oracle (dense forward mode, AD-free),
# the torch.func oracle being too slow there
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,Emax,B", [(29, 25, 6), (57, 49, 3)])
def test_large_n_against_c_oracle(n, Emax, B):
    import ctypes as C
    import coulombgas_amd as cg
    from coulombgas_amd.build import build_oracle
    lib = C.CDLL(build_oracle())
    lib.cgo_mcmc.restype = C.c_double
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    dim, L = 2, box_length(n, 2)
    rng = np.random.default_rng(n)
    sp = orbitals(2, Emax)
    theta = np.load(GOLDEN_DIR + "/shipped_n%d_rs10.npz" % n)["theta"]          # trained flow of the shipped run
    x = walkers(rng, B, n, dim, L)
    sidx = state_indices(rng, B, n, sp.shape[0])
    flow = cg.FermiNet(2, 16, 16, L)
    eng = flow.engine(n, dim, sp)
    eng.set_params(theta)
    out = np.zeros((B, 3))
    lib.cgo_logpsi(n, dim, 2, 16, 16, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(x), B, p(out))
    lphi, hld = eng.logphi_logjacdet(x, sidx)
    assert np.abs(lphi[:, 0] - out[:, 0]).max() < 1e-10 * np.abs(out[:, 0]).max()
    assert np.abs(np.angle(np.exp(1j * (lphi[:, 1] - out[:, 1])))).max() < 1e-10
    assert np.abs(hld - out[:, 2]).max() < 1e-11
    steps = 3
    noise = rng.standard_normal((steps, B, n, dim)); unif = rng.uniform(size=(steps, B))
    xg, lpg, nacc = eng.mcmc(x, sidx, steps, 0.1, noise=noise, unif=unif)
    xc = x.copy(); lpc = np.zeros(B)
    rate = lib.cgo_mcmc(n, dim, 2, 16, 16, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(xc), B, steps, C.c_double(0.1),
                        p(noise), p(unif), p(lpc))
    assert nacc / (steps * B) == pytest.approx(rate, abs=1e-15)
    assert np.abs(xg - xc).max() < 1e-12 and np.abs(lpg - lpc).max() < 1e-9 * max(1.0, np.abs(lpc).max())


@pytest.mark.parametrize("n,Emax,B,ws", [(29, 25, 5, 1.5), (57, 49, 3, 1.2), (20, 25, 4, 2.0), (33, 25, 3, 1.5), (45, 49, 2, 1.2), (25, 25, 3, 1.5),
                                         (32, 25, 3, 1.5), (40, 25, 2, 1.3), (36, 25, 2, 2.0), (56, 49, 2, 1.2)])
def test_large_n_with_row_exchanges_against_c_oracle(n, Emax, B, ws):
    """The concurrent, flag-decoupled LU pair (cg_blocked_lu_dual2: rows never move, GEMM-only helpers) with a flow far from the identity
    (random weights of standard deviation ws: the Jacobian is not diagonally dominant, the threshold test fails and the pivots leave the
    diagonal -- at N > 64 also the register slot the panel code reads, so that a lane's two rows trade slots); ragged last panels and
    blocks (N = 40, 66, 72), one row per lane up to its limit (N = 64), two rows per lane right above it (N = 66 ... 114)."""
    import ctypes as C
    import coulombgas_amd as cg
    from coulombgas_amd.build import build_oracle
    lib = C.CDLL(build_oracle())
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    dim, L = 2, box_length(n, 2)
    rng = np.random.default_rng(100 + n)
    sp = orbitals(2, Emax)
    theta = flow_theta(rng, 2, 16, 16, dim, ws, 0.2)
    x = walkers(rng, B, n, dim, L)
    sidx = state_indices(rng, B, n, sp.shape[0])
    flow = cg.FermiNet(2, 16, 16, L)
    eng = flow.engine(n, dim, sp)
    eng.set_params(theta)
    out = np.zeros((B, 3))
    lib.cgo_logpsi(n, dim, 2, 16, 16, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(x), B, p(out))
    lphi, hld = eng.logphi_logjacdet(x, sidx)
    assert np.abs(lphi[:, 0] - out[:, 0]).max() < 1e-9 * np.abs(out[:, 0]).max()
    assert np.abs(np.angle(np.exp(1j * (lphi[:, 1] - out[:, 1])))).max() < 1e-9
    assert np.abs(hld - out[:, 2]).max() < 1e-9 * max(1.0, np.abs(out[:, 2]).max())
    J = np.asarray(eng.flow_jacobian(x)).reshape(B, n * dim, n * dim)
    d = np.abs(np.diagonal(J, axis1=1, axis2=2)); off = np.abs(J).sum(-1) - d
    assert (off > d).any()                              # the case does exercise pivoting: J is not diagonally dominant


def test_beyond_the_lds_limit_runs_on_the_general_path():
    """Maximum size: at n = 72 (N = 144) J alone is 162 KB, more than the 160 KB of LDS; the same depth-2 model then runs
    on the general path (HBM workspace) with identical results (C oracle) instead of failing."""
    import ctypes as C
    import coulombgas_amd as cg
    from coulombgas_amd.build import build_oracle
    lib = C.CDLL(build_oracle())
    p = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    n, dim, B = 72, 2, 2
    L = box_length(n, dim)
    rng = np.random.default_rng(n)
    sp = orbitals(2, 49)
    theta = np.load(GOLDEN_DIR + "/shipped_n57_rs10.npz")["theta"]
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, sp.shape[0])
    eng = cg.FermiNet(2, 16, 16, L).engine(n, dim, sp)
    eng.set_params(theta)
    assert eng.launch_info()["fast"] == 0
    out = np.zeros((B, 3))
    lib.cgo_logpsi(n, dim, 2, 16, 16, C.c_double(L), p(theta), p(sp), sp.shape[0], p(sidx), p(x), B, p(out))
    lphi, hld = eng.logphi_logjacdet(x, sidx)
    assert np.abs(lphi[:, 0] - out[:, 0]).max() < 1e-10 * np.abs(out[:, 0]).max()
    assert np.abs(np.angle(np.exp(1j * (lphi[:, 1] - out[:, 1])))).max() < 1e-10
    assert np.abs(hld - out[:, 2]).max() < 1e-11
    xg, lpg, nacc = eng.mcmc(x, sidx, 2, 0.1, seed=5)
    assert np.isfinite(xg).all() and np.abs(lpg - eng.logp(xg, sidx)).max() < 1e-9 * np.abs(lpg).max()


# ---------------------------------------------------------------------------------------------
# general-depth path: the depth-3 networks of the reference's own tests (tests/test_flow.py:42, tests/test_logpsi.py:29)
# ---------------------------------------------------------------------------------------------
GCASES = [(7, 3, 3, 16, 16, 1.234), (5, 2, 4, 8, 4, 2.0), (13, 2, 3, 16, 16, None)]


@pytest.mark.parametrize("n,dim,depth,hs,ht,L", GCASES)
def test_general_depth_against_oracle(n, dim, depth, hs, ht, L):
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    from torch.func import jacfwd
    L = box_length(n, dim) if L is None else L
    rng = np.random.default_rng(depth)
    sp = orbitals(dim)
    B = 2
    theta = flow_theta(rng, depth, hs, ht, dim, 0.3, 0.1)
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, sp.shape[0]); v = rng.standard_normal(x.shape)
    flow = cg.FermiNet(depth, hs, ht, L)
    rflow = R.FermiNet(depth, hs, ht, L); rparams = R.flow_unravel(R.T(theta), depth, hs, ht, dim)
    z = flow.apply(theta, None, x)
    J = flow.engine(n, dim).flow_jacobian(x)
    logpsi = cg.make_logpsi(flow, sp, L)
    out = logpsi(x, theta, sidx)
    r_logpsi = R.make_logpsi(rflow, sp, L)
    sb = torch.as_tensor(sidx.astype(np.int64))
    for b in range(B):
        xb = R.T(x[b])
        assert np.abs(z[b] - rflow.apply(rparams, xb).numpy()).max() < 1e-12 * max(1.0, np.abs(z[b]).max())
        Jr = jacfwd(lambda xf: rflow.apply(rparams, xf.reshape(n, dim)).reshape(-1))(xb.reshape(-1)).numpy()
        assert np.abs(J[b] - Jr).max() < 1e-12
        ref = r_logpsi(xb, rparams, sb[b]).numpy()
        assert abs(out[b, 0] - ref[0]) < 1e-11 * max(1.0, abs(ref[0]))
        assert abs(np.angle(np.exp(1j * (out[b, 1] - ref[1])))) < 1e-11
    # flow equivariances of tests/test_flow.py:25,32,38 through the GPU path
    image = rng.integers(-5, 6, size=(n, dim)) * L
    assert np.allclose(flow.apply(theta, None, x[0] + image), z[0] + image, atol=1e-9)
    shift = rng.standard_normal(dim)
    assert np.allclose(flow.apply(theta, None, x[0] + shift), z[0] + shift, atol=1e-9)
    perm = rng.permutation(n)
    assert np.allclose(flow.apply(theta, None, x[0][perm]), z[0][perm], atol=1e-12)
    if n <= 7:
        r_logphi, r_logjacdet = R.make_logphi_logjacdet(rflow, sp, L)
        logphi, logjacdet = cg.make_logphi_logjacdet(flow, sp, L)
        for kw, rkw in ((dict(), dict()), (dict(hutchinson=True, logphi=logphi, logjacdet=logjacdet), dict(hutchinson=True, logphi=r_logphi, logjacdet=r_logjacdet))):
            g, l = cg.make_logpsi_grad_laplacian(logpsi, **kw)[1](x, theta, sidx, v)
            gr, lr = R.make_logpsi_grad_laplacian(r_logpsi, **rkw)[1](R.T(x), rparams, sb, R.T(v))
            assert np.abs(g - gr.numpy()).max() < 1e-10 * np.abs(gr.numpy()).max()
            assert np.abs(l - lr.numpy()).max() < 1e-9 * np.abs(lr.numpy()).max()
    # chain with supplied noise
    steps = 4
    noise = rng.standard_normal((steps,) + x.shape); unif = rng.uniform(size=(steps, B))
    logp = cg.make_logp(logpsi)
    x_new, rate = cg.mcmc(logp.bind(theta, sidx), x, 0, steps, 0.1, noise=noise, unif=unif)
    r_logp = R.make_logp(r_logpsi)
    xr, _, rate_r = R.mcmc(lambda xx: r_logp(xx, rparams, sb), R.T(x), R.T(noise), R.T(unif), steps, 0.1)
    assert rate == pytest.approx(rate_r, abs=1e-15) and np.abs(x_new - xr.numpy()).max() < 1e-12
    # theta-VJP and per-sample scores of the general path (dual-number reverse passes, cg_generic.hpp) vs jacrev of the oracle
    if n <= 7:
        w_re, w_im = rng.standard_normal(B), rng.standard_normal(B)
        eng = flow.engine(n, dim, sp); eng.set_params(theta)
        g = eng.param_vjp(x, sidx, w_re, w_im)
        lpt = lambda xb, th, sbb: r_logpsi(xb, R.flow_unravel(th, depth, hs, ht, dim), sbb)

        def S(th):
            o = torch.stack([lpt(R.T(x[b]), th, sb[b]) for b in range(B)])
            return (R.T(w_re) * o[:, 0] + R.T(w_im) * o[:, 1]).sum()
        gr = torch.func.grad(S)(R.T(theta)).numpy()
        assert np.abs(g - gr).max() < 1e-10 * max(1.0, np.abs(gr).max())
        qs = eng.quantum_score(x, sidx)
        qr = R.make_quantum_score(lpt)(R.T(x), R.T(theta), sb).numpy()
        assert np.abs(qs - qr).max() < 1e-10 * max(1.0, np.abs(qr).max())


@pytest.mark.parametrize("case,depth", [((13, 2, 16, 16, None, 0.2, 0.1), 2), ((5, 2, 8, 4, 2.0, 0.4, 0.2), 3)])
def test_quantum_fisher_and_sr_update(case, depth):
    """cg_quantum_fisher (scores + f64 MFMA SYRK on the device) and the hybrid SR update (src/sr.py:56-122) against the
    oracle: Re(S^H S)/B and mean(S) from jacrev scores, centring, damped solve, norm clip."""
    import coulombgas_amd as cg
    from oracle import cg_ref as R
    n, dim, hs, ht, L, ws, bs = case
    L = box_length(n, dim) if L is None else L
    rng = np.random.default_rng(29)
    sp = orbitals(dim)
    B = 9
    theta = flow_theta(rng, depth, hs, ht, dim, ws, bs)
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, sp.shape[0])
    flow = cg.FermiNet(depth, hs, ht, L)
    logpsi = cg.make_logpsi(flow, sp, L)
    eng = flow.engine(n, dim, sp); eng.set_params(theta)
    F, sm = eng.quantum_fisher(x, sidx)
    rflow = R.FermiNet(depth, hs, ht, L)
    r_logpsi = R.make_logpsi(rflow, sp, L)
    lpt = lambda xb, th, sbb: r_logpsi(xb, R.flow_unravel(th, depth, hs, ht, dim), sbb)
    qs = R.make_quantum_score(lpt)(R.T(x), R.T(theta), torch.as_tensor(sidx.astype(np.int64))).numpy()
    Fr = (qs.conj().T @ qs).real / B
    assert np.abs(F - Fr).max() < 1e-10 * np.abs(Fr).max() and np.abs(F - F.T).max() == 0.0
    assert np.abs(sm - qs.mean(axis=0)).max() < 1e-10 * np.abs(qs).max()
    # classical Fisher matrix on the device: real SYRK; blocked Cholesky of the damped matrices
    cs = rng.standard_normal((37, 53))
    Fc = eng.fisher_real(cs)
    assert np.abs(Fc - cs.T @ cs / 37).max() < 1e-13 * np.abs(Fc).max() and np.abs(Fc - Fc.T).max() == 0.0
    for M, mc in ((Fc + 1e-3 * np.eye(53), cs.mean(axis=0) + 0j), (Fr + 1e-3 * np.eye(theta.size), qs.mean(axis=0))):
        Lg = eng.cholesky(M)
        Lr = np.linalg.cholesky(M)
        assert np.abs(Lg - Lr).max() < 1e-9 * np.abs(Lr).max() and np.abs(Lg @ Lg.T - M).max() < 1e-12 * np.abs(M).max()
        # the whole damped solve on the device (shift, factor, both substitutions) against LAPACK; the input stays intact
        rhs = rng.standard_normal(M.shape[0]); M0 = M.copy()
        xg = eng.spd_solve(M, rhs, 2e-3)
        xr = np.linalg.solve(M + 2e-3 * np.eye(M.shape[0]), rhs)
        assert np.abs(xg - xr).max() < 1e-9 * np.abs(xr).max() and np.array_equal(M, M0)
        xg = eng.spd_solve(M, rhs, 2e-3, center=mc)            # centred: the covariance of the scores (src/sr.py:88)
        xr = np.linalg.solve(M - (mc.conj()[:, None] * mc).real + 2e-3 * np.eye(M.shape[0]), rhs)
        assert np.abs(xg - xr).max() < 1e-8 * np.abs(xr).max() and np.array_equal(M, M0)
    from coulombgas_amd._lib import CoulombGasError
    with pytest.raises(CoulombGasError):
        eng.cholesky(np.diag([1.0, -1.0, 2.0]))
    with pytest.raises(CoulombGasError):
        eng.spd_solve(np.diag([1.0, -1.0, 2.0]), np.ones(3))
    fishers_fn, opt = cg.hybrid_fisher_sr(None, cg.make_quantum_score(logpsi), 1e-3, 1e-3)
    params_flow = flow.unravel(theta, dim)
    cf, qf, qm = fishers_fn(None, params_flow, sidx, x)
    g = rng.standard_normal(theta.size)
    (_, u_flow), _ = opt.update((None, flow.unravel(g, dim)), opt.init(None), (cf, qf, qm))
    ref = R.hybrid_fisher_sr_update(np.zeros((B, 1)) + 1.0, qs, np.ones(1), g, 1e-3, 1e-3)[4]
    assert np.abs(flow.ravel(u_flow, dim) - ref).max() < 1e-6 * np.abs(ref).max()


def test_training_lowers_the_energy():
    """End to end through the C-ABI (sampling call, Hutchinson-split local energies, theta-VJP, quantum Fisher matrix, SR
    step; main.py:216-384 mirror): starting from an identity-like flow the variational energy of the n=5 ground state
    goes down by many standard errors within a few epochs."""
    import coulombgas_amd as cg
    n, dim, rs = 5, 2, 5.0
    L = box_length(n, dim)
    sp = orbitals(dim)
    flow = cg.FermiNet(2, 16, 16, L)
    p0 = flow.init(1, np.zeros((n, dim)))
    samp = cg.GroundStateSampler(n, sp.shape[0])
    rows = []
    pv, pf, _ = cg.train(flow, p0, sp, n, dim, L, rs=rs, beta=1 / (4 * 0.15), batch=4096, epochs=14, sampler=samp,
                         log_prob=samp.log_prob, sr=(1e-3, 1e-3), mc_therm=5, mc_steps=30, seed=3, log=rows.append)
    v = np.array([[float(t) for t in r.split()] for r in rows])
    E, Estd = v[:, 3], v[:, 4]
    assert np.isfinite(v).all() and (v[:, -1] > 0.2).all()
    assert np.median(E[-4:]) < E[0] - 4 * Estd[0], (E, Estd)


def test_finite_temperature_training_on_device():
    """The whole finite-temperature loop of main.py:216-384 with BOTH parameter sets trained and nothing on the host but O(P)
    vectors: the Transformer samples the occupations on the GPU (cg_van_sample), log p, the classical scores / their weighted
    VJP (jax.jacrev(classical_lossfn), main.py:277) and the classical Fisher matrix (src/sr.py:77-79) come from the device's
    reverse pass, the flow side as in test_training_lowers_the_energy.  The free energy F = <log p / beta + E_loc> of n = 5
    electrons at Theta = 0.3 goes down by many standard errors, the entropy stays positive, both parameter sets move."""
    import coulombgas_amd as cg
    n, dim, rs, Theta = 5, 2, 5.0, 0.3
    L = box_length(n, dim)
    sp = orbitals(dim)[:21]
    M = sp.shape[0]
    flow = cg.FermiNet(2, 16, 16, L)
    p0 = flow.init(1, np.zeros((n, dim)))
    van = cg.Transformer(M, 2, 16, 4, 32)
    pv0 = van.init(5, sp[:n])
    sampler, log_prob = cg.make_autoregressive_sampler(van, sp, n, M)          # train() attaches its engine
    rows = []
    pv, pf, _ = cg.train(flow, p0, sp, n, dim, L, rs=rs, beta=1 / (4 * Theta), batch=4096, epochs=12, sampler=sampler,
                         log_prob=log_prob, params_van=pv0, sr=(1e-3, 1e-3), mc_therm=5, mc_steps=30, seed=4, log=rows.append)
    v = np.array([[float(t) for t in r.split()] for r in rows])
    F, Fstd, S = v[:, 1], v[:, 2], v[:, 9]
    assert np.isfinite(v).all() and (v[:, -1] > 0.2).all()
    assert np.median(F[-3:]) < F[0] - 4 * Fstd[0], (F, Fstd)
    assert (S > 0).all()
    dv = max(np.abs(pv[m][l] - pv0[m][l]).max() for m in pv0 for l in pv0[m])
    df = max(np.abs(np.asarray(pf[m][l]) - np.asarray(p0[m][l])).max() for m in p0 for l in p0[m])
    assert dv > 0 and df > 0


def test_full_size_properties():
    """BASELINE config 2/3 at full size (n=13, dim=2, 8192 walkers) through size-independent properties: lattice /
    translation / permutation invariance of |Psi|^2 (tests/test_logpsi.py:45,54,72,77), block rows of J summing to the
    identity (translation equivariance of the flow, tests/test_flow.py:32), the chain's bookkeeping, determinism, Ewald
    invariances, linearity of the theta-VJP in its weights, and the score matrix reproducing the VJP."""
    import coulombgas_amd as cg
    n, dim, B = 13, 2, 8192
    L = box_length(n, dim)
    rng = np.random.default_rng(8192)
    sp = orbitals(dim)
    theta = flow_theta(rng, 2, 16, 16, dim, 0.2, 0.1)
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, sp.shape[0])
    eng = cg.FermiNet(2, 16, 16, L).engine(n, dim, sp); eng.set_params(theta)
    lp = eng.logp(x, sidx)
    assert np.isfinite(lp).all()
    image = rng.integers(-3, 4, size=(B, n, dim)) * L
    assert np.abs(eng.logp(x + image, sidx) - lp).max() < 1e-8 * np.abs(lp).max()
    shift = rng.standard_normal((B, 1, dim))
    assert np.abs(eng.logp(x + shift, sidx) - lp).max() < 1e-8 * np.abs(lp).max()
    perm = rng.permutation(n)
    assert np.abs(eng.logp(x[:, perm], sidx) - lp).max() < 1e-8 * np.abs(lp).max()
    J = eng.flow_jacobian(x[:512]).reshape(512, n, dim, n, dim)
    assert np.abs(J.sum(axis=3) - np.eye(dim)).max() < 1e-12
    x1, lp1, na1 = eng.mcmc(x, sidx, 10, 0.1, seed=77)
    x2, lp2, na2 = eng.mcmc(x, sidx, 10, 0.1, seed=77)
    assert np.array_equal(x1, x2) and np.array_equal(lp1, lp2) and na1 == na2 and 0.3 < na1 / (10 * B) < 0.9
    assert np.abs(lp1 - eng.logp(x1, sidx)).max() < 1e-9 * np.abs(lp1).max()
    eng.set_ewald(10, cg.kpoints(dim, 15), 10.0)
    V = eng.ewald(x)
    assert np.abs(eng.ewald(x + image) - V).max() < 1e-9 * np.abs(V).max() and np.abs(eng.ewald(x[:, perm]) - V).max() < 1e-10 * np.abs(V).max()
    w1, w2, w3, w4 = (rng.standard_normal(B) for _ in range(4))
    ga = eng.param_vjp(x, sidx, w1, w2, use_scores=False); gb = eng.param_vjp(x, sidx, w3, w4, use_scores=False)
    gc = eng.param_vjp(x, sidx, 2.0 * w1 - w3, 2.0 * w2 - w4, use_scores=False)
    assert np.abs(gc - (2.0 * ga - gb)).max() < 1e-9 * np.abs(ga).max()
    assert np.abs(eng.param_vjp(x, sidx, w1, w2) - ga).max() < 1e-9 * np.abs(ga).max()      # from the resident score matrix
    F, sm = eng.quantum_fisher(x, sidx)
    assert np.abs(F - F.T).max() == 0.0 and np.linalg.eigvalsh(F).min() > -1e-9 * np.abs(F).max()
    assert np.abs(eng.param_vjp(x, sidx, np.full(B, 1.0 / B), np.zeros(B)) - sm.real).max() < 1e-10 * np.abs(sm).max()
    # cg_grad_laplacian at the full batch (BASELINE config 3: one walker per workgroup, 8192 workgroups): deterministic, independent of
    # the batch a walker is evaluated in, invariant under lattice translations, the two Hutchinson modes share their gradient, and a
    # random sample of walkers against the oracle
    from oracle import cg_ref as R
    v = rng.standard_normal(x.shape)
    g2, l2 = eng.grad_laplacian(x, sidx, 2, v)
    g2b, l2b = eng.grad_laplacian(x, sidx, 2, v)
    assert np.isfinite(l2).all() and np.isfinite(g2).all() and np.array_equal(g2, g2b) and np.array_equal(l2, l2b)
    pick = rng.choice(B, 24, replace=False)
    gs, ls = eng.grad_laplacian(x[pick], sidx[pick], 2, v[pick])
    assert np.array_equal(gs, g2[pick]) and np.array_equal(ls, l2[pick])
    gi, li = eng.grad_laplacian(x + image, sidx, 2, v)
    assert np.abs(gi - g2).max() < 1e-7 * np.abs(g2).max() and np.abs(li - l2).max() < 1e-6 * np.abs(l2).max()
    g1, _ = eng.grad_laplacian(x, sidx, 1, v)
    assert np.array_equal(g1, g2)
    rflow = R.FermiNet(2, 16, 16, L)
    rparams = R.flow_unravel(R.T(theta), 2, 16, 16, dim)
    r_logpsi = R.make_logpsi(rflow, sp, L)
    r_logphi, r_logjacdet = R.make_logphi_logjacdet(rflow, sp, L)
    _, rfn = R.make_logpsi_grad_laplacian(r_logpsi, hutchinson=True, logphi=r_logphi, logjacdet=r_logjacdet)
    gr, lr = rfn(R.T(x[pick]), rparams, torch.as_tensor(sidx[pick].astype(np.int64)), R.T(v[pick]))
    assert np.abs(gs - gr.numpy()).max() < 1e-10 * np.abs(gr.numpy()).max()
    assert np.abs(ls - lr.numpy()).max() < 1e-9 * np.abs(lr.numpy()).max()


def _load_van(name):
    z = np.load(GOLDEN_DIR + "/" + name)
    pv = {}
    for k in z.files:
        if "|" in k:
            m, l = k.split("|"); pv.setdefault(m, {})[l] = z[k]
    return pv


@pytest.mark.parametrize("n,rs,Emax,fixture,van,B,rounds,tol", [
    (29, 10.0, 25, "shipped_n29_rs10.npz", "shipped_n29_rs10_van.npz", 512, 8, 0.005),
    (29, 1.0, 25, "shipped_n29_rs1.npz", "shipped_n29_rs1_van.npz", 512, 16, 0.005),      # BASELINE config 4 (rs = 1)
    # BASELINE config 5 (Emax = 49).  Measured with this build: E -9.50 +- 0.004 vs published -9.5518 (0.53 %), K 1.72 vs 1.555,
    # V -11.22 vs -11.107, acceptance 0.23 vs 0.257 -- stable over 40 sampling rounds and identical in the oracle (the n = 57
    # golden vectors pin the HIP path to it at 1e-10): the v1-formulas-vs-shipped-data gap of SURVEY App. B6, larger at n = 57.
    (57, 10.0, 49, "shipped_n57_rs10.npz", "shipped_n57_rs10_van.npz", 256, 12, 0.01),
])
def test_shipped_model_reproduces_published_energies(n, rs, Emax, fixture, van, B, rounds, tol):
    """Statistical end-to-end KATs with the shipped final models (Transformer density matrix + trained flow) of three
    production runs: host sampler -> GPU Metropolis chains -> GPU local energies (Hutchinson-split, device-resident step)
    reproduce the E and F of the last published data.txt row within `tol` (0.5 %; 1 % at n = 57) or 4 standard errors of this
    sample, whichever is larger (v1 formulas + shipped parameters agree with the published E, F to ~0.15 %; K and V separately are only
    loose, App. B6).  The published row comes from 8192 walkers."""
    import coulombgas_amd as cg
    dim = 2
    L, beta = box_length(n, dim), 1 / (4 * 0.15)
    sp = orbitals(2, Emax)
    fix = np.load(GOLDEN_DIR + "/" + fixture)
    pv = _load_van(van)
    vanm = cg.Transformer(sp.shape[0], 2, 16, 4, 32)
    flow = cg.FermiNet(2, 16, 16, L)
    # the density matrix samples and evaluates on the same GPU: state indices and log-probabilities never visit the host
    sampler, log_prob = cg.make_autoregressive_sampler(vanm, sp, n, sp.shape[0], engine=flow.engine(n, dim, sp))
    pf = flow.unravel(fix["theta"], dim)
    logpsi0 = cg.make_logpsi(flow, sp, L)
    logphi, logjac = cg.make_logphi_logjacdet(flow, sp, L)
    logp = cg.make_logp(logpsi0)
    logpsi, lgl = cg.make_logpsi_grad_laplacian(logpsi0, hutchinson=True, logphi=logphi, logjacdet=logjac)
    G = cg.kpoints(dim, 15)
    Vconst = n * rs / L * cg.Madelung(dim, 10, G)
    loss = cg.make_loss(log_prob, logpsi, lgl, 10, G, L, rs, Vconst, beta)
    from coulombgas_amd.engine import DeviceArray
    x = DeviceArray.from_numpy(flow.engine(n, dim, sp), fix["x"][:B])          # shipped (already thermalised) walkers, kept in HBM
    key = np.random.SeedSequence(5)
    acc_m = {k: [] for k in ("E_mean", "E2_mean", "F_mean", "F2_mean")}
    for it in range(rounds + 2):
        key, sidx, x, acc = cg.sample_stateindices_and_x(key, sampler, pv, logp, x, pf, 50, 0.1, L)
        if it >= 2:
            obs, _, _ = loss(pv, pf, sidx, x, key)
            for k in acc_m:
                acc_m[k].append(obs[k])
    row = fix["data_row"]                                          # epoch F F_std E E_std K K_std V V_std S S_std accept
    ns = B * rounds
    E, F = np.mean(acc_m["E_mean"]), np.mean(acc_m["F_mean"])
    sE = np.sqrt(max(np.mean(acc_m["E2_mean"]) - E * E, 0.0) / ns); sF = np.sqrt(max(np.mean(acc_m["F2_mean"]) - F * F, 0.0) / ns)
    E, F, sE, sF = E / rs ** 2, F / rs ** 2, sE / rs ** 2, sF / rs ** 2
    print("shipped n=%d rs=%g: E %.4f +- %.4f (published %.4f)  F %.4f +- %.4f (published %.4f)  accept %.3f (published %.3f)"
          % (n, rs, E, sE, row[3], F, sF, row[1], acc, row[11]))
    assert 0.15 < acc < 0.65
    assert abs(E - row[3]) < max(tol * abs(row[3]), 4 * sE), (E, sE, row[3])
    assert abs(F - row[1]) < max(tol * abs(row[1]), 4 * sF), (F, sF, row[1])


def test_epoch1_row_of_the_production_run():
    """The first row of data/n_29_..._rs_10.0/data.txt (flow ~ identity: freshly initialised N(0, 0.01^2) weights, pretrained
    Transformer): K 0.6197, V -4.8706, acceptance 0.5369 at the published proposal width -- the cleanest reference-held
    known answer for sampler + Slater determinant + Ewald sum, here through the HIP path at BASELINE config 4's
    per-GPU batch: Transformer sampler (host) -> 10 thermalisation rounds -> cg_mcmc / cg_grad_laplacian / cg_ewald."""
    import coulombgas_amd as cg
    n, dim, rs, B = 29, 2, 10.0, 2048
    L, beta = box_length(n, dim), 1 / (4 * 0.15)
    sp = orbitals(2, 25)
    pv = _load_van("pretrained_van_n29.npz")
    vanm = cg.Transformer(sp.shape[0], 2, 16, 4, 32)
    flow = cg.FermiNet(2, 16, 16, L)
    sampler, log_prob = cg.make_autoregressive_sampler(vanm, sp, n, sp.shape[0], engine=flow.engine(n, dim, sp))
    pf = flow.init(7, np.zeros((n, dim)))                          # src/flow.py:6-14: N(0, 0.01^2) weights, zero biases
    logpsi0 = cg.make_logpsi(flow, sp, L)
    logphi, logjac = cg.make_logphi_logjacdet(flow, sp, L)
    logp = cg.make_logp(logpsi0)
    logpsi, lgl = cg.make_logpsi_grad_laplacian(logpsi0, hutchinson=True, logphi=logphi, logjacdet=logjac)
    G = cg.kpoints(dim, 15)
    Vconst = n * rs / L * cg.Madelung(dim, 10, G)
    loss = cg.make_loss(log_prob, logpsi, lgl, 10, G, L, rs, Vconst, beta)
    x = np.random.default_rng(11).uniform(0, L, (B, n, dim))       # main.py:236
    key = np.random.SeedSequence(12)
    K, V, A = [], [], []
    for it in range(10 + 3):                                       # main.py:241-246 thermalisation, then three "epochs" of sampling
        key, sidx, x, acc = cg.sample_stateindices_and_x(key, sampler, pv, logp, x, pf, 50, 0.1, L)
        if it >= 10:
            obs, _, _ = loss(pv, pf, sidx, x, key)
            K.append(obs["K_mean"] / rs ** 2); V.append(obs["V_mean"] / rs ** 2); A.append(acc)
    row = np.load(GOLDEN_DIR + "/shipped_n29_rs10.npz")["data_row_epoch1"]
    print("epoch-1 KAT n=29 rs=10: K %.5f (published %.5f)  V %.4f (published %.4f)  accept %.4f (published %.4f)"
          % (np.mean(K), row[5], np.mean(V), row[7], np.mean(A), row[11]))
    assert abs(np.mean(K) - row[5]) < 0.005 * row[5]
    assert abs(np.mean(V) - row[7]) < 0.005 * abs(row[7])
    assert abs(np.mean(A) - row[11]) < 0.01


# ---------------------------------------------------------------------------------------------
# device-resident optimisation step (K8 eloc_reduce, clip weights, scores, Fisher, solve all in HBM)
# ---------------------------------------------------------------------------------------------
def test_local_energy_kernels_against_oracle():
    """cg_local_energy / cg_abs_dev / cg_clip_weights (src/VMC.py:39-58, 63-64, 72-73) against the oracle's
    observables_and_weights on supplied grad / lap / V, including a batch that is not a multiple of the 16 walkers a
    workgroup handles, values on both clip bounds, and the NULL logp_states (zeros) form."""
    from oracle import cg_ref as R
    from coulombgas_amd.engine import Engine, DeviceArray
    n, dim, B = 7, 2, 77
    L = box_length(n, dim)
    rng = np.random.default_rng(5)
    eng = Engine(n, dim, 2, 16, 16, L, orbitals(dim))
    grad = rng.standard_normal((B, n, dim)) + 1j * rng.standard_normal((B, n, dim))
    lap = 3.0 * rng.standard_normal(B) + 1j * rng.standard_normal(B)
    lap[:6] *= 40.0                                                   # outliers: both clip bounds are hit
    V = rng.standard_normal(B)
    Vconst, beta = -1.75, 1 / (4 * 0.15)
    for lps in (-5.0 + rng.standard_normal(B), None):
        g_d = eng.scratch("t_grad", (B, n, dim), complex_pairs=True).upload(np.stack([grad.real, grad.imag], axis=-1))
        l_d = eng.scratch("t_lap", (B,), complex_pairs=True).upload(np.stack([lap.real, lap.imag], axis=-1))
        V_d = DeviceArray.from_numpy(eng, V)
        lps_d = None if lps is None else DeviceArray.from_numpy(eng, lps)
        eloc, floc, mom = eng.local_energy_d(g_d, l_d, V_d, lps_d, Vconst, beta)
        lps_r = np.zeros(B) if lps is None else lps
        robs, rE, rF, rFc, rEc = R.observables_and_weights(R.T(lps_r), torch.as_tensor(grad), torch.as_tensor(lap), R.T(V), Vconst, beta)
        keys = ("K_mean", "K2_mean", "V_mean", "V2_mean", "E_mean", "E2_mean", "F_mean", "F2_mean", "S_mean", "S2_mean")
        m = np.asarray(mom)
        for k, v in zip(keys, m):
            assert v == pytest.approx(float(robs[k]), rel=1e-12, abs=1e-12), k
        assert np.abs(np.asarray(eloc) - rE.numpy()).max() < 1e-12 * np.abs(rE.numpy()).max()
        assert np.abs(np.asarray(floc) - rF.numpy()).max() < 1e-12 * np.abs(rF.numpy()).max()
        tvE = eng.abs_dev_d(eloc, (mom, 4), "t_tvE")
        assert float(np.asarray(tvE)[0]) == pytest.approx(float((rE - robs["E_mean"]).abs().mean()), rel=1e-12)
        w_re, w_im = eng.clip_weights_d(eloc, (mom, 4), tvE, 2.0 / B, "t_w")
        Ec = rEc.numpy()
        assert (Ec != rE.numpy()).sum() >= 2                          # the clip is active in this sample
        assert np.abs(np.asarray(w_re) - 2.0 / B * Ec.real).max() < 1e-13 * np.abs(Ec).max()
        assert np.abs(np.asarray(w_im) - 2.0 / B * Ec.imag).max() < 1e-13 * np.abs(Ec).max()
        tvF = eng.abs_dev_d(floc, (mom, 6), "t_tvF")
        assert float(np.asarray(tvF)[0]) == pytest.approx(float((rF - robs["F_mean"]).abs().mean()), rel=1e-12)
        wf_re, _ = eng.clip_weights_d(floc, (mom, 6), tvF, 1.0 / B, "t_wf")
        assert np.abs(np.asarray(wf_re) - rFc.numpy() / B).max() < 1e-13 * np.abs(rFc.numpy()).max()
    # in-library normal stream for the Hutchinson probe: deterministic, offset = position in the stream, N(0,1) moments
    a = np.asarray(eng.randn_d("t_r1", (4096, 16), seed=11, offset=0)).ravel()
    b = np.asarray(eng.randn_d("t_r2", (4096, 16), seed=11, offset=0)).ravel()
    c = np.asarray(eng.randn_d("t_r3", (1000,), seed=11, offset=500))
    assert np.array_equal(a, b) and np.array_equal(c, a[500:1500])
    assert abs(a.mean()) < 0.02 and abs(a.std() - 1.0) < 0.02 and abs((a ** 4).mean() - 3.0) < 0.15
    eng.close()


def test_device_resident_step_equals_host_array_step():
    """One optimisation step of main.py:270-307 with the walkers as a DeviceArray (nothing of size O(B) or O(P^2) crosses
    PCIe) gives the numbers of the same step fed with numpy arrays: observables, theta-gradient, score, Fisher matrix,
    SR update -- and the accumulated Fisher matrix over two accumulation steps is the mean of the two."""
    import coulombgas_amd as cg
    from coulombgas_amd.engine import DeviceArray
    case = (13, 2, 16, 16, None, 0.1, 0.05)
    n, dim = case[:2]
    B = 48
    s = _setup(case, B, seed=23)
    rng = s["rng"]
    L, rs, kappa, beta = s["L"], 10.0, 10, 1 / (4 * 0.15)
    G = cg.kpoints(dim, 15)
    Vconst = n * rs / L * cg.Madelung(dim, kappa, G)
    lps = -3.0 + rng.standard_normal(B)
    log_prob = lambda pv, si: lps
    v = rng.standard_normal(s["x"].shape)
    lp0 = cg.make_logpsi(s["flow"], s["sp"], L)
    logphi, logjac = cg.make_logphi_logjacdet(s["flow"], s["sp"], L)
    logpsi, lgl = cg.make_logpsi_grad_laplacian(lp0, hutchinson=True, logphi=logphi, logjacdet=logjac)
    loss = cg.make_loss(log_prob, logpsi, lgl, kappa, G, L, rs, Vconst, beta)
    fishers_fn, opt = cg.hybrid_fisher_sr(None, cg.make_quantum_score(lp0), 1e-3, 1e-3)
    params = s["flow"].unravel(s["theta"], dim)
    eng = s["flow"].engine(n, dim, s["sp"])
    res = []
    for x in (s["x"], DeviceArray.from_numpy(eng, s["x"])):
        obs, closs, qloss = loss(None, params, s["sidx"], x, v)
        g, sc = qloss.grad(params, as_pytree=False, reduce=True)
        cf, qf, qm = fishers_fn(None, params, s["sidx"], x)
        qf_h = np.asarray(qf).copy()
        (_, upd), _ = opt.update((None, s["flow"].unravel(g - obs["E_mean"] * sc, dim)), None, (cf, qf, qm))
        res.append((obs, g, sc, qf_h, qm, s["flow"].ravel(upd, dim)))
    for a, b in zip(res[0], res[1]):
        if isinstance(a, dict):
            assert a == b
        else:
            assert np.array_equal(a, b)
    # the solve left the Fisher matrix intact, and matches LAPACK
    P = s["theta"].size
    qf_h, qm = res[0][3], res[0][4]
    ref = np.linalg.solve(qf_h - np.outer(qm.real, qm.real) - np.outer(qm.imag, qm.imag) + 1e-3 * np.eye(P), res[0][1] - res[0][0]["E_mean"] * res[0][2])
    gn = float(np.dot(res[0][1] - res[0][0]["E_mean"] * res[0][2], ref))
    ref *= -min(np.sqrt(1e-3 / gn), 1.0)
    assert np.abs(res[0][5] - ref).max() < 1e-7 * np.abs(ref).max()
    # accumulation over two steps on the device (main.py:285-305)
    from coulombgas_amd.driver import make_update
    seen = {}
    spy = cg.sr.GradientTransformation(lambda p: None, lambda gr, st, params=None: (seen.setdefault("fish", params), ((None, s["flow"].unravel(np.zeros(P), dim)), st))[1])
    update = make_update(loss, spy, 2, fishers_fn)
    xs = [DeviceArray.from_numpy(eng, s["x"]), DeviceArray.from_numpy(eng, walkers(rng, B, n, dim, L))]
    Fs = []
    for xd in xs:
        Fs.append(np.asarray(fishers_fn(None, params, s["sidx"], xd)[1]).copy())
    acc = update.new_acc()
    for a, xd in enumerate(xs):
        _, _, _, acc = update(None, params, None, s["sidx"], xd, v, acc, a == 1)
    assert np.abs(np.asarray(seen["fish"][1]) - 0.5 * (Fs[0] + Fs[1])).max() < 1e-15 * np.abs(Fs[0]).max()


def test_grad_laplacian_n57_all_memory_placements():
    """BASELINE config 5 size (n = 57, Emax = 49): cg_grad_laplacian, cg_param_vjp, cg_ewald and the scores on shipped walkers and
    parameters; Hutchinson and Hutchinson-split against the golden vectors (tests/test_gpu_golden.py) and, here, the two
    modes against each other through their common gradient, plus the VJP against the resident-score path."""
    from coulombgas_amd.engine import Engine
    g = np.load(GOLDEN_DIR + "/golden_n57_d2.npz")
    n, dim, L = int(g["n"]), int(g["dim"]), float(g["L"])
    fix = np.load(GOLDEN_DIR + "/shipped_n57_rs10.npz")
    eng = Engine(n, dim, 2, 16, 16, L, g["sp_indices"])
    eng.set_params(g["theta"])
    B = 24
    rng = np.random.default_rng(57)
    x = fix["x"][:B]; sidx = state_indices(rng, B, n, g["sp_indices"].shape[0]); v = rng.standard_normal(x.shape)
    g1, l1 = eng.grad_laplacian(x, sidx, 1, v)
    g2, l2 = eng.grad_laplacian(x, sidx, 2, v)
    assert np.isfinite(l1).all() and np.isfinite(l2).all() and np.array_equal(g1, g2)
    w1, w2 = rng.standard_normal(B), rng.standard_normal(B)
    ga = eng.param_vjp(x, sidx, w1, w2, use_scores=False)
    gb = eng.param_vjp(x, sidx, w1, w2)
    assert np.abs(ga - gb).max() < 1e-9 * np.abs(ga).max()
    # Ewald sum at n = 57 (1596 pairs x 708 G in the reference's pair form) against the C oracle's pair-form restatement
    import ctypes as C
    from coulombgas_amd.build import build_oracle
    olib = C.CDLL(build_oracle())
    eng.set_ewald(10, g["G"], 10.0)
    V = eng.ewald(x)
    Vr = np.zeros(B)
    Gl = np.ascontiguousarray(g["G"], dtype=np.int64); xc = np.ascontiguousarray(x)
    olib.cgo_ewald(n, dim, C.c_double(L), C.c_double(10.0), C.c_double(10.0), Gl.ctypes.data_as(C.c_void_p), Gl.shape[0],
                   xc.ctypes.data_as(C.c_void_p), B, Vr.ctypes.data_as(C.c_void_p))
    assert np.abs(V - Vr).max() < 1e-10 * np.abs(Vr).max()
    eng.close()


@pytest.mark.parametrize("name", ["golden_n57_d2.npz", "golden_n49_d2.npz"])
def test_two_launch_chunks_at_the_production_batch(name, monkeypatch):
    """BASELINE config 5's per-GPU batch (512 walkers) at n = 57 and n = 49 with the derivative kernels forced into launches of ONE
    workgroup per CU (256 walkers; CG_BIG_ROUNDS / CG_VJP_PER_CU / CG_LAP_PER_CU = 1: csrc/cg_k_big.hip, cg_k_derivs.inc -- the
    default is up to four rounds of workgroups per launch), so walkers 256 ... 511 go through a SECOND launch that reuses the
    workspace slots of the first, in the planned kernels (Hutchinson modes, scores) and in the first generation (theta-VJP without
    resident scores takes the same score kernel).  (i) determinism; (ii) batch independence bit for bit: rows 256 ... of the full
    call == the call on x[256:] alone; (iii) the oracle's golden walkers placed INSIDE the second chunk (and one in the first)
    reproduce the golden gradient / Laplacian / theta-VJP; (iv) the per-sample scores of the second chunk == those of the walkers
    alone."""
    from coulombgas_amd.engine import Engine
    for k in ("CG_BIG_ROUNDS", "CG_VJP_PER_CU", "CG_LAP_PER_CU"):
        monkeypatch.setenv(k, "1")
    g = np.load(GOLDEN_DIR + "/" + name)
    n, dim, L = int(g["n"]), int(g["dim"]), float(g["L"])
    M = g["sp_indices"].shape[0]
    eng = Engine(n, dim, 2, 16, 16, L, g["sp_indices"])
    eng.set_params(g["theta"])
    B = 512
    rng = np.random.default_rng(n)
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, M); v = rng.standard_normal(x.shape)
    rows = {7: 0, 300: 0, 511: 1}                            # batch row -> golden walker
    for r, k in rows.items():
        x[r], sidx[r], v[r] = g["x"][k], g["state_idx"][k], g["v"][k]
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    gr, lp = eng.grad_laplacian(x, sidx, 2, v)
    gr2, lp2 = eng.grad_laplacian(x, sidx, 2, v)
    assert np.array_equal(gr, gr2) and np.array_equal(lp, lp2)
    gt, lt = eng.grad_laplacian(x[256:], sidx[256:], 2, v[256:])
    assert np.array_equal(gr[256:], gt) and np.array_equal(lp[256:], lt)
    for r, k in rows.items():
        assert rel(gr[r], g["grad_split"][k]) < 1e-10 and rel(lp[r], g["lap_split"][k]) < 1e-9
    gh, lh = eng.grad_laplacian(x, sidx, 1, v)
    for r, k in rows.items():
        assert rel(gh[r], g["grad_hutch"][k]) < 1e-10 and rel(lh[r], g["lap_hutch"][k]) < 1e-9
    # theta-VJP with the golden weights on the golden walkers of the SECOND chunk only
    w_re, w_im = np.zeros(B), np.zeros(B)
    w_re[300], w_im[300], w_re[511], w_im[511] = g["w_re"][0], g["w_im"][0], g["w_re"][1], g["w_im"][1]
    assert rel(eng.param_vjp(x, sidx, w_re, w_im), g["vjp"]) < 1e-10
    assert rel(eng.param_vjp(x, sidx, w_re, w_im, use_scores=False), g["vjp"]) < 1e-10
    sc = eng.quantum_score(x, sidx)
    sc2 = eng.quantum_score(x, sidx)
    st = eng.quantum_score(x[256:], sidx[256:])
    assert np.array_equal(sc, sc2) and np.array_equal(sc[256:], st) and np.isfinite(sc).all()
    eng.close()


@pytest.mark.parametrize("n,B,mode", [(57, 300, 2), (29, 70, 1), (20, 9, 2), (13, 33, 2), (29, 5, 0), (29, 1, 2), (16, 3, 1), (8, 2, 2)])
def test_grad_laplacian_and_scores_in_one_call(n, B, mode, monkeypatch):
    """cg_grad_laplacian_scores (the pair of calls an optimisation step makes on the same walkers: src/VMC.py:35, then the jacrev of
    main.py:278): gradient, Laplacian and the resident per-sample scores are those of cg_grad_laplacian + cg_scores_compute BIT FOR BIT --
    where the fused kernel of csrc/cg_k_big.hip serves (n > 16, Hutchinson modes; n = 57 with a second launch chunk, n = 20 with a
    different placement plan) and where the call falls back to the two kernels (n = 13, exact mode).  The scores are compared through
    one-hot theta-VJPs (single rows of the score matrix, real and imaginary part), their batch mean and the Fisher matrix."""
    from coulombgas_amd.engine import Engine, DeviceArray
    from bench import synthetic
    monkeypatch.setenv("CG_BIG_ROUNDS", "1")                   # n = 57: 256 walkers per launch
    monkeypatch.setenv("CG_SMALL_FUSED_CHUNK", "20")           # n = 13, B = 33: a second launch that reuses the stash slots of the first
    Emax = 49 if n > 40 else 25
    L, sp, theta, sidx, x = synthetic(n, 2, B, Emax, 3)
    rng = np.random.default_rng(100 + n)
    theta = theta + 0.05 * rng.standard_normal(theta.shape)
    eng = Engine(n, 2, 2, 16, 16, L, sp); eng.set_params(theta)
    P = eng.P
    x_d = DeviceArray.from_numpy(eng, x); s_d = DeviceArray.from_numpy(eng, sidx, np.int32)
    v_d = DeviceArray.from_numpy(eng, rng.standard_normal(x.shape)) if mode else None
    rows = sorted({0, B // 2, B - 1})

    def score_probe():
        out = []
        acc = eng.scratch("probe_acc", (3 * P + P * P,))
        for r in rows:
            for part in (0, 1):
                w = np.zeros((2, B)); w[part, r] = 1.0
                eng.scores_vjp_d(DeviceArray.from_numpy(eng, w[0]), DeviceArray.from_numpy(eng, w[1]), acc, 0)
                out.append(eng.to_host(acc)[:P].copy())
        eng.scores_mean_d(acc, P)
        eng.scores_fisher_d(acc, 3 * P, P)
        out.append(eng.to_host(acc)[P:].copy())
        return out

    g_d, l_d = eng.grad_laplacian_d(x_d, s_d, mode, v_d)
    g0, l0 = eng.to_host(g_d).copy(), eng.to_host(l_d).copy()
    eng.scores_compute_d(x_d, s_d)
    ref = score_probe()
    x_d.version += 1                                           # the resident scores no longer count as those of x_d
    g_d, l_d = eng.grad_laplacian_d(x_d, s_d, mode, v_d, with_scores=True)
    g1, l1 = eng.to_host(g_d).copy(), eng.to_host(l_d).copy()
    if mode:
        assert eng._score_key_d is not None                    # ... the fused call has left them there
    eng.scores_compute_d(x_d, s_d)                             # (a look-up after a fused call; the computation in exact mode)
    got = score_probe()
    assert np.isfinite(g0).all() and np.isfinite(l0).all() and all(np.isfinite(a).all() for a in ref)
    assert np.array_equal(g0, g1) and np.array_equal(l0, l1)
    assert len(ref) == len(got) and all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert max(np.abs(a).max() for a in ref[:-1]) > 0
    if mode and B <= 9:
        # the same entry point with HOST pointers (staged arguments: the two calls one after the other inside the library)
        from coulombgas_amd._lib import lib, check
        import ctypes as C
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        xh, sh_, vh = np.ascontiguousarray(x), np.ascontiguousarray(sidx, dtype=np.int32), np.ascontiguousarray(eng.to_host(v_d))
        gh, lh = np.empty((B, n, 2, 2)), np.empty((B, 2))
        check(lib().cg_grad_laplacian_scores(eng._ctx, p(xh), p(sh_), B, mode, p(vh), p(gh), p(lh)), eng._ctx)
        eng._score_key = eng._score_key_d = None
        assert np.array_equal(gh.reshape(-1), np.asarray(g0).view(np.float64).reshape(-1)) and np.array_equal(lh.reshape(-1), np.asarray(l0).view(np.float64).reshape(-1))
        assert all(np.array_equal(a, b) for a, b in zip(ref, score_probe()))
    eng.close()


# ---------------------------------------------------------------------------------------------
# f2: the autoregressive Transformer density matrix on the device (cg_van_sample / cg_van_log_prob)
# ---------------------------------------------------------------------------------------------
def test_transformer_density_matrix_on_device():
    """src/autoregressive.py:50-96 + src/sampler.py:4-50 as device kernels (one wave per sample, key / value cache in LDS):
    (i) the reference's own KAT tests/test_sampler.py:40-69: the probabilities of all C(10,4) ordered occupations sum to one;
    (ii) log-probabilities against the oracle's torch restatement with perturbed weights; (iii) the sampler with supplied
    uniforms against the host restatement of jax.random.categorical, and its by-product log-probabilities; (iv) the Philox
    sampler's empirical frequencies against the exact probabilities; (v) the shipped pretrained n = 13 model: F, E, S of its
    published data.txt row; (vi) the shipped n = 57 model (M = 149 orbitals) against the host implementation."""
    import itertools
    import coulombgas_amd as cg
    from coulombgas_amd.engine import Engine
    from oracle import cg_ref as R
    dim = 2
    # (i) + (iv) n = 4 electrons in M = 10 orbitals
    n, M = 4, 10
    sp10 = orbitals(2)[-M:]
    eng = Engine(n, dim, 2, 16, 16, box_length(n, dim), sp10)
    van = cg.Transformer(M, 2, 16, 4, 32)
    rng = np.random.default_rng(4)
    params = van.init(rng, sp10[:n])
    for mod in params:
        for leaf in params[mod]:
            params[mod][leaf] = params[mod][leaf] + 0.3 * rng.standard_normal(params[mod][leaf].shape)
    sampler, log_prob = cg.make_autoregressive_sampler(van, sp10, n, M, engine=eng)
    states = np.array(list(itertools.combinations(range(M), n)), dtype=np.int32)
    lp_all = log_prob(params, states)
    assert np.exp(lp_all).sum() == pytest.approx(1.0, abs=1e-12)
    tp = {m: {l: R.T(v) for l, v in params[m].items()} for m in params}
    lr = np.array([float(R.autoregressive_log_prob(tp, torch.as_tensor(s.astype(np.int64)), R.T(sp10), 2, 4)) for s in states])
    assert np.abs(lp_all - lr).max() < 1e-12 * np.abs(lr).max()
    B = 200000
    s_d = sampler(params, 11, B)
    s = np.asarray(s_d)
    assert s.shape == (B, n) and (np.diff(s, axis=1) > 0).all() and s.min() >= 0 and s.max() < M
    assert np.abs(np.asarray(log_prob(params, s_d)) - log_prob(params, s)).max() < 1e-12           # by-product of the sampling pass
    code = {tuple(st): i for i, st in enumerate(states.tolist())}
    counts = np.bincount([code[tuple(r)] for r in s.tolist()], minlength=len(states))
    p = np.exp(lr)
    chi2 = ((counts - B * p) ** 2 / (B * p)).sum()
    assert chi2 < len(states) + 6 * np.sqrt(2 * len(states)), chi2                               # chi-square, 209 d.o.f.
    # (iii) supplied uniforms: identical draws to the host restatement
    hs, hlp = make_host_sampler(van, sp10, n, M)
    u = rng.uniform(size=(512, n, M))
    assert np.array_equal(np.asarray(sampler(params, 0, 512, unif=u)), hs(params, 0, 512, unif=u))
    eng.close()
    # (ii) n = 5, M = 12 against the oracle
    n, M = 5, 12
    sp = orbitals(2)[-M:]
    eng = Engine(n, dim, 2, 16, 16, box_length(n, dim), sp)
    van = cg.Transformer(M, 2, 16, 4, 32)
    params = van.init(rng, sp[:n])
    for mod in params:
        for leaf in params[mod]:
            params[mod][leaf] = params[mod][leaf] + 0.3 * rng.standard_normal(params[mod][leaf].shape)
    sampler, log_prob = cg.make_autoregressive_sampler(van, sp, n, M, engine=eng)
    s = np.asarray(sampler(params, 2, 64))
    tp = {m: {l: R.T(v) for l, v in params[m].items()} for m in params}
    lr = np.array([float(R.autoregressive_log_prob(tp, torch.as_tensor(r.astype(np.int64)), R.T(sp), 2, 4)) for r in s])
    assert np.abs(log_prob(params, s) - lr).max() < 1e-12 * np.abs(lr).max()
    eng.close()
    # (v) shipped pretrained model, n = 13 (published free-fermion row: epoch, F, F_std, E, E_std, S, S_std)
    n, Theta = 13, 0.15
    L, beta = box_length(n, dim), 1 / (4 * Theta)
    spt = orbitals(2, 25)
    z = np.load(GOLDEN_DIR + "/pretrained_van_n13.npz")
    pv = _load_van("pretrained_van_n13.npz")
    eng = Engine(n, dim, 2, 16, 16, L, spt)
    van = cg.Transformer(spt.shape[0], 2, 16, 4, 32)
    sampler, log_prob = cg.make_autoregressive_sampler(van, spt, n, spt.shape[0], engine=eng)
    Es = (2 * np.pi / L) ** 2 * (spt ** 2).sum(-1)                                    # src/freefermion/pretraining.py:54
    B = 65536
    s_d = sampler(pv, 1, B)
    s, lp = np.asarray(s_d), np.asarray(log_prob(pv, s_d))
    F = lp / beta + Es[s].sum(-1)
    row = z["data_row_last"]
    print("pretrained n=13 on the device: F %.6f +- %.6f (published %.6f)  E %.4f (published %.4f)  S %.4f (published %.4f)"
          % (F.mean(), F.std() / np.sqrt(B), row[1], Es[s].sum(-1).mean(), row[3], -lp.mean(), row[5]))
    assert abs(F.mean() - row[1]) < 5 * np.hypot(F.std() / np.sqrt(B), row[2])
    assert abs(Es[s].sum(-1).mean() - row[3]) < 5 * np.hypot(Es[s].sum(-1).std() / np.sqrt(B), row[4])
    assert abs(-lp.mean() - row[5]) < 5 * np.hypot(lp.std() / np.sqrt(B), row[6])
    eng.close()
    # (vi) shipped n = 57 model: M = 149 orbitals (three lanes-rounds of logits, 57-position attention)
    n = 57
    sp49 = orbitals(2, 49)
    pv = _load_van("shipped_n57_rs10_van.npz")
    eng = Engine(n, dim, 2, 16, 16, box_length(n, dim), sp49)
    van = cg.Transformer(sp49.shape[0], 2, 16, 4, 32)
    sampler, log_prob = cg.make_autoregressive_sampler(van, sp49, n, sp49.shape[0], engine=eng)
    _, hlp = make_host_sampler(van, sp49, n, sp49.shape[0])
    s = np.asarray(sampler(pv, 5, 96))
    assert (np.diff(s, axis=1) > 0).all() and s.max() < sp49.shape[0]
    ref = hlp(pv, s)
    assert np.abs(log_prob(pv, s) - ref).max() < 1e-11 * np.abs(ref).max()
    eng.close()


def test_classical_fisher_and_blocked_cholesky_at_ragged_sizes():
    """src/sr.py:36 / :38-41 at sizes that exercise every edge of the blocked kernels: 64 x 64 wave blocks of the SYRK with ragged
    last blocks and a sliced batch, the 64-column Cholesky panels (MFMA panel solve with the inverse diagonal block, 64 x 64
    trailing blocks), one-launch-per-block triangular solves."""
    from coulombgas_amd.engine import Engine
    rng = np.random.default_rng(8)
    eng = Engine(3, 2, 2, 16, 16, box_length(3, 2), orbitals(2))
    for B, P in ((1000, 333), (130, 64), (257, 129), (50, 500), (4099, 191)):
        S = rng.standard_normal((B, P))
        F = eng.fisher_real(S)
        ref = S.T @ S / B
        assert np.abs(F - ref).max() < 1e-13 * np.abs(ref).max() * np.sqrt(B) and np.array_equal(F, F.T), (B, P)
    for P in (64, 65, 127, 256, 257, 333, 512, 513, 700, 769, 1153):       # (256-column outer blocks: look-ahead from three of them on)
        S = rng.standard_normal((2 * P, P))
        M = S.T @ S / (2 * P) + 1e-3 * np.eye(P)
        Lg = eng.cholesky(M)
        Lr = np.linalg.cholesky(M)
        assert np.abs(np.tril(Lg) - Lr).max() < 1e-10 * np.abs(Lr).max(), P
        rhs = rng.standard_normal(P)
        xg = eng.spd_solve(M, rhs, 1e-3)
        xr = np.linalg.solve(M + 1e-3 * np.eye(P), rhs)
        assert np.abs(xg - xr).max() < 1e-10 * np.abs(xr).max(), P
    # a non-positive pivot deep inside the matrix (fourth outer block, second 64-column step) is reported, not factored through
    from coulombgas_amd._lib import CoulombGasError
    d = np.ones(900); d[840] = -1.0
    for fn in (lambda: eng.cholesky(np.diag(d)), lambda: eng.spd_solve(np.diag(d), np.ones(900))):
        with pytest.raises(CoulombGasError, match="not positive definite"):
            fn()
    eng.close()


def test_transformer_reverse_pass_on_device():
    """jax.vmap(jax.grad(log_prob)) (src/sampler.py:52-65) and jax.jacrev of the weighted log-probability sums (main.py:277) from
    the device's hand-written reverse pass (cg_van_scores_*): per-sample scores against torch autograd through the oracle's
    restatement and against the host numpy backward; the weighted VJP; the classical Fisher matrix in ravel_pytree order;
    the shipped n = 57 model (M = 149: three lane-rounds of logits, stash in HBM)."""
    import coulombgas_amd as cg
    from coulombgas_amd.engine import Engine
    from coulombgas_amd.sr import ravel_pytree, _ravel_batched
    from oracle import cg_ref as R
    dim = 2
    rng = np.random.default_rng(14)
    for n, M, nl, ms, nh, hsz, B in ((5, 12, 2, 16, 4, 32, 70), (4, 10, 1, 8, 2, 16, 33), (3, 70, 3, 32, 4, 48, 20)):
        sp = orbitals(2)[-M:] if M <= 25 else orbitals(2, 25)[:M] if M <= 81 else orbitals(2, 49)[:M]
        eng = Engine(n, dim, 2, 16, 16, box_length(n, dim), sp)
        van = cg.Transformer(M, nl, ms, nh, hsz)
        params = van.init(rng, sp[:n])
        for mod in params:
            for leaf in params[mod]:
                params[mod][leaf] = params[mod][leaf] + 0.3 * rng.standard_normal(params[mod][leaf].shape)
        sampler, log_prob = cg.make_autoregressive_sampler(van, sp, n, M, engine=eng)
        _, hlp = make_host_sampler(van, sp, n, M)
        s_d = sampler(params, 3, B)
        s = np.asarray(s_d)
        sc = log_prob.grad(params, s_d)
        S = np.asarray(sc)                                                    # (B, P), ravel_pytree order
        Sh = _ravel_batched(hlp.grad(params, s))
        scale = np.abs(Sh).max()
        assert S.shape == Sh.shape and np.abs(S - Sh).max() < 1e-12 * scale, (n, M, np.abs(S - Sh).max() / scale)
        # torch autograd through the oracle's restatement of src/autoregressive.py + src/sampler.py (first 6 samples)
        for b in range(6):
            tp = {m: {l: R.T(v).clone().requires_grad_(True) for l, v in params[m].items()} for m in params}
            lp = R.autoregressive_log_prob(tp, torch.as_tensor(s[b].astype(np.int64)), R.T(sp), nl, nh)
            lp.backward()
            gref = ravel_pytree({m: {l: tp[m][l].grad.numpy() for l in tp[m]} for m in tp})[0]
            assert np.abs(S[b] - gref).max() < 1e-12 * scale, (n, M, b, np.abs(S[b] - gref).max() / scale)
        # per-leaf view
        tree = sc.tree()
        href = hlp.grad(params, s)
        for mod in href:
            for leaf in href[mod]:
                assert tree[mod][leaf].shape == href[mod][leaf].shape
                assert np.abs(tree[mod][leaf] - href[mod][leaf]).max() < 1e-12 * scale
        w = rng.standard_normal(B) / B
        gv = ravel_pytree(log_prob.vjp(params, s_d, w))[0]
        assert np.abs(gv - w @ Sh).max() < 1e-12 * np.abs(w @ Sh).max()
        gv2 = ravel_pytree(log_prob.vjp(params, s, w))[0]                      # host samples are uploaded
        assert np.array_equal(gv, gv2)
        F = eng.to_host(sc.fisher_d())
        Fh = Sh.T @ Sh / B
        assert np.abs(F - Fh).max() < 1e-12 * np.abs(Fh).max()
        eng.close()
    n = 57
    sp49 = orbitals(2, 49)
    pv = _load_van("shipped_n57_rs10_van.npz")
    eng = Engine(n, dim, 2, 16, 16, box_length(n, dim), sp49)
    van = cg.Transformer(sp49.shape[0], 2, 16, 4, 32)
    sampler, log_prob = cg.make_autoregressive_sampler(van, sp49, n, sp49.shape[0], engine=eng)
    _, hlp = make_host_sampler(van, sp49, n, sp49.shape[0])
    s_d = sampler(pv, 5, 40)
    S = np.asarray(log_prob.grad(pv, s_d))
    Sh = _ravel_batched(hlp.grad(pv, np.asarray(s_d)))
    assert np.abs(S - Sh).max() < 1e-11 * np.abs(Sh).max(), np.abs(S - Sh).max() / np.abs(Sh).max()
    eng.close()


@pytest.mark.parametrize("n,M,B,packed", [(5, 12, 70, 1), (24, 40, 33, 1), (33, 81, 20, 1), (64, 113, 9, 1), (13, 81, 37, 1), (17, 40, 9, 1),
                                          (19, 81, 7, 1), (3, 12, 50, 1), (13, 81, 11, 0), (5, 12, 70, 0), (24, 40, 33, 0), (33, 81, 20, 0), (29, 81, 5, 1)])
def test_transformer_positions_in_parallel_reverse_pass(n, M, B, packed, monkeypatch):
    """csrc/cg_van_par.hpp (the per-sample gradient of log p with the positions on the lanes of a wave, weight gradients on the matrix
    cores; shipped architecture, the default from n = 20 on), forced at ragged sizes -- fewer orbitals than one 16-column tile, n = 64
    (every lane but one a position), orbital counts that are no multiple of 16 -- against the host numpy backward at 1e-12 and
    against torch autograd through the oracle's restatement (src/sampler.py:40-46, 65).  Up to 33 particles the PACKED variant runs
    (64 / (n - 1) samples share a wave: 2 at n = 24 ... 33, 16 at n = 5, 5 at n = 13, 3 at n = 19 with ten lanes idle, 32 at n = 3; batches that are no
    multiple of that); packed = 0 forces one sample per wave there."""
    import coulombgas_amd as cg
    from coulombgas_amd.engine import Engine
    from coulombgas_amd.sr import ravel_pytree, _ravel_batched
    from oracle import cg_ref as R
    monkeypatch.setenv("CG_VAN_PAR", "1"); monkeypatch.setenv("CG_VAN_PACKED", str(packed))
    rng = np.random.default_rng(100 + n)
    sp = orbitals(2)[-M:] if M <= 25 else orbitals(2, 25)[:M] if M <= 81 else orbitals(2, 36)[:M]
    eng = Engine(n, 2, 2, 16, 16, box_length(n, 2), sp)
    van = cg.Transformer(M, 2, 16, 4, 32)
    params = van.init(rng, sp[:n])
    for mod in params:
        for leaf in params[mod]:
            params[mod][leaf] = params[mod][leaf] + 0.3 * rng.standard_normal(params[mod][leaf].shape)
    sampler, log_prob = cg.make_autoregressive_sampler(van, sp, n, M, engine=eng)
    _, hlp = make_host_sampler(van, sp, n, M)
    s_d = sampler(params, 3, B)
    s = np.asarray(s_d)
    S = np.asarray(log_prob.grad(params, s_d))
    Sh = _ravel_batched(hlp.grad(params, s))
    scale = np.abs(Sh).max()
    assert S.shape == Sh.shape and np.abs(S - Sh).max() < 1e-12 * scale, np.abs(S - Sh).max() / scale
    for b in range(2):
        tp = {m: {l: R.T(v).clone().requires_grad_(True) for l, v in params[m].items()} for m in params}
        lp = R.autoregressive_log_prob(tp, torch.as_tensor(s[b].astype(np.int64)), R.T(sp), 2, 4)
        lp.backward()
        gref = ravel_pytree({m: {l: tp[m][l].grad.numpy() for l in tp[m]} for m in tp})[0]
        assert np.abs(S[b] - gref).max() < 1e-12 * scale
    eng.close()


@pytest.mark.parametrize("finite_T", [False, True])
def test_resumed_run_retraces_the_uninterrupted_one(finite_T, tmp_path):
    """main.py:217-223 + 374-381 on the device path: a run interrupted after its checkpoint at epoch 2 and resumed (walkers, keys,
    parameters, optimizer state from the file) produces epochs 3 and 4 of the uninterrupted run -- rows and parameters bit for bit;
    zero temperature (SR on the flow) and finite temperature (the Transformer density matrix sampled and trained on the GPU too)."""
    import coulombgas_amd as cg
    n, dim = 13, 2
    L = box_length(n, dim); sp = orbitals(2, 25)
    flow = cg.FermiNet(2, 16, 16, L)
    p0 = flow.init(5, np.zeros((n, dim)))
    kw = dict(rs=10.0, beta=1 / (4 * 0.15), batch=48, mc_therm=2, mc_steps=10, seed=11, sr=(1e-3, 1e-3), ckpt_every=2)
    if finite_T:
        van = cg.Transformer(sp.shape[0], 2, 16, 4, 32)
        pv = van.init(np.random.default_rng(3), sp[:n])
        make = lambda: cg.make_autoregressive_sampler(van, sp, n, sp.shape[0])
    else:
        pv = None
        samp = cg.GroundStateSampler(n, sp.shape[0])
        make = lambda: (samp, samp.log_prob)

    def run(epochs, path, **more):
        sampler, log_prob = make()
        return cg.train(flow, p0, sp, n, dim, L, epochs=epochs, sampler=sampler, log_prob=log_prob, params_van=pv, ckpt_path=str(path), **kw, **more)
    vA, fA, rowsA = run(4, tmp_path / "a")
    run(2, tmp_path / "b")
    vB, fB, rowsB = run(4, tmp_path / "b", epoch_finished=2)
    assert len(rowsA) == 4 and rowsB == rowsA[2:]
    assert np.array_equal(flow.ravel(fA, dim), flow.ravel(fB, dim))
    if finite_T:
        from coulombgas_amd.sr import ravel_pytree
        assert np.array_equal(ravel_pytree(vA)[0], ravel_pytree(vB)[0])
        assert not np.array_equal(ravel_pytree(vA)[0], ravel_pytree(pv)[0])


def test_freefermion_pretraining_on_device():
    """f4 (src/freefermion/pretraining.py:34-108) with the density matrix sampled and evaluated on the GPU and the classical
    Fisher matrix formed there: natural-gradient pre-training of a small Transformer lowers F = <log p / beta + E> towards the
    exact canonical free energy of non-interacting fermions (brute-force enumeration), never below it."""
    import itertools
    import coulombgas_amd as cg
    from coulombgas_amd.engine import Engine
    from coulombgas_amd.freefermion import exact_free_energy
    n, Theta, dim = 4, 0.15, 2
    L, beta = box_length(n, dim), 1 / (4 * Theta)
    sp10 = orbitals(2, 25)[-10:]
    Es = (2 * np.pi / L) ** 2 * (sp10 ** 2).sum(-1)
    Etot = np.array([Es[list(c)].sum() for c in itertools.combinations(range(10), n)])
    w = np.exp(-beta * (Etot - Etot.min()))
    F_exact = Etot.min() - np.log(w.sum()) / beta
    assert exact_free_energy(Es, n, beta)[0] == pytest.approx(F_exact, rel=1e-13)
    eng = Engine(n, dim, 2, 16, 16, L, sp10)
    van = cg.Transformer(10, 1, 8, 2, 16)
    p0 = van.init(5, sp10[:n])
    pv, rows = cg.pretrain(van, p0, n, dim, Theta, sp10, 11, sr=True, damping=1e-3, max_norm=1e-2, batch=4096, epoch=60, engine=eng)
    v = np.array([[float(t) for t in r.split()] for r in rows])
    assert v.shape == (60, 7) and np.isfinite(v).all()
    assert v[-5:, 1].mean() < v[0, 1] - 5 * v[0, 2]               # F went down by many standard errors
    assert v[-5:, 1].mean() > F_exact - 5 * v[-5:, 2].mean()      # ... and respects the variational bound
    print("pre-training on the device: F %.5f -> %.5f (exact %.5f)" % (v[0, 1], v[-5:, 1].mean(), F_exact))
    eng.close()


@pytest.mark.timeout(300)
def test_panel_inverses_with_full_pivoting_activity(tmp_path):
    """The register-tiled, panel-blocked Gauss-Jordan inverses of the derivative kernels at n > 16 (csrc/cg_linalg.hpp: cg_inverse_panel_real /
    _complex) on matrices WITHOUT diagonal dominance -- the flow Jacobians of the parity tests are near the identity and keep the natural
    pivots, so the rows-never-move bookkeeping, the used-row masks and the partial last panels are exercised here: the development harness
    (tools/lu_bench/tile_inv_bench.hip) is compiled and run at every tile shape; it checks |A^-1 A - I| and |D^-1 D - I| < 1e-9 itself."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tile_inv_bench")
    subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-o", exe, os.path.join(root, "tools", "lu_bench", "tile_inv_bench.hip")])
    for n, nt in ((29, 512), (29, 256), (33, 512), (49, 512), (57, 512), (64, 512), (17, 256), (32, 512)):
        r = subprocess.run([exe, str(n), str(nt), "2", "1", "1"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (n, nt, r.stdout[-500:], r.stderr[-500:])
