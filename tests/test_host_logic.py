"""Host-side logic of the reference-API mirror (CPU; the GPU engine is replaced by the host emulation of the
device code, tests/emul_engine.py)."""
import numpy as np
import pytest
import torch

import coulombgas_amd as cg
from coulombgas_amd import flow as fl, vmc
from oracle import cg_ref as R
from tests import emul_engine
from tests.host_transformer import make_host_sampler
from tests.common import orbitals, box_length, flow_theta, state_indices, walkers, GOLDEN


def test_parameter_tree_matches_haiku_layout():
    """names, shapes, ravel order, P = 1074 for depth 2 / 16 / 16 / dim 2."""
    order = fl.ravel_order(2, 16, 16, 2)
    assert [(n, l, s) for n, l, s in order] == [
        ("fermi_net/linear", "b", (2,)), ("fermi_net/linear", "w", (16, 2)),
        ("fermi_net/~/linear", "b", (16,)), ("fermi_net/~/linear", "w", (9, 16)),
        ("fermi_net/~/linear_1", "b", (16,)), ("fermi_net/~/linear_1", "w", (48, 16)),
        ("fermi_net/~/linear_2", "b", (16,)), ("fermi_net/~/linear_2", "w", (5, 16))]
    assert sum(int(np.prod(s)) for _, _, s in order) == 1074
    f = cg.FermiNet(2, 16, 16, 9.5)
    p = f.init(0, np.zeros((29, 2)))
    th = f.ravel(p, 2)
    assert th.shape == (1074,) and np.all(p["fermi_net/linear"]["b"] == 0) and 0.005 < p["fermi_net/~/linear_1"]["w"].std() < 0.02
    q = f.unravel(th, 2)
    assert all(np.array_equal(q[n][l], p[n][l]) for n, l, _ in order)
    shipped = np.load(GOLDEN + "/shipped_n29_rs10.npz")["theta"]
    assert f.unravel(shipped, 2)["fermi_net/~/linear_2"]["w"].shape == (5, 16)
    with pytest.raises(ValueError):
        cg.FermiNet(1, 16, 16, 1.0)
    # same ordering as the oracle's restatement of ravel_pytree
    assert [(n, l) for n, l, _ in order] == [(n, l) for n, l, _ in R.flow_ravel_order(2, 16, 16, 2)]
    assert [(n, l) for n, l, _ in fl.ravel_order(3, 16, 16, 3)] == [(n, l) for n, l, _ in R.flow_ravel_order(3, 16, 16, 3)]


def test_complex_clip_is_lexicographic():
    """SURVEY App. B4 (src/VMC.py:73)"""
    a = np.array([1 + 5j, -3 + 2j, 4 - 1j, 0.5 + 0j, 2 + 7j, -1 - 7j])
    out = vmc.complex_clip(a, -1.0, 2.0)
    assert np.array_equal(out, np.array([1 + 5j, -1 + 0j, 2 + 0j, 0.5 + 0j, 2 + 0j, -1 + 0j]))
    ref = R.complex_clip(torch.as_tensor(a), -1.0, 2.0).numpy()
    assert np.array_equal(out, ref)


def test_kpoints_madelung_shard():
    G = cg.kpoints(2, 15)
    assert G.shape == (708, 2) and np.array_equal(G, R.kpoints(2, 15))
    assert cg.kpoints(3, 4).shape[1] == 3
    assert cg.Madelung(2, 10, G) == pytest.approx(-3.900264920056, abs=1e-11)
    assert cg.Madelung(3, 7, cg.kpoints(3, 7)) == pytest.approx(R.Madelung(3, 7, R.kpoints(3, 7)), rel=1e-13)
    x = np.arange(24.0).reshape(2, 3, 4)
    assert np.array_equal(cg.shard(x, rank=1, world=2), x[1])
    with pytest.raises(ValueError):
        cg.shard(x, rank=0, world=4)
    assert cg.replicate({"a": 1}, 8) == {"a": 1}


def _problem(B=6, seed=17):
    n, dim, hs, ht = 13, 2, 16, 16
    L = box_length(n, dim)
    rng = np.random.default_rng(seed)
    sp = orbitals(2)
    theta = flow_theta(rng, 2, hs, ht, dim, 0.1, 0.05)
    return dict(n=n, dim=dim, hs=hs, ht=ht, L=L, sp=sp, theta=theta, x=walkers(rng, B, n, dim, L),
                sidx=state_indices(rng, B, n, sp.shape[0]), rng=rng, rs=10.0, kappa=10, beta=1 / (4 * 0.15),
                logp_states=-5.0 + rng.standard_normal(B), v=rng.standard_normal((B, n, dim)))


def build_loss(pb, comm=None):
    flow = cg.FermiNet(2, pb["hs"], pb["ht"], pb["L"])
    G = cg.kpoints(pb["dim"], 15)
    Vconst = pb["n"] * pb["rs"] / pb["L"] * cg.Madelung(pb["dim"], pb["kappa"], G)
    logpsi_novmap = cg.make_logpsi(flow, pb["sp"], pb["L"])
    logphi, logjacdet = cg.make_logphi_logjacdet(flow, pb["sp"], pb["L"])
    logpsi, lgl = cg.make_logpsi_grad_laplacian(logpsi_novmap, hutchinson=True, logphi=logphi, logjacdet=logjacdet)
    return cg.make_loss(lambda pv, si: pv, logpsi, lgl, pb["kappa"], G, pb["L"], pb["rs"], Vconst, pb["beta"], comm=comm), G, Vconst


def test_make_loss_host_algebra(monkeypatch):
    """src/VMC.py:31-80 + main.py:277-278 with the emulated engine: observables, clip, values, gradients vs the oracle."""
    emul_engine.install(monkeypatch)
    pb = _problem()
    obs_fn, G, Vconst = build_loss(pb)
    obs, closs, qloss = obs_fn(pb["logp_states"], pb["theta"], pb["sidx"], pb["x"], pb["v"])
    qv = qloss(pb["theta"]); cv = closs(pb["logp_states"])
    g_grad, g_score = qloss.grad(pb["theta"], as_pytree=False)
    n, dim, hs, ht, L = pb["n"], pb["dim"], pb["hs"], pb["ht"], pb["L"]
    rflow = R.FermiNet(2, hs, ht, L); rparams = R.flow_unravel(R.T(pb["theta"]), 2, hs, ht, dim)
    r_logpsi = R.make_logpsi(rflow, pb["sp"], L)
    r_logphi, r_logjacdet = R.make_logphi_logjacdet(rflow, pb["sp"], L)
    _, rfn = R.make_logpsi_grad_laplacian(r_logpsi, hutchinson=True, logphi=r_logphi, logjacdet=r_logjacdet)
    sb = torch.as_tensor(pb["sidx"].astype(np.int64))
    gr, lr = rfn(R.T(pb["x"]), rparams, sb, R.T(pb["v"]))
    pot = R.potential_energy(R.T(pb["x"]), pb["kappa"], G, L, pb["rs"])
    robs, Eloc, Floc, Fc, Ec = R.observables_and_weights(R.T(pb["logp_states"]), gr, lr, pot, Vconst, pb["beta"])
    assert set(obs) == set(robs)
    for k in robs:
        assert obs[k] == pytest.approx(float(robs[k]), rel=1e-9, abs=1e-9), k
    lpt = lambda xb, th, sbb: r_logpsi(xb, R.flow_unravel(th, 2, hs, ht, dim), sbb)
    v0, v1, dg, ds = R.quantum_loss_and_grads(lpt, R.T(pb["theta"]), R.T(pb["x"]), sb, Ec)
    assert qv[0] == pytest.approx(float(v0), rel=1e-9, abs=1e-9) and qv[1] == pytest.approx(float(v1), rel=1e-10)
    assert np.abs(g_grad - dg.numpy()).max() < 1e-7 * max(1.0, np.abs(dg.numpy()).max())
    assert np.abs(g_score - ds.numpy()).max() < 1e-10 * max(1.0, np.abs(ds.numpy()).max())
    assert cv[0] == pytest.approx(float((R.T(pb["logp_states"]) * Fc).mean()), rel=1e-10)
    assert np.allclose(closs.weights, Fc.numpy() / len(Fc))
    tree_g, tree_s = qloss.grad(pb["theta"])
    assert tree_g["fermi_net/~/linear_1"]["w"].shape == (48, 16)
    only = cg.make_observable(lambda pv, si: pv, *[None] * 0, **{}) if False else None   # alias exists
    assert callable(cg.make_observable)


def test_make_loss_asks_for_the_scores_with_the_laplacian(monkeypatch):
    """make_loss runs the grad / Laplacian of a step as ONE call that also leaves the per-sample scores of the same walkers resident
    (cg_grad_laplacian_scores on the device: the set-up of the two is shared); quantum_lossfn.grad then only looks them up.
    make_observable has no gradient to follow and must not ask; an engine without the keyword (a foreign logpsi_grad_laplacian)
    is called the reference's way."""
    emul_engine.install(monkeypatch)
    from tests.emul_engine import EmulEngine
    import coulombgas_amd.logpsi as lp_mod
    monkeypatch.setattr(lp_mod, "_is_device", lambda a: True)          # the emulated engine's arrays are numpy arrays: take the device branch
    asked, computed = [], []
    orig_gl, orig_sc = EmulEngine.grad_laplacian_d, EmulEngine.scores_compute_d
    monkeypatch.setattr(EmulEngine, "grad_laplacian_d", lambda self, x, s, mode, v=None, with_scores=False:
                        (asked.append(with_scores), orig_gl(self, x, s, mode, v, with_scores=with_scores))[1])
    monkeypatch.setattr(EmulEngine, "scores_compute_d", lambda self, x, s: (computed.append(1), orig_sc(self, x, s))[1])
    pb = _problem()
    obs_fn, G, Vconst = build_loss(pb)
    obs, closs, qloss = obs_fn(pb["logp_states"], pb["theta"], pb["sidx"], pb["x"], pb["v"])
    assert asked == [True] and len(computed) == 1
    g1 = qloss.grad(pb["theta"], as_pytree=False)
    # the same through the two separate calls
    flow = cg.FermiNet(2, pb["hs"], pb["ht"], pb["L"])
    logpsi_novmap = cg.make_logpsi(flow, pb["sp"], pb["L"])
    logphi, logjacdet = cg.make_logphi_logjacdet(flow, pb["sp"], pb["L"])
    logpsi, lgl = cg.make_logpsi_grad_laplacian(logpsi_novmap, hutchinson=True, logphi=logphi, logjacdet=logjacdet)
    sep = cg.make_loss(lambda pv, si: pv, logpsi, lgl, pb["kappa"], G, pb["L"], pb["rs"], Vconst, pb["beta"], fuse_scores=False)
    del asked[:]
    obs2, _, qloss2 = sep(pb["logp_states"], pb["theta"], pb["sidx"], pb["x"], pb["v"])
    g2 = qloss2.grad(pb["theta"], as_pytree=False)
    assert asked == [False] and obs2 == obs and all(np.array_equal(a, b) for a, b in zip(g1, g2))
    del asked[:]
    only_obs = cg.make_observable(lambda pv, si: pv, logpsi, lgl, pb["kappa"], G, pb["L"], pb["rs"], Vconst, pb["beta"])
    assert only_obs(pb["logp_states"], pb["theta"], pb["sidx"], pb["x"], pb["v"]) == obs and asked == [False]
    plain = lambda x, params, state_indices, key: lgl(x, params, state_indices, key)          # the reference's four-argument form
    plain.wf = lgl.wf
    four = cg.make_loss(lambda pv, si: pv, logpsi, plain, pb["kappa"], G, pb["L"], pb["rs"], Vconst, pb["beta"])
    assert four(pb["logp_states"], pb["theta"], pb["sidx"], pb["x"], pb["v"])[0] == obs


def test_sample_stateindices_and_x(monkeypatch):
    """src/VMC.py:8-25: key split, sampler call, chain, wrap into [0, L)."""
    emul_engine.install(monkeypatch)
    pb = _problem(B=8)
    flow = cg.FermiNet(2, 16, 16, pb["L"])
    logp = cg.make_logp(cg.make_logpsi(flow, pb["sp"], pb["L"]))
    calls = {}

    def sampler(params_van, key_state, batch):
        calls["batch"] = batch; calls["key"] = key_state
        return pb["sidx"][:batch]
    key = np.random.SeedSequence(42)
    key2, s, x, rate = cg.sample_stateindices_and_x(key, sampler, None, logp, pb["x"] + 3 * pb["L"], pb["theta"], 5, 0.1, pb["L"])
    assert calls["batch"] == 8 and isinstance(key2, np.random.SeedSequence)
    assert s.dtype == np.int32 and s.shape == (8, 13) and x.shape == (8, 13, 2)
    assert (x >= 0).all() and (x < pb["L"]).all() and 0.0 <= rate <= 1.0
    # deterministic in the key
    _, _, x_again, rate_again = cg.sample_stateindices_and_x(np.random.SeedSequence(42), sampler, None, logp, pb["x"] + 3 * pb["L"],
                                                             pb["theta"], 5, 0.1, pb["L"])
    assert np.array_equal(x, x_again) and rate == rate_again
    with pytest.raises(TypeError):
        cg.mcmc(lambda xx: xx, pb["x"], 0, 3, 0.1)          # an opaque callable cannot run in the GPU chain: no fallback


def test_hybrid_fisher_sr_against_oracle(monkeypatch):
    """src/sr.py:56-122 (+ fisher_sr :13-52): Fisher matrices, centring, damped solves and the norm clip of the host mirror
    (engine = host emulation of the device code) against the oracle's statement-by-statement restatement, with the
    per-sample scores of the oracle's jacrev."""
    emul_engine.install(monkeypatch)
    n, dim, hs, ht, B = 5, 2, 4, 4, 12
    L = 2.0
    rng = np.random.default_rng(23)
    sp = orbitals(dim)
    theta = flow_theta(rng, 2, hs, ht, dim, 0.4, 0.2)
    x = walkers(rng, B, n, dim, L); sidx = state_indices(rng, B, n, sp.shape[0])
    flow = cg.FermiNet(2, hs, ht, L)
    params_flow = flow.unravel(theta, dim)
    logpsi = cg.make_logpsi(flow, sp, L)
    qscore = cg.make_quantum_score(logpsi)
    Pv = 7
    cs = rng.standard_normal((B, Pv))
    classical_score_fn = lambda params_van, state_indices: {"b": cs[:, 4:], "a": cs[:, :4].reshape(B, 2, 2)}   # a pytree
    damping, max_norm = 1e-3, 1e-3
    fishers_fn, opt = cg.hybrid_fisher_sr(classical_score_fn, qscore, damping, max_norm)
    cf, qf, qm = fishers_fn(None, params_flow, sidx, x)
    g_van = {"a": rng.standard_normal((2, 2)), "b": rng.standard_normal(3)}
    g_flow = flow.unravel(rng.standard_normal(theta.size), dim)
    (u_van, u_flow), _ = opt.update((g_van, g_flow), opt.init(None), (cf, qf, qm))
    # oracle
    rflow = R.FermiNet(2, hs, ht, L)
    r_logpsi = R.make_logpsi(rflow, sp, L)
    lpt = lambda xb, th, sbb: r_logpsi(xb, R.flow_unravel(th, 2, hs, ht, dim), sbb)
    qs = R.make_quantum_score(lpt)(R.T(x), R.T(theta), torch.as_tensor(sidx.astype(np.int64))).numpy()
    gv = np.concatenate([g_van["a"].ravel(), g_van["b"].ravel()])
    rcf, rqf, rqm, ruv, ruf = R.hybrid_fisher_sr_update(cs, qs, gv, flow.ravel(g_flow, dim), damping, max_norm)
    assert np.abs(cf - rcf).max() < 1e-13 and np.abs(qf - rqf).max() < 1e-9 * np.abs(rqf).max() and np.abs(qm - rqm).max() < 1e-10 * np.abs(rqm).max()
    assert np.abs(np.concatenate([u_van["a"].ravel(), u_van["b"].ravel()]) - ruv).max() < 1e-9 * np.abs(ruv).max()
    assert np.abs(flow.ravel(u_flow, dim) - ruf).max() < 1e-7 * np.abs(ruf).max()
    new = cg.apply_updates(params_flow, u_flow)
    assert np.allclose(flow.ravel(new, dim), theta + flow.ravel(u_flow, dim))
    # fisher_sr (purely classical natural gradient)
    upd, _ = cg.fisher_sr(classical_score_fn, damping, max_norm).update(g_van, None, (None, sidx))
    assert np.abs(np.concatenate([upd["a"].ravel(), upd["b"].ravel()]) - R.sr_solve_and_clip(rcf, gv, damping, max_norm)).max() < 1e-9 * np.abs(ruv).max()


def test_training_loop_runs_and_logs(monkeypatch):
    """main.py:216-384 mirror (coulombgas_amd/driver.py) end to end on the host emulation: thermalisation, two epochs with
    gradient accumulation, both optimizers, data.txt row format (main.py:367-372)."""
    emul_engine.install(monkeypatch)
    n, dim, L = 4, 2, 2.0
    sp = orbitals(dim)
    flow = cg.FermiNet(2, 4, 4, L)
    p0 = flow.init(3, np.zeros((n, dim)))
    samp = cg.GroundStateSampler(n, sp.shape[0])
    for kw in (dict(sr=(1e-3, 1e-3)), dict(optimizer=cg.adam(1e-2))):
        pv, pf, rows = cg.train(flow, p0, sp, n, dim, L, rs=2.0, beta=1 / (4 * 0.15), batch=8, epochs=2, sampler=samp,
                                log_prob=samp.log_prob, mc_therm=1, mc_steps=3, acc_steps=2, seed=1, **kw)
        assert pv is None and len(rows) == 2
        vals = [float(v) for v in rows[-1].split()]
        assert len(vals) == 12 and int(vals[0]) == 2 and all(np.isfinite(vals)) and 0.0 <= vals[-1] <= 1.0
        assert vals[9] == 0.0                                   # entropy of the zero-temperature sampler
        assert abs(vals[1] - vals[3]) < 1e-6                    # F = E when S = 0
        assert not np.array_equal(flow.ravel(pf, dim), flow.ravel(p0, dim))
    # finite temperature: the reference's Transformer density matrix trained together with the flow (hybrid SR, main.py:179-184)
    M = 8
    spm = sp[-M:]
    van = cg.Transformer(M, 1, 8, 2, 16)
    pv0 = van.init(2, spm[:n])
    tsamp, tlogp = make_host_sampler(van, spm, n, M)
    pv, pf, rows = cg.train(flow, p0, spm, n, dim, L, rs=2.0, beta=1 / (4 * 0.15), batch=8, epochs=2, sampler=tsamp, log_prob=tlogp,
                            params_van=pv0, sr=(1e-3, 1e-3), mc_therm=1, mc_steps=3, acc_steps=2, seed=1)
    vals = [float(v) for v in rows[-1].split()]
    assert all(np.isfinite(vals)) and vals[9] > 0.0            # entropy > 0 now
    assert any(not np.array_equal(pv[m][l], pv0[m][l]) for m in pv for l in pv[m])
    assert not np.array_equal(flow.ravel(pf, dim), flow.ravel(p0, dim))


def test_update_averages_fishers_over_accumulation_steps():
    """main.py:285-305: the Fisher matrices handed to the optimizer are the MEAN over the accumulation steps.  fishers_fn
    returns an engine-owned buffer that the next accumulation step overwrites (as the GPU engine's output buffers are): the
    accumulator must copy on the first step instead of aliasing it (round-1 bug: F_2 instead of (F_1 + F_2) / 2)."""
    from coulombgas_amd.driver import make_update
    P = 5
    rng = np.random.default_rng(0)
    mats = [rng.standard_normal((P, P)) for _ in range(2)]
    means = [rng.standard_normal(P) + 1j * rng.standard_normal(P) for _ in range(2)]
    out_buffer, calls, seen = np.zeros((P, P)), [0], {}

    def fishers_fn(params_van, params_flow, state_indices, x):
        k = calls[0]; calls[0] += 1
        out_buffer[...] = mats[k]                      # the same array object every call
        return None, out_buffer, means[k]

    def observable_and_lossfn(params_van, params_flow, state_indices, x, key):
        q = lambda p: (0.0, 0.0)
        q.grad = lambda p, reduce=False: ({"w": np.ones(P)}, {"w": np.zeros(P)})
        return {k: 1.0 for k in cg.driver.DATA_KEYS}, None, q

    def opt_update(grads, state, params=None):
        seen["fish"] = params
        return (None, {"w": np.zeros(P)}), state
    opt = cg.sr.GradientTransformation(lambda p: None, opt_update)
    update = make_update(observable_and_lossfn, opt, 2, fishers_fn)
    acc = update.new_acc()
    pf = {"w": np.zeros(P)}
    for a in range(2):
        _, pf, _, acc = update(None, pf, None, None, np.zeros((3, 2, 2)), None, acc, a == 1)
    cf, qf, qm = seen["fish"]
    assert cf is None
    assert np.allclose(qf, 0.5 * (mats[0] + mats[1]), rtol=0, atol=1e-15)
    assert np.allclose(qm, 0.5 * (means[0] + means[1]), rtol=0, atol=1e-15)


def test_train_checkpoint_and_resume(monkeypatch, tmp_path):
    """main.py:217-223, 374-381: train() writes {"keys", "x", "params_van", "params_flow", "opt_state"} with the reference's
    leading device axis on x, and a run resumed from that file continues with the saved walkers, parameters and Adam moments."""
    emul_engine.install(monkeypatch)
    n, dim, L = 4, 2, 2.0
    sp = orbitals(dim)
    flow = cg.FermiNet(2, 4, 4, L)
    p0 = flow.init(3, np.zeros((n, dim)))
    samp = cg.GroundStateSampler(n, sp.shape[0])
    kw = dict(rs=2.0, beta=1 / (4 * 0.15), batch=8, sampler=samp, log_prob=samp.log_prob, mc_therm=1, mc_steps=3, seed=1,
              ckpt_path=str(tmp_path), ckpt_every=2)
    _, pf2, rows2 = cg.train(flow, p0, sp, n, dim, L, epochs=2, optimizer=cg.adam(1e-2), **kw)
    ck = cg.load_data(cg.ckpt_filename(2, str(tmp_path)))
    assert set(ck) == {"keys", "x", "params_van", "params_flow", "opt_state"}
    assert ck["x"].shape == (1, 8, n, dim) and ck["opt_state"]["count"] == 2
    assert np.array_equal(flow.ravel(ck["params_flow"], dim), flow.ravel(pf2, dim))
    rows = []
    _, pf4, rows4 = cg.train(flow, p0, sp, n, dim, L, epochs=4, optimizer=cg.adam(1e-2), epoch_finished=2, log=rows.append, **kw)
    assert [int(r.split()[0]) for r in rows4] == [3, 4] and rows == rows4
    assert not np.array_equal(flow.ravel(pf4, dim), flow.ravel(pf2, dim))
    assert cg.load_data(cg.ckpt_filename(4, str(tmp_path)))["opt_state"]["count"] == 4
    # ... and retraces an uninterrupted run draw for draw (the run goes on from the key it writes into the file)
    _, pfA, rowsA = cg.train(flow, p0, sp, n, dim, L, epochs=4, optimizer=cg.adam(1e-2), **dict(kw, ckpt_path=str(tmp_path / "straight")))
    assert rowsA[2:] == rows4 and np.array_equal(flow.ravel(pfA, dim), flow.ravel(pf4, dim))


def test_checkpoint_interop(tmp_path):
    """src/checkpoint.py mirror: jax-pickled pytrees (main.py:374-381) load as numpy without jax (the array
    reconstructor is mapped to numpy, optimizer-state classes to stubs); save_data round-trips; data.txt columns."""
    import sys, types, pickle, os
    from coulombgas_amd import checkpoint as ck

    # a pickle in the format jax writes: arrays reduce to jax._src.array._reconstruct_array(np-reconstructor, args, state, aval)
    fake = types.ModuleType("jax._src.array")
    class FakeJaxArray:
        def __init__(self, a): self.a = np.asarray(a)
        def __reduce__(self):
            fun, args, state = self.a.__reduce__()
            return (fake._reconstruct_array, (fun, args, state, ("aval", self.a.shape)))
    def _reconstruct_array(fun, args, arr_state, aval_state):       # never called on load: our unpickler intercepts the name
        raise AssertionError("the jax reconstructor must not be imported")
    _reconstruct_array.__module__ = "jax._src.array"; _reconstruct_array.__qualname__ = "_reconstruct_array"
    fake._reconstruct_array = _reconstruct_array
    optx = types.ModuleType("optax._src.base")
    class EmptyState(tuple): pass
    EmptyState.__module__ = "optax._src.base"; EmptyState.__qualname__ = "EmptyState"
    optx.EmptyState = EmptyState
    mods = {"jax": types.ModuleType("jax"), "jax._src": types.ModuleType("jax._src"), "jax._src.array": fake,
            "optax": types.ModuleType("optax"), "optax._src": types.ModuleType("optax._src"), "optax._src.base": optx}
    x = np.random.default_rng(0).uniform(size=(2, 3, 5, 2))
    tree = {"x": FakeJaxArray(x), "params_flow": {"fermi_net/linear": {"b": FakeJaxArray(np.arange(2.0)), "w": FakeJaxArray(np.ones((16, 2)))}},
            "opt_state": EmptyState(), "keys": FakeJaxArray(np.arange(4, dtype=np.uint32).reshape(2, 2))}
    fn = ck.ckpt_filename(100, str(tmp_path))
    assert fn.endswith("epoch_000100.pkl")
    sys.modules.update(mods)
    try:
        with open(fn, "wb") as f:
            pickle.dump(tree, f)
    finally:
        for k in mods: sys.modules.pop(k, None)
    got = cg.load_data(fn)
    assert isinstance(got["x"], np.ndarray) and np.array_equal(got["x"], x) and got["keys"].dtype == np.uint32
    assert np.array_equal(got["params_flow"]["fermi_net/linear"]["w"], np.ones((16, 2)))
    out = {"x": x, "params_flow": {"a": {"b": np.zeros(3)}}, "epoch": 7}
    cg.save_data(out, fn)
    back = cg.load_data(fn)
    assert np.array_equal(back["x"], x) and back["epoch"] == 7 and np.array_equal(back["params_flow"]["a"]["b"], np.zeros(3))
    with open(os.path.join(str(tmp_path), "data.txt"), "w") as f:
        f.write("     1  -4.870600  0.002007  -4.870600  0.002007  0.619739  0.000123  -5.490339  0.002010  0.000000  0.000000  0.5369\n")
    lg = ck.load_log(os.path.join(str(tmp_path), "data.txt"))
    assert lg["E"][0] == -4.8706 and lg["accept_rate"][0] == 0.5369
    # container-only cross-check against a shipped checkpoint and its committed fixture slice
    import glob
    d = glob.glob("/root/reference/data/n_29_dim_2_rs_10.0_*")
    if d:
        c = cg.load_data(cg.ckpt_filename(3000, d[0]))
        fix = np.load(GOLDEN + "/shipped_n29_rs10.npz")
        th = np.concatenate([np.asarray(c["params_flow"][k][l]).ravel() for k in sorted(c["params_flow"]) for l in ("b", "w")])
        assert np.array_equal(th, fix["theta"]) and np.array_equal(c["x"].reshape(-1, 29, 2)[:512], fix["x"])


def _load_van(name):
    z = np.load(GOLDEN + "/" + name)
    pv = {}
    for k in z.files:
        if "|" in k:
            m, l = k.split("|"); pv.setdefault(m, {})[l] = z[k]
    return pv, z


def test_autoregressive_sampler_kats():
    """src/autoregressive.py + src/sampler.py (numpy forward): (i) tests/test_sampler.py:40-69 -- the conditional
    probabilities sum to one over all C(10,4) ordered occupations; (ii) the mask of src/sampler.py:72-91; (iii) the
    shipped pretrained free-fermion model (n=13): F = <log p / beta + E> reproduces the published value of its data.txt."""
    import itertools
    sp10 = orbitals(2)[-10:]
    van = cg.Transformer(10, 2, 16, 4, 32)
    params = van.init(0, sp10[:4])
    mask_fn, sampler, log_prob = make_host_sampler(van, sp10, 4, 10, mask_fn=True)
    si = np.array(list(itertools.combinations(range(10), 4)))
    assert np.exp(log_prob(params, si)).sum() == pytest.approx(1.0, abs=1e-12)
    m = mask_fn(np.array([1, 4, 5, 7])).astype(int)
    assert m.tolist() == [[1, 1, 1, 1, 1, 1, 1, 0, 0, 0], [0, 0, 1, 1, 1, 1, 1, 1, 0, 0], [0, 0, 0, 0, 0, 1, 1, 1, 1, 0], [0, 0, 0, 0, 0, 0, 1, 1, 1, 1]]
    s = sampler(params, 3, 64)
    assert s.shape == (64, 4) and (np.diff(s, axis=1) > 0).all() and s.min() >= 0 and s.max() < 10
    with pytest.raises(ValueError):
        cg.Transformer(10, 2, 16, 3, 32)
    # shipped pretrained model
    pv, z = _load_van("pretrained_van_n13.npz")
    n, Theta = 13, 0.15
    L, beta = np.sqrt(np.pi * n), 1 / (4 * Theta)
    spt = orbitals(2, 25)
    Es = (2 * np.pi / L) ** 2 * (spt ** 2).sum(-1)                                    # src/freefermion/pretraining.py:54
    van = cg.Transformer(spt.shape[0], 2, 16, 4, 32)
    sampler, log_prob = make_host_sampler(van, spt, n, spt.shape[0])
    B = 4096
    s = sampler(pv, 1, B)
    lp = log_prob(pv, s)
    F = lp / beta + Es[s].sum(-1)
    row = z["data_row_last"]                                                          # epoch, F, F_std, E, E_std, S, S_std
    assert abs(F.mean() - row[1]) < 5 * np.hypot(F.std() / np.sqrt(B), row[2])
    assert abs(Es[s].sum(-1).mean() - row[3]) < 5 * np.hypot(Es[s].sum(-1).std() / np.sqrt(B), row[4])
    assert abs(-lp.mean() - row[5]) < 5 * np.hypot(lp.std() / np.sqrt(B), row[6])


def test_autoregressive_gradients_vs_torch_autograd():
    """jax.grad(log_prob) of src/sampler.py:65 (classical score) and the weighted VJP: hand-written numpy reverse pass of
    coulombgas_amd/autoregressive.py against autograd of the oracle's torch restatement."""
    n, M, B = 5, 12, 6
    sp = orbitals(2)[-M:]
    van = cg.Transformer(M, 2, 16, 4, 32)
    rng = np.random.default_rng(4)
    params = van.init(rng, sp[:n])
    for mod in params:                                       # larger weights than the init so that every path matters
        for leaf in params[mod]:
            params[mod][leaf] = params[mod][leaf] + 0.3 * rng.standard_normal(params[mod][leaf].shape)
    sampler, log_prob = make_host_sampler(van, sp, n, M)
    s = sampler(params, 2, B)
    tp = {m: {l: R.T(v).requires_grad_(True) for l, v in params[m].items()} for m in params}
    lps = [R.autoregressive_log_prob(tp, torch.as_tensor(s[b].astype(np.int64)), R.T(sp), 2, 4) for b in range(B)]
    assert np.abs(log_prob(params, s) - np.array([float(v) for v in lps])).max() < 1e-12
    leaves = [(m, l) for m in sorted(tp) for l in sorted(tp[m])]
    g = log_prob.grad(params, s)
    for b in range(B):
        gr = torch.autograd.grad(lps[b], [tp[m][l] for m, l in leaves], retain_graph=True)
        for (m, l), r in zip(leaves, gr):
            assert np.abs(g[m][l][b] - r.numpy()).max() < 1e-12 * max(1.0, np.abs(r.numpy()).max()), (m, l)
    w = rng.standard_normal(B)
    v = log_prob.vjp(params, s, w)
    for m, l in leaves:
        assert np.abs(v[m][l] - np.tensordot(w, g[m][l], axes=1)).max() < 1e-12
    sc = cg.make_classical_score(log_prob)(params, s)
    from coulombgas_amd.sr import _ravel_batched
    assert _ravel_batched(sc).shape == (B, sum(int(np.prod(params[m][l].shape)) for m, l in leaves))


def test_freefermion_pretraining_and_exact_free_energy():
    """src/freefermion/pretraining.py mirror: (i) the double-precision canonical recursion for F, E, S against brute-force
    enumeration and against the published (sampled) free energies of the shipped pretrained models; (ii) natural-gradient
    pre-training of a small density matrix lowers F towards the exact value (variational bound)."""
    import itertools
    from coulombgas_amd.freefermion import exact_free_energy, make_loss
    sp = orbitals(2, 25)
    n, Theta = 4, 0.15
    L, beta = np.sqrt(np.pi * n), 1 / (4 * Theta)
    sp10 = sp[-10:]
    Es = (2 * np.pi / L) ** 2 * (sp10 ** 2).sum(-1)
    Etot = np.array([Es[list(c)].sum() for c in itertools.combinations(range(10), n)])
    w = np.exp(-beta * (Etot - Etot.min())); Z = w.sum()
    F_bf = Etot.min() - np.log(Z) / beta; E_bf = (Etot * w).sum() / Z
    F, E, S = exact_free_energy(Es, n, beta)
    assert F == pytest.approx(F_bf, rel=1e-13) and E == pytest.approx(E_bf, rel=1e-12) and S == pytest.approx(beta * (E_bf - F_bf), rel=1e-10)
    for nn, pub in ((13, 24.811018), (29, 54.701225)):            # data/freefermion/pretraining/*/*/data.txt:5000 (trained, sampled)
        LL = np.sqrt(np.pi * nn)
        Fx = exact_free_energy((2 * np.pi / LL) ** 2 * (sp ** 2).sum(-1), nn, beta)[0]
        assert abs(Fx - pub) < 2e-4
    van = cg.Transformer(10, 1, 8, 2, 16)
    p0 = van.init(5, sp10[:n])
    pv, rows = cg.pretrain(van, p0, n, 2, Theta, sp10, 11, sr=True, damping=1e-3, max_norm=1e-2, batch=1024, epoch=40,
                          density_matrix=make_host_sampler(van, sp10, n, 10))
    v = np.array([[float(t) for t in r.split()] for r in rows])
    assert v.shape == (40, 7) and np.isfinite(v).all()
    assert v[-5:, 1].mean() < v[0, 1] - 5 * v[0, 2]               # F went down by many standard errors
    assert v[-5:, 1].mean() > F - 5 * v[-5:, 2].mean()            # ... and respects the variational bound
    _, log_prob = make_host_sampler(van, sp10, n, 10)
    loss = make_loss(log_prob, Es, beta)
    s = np.array(list(itertools.combinations(range(10), n)))[:32]
    val, aux = loss(pv, s)
    assert np.isfinite(val) and set(aux) == {"E_mean", "E_std", "F_mean", "F_std", "S_mean", "S_std"}


def test_transformer_flat_parameter_order_and_missing_engine():
    """Host side of the device Transformer: flat_params (the order of cg_van_set_params, include/coulombgas.h) and unflat_params
    are inverse to each other, also with leading axes (per-sample gradients); the permutation DeviceScores hands to
    cg_van_scores_fisher maps ravel_pytree order (jax.flatten_util: sorted keys) to that flat order; and the density-matrix
    closures refuse to run without a GPU engine unless the numpy restatement is asked for explicitly."""
    import coulombgas_amd as cg
    from coulombgas_amd.autoregressive import flat_params, unflat_params
    from coulombgas_amd.sr import ravel_pytree
    van = cg.Transformer(10, 2, 16, 4, 32)
    sp10 = orbitals(2)[:10]
    params = van.init(3, sp10[:4])
    flat = flat_params(van, params, 2)
    count = sum(int(np.prod(v.shape)) for m in params.values() for v in m.values())
    assert flat.shape == (count,)
    back = unflat_params(van, flat, 2)
    for m in params:
        for l in params[m]:
            assert np.array_equal(back[m][l], params[m][l])
    batched = unflat_params(van, np.stack([flat, 2 * flat]), 2)              # leading batch axis
    assert batched[van.name]["x1hat"].shape == (2, 10) and np.array_equal(batched[van.name + "/output_mlp"]["w"][1], 2 * params[van.name + "/output_mlp"]["w"])
    perm = ravel_pytree(unflat_params(van, np.arange(count), 2))[0].astype(int)
    assert sorted(perm.tolist()) == list(range(count))
    assert np.array_equal(ravel_pytree(params)[0], flat[perm])
    sampler, log_prob = cg.make_autoregressive_sampler(van, sp10, 4, 10)
    with pytest.raises(RuntimeError):
        sampler(params, 0, 8)
    with pytest.raises(RuntimeError):
        log_prob(params, np.array([[0, 1, 2, 3]]))


def test_sp_orbitals_and_twist_sort_serve_the_pinned_tables():
    """src/__init__.py:1 exports sp_orbitals / twist_sort and main.py:79-90 calls them: the package serves the pinned tables
    (coulombgas_amd/data, written from the reference's own function) under those names and refuses every other grid."""
    import coulombgas_amd as cg
    for Emax, M in ((25, 81), (36, 113), (49, 149)):
        sp_indices, Es = cg.sp_orbitals(2, Emax)                       # main.py:79
        assert sp_indices.shape == (M, 2) and sp_indices.dtype == np.int64 and Es.shape == (M,)
        assert (np.diff(Es) >= 0).all() and Es[-1] <= Emax and np.array_equal(Es, (sp_indices ** 2).sum(-1))
        assert len({tuple(r) for r in sp_indices}) == M
        assert Es[12] == 4 and Es[28] == 9                               # closed shells n = 13, 29 (main.py:80 Ef = Es[n-1])
        tw, Etw = cg.twist_sort(sp_indices, np.array([0.25, 0.25]))     # main.py:88
        assert (np.diff(Etw) >= 0).all()
        assert np.array_equal(tw[::-1], orbitals(2, Emax))              # main.py:90 == the table every engine test uses
        with pytest.raises(ValueError):
            cg.twist_sort(sp_indices, np.array([0.0, 0.5]))
        with pytest.raises(ValueError):
            cg.twist_sort(sp_indices[::-1], np.array([0.25, 0.25]))
    idx3, Es3 = cg.sp_orbitals(3)                                        # tests/test_slater.py:17
    assert idx3.shape == (1935, 3) and (np.diff(Es3) >= 0).all() and Es3[6] == 1 and Es3[7] == 2
    t3, E3 = cg.twist_sort(idx3, np.array([0.1, 0.2, 0.3]))
    assert (np.diff(E3) >= 0).all() and t3.shape == idx3.shape
    for bad in ((2, 30), (3, 25), (1, 25)):
        with pytest.raises(ValueError):
            cg.sp_orbitals(*bad)
