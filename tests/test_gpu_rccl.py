"""The device-buffer collectives on hardware (run with -m gpu): a WORLD-1 RCCL communicator executes every all-reduce site of an
optimisation step -- the only multi-GPU evidence obtainable on a one-GPU box.  Mirrors jax.lax.pmean of src/VMC.py:44-53,
main.py:280 and src/sr.py:73-82 as implemented by coulombgas_amd.comm.RcclComm.pmean_d (in-place ncclAllReduce + 1/N on the
library's stream, no host staging)."""
import json

import numpy as np
import pytest

from tests.common import orbitals, box_length

pytestmark = pytest.mark.gpu


def _engine(n=13, dim=2):
    import coulombgas_amd as cg
    L = box_length(n, dim)
    sp = orbitals(dim)
    flow = cg.FermiNet(2, 16, 16, L)
    return flow, flow.engine(n, dim, sp), sp, L


def test_pmean_d_on_device_arrays_views_and_a_fisher_sized_buffer():
    from coulombgas_amd.comm import RcclComm
    from coulombgas_amd.engine import DeviceArray
    flow, eng, sp, L = _engine()
    comm = RcclComm(eng, 0, 1)
    rng = np.random.default_rng(0)
    # (i) a plain DeviceArray and a complex one: the mean over one rank is the array itself, bit for bit; the version is bumped
    a = rng.standard_normal(1000); d = DeviceArray.from_numpy(eng, a); v0 = d.version
    assert comm.pmean_d(d) is d and d.version == v0 + 1 and np.array_equal(np.asarray(d), a)
    c = DeviceArray(eng, (37,), complex_pairs=True); c.upload(rng.standard_normal((37, 2)))
    before = np.asarray(c).copy(); comm.pmean_d(c); assert np.array_equal(np.asarray(c), before)
    # (ii) the packed buffers of the update: [grad (P) | mean score (2P)] whole, and [Fisher (P^2) | mean score (2P)] -- the matrix
    # through a DeviceView at index 0, the score part by (count, index) at a non-zero offset
    P = eng.P
    pack = DeviceArray.from_numpy(eng, rng.standard_normal(P * P + 2 * P))
    ref = np.asarray(pack).copy()
    comm.pmean_d(pack)
    comm.pmean_d(pack, count=2 * P, index=P * P)
    view = eng.view(pack, P * P, (P, 2))
    comm.pmean_d(view)
    comm.pmean_d(eng.view(pack, 0, (P, P)))
    assert np.array_equal(np.asarray(pack), ref)
    with pytest.raises(IndexError):
        comm.pmean_d(pack, count=2 * P + 1, index=P * P)
    with pytest.raises(IndexError):
        comm.pmean_d(eng.view(pack, P * P, (P, 2)), count=2 * P + 8)
    # (iii) the classical Fisher matrix of the shipped Transformer (P_van = 5907: 279 MB) in one all-reduce
    Pv = 5907
    big = DeviceArray(eng, (Pv, Pv))
    row = rng.standard_normal(Pv)
    host = np.empty((Pv, Pv)); host[:] = row[None, :]; host += np.arange(Pv)[:, None] * 1e-3
    big.upload(host)
    comm.pmean_d(big)
    assert np.array_equal(big.numpy(0, Pv), host[0]) and np.array_equal(big.numpy((Pv - 1) * Pv, Pv), host[-1])
    assert np.array_equal(big.numpy(2953 * Pv + 17, 1000), host[2953, 17:1017])
    big.free(); comm.close()


def test_training_with_a_world1_rccl_communicator_equals_the_null_communicator():
    """two SR epochs of train() with every pmean site going through ncclAllReduce on device buffers (world 1) against the same run
    with NullComm: identical data.txt rows and parameters, bit for bit (a one-rank sum followed by x 1/1 is exact)."""
    import coulombgas_amd as cg
    from coulombgas_amd.comm import RcclComm, NullComm
    n, dim = 13, 2
    out = {}
    for name in ("null", "rccl"):
        flow, eng, sp, L = _engine(n, dim)
        comm = NullComm() if name == "null" else RcclComm(eng, 0, 1)
        samp = cg.GroundStateSampler(n, sp.shape[0])
        p0 = flow.init(3, np.zeros((n, dim)))
        pv, pf, rows = cg.train(flow, p0, sp, n, dim, L, rs=10.0, beta=1 / (4 * 0.15), batch=256, epochs=2, sampler=samp,
                                log_prob=samp.log_prob, sr=(1e-3, 1e-3), mc_therm=2, mc_steps=10, acc_steps=2, seed=11, comm=comm)
        out[name] = (rows, flow.ravel(pf, dim))
        comm.close()
    assert out["null"][0] == out["rccl"][0]
    assert np.array_equal(out["null"][1], out["rccl"][1])
    assert all(np.isfinite([float(v) for v in r.split()]).all() for r in out["rccl"][0])


def test_bench_distributed_branch_without_torch_distributed(request):
    """bench.py's N > 1 code path, launched once per session (tests/conftest.py) as plain `python bench.py --gpus 1` with
    RANK / WORLD_SIZE / MASTER_* in the environment and CG_FORCE_DIST=1: the library's own RCCL communicator (id exchange of
    coulombgas_amd.comm, no torch.distributed), the accept rate reduced on the device, barrier / MAX through cg_allreduce_mean."""
    r = getattr(request.config, "_cg_bench_dist", None)
    assert r, "the session launch did not run (pytest -m gpu on a GPU box)"
    assert r["rc"] == 0, r["out"][-2000:] + r["err"]
    line = [l for l in r["out"].splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["comm"] == "rccl via cg_allreduce_mean" and j["n_gpus"] == 1 and j["finite"] and j["value"] > 1e6
    assert 0.3 < j["accept_rate"] < 0.95


def test_accept_rate_on_the_device_equals_the_host_count():
    """cg_mcmc_accept_rate (count / denom formed on the device, all-reduced there by a world-1 communicator) against the counter
    read back by cg_mcmc_accepts, and through coulombgas_amd.mcmc with RcclComm / NullComm"""
    import coulombgas_amd as cg
    from coulombgas_amd.comm import RcclComm, NullComm
    flow, eng, sp, L = _engine()
    n, dim, B, steps = 13, 2, 512, 7
    rng = np.random.default_rng(5)
    theta = flow.ravel(flow.init(3, np.zeros((n, dim))), dim)
    logp = cg.make_logp(cg.make_logpsi(flow, sp, L))
    x = rng.uniform(0, L, (B, n, dim)); sidx = np.tile(np.arange(sp.shape[0] - n, sp.shape[0], dtype=np.int32), (B, 1))
    comm = RcclComm(eng, 0, 1)
    rates = []
    for cm in (NullComm(), comm):
        _, rate = cg.mcmc(logp.bind(flow.unravel(theta, dim), sidx), x, 123, steps, 0.1, comm=cm)
        rates.append(rate)
        assert rate == eng.mcmc_accepts() / float(steps * B)
    assert rates[0] == rates[1] and 0.2 < rates[0] < 1.0
    assert comm.pmax(3.25) == 3.25
    comm.barrier(); comm.close()
