"""N > 1 path on CPU: world_size-2 gloo job (one process per 'device') vs the single-process result on the full
batch.  Covers the pmean sites of src/MCMC.py:39, src/VMC.py:46-53,63,72 and main.py:280 as implemented by
coulombgas_amd.comm, and the per-rank sharding of walkers (main.py:231-237)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_match_single_process(tmp_path, monkeypatch):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # every rank holds the same reduced values
    for k in ("obs", "qv", "g", "s", "rate", "cf", "qf", "qm"):
        assert np.array_equal(r0[k], r1[k]), k
    assert not np.array_equal(r0["x"], r1["x"])                 # different walkers per rank
    # Fisher matrices accumulated per rank and all-reduced once == all-reduced per accumulation step (the pmean is linear)
    assert float(r0["defer_err"]) < 1e-14 and float(r1["defer_err"]) < 1e-14
    # single process, full batch
    from tests import emul_engine
    from tests.test_host_logic import _problem, build_loss
    emul_engine.install(monkeypatch)
    pb = _problem(B=8)
    obs_fn, G, Vconst = build_loss(pb)
    obs, closs, qloss = obs_fn(pb["logp_states"], pb["theta"], pb["sidx"], pb["x"], pb["v"])
    full = np.array([obs[k] for k in sorted(obs)])
    assert np.abs(r0["obs"] - full).max() < 1e-11 * np.abs(full).max()      # mean of shard means == batch mean
    qv = np.array(qloss(pb["theta"]))
    assert np.abs(r0["qv"] - qv).max() < 1e-10 * max(1.0, np.abs(qv).max())  # same clip width (pmean'd tv) on both paths
    g, s = qloss.grad(pb["theta"], as_pytree=False)
    assert np.abs(r0["g"] - g).max() < 1e-10 * max(1.0, np.abs(g).max())
    assert np.abs(r0["s"] - s).max() < 1e-11 * max(1.0, np.abs(s).max())
    assert 0.0 <= float(r0["rate"]) <= 1.0
    # SR: mean over ranks of the per-rank Fisher blocks == Fisher matrix of the full batch (src/sr.py:70-76)
    import coulombgas_amd as cg
    flow = cg.FermiNet(2, 16, 16, pb["L"])
    qscore = cg.make_quantum_score(cg.make_logpsi(flow, pb["sp"], pb["L"]))
    cs_full = np.random.default_rng(99).standard_normal((pb["x"].shape[0], 5))
    fishers_fn, _ = cg.hybrid_fisher_sr(lambda pv, si: cs_full, qscore, 1e-3, 1e-3)
    cf, qf, qm = fishers_fn(None, flow.unravel(pb["theta"], pb["x"].shape[-1]), pb["sidx"], pb["x"])
    assert np.abs(r0["cf"] - cf).max() < 1e-12 * np.abs(cf).max()
    assert np.abs(r0["qf"] - qf).max() < 1e-10 * np.abs(qf).max() and np.abs(r0["qm"] - qm).max() < 1e-10 * np.abs(qm).max()



@pytest.mark.timeout(600)
def test_checkpoint_at_world_3_holds_every_ranks_walkers_bit_for_bit(tmp_path):
    """main.py:374-381 at a world size that is not a power of two: the gather behind train(ckpt_path=...) is a SUM all-reduce of
    zero-padded slots (comm.allgather), so slot r of the file is rank r's array bit for bit (a mean all-reduce of world * x is
    not, at world 3), and a resumed run picks its own slot up again."""
    import coulombgas_amd as cg
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    port = 31500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path), "ckpt"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ck = cg.load_data(cg.ckpt_filename(2, str(tmp_path / "ck")))
    assert np.asarray(ck["x"]).shape == (3, 4, 4, 2)
    for rank in range(3):
        mine = np.load(tmp_path / ("ckrank%d.npz" % rank))
        assert np.array_equal(np.asarray(ck["x"])[rank], mine["x"]), rank            # bit for bit
        assert np.array_equal(np.asarray(ck["keys"])[rank], mine["keys"].astype(np.uint32))
    assert not np.array_equal(np.asarray(ck["x"])[0], np.asarray(ck["x"])[1])
    rows = [np.load(tmp_path / ("ckresume%d.npz" % rank))["rows"] for rank in range(3)]
    assert all(len(rw) == 1 and rw[0].split()[0] == "3" for rw in rows) and rows[0][0] == rows[1][0] == rows[2][0]
    # the inexact form the gather replaced, for the record: a mean all-reduce of world * x rounds twice at world 3
    x = np.asarray(ck["x"])[1]
    assert not np.array_equal((x * 3.0) / 3.0, x)


def _rdzv_rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port - 1), CG_RDZV_PORT=str(port))   # the port that was probed free is the one listened on
    from coulombgas_amd.comm import tcp_broadcast_bytes
    out = []
    for k in range(2):                                   # two exchanges in a row (a second communicator): sequence numbers keep them apart
        payload = bytes((i + k) % 256 for i in range(128)) if rank == 0 else None
        out.append(tcp_broadcast_bytes(payload, rank, world, timeout=60.0))
    q.put((rank, out))


def _stray_client(port, stop):
    # a port scan / a rank of another job: connects, sends garbage or a hello with the wrong token, never one of this job's ranks
    import socket, struct, time
    k = 0
    while not stop.is_set():
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=1.0) as s:
                s.settimeout(1.0)
                s.sendall(b"GET / HTTP/1.0\r\n\r\n" if k % 2 else b"CGID" + b"x" * 16 + struct.pack("<ii", 1, 0))
                try:
                    s.recv(128)
                except OSError:
                    pass
        except OSError:
            pass
        k += 1
        time.sleep(0.02)


@pytest.mark.timeout(120)
def test_rccl_id_exchange_over_tcp_world3():
    """the rendezvous of coulombgas_amd.comm.RcclComm (rank 0 serves the 128-byte id on its rendezvous port to the peers that
    present the job token, their rank and the exchange's sequence number; the peers retry until it is up) between three processes
    on the CPU, twice in a row, while a stray client keeps connecting with garbage and with another job's token: every rank ends
    with rank 0's bytes of each exchange; no torch anywhere"""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    stop = ctx.Event()
    stray = ctx.Process(target=_stray_client, args=(port, stop))
    stray.start()
    world = 3
    procs = [ctx.Process(target=_rdzv_rank, args=(r, world, port, q)) for r in (2, 1, 0)]      # the peers start first
    for p in procs:
        p.start()
    try:
        got = dict(q.get(timeout=100) for _ in range(world))
    finally:
        stop.set()
    for p in procs + [stray]:
        p.join(30); assert p.exitcode == 0
    for k in range(2):
        assert all(got[r][k] == bytes((i + k) % 256 for i in range(128)) for r in range(world))
    import coulombgas_amd.comm as cm
    assert "import torch" not in open(cm.__file__).read()
