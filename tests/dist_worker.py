"""Worker for tests/test_distributed_gloo.py: one rank of a world_size-N gloo job (CPU).  Each rank owns a shard of
the walker batch (main.py:231-236), runs the host logic with the emulated engine and the TorchDistComm, and writes
what it computed; rank 0's parent compares with the single-process result on the full batch."""
import json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ckpt_main(out_dir):
    """mode "ckpt": train() with a checkpoint at a world size that is not a power of two (main.py:374-381): every rank records the
    walkers it hands to the gather; the parent compares them with the file bit for bit and resumes from it"""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import coulombgas_amd as cg
    import coulombgas_amd.driver as drv
    from coulombgas_amd.comm import set_comm
    from tests.torch_comm import TorchDistComm
    from tests import emul_engine
    from tests.common import orbitals

    class MP:
        def setattr(self, obj, name, val):
            setattr(obj, name, val)
    emul_engine.install(MP())
    comm = TorchDistComm()
    set_comm(comm)
    seen = []
    real = drv.allgather
    drv.allgather = lambda cm, a: (seen.append(np.array(a, copy=True)), real(cm, a))[1]
    n, dim, L = 4, 2, 2.0
    sp = orbitals(dim)
    flow = cg.FermiNet(2, 4, 4, L)
    p0 = flow.init(3, np.zeros((n, dim)))
    samp = cg.GroundStateSampler(n, sp.shape[0])
    kw = dict(rs=2.0, beta=1 / (4 * 0.15), batch=4, sampler=samp, log_prob=samp.log_prob, mc_therm=1, mc_steps=3, seed=1,
              ckpt_path=os.path.join(out_dir, "ck"), ckpt_every=2, comm=comm)
    cg.train(flow, p0, sp, n, dim, L, epochs=2, optimizer=cg.adam(1e-2), **kw)
    np.savez(os.path.join(out_dir, "ckrank%d.npz" % rank), x=seen[0], keys=seen[1])
    dist.barrier()
    seen.clear()
    _, _, rows = cg.train(flow, p0, sp, n, dim, L, epochs=3, optimizer=cg.adam(1e-2), epoch_finished=2, **kw)     # resume: rank r continues with slot r
    np.savez(os.path.join(out_dir, "ckresume%d.npz" % rank), rows=np.array(rows))
    dist.barrier()
    dist.destroy_process_group()


def main():
    out_dir = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "ckpt":
        return ckpt_main(out_dir)
    import torch.distributed as dist
    dist.init_process_group("gloo")           # MASTER_ADDR=127.0.0.1 from the launcher
    rank, world = dist.get_rank(), dist.get_world_size()
    import coulombgas_amd as cg
    import coulombgas_amd.flow as fl
    from coulombgas_amd.comm import set_comm
    from tests.torch_comm import TorchDistComm
    from tests import emul_engine
    from tests.test_host_logic import _problem, build_loss

    class MP:       # minimal monkeypatch stand-in
        def setattr(self, obj, name, val):
            setattr(obj, name, val)
    emul_engine.install(MP())
    comm = TorchDistComm()
    set_comm(comm)
    pb = _problem(B=8)
    B = pb["x"].shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)                                   # this rank's walkers
    obs_fn, G, Vconst = build_loss(pb)                                     # uses get_comm()
    obs, closs, qloss = obs_fn(pb["logp_states"][sl], pb["theta"], pb["sidx"][sl], pb["x"][sl], pb["v"][sl])
    qv = comm.pmean(np.array(qloss(pb["theta"])))                      # per-device means, pmean'd like main.py:280
    g, s = qloss.grad(pb["theta"], as_pytree=False, reduce=True)           # main.py:278 + the packed pmean of :280 (the driver's path)
    # sampling call: distinct streams per rank, pmean'd accept rate (src/MCMC.py:39)
    flow = cg.FermiNet(2, 16, 16, pb["L"])
    logp = cg.make_logp(cg.make_logpsi(flow, pb["sp"], pb["L"]))
    key = np.random.SeedSequence(7).spawn(world)[rank]                      # jax.random.split(key, num_devices), main.py:237
    _, sidx, x, rate = cg.sample_stateindices_and_x(key, lambda pv, k, b: pb["sidx"][sl], None, logp, pb["x"][sl], pb["theta"], 4, 0.1, pb["L"])
    # SR Fisher matrices: per-rank blocks all-reduced in fishers_fn (src/sr.py:70-76)
    qscore = cg.make_quantum_score(cg.make_logpsi(flow, pb["sp"], pb["L"]))
    cs_full = np.random.default_rng(99).standard_normal((pb["x"].shape[0], 5))
    fishers_fn, _ = cg.hybrid_fisher_sr(lambda pv, si: cs_full[sl], qscore, 1e-3, 1e-3)
    cf, qf, qm = fishers_fn(None, flow.unravel(pb["theta"], pb["x"].shape[-1]), pb["sidx"][sl], pb["x"][sl])
    cf, qf = np.array(cf), np.array(qf)                                    # (the engine's buffers are overwritten by the next call)
    # two accumulation steps: all-reducing each step's matrices (src/sr.py:70-76 as written) against accumulating this rank's matrices
    # and all-reducing the sums once (make_update at acc_steps > 1)
    pf = flow.unravel(pb["theta"], pb["x"].shape[-1])
    xb = pb["x"][sl] + 0.05
    copy3 = lambda f: (np.array(f[0]), np.array(f[1]), np.array(f[2]))
    e0 = copy3(fishers_fn(None, pf, pb["sidx"][sl], pb["x"][sl])); e1 = copy3(fishers_fn(None, pf, pb["sidx"][sl], xb))
    each = [a + b for a, b in zip(e0, e1)]
    l0 = copy3(fishers_fn(None, pf, pb["sidx"][sl], pb["x"][sl], reduce=False)); l1 = copy3(fishers_fn(None, pf, pb["sidx"][sl], xb, reduce=False))
    once = fishers_fn.reduce_accumulated(tuple(a + b for a, b in zip(l0, l1)))
    defer_err = max(float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(a)).max()) for a, b in zip(each, once))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), obs=np.array([obs[k] for k in sorted(obs)]), qv=np.array(qv), g=g, s=s,
             rate=rate, x=x, tvE=float(np.abs(obs_fn.Eloc - obs["E_mean"]).mean()), cf=cf, qf=qf, qm=qm, defer_err=defer_err)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
