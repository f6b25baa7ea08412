"""torch.distributed (gloo) communicator of the multi-process CPU tests (tests/test_distributed_gloo.py, tests/dist_worker.py): the
same pmean / pmean_d interface as coulombgas_amd.comm.RcclComm on the numpy handles of the CPU test engine.  Test infrastructure:
the package itself does not import torch."""
import numpy as np


class TorchDistComm:
    def __init__(self, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device          # None: host tensors (gloo); "cuda": nccl(=RCCL) through torch

    def pmean(self, a):
        import torch
        t = torch.from_numpy(np.array(a, dtype=np.float64, ndmin=1, copy=True))
        if self.device is not None:
            t = t.to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            out = (t / self.world).cpu().numpy()
            return out.reshape(np.shape(a)) if np.ndim(a) else float(out[0])
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        out = (t / self.world).numpy()
        return out.reshape(np.shape(a)) if np.ndim(a) else float(out[0])

    def psum(self, a):
        import torch
        t = torch.from_numpy(np.array(a, dtype=np.float64, ndmin=1, copy=True))
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        out = t.cpu().numpy()
        return out.reshape(np.shape(a)) if np.ndim(a) else float(out[0])

    def pmean_d(self, a, count=None, index=0):
        """array handles of the CPU test engine are numpy arrays: all-reduce in place"""
        if not isinstance(a, np.ndarray):
            raise TypeError("TorchDistComm.pmean_d works on the numpy handles of the CPU test engine; device arrays of a GPU "
                            "engine are reduced by RcclComm (got %s)" % type(a).__name__)
        flat = a.reshape(-1)
        if index < 0 or (count is not None and index + count > flat.size):
            raise IndexError("pmean_d: [%d, %d) outside an array of %d elements" % (index, index + (count or 0), flat.size))
        n = flat.size - index if count is None else count
        flat[index:index + n] = self.pmean(flat[index:index + n])
        return a

    def close(self):
        pass

    def accept_rate(self, engine, denom):
        return self.pmean(engine.mcmc_accepts() / float(denom) if denom else 0.0)

    def pmax(self, v):
        from coulombgas_amd.comm import allgather
        return float(np.max(allgather(self, np.array([float(v)]))))

    def barrier(self):
        self.dist.barrier()
