"""Batch-axis mean over the GPUs of a node: the counterpart of jax.lax.pmean(axis_name="p")
(src/MCMC.py:39, src/VMC.py:46-53,63,72, main.py:280).  One process per GPU.

  NullComm        world size 1 (identity)
  RcclComm        RCCL all-reduce over xGMI through the C-ABI (cg_allreduce_mean); the 128-byte unique id is
                  exchanged through torch.distributed's store/broadcast (plumbing only)
  TorchDistComm   torch.distributed all_reduce on host tensors (gloo) -- used by the CPU tests of the
                  multi-process host logic
"""
import ctypes as C
import os
import numpy as np


class NullComm:
    rank, world = 0, 1

    def pmean(self, a):
        return a

    def pmean_d(self, a, count=None, index=0):
        """in-place mean over the ranks of a device-resident array (or of `count` elements from `index`): nothing to do"""
        return a

    def close(self):
        pass


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


class RcclComm:
    """RCCL communicator bound to an Engine's device/stream."""

    def __init__(self, engine, rank, world, unique_id=None, exchange=None):
        from ._lib import lib, check
        self.engine, self.rank, self.world = engine, rank, world
        uid = (C.c_char * 128)()
        if unique_id is None:
            if rank == 0:
                check(lib().cg_comm_unique_id(uid), None)
            if world == 1 and exchange is None and os.environ.get("CG_FORCE_DIST") != "1":
                raw = bytes(uid.raw)
            else:
                if exchange is None:
                    exchange = _torch_broadcast_bytes
                raw = exchange(bytes(uid.raw) if rank == 0 else None)
            uid = (C.c_char * 128).from_buffer_copy(raw)
        else:
            uid = (C.c_char * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        check(lib().cg_comm_create(C.byref(h), engine._ctx, rank, world, uid), None)
        self._h = h
        self._buf = None

    def pmean(self, a):
        from ._lib import lib, check
        arr = np.array(a, dtype=np.float64, ndmin=1, copy=True)
        flat = np.ascontiguousarray(arr.ravel())
        if self._buf is None or self._buf.nbytes < flat.nbytes:
            self._buf = self.engine.alloc((max(flat.size, 16),))
        check(lib().cg_memcpy_h2d(self.engine._ctx, self._buf.ptr, flat.ctypes.data_as(C.c_void_p), flat.nbytes), self.engine._ctx)
        check(lib().cg_allreduce_mean(self._h, self._buf.ptr, flat.size), self.engine._ctx)
        check(lib().cg_memcpy_d2h(self.engine._ctx, flat.ctypes.data_as(C.c_void_p), self._buf.ptr, flat.nbytes), self.engine._ctx)
        out = flat.reshape(arr.shape)
        return out.reshape(np.shape(a)) if np.ndim(a) else float(out[0])

    def pmean_d(self, a, count=None, index=0):
        """in-place RCCL all-reduce (mean) of a DeviceArray, or of `count` doubles from element `index` of its buffer, on the
        engine's stream: no host staging (main.py:280, src/VMC.py:46-53, src/sr.py:73-82)"""
        from ._lib import lib, check
        base = a.base                                        # a DeviceArray is its own base (index 0); a DeviceView a window
        total = base.size * (2 if base.complex_pairs else 1) - a.index        # doubles from a's first element to the buffer's end
        n = (a.size * (2 if a.complex_pairs else 1) - index) if count is None else count
        if index < 0 or n < 0 or index + n > total:
            raise IndexError("pmean_d: [%d, %d) outside a device buffer of %d doubles" % (index, index + n, total))
        check(lib().cg_allreduce_mean(self._h, a.ptr_at(index), int(n)), self.engine._ctx)
        a.version += 1
        return a

    def close(self):
        if self._h is not None:
            from ._lib import lib
            lib().cg_comm_destroy(self._h)
            self._h = None


def _torch_broadcast_bytes(payload):
    import torch
    import torch.distributed as dist
    t = torch.zeros(128, dtype=torch.uint8)
    if dist.get_rank() == 0:
        t = torch.tensor(list(payload), dtype=torch.uint8)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist())


def allgather(comm, a):
    """(world, ...) array of every rank's `a` (same shape on all ranks), through the mean all-reduce the communicators
    provide: rank r contributes world * a in slot r.  Used off the hot path only (checkpoints, main.py:374-381)."""
    a = np.asarray(a, dtype=np.float64)
    if comm.world == 1:
        return a[None]
    buf = np.zeros((comm.world,) + a.shape)
    buf[comm.rank] = a * comm.world
    return np.asarray(comm.pmean(buf)).reshape(buf.shape)


_COMM = NullComm()


def set_comm(comm):
    global _COMM
    _COMM = comm if comm is not None else NullComm()


def get_comm():
    return _COMM
