"""Batch-axis mean over the GPUs of a node: the counterpart of jax.lax.pmean(axis_name="p")
(src/MCMC.py:39, src/VMC.py:46-53,63,72, main.py:280).  One process per GPU.

  NullComm        world size 1 (identity)
  RcclComm        RCCL all-reduce over xGMI through the C-ABI (cg_allreduce_mean).  The 128-byte unique id travels from rank 0
                  to the others in one TCP exchange on MASTER_ADDR : MASTER_PORT + 1 (tcp_broadcast_bytes); barrier and maxima
                  go through the same all-reduce.  Nothing here imports torch: the reference needs nothing but its array
                  library for the collectives (src/utils.py:4-8, main.py:214-239) and neither does this -- a launcher such as
                  torch.distributed.run only provides RANK / WORLD_SIZE / MASTER_* in the environment.
(the gloo communicator of the multi-process CPU tests lives in tests/torch_comm.py)
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

    def accept_rate(self, engine, denom):
        """accepted moves of the engine's last chain / denom (src/MCMC.py:37-39)"""
        return engine.mcmc_accepts() / float(denom) if denom else 0.0

    def pmax(self, v):
        return float(v)

    def barrier(self):
        pass

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
                    exchange = lambda payload: tcp_broadcast_bytes(payload, rank, world)
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

    def accept_rate(self, engine, denom):
        """src/MCMC.py:37-39: the rate is formed from the device counter and averaged over the ranks on the device (8 bytes come back)"""
        return engine.mcmc_accept_rate(denom, self._h) if denom else 0.0

    def pmax(self, v):
        """maximum of a host scalar over the ranks (e.g. the elapsed time of a benchmark)"""
        return float(np.max(allgather(self, np.array([float(v)]))))

    def barrier(self):
        """every rank has reached this point and its stream is drained: a one-element all-reduce"""
        self.pmean(0.0)

    def close(self):
        if self._h is not None:
            from ._lib import lib
            lib().cg_comm_destroy(self._h)
            self._h = None


def tcp_broadcast_bytes(payload, rank, world, addr=None, port=None, timeout=300.0):
    """Rank 0 hands `payload` (the 128-byte RCCL id) to the other ranks of the job: it listens on port MASTER_PORT + 1 (or
    CG_RDZV_PORT; MASTER_PORT itself belongs to the launcher's own store under torch.distributed.run) and sends the bytes to each
    of the world - 1 peers that connect; the peers retry until it is up.  Returns the payload on every rank."""
    import socket, time
    if world == 1:
        return payload
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ.get("CG_RDZV_PORT") or int(os.environ.get("MASTER_PORT", "29500")) + 1)
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            srv.bind(("", port))
        except OSError as e:
            raise RuntimeError("coulombgas_amd.comm: cannot listen on port %d for the RCCL id exchange (%s); set CG_RDZV_PORT" % (port, e))
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(payload)
        finally:
            srv.close()
        return payload
    deadline = time.time() + timeout
    while True:
        try:
            s = socket.create_connection((addr, port), timeout=10.0)
            break
        except OSError:
            if time.time() > deadline:
                raise RuntimeError("coulombgas_amd.comm: rank %d could not reach rank 0 at %s:%d" % (rank, addr, port))
            time.sleep(0.05)
    buf = b""
    with s:
        while len(buf) < len(payload or b"") or (payload is None and len(buf) < 128):
            chunk = s.recv(128 - len(buf))
            if not chunk:
                raise RuntimeError("coulombgas_amd.comm: rank 0 closed the id exchange early (%d of 128 bytes)" % len(buf))
            buf += chunk
    return buf


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
