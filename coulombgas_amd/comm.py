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

    def psum(self, a):
        return a

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

    def _reduce_host(self, a, fn):
        from ._lib import lib, check
        arr = np.array(a, dtype=np.float64, ndmin=1, copy=True)
        flat = np.ascontiguousarray(arr.ravel())
        if self._buf is None or self._buf.nbytes < flat.nbytes:
            self._buf = self.engine.alloc((max(flat.size, 16),))
        check(lib().cg_memcpy_h2d(self.engine._ctx, self._buf.ptr, flat.ctypes.data_as(C.c_void_p), flat.nbytes), self.engine._ctx)
        check(fn(self._h, self._buf.ptr, flat.size), self.engine._ctx)
        check(lib().cg_memcpy_d2h(self.engine._ctx, flat.ctypes.data_as(C.c_void_p), self._buf.ptr, flat.nbytes), self.engine._ctx)
        out = flat.reshape(arr.shape)
        return out.reshape(np.shape(a)) if np.ndim(a) else float(out[0])

    def pmean(self, a):
        from ._lib import lib
        return self._reduce_host(a, lib().cg_allreduce_mean)

    def psum(self, a):
        """sum over the ranks of a host array (staged through a device buffer); exact where all but one rank contribute zeros"""
        from ._lib import lib
        return self._reduce_host(a, lib().cg_allreduce_sum)

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
        """src/MCMC.py:37-39: the rate is formed from the device counter and averaged over the ranks on the device (8 bytes come back).
        An engine other than the one this communicator is bound to (a second size in the same process): its count goes through the
        host-staged mean instead -- the device-side path needs the communicator's own context."""
        if not denom:
            return 0.0
        if engine is not self.engine:
            return float(self.pmean(engine.mcmc_accepts() / float(denom)))
        return engine.mcmc_accept_rate(denom, self._h)

    def pmax(self, v):
        """maximum of a host scalar over the ranks (e.g. the elapsed time of a benchmark): exact, every rank's value comes back bit for bit"""
        return float(np.max(allgather(self, np.array([float(v)]))))

    def barrier(self):
        """every rank has reached this point and its stream is drained: a one-element all-reduce"""
        self.pmean(0.0)

    def close(self):
        if self._h is not None:
            from ._lib import lib
            lib().cg_comm_destroy(self._h)
            self._h = None


_RDZV_SEQ = [0]          # exchanges this process has taken part in: the second communicator of a job does not answer the first one's peers


def _rdzv_token(world):
    """16 bytes every rank of ONE job derives alike and another job does not: CG_RDZV_TOKEN, else the launcher's run id, else the
    rendezvous address itself"""
    import hashlib
    key = os.environ.get("CG_RDZV_TOKEN") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    key += "|%s|%s|%d" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29500"), world)
    return hashlib.sha256(key.encode()).digest()[:16]


def tcp_broadcast_bytes(payload, rank, world, addr=None, port=None, timeout=300.0, seq=None):
    """Rank 0 hands `payload` (the 128-byte RCCL id) to the other ranks of the job.  It listens on MASTER_ADDR : MASTER_PORT + 1 (or
    CG_RDZV_PORT; MASTER_PORT itself belongs to the launcher's own store under torch.distributed.run); a peer opens with a 28-byte
    hello -- magic, the job token (_rdzv_token), its rank, the sequence number of the exchange -- and rank 0 answers every distinct
    rank 1 ... world - 1 exactly once and drops anything else (a port scan, a rank of another job on an adjacent port, a fast rank
    already in its next exchange), until all peers are served or `timeout` expires.  The peers retry until rank 0 is up and read
    with a timeout of their own.  Returns the payload on every rank."""
    import socket, struct, time
    if world == 1:
        return payload
    if seq is None:
        seq = _RDZV_SEQ[0]
        _RDZV_SEQ[0] += 1
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or os.environ.get("CG_RDZV_PORT") or int(os.environ.get("MASTER_PORT", "29500")) + 1)
    token = _rdzv_token(world)
    hello_len = 4 + 16 + 4 + 4
    deadline = time.time() + timeout
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            try:
                srv.bind((addr, port))                    # the rendezvous address only, not every interface
            except OSError:
                srv.bind(("", port))                      # (MASTER_ADDR is not an address of this host's own: a NAT'ed launcher)
        except OSError as e:
            srv.close()
            raise RuntimeError("coulombgas_amd.comm: cannot listen on port %d for the RCCL id exchange (%s); set CG_RDZV_PORT" % (port, e))
        srv.listen(max(world, 8))
        served = set()
        try:
            while len(served) < world - 1:
                left = deadline - time.time()
                if left <= 0:
                    raise RuntimeError("coulombgas_amd.comm: id exchange %d timed out with ranks %s still missing"
                                       % (seq, sorted(set(range(1, world)) - served)))
                srv.settimeout(left)
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    conn.settimeout(5.0)
                    try:
                        hello = b""
                        while len(hello) < hello_len:
                            chunk = conn.recv(hello_len - len(hello))
                            if not chunk:
                                break
                            hello += chunk
                        if len(hello) != hello_len or hello[:4] != b"CGID" or hello[4:20] != token:
                            continue                      # not one of this job's ranks
                        r, q = struct.unpack("<ii", hello[20:])
                        if q != seq or r < 1 or r >= world or r in served:
                            continue                      # another exchange / a rank that has its bytes already
                        conn.sendall(payload)
                        served.add(r)
                    except OSError:
                        continue
        finally:
            srv.close()
        return payload
    want = 128
    while True:
        if time.time() > deadline:
            raise RuntimeError("coulombgas_amd.comm: rank %d did not get the RCCL id from rank 0 at %s:%d (exchange %d)" % (rank, addr, port, seq))
        try:
            s = socket.create_connection((addr, port), timeout=10.0)
        except OSError:
            time.sleep(0.05)
            continue
        buf = b""
        try:
            with s:
                s.settimeout(30.0)
                s.sendall(b"CGID" + token + struct.pack("<ii", rank, seq))
                while len(buf) < want:
                    chunk = s.recv(want - len(buf))
                    if not chunk:
                        break
                    buf += chunk
        except OSError:
            buf = b""
        if len(buf) == want:
            return buf
        time.sleep(0.05)              # dropped (rank 0 still in an earlier exchange, or not rank 0 at all): again


def allgather(comm, a):
    """(world, ...) array of every rank's `a` (same shape on all ranks): rank r puts `a` in slot r of a zero-filled buffer and the
    buffers are SUMMED over the ranks -- x + 0 + ... + 0 is x bit for bit at any world size (a mean all-reduce of world * a is not:
    it rounds twice unless the world is a power of two).  Communicators without `psum` fall back to that form for power-of-two
    worlds only.  Used off the hot path (checkpoints, main.py:374-381; timing maxima)."""
    a = np.asarray(a, dtype=np.float64)
    if comm.world == 1:
        return a[None]
    buf = np.zeros((comm.world,) + a.shape)
    if hasattr(comm, "psum"):
        buf[comm.rank] = a
        return np.asarray(comm.psum(buf)).reshape(buf.shape)
    if comm.world & (comm.world - 1):
        raise NotImplementedError("allgather over %d ranks needs a communicator with psum (an exact sum all-reduce)" % comm.world)
    buf[comm.rank] = a * comm.world
    return np.asarray(comm.pmean(buf)).reshape(buf.shape)


_COMM = NullComm()


def set_comm(comm):
    global _COMM
    _COMM = comm if comm is not None else NullComm()


def get_comm():
    return _COMM
