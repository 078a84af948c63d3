"""Multi-GPU plumbing of the scoring path without PyTorch: one process per GPU of ONE node, RCCL for the data.

The path shards by locus (sharding.py) and has a single exchange step, the gather of the result records to rank 0
(SURVEY.md section 8e).  That gather runs device to device over xGMI through the library's C ABI
(`advntr_comm_*`, csrc/abi_comm.h: grouped ncclSend / ncclRecv).  What RCCL needs from the host is the 128-byte
unique id handed from rank 0 to the others; `FileRendezvous` does that through a directory under /tmp (the ranks of
one node share a file system), keyed by the launcher's MASTER_PORT and process id, so it works the same under
`python -m torch.distributed.run` (which only sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and starts the
processes -- nothing of torch is imported here) and under bench.py's own spawner.

Two communicators with one interface (rank, world, barrier, allreduce_max, allgather_i64, gather_bytes):
  RcclComm  -- the product path on a multi-GPU node; also gathers a DeviceBatch's records without a host round trip;
  HostComm  -- the same calls over the rendezvous directory only: for CPU-side tests of the sharding logic and for
               several ranks sharing one GPU (a single-GPU test box), where RCCL refuses two ranks on one device.

The reference has no counterpart (serial loop over loci, /root/reference/advntr/genome_analyzer.py:280-297).
"""
import ctypes
import os
import shutil
import struct
import time

import numpy as np


class FileRendezvous(object):
    """Small blobs between the ranks of one node through files: put / get by name, all-gather, barrier."""

    def __init__(self, rank, world, directory=None, timeout=900.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if directory is None:
            directory = os.environ.get("ADVNTR_RDZV_DIR")
        if directory is None:
            key = "%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), os.environ.get("TORCHELASTIC_RUN_ID", "x"))
            directory = os.path.join("/tmp", "advntr_rdzv_%d_%s" % (os.getuid(), key))
        self.dir = directory
        os.makedirs(self.dir, exist_ok=True)
        self._seq = 0

    def put(self, name, data):
        tmp = os.path.join(self.dir, ".%s.%d.tmp" % (name, self.rank))
        with open(tmp, "wb") as fh:
            fh.write(data)
        os.rename(tmp, os.path.join(self.dir, name))           # atomic: a reader never sees a partial file

    def get(self, name):
        path = os.path.join(self.dir, name)
        t0 = time.monotonic()
        delay = 0.0002
        while True:
            try:
                with open(path, "rb") as fh:
                    return fh.read()
            except FileNotFoundError:
                if time.monotonic() - t0 > self.timeout:
                    raise TimeoutError("rendezvous: rank %d waited %.0f s for %s" % (self.rank, self.timeout, path))
                time.sleep(delay)
                delay = min(delay * 1.5, 0.005)

    def allgather(self, data):
        """Every rank contributes bytes; every rank gets the list in rank order.  Collective: same call order on all ranks."""
        self._seq += 1
        self.put("ag%d.%d" % (self._seq, self.rank), data)
        return [self.get("ag%d.%d" % (self._seq, r)) for r in range(self.world)]

    def gather(self, data, root=0):
        self._seq += 1
        self.put("g%d.%d" % (self._seq, self.rank), data)
        if self.rank != root:
            return None
        return [self.get("g%d.%d" % (self._seq, r)) for r in range(self.world)]

    def broadcast(self, data, root=0):
        self._seq += 1
        if self.rank == root:
            self.put("b%d" % self._seq, data)
            return data
        return self.get("b%d" % self._seq)

    def barrier(self):
        self.allgather(b"")

    def close(self):
        """Collective.  Rank 0 removes the directory once every rank has said goodbye."""
        try:
            self.put("bye.%d" % self.rank, b"")
            if self.rank == 0:
                for r in range(self.world):
                    self.get("bye.%d" % r)
                shutil.rmtree(self.dir, ignore_errors=True)
        except (OSError, TimeoutError):
            pass


class HostComm(object):
    """The communicator interface over the rendezvous directory only (no GPU involved)."""
    backend = "host"
    fallback_reason = None

    def __init__(self, rdzv):
        self.rdzv, self.rank, self.world = rdzv, rdzv.rank, rdzv.world

    def barrier(self):
        self.rdzv.barrier()

    def allreduce_max(self, x):
        return max(struct.unpack("<d", b)[0] for b in self.rdzv.allgather(struct.pack("<d", float(x))))

    def allgather_i64(self, x):
        return [struct.unpack("<q", b)[0] for b in self.rdzv.allgather(struct.pack("<q", int(x)))]

    def gather_bytes(self, data, root=0):
        """list of every rank's bytes on `root` (rank order), None elsewhere."""
        return self.rdzv.gather(bytes(data), root)

    def gather_results(self, batch, counts, root=0):
        """The records of a device batch to `root` through host memory and the rendezvous directory (what RcclComm does
        device to device): (logp, summary) of all ranks in rank order on root, (None, None) elsewhere."""
        logp, summ = batch.fetch()
        parts = self.rdzv.gather(logp.tobytes() + summ.tobytes(), root)
        if self.rank != root:
            return None, None
        lps, sms = [], []
        for p, c in zip(parts, counts):
            c = int(c)
            lps.append(np.frombuffer(p[:8 * c], np.float64))
            sms.append(np.frombuffer(p[8 * c:], np.int32).reshape(c, -1))
        return np.concatenate(lps), np.concatenate(sms)

    def close(self):
        self.rdzv.close()


class RcclComm(HostComm):
    """RCCL communicator of the library (advntr_comm_*), one rank per GPU; the current device must be set before."""
    backend = "rccl"

    def __init__(self, rdzv):
        from . import _lib
        HostComm.__init__(self, rdzv)
        self._lib = _lib
        L = _lib.load()
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(L.advntr_comm_unique_id(ctypes.addressof(uid)))
        blob = rdzv.broadcast(uid.raw, 0)
        self._h = L.advntr_comm_create(self.rank, self.world, blob)
        if not self._h:
            raise _lib.EngineError(_lib.ERR_DEVICE, _lib.last_error())

    def barrier(self):
        self._lib.check(self._lib.load().advntr_comm_barrier(self._h))

    def allreduce_max(self, x):
        v = ctypes.c_double(float(x))
        self._lib.check(self._lib.load().advntr_comm_allreduce_max_f64(self._h, ctypes.byref(v)))
        return v.value

    def allgather_i64(self, x):
        out = np.zeros(self.world, np.int64)
        self._lib.check(self._lib.load().advntr_comm_allgather_i64(self._h, int(x), out.ctypes.data))
        return [int(v) for v in out]

    def gather_bytes(self, data, root=0):
        data = bytes(data)
        counts = np.array(self.allgather_i64(len(data)), np.int64)
        dst = ctypes.create_string_buffer(max(int(counts.sum()), 1)) if self.rank == root else None
        self._lib.check(self._lib.load().advntr_comm_gather_bytes(
            self._h, root, data if data else None, counts.ctypes.data, ctypes.addressof(dst) if dst is not None else None))
        if self.rank != root:
            return None
        out, at = [], 0
        for c in counts:
            out.append(dst.raw[at:at + int(c)])
            at += int(c)
        return out

    def gather_results_start(self, batch, counts, root=0):
        """Queue the gather of `batch`'s records behind the launches issued so far; returns immediately."""
        self._counts = np.ascontiguousarray(counts, np.int64)
        self._root = root
        self._lib.check(self._lib.load().advntr_comm_gather_results_start(self._h, batch._h, root, self._counts.ctypes.data))

    def gather_results_finish(self, fetch=True):
        """Wait for the gather; on the root (logp, summary) of all ranks in rank order when fetch is set."""
        logp = summ = None
        if self.rank == self._root and fetch:
            total = int(self._counts.sum())
            logp = np.zeros(total, np.float64)
            summ = np.zeros((total, self._lib.SUMMARY_INTS), np.int32)
        self._lib.check(self._lib.load().advntr_comm_gather_results_finish(
            self._h, None if logp is None else logp.ctypes.data, None if summ is None else summ.ctypes.data))
        return logp, summ

    def close(self):
        if getattr(self, "_h", None):
            self._lib.load().advntr_comm_destroy(self._h)
            self._h = None
        HostComm.close(self)


def env_world():
    """(rank, local_rank, world) from the launcher's environment; (0, 0, 1) for a plain single process."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend=None, set_device=True):
    """Join the job the launcher described (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT): bind this rank's GPU and
    return a communicator -- RcclComm by default, HostComm with backend="host" (or ADVNTR_DIST_BACKEND=host).  Returns
    None for a single process that was not started by a launcher."""
    rank, local, world = env_world()
    if world <= 1 and "RANK" not in os.environ:
        return None
    backend = backend or os.environ.get("ADVNTR_DIST_BACKEND", "rccl")
    if backend not in ("rccl", "host"):
        raise ValueError("backend must be 'rccl' or 'host', not %r" % (backend,))
    if set_device:
        from . import _lib
        n = _lib.load().advntr_device_count()
        if n > 0:
            # RCCL wants one rank per GPU; the host backend lets ranks share the GPUs that exist
            if backend == "rccl" and local >= n:
                raise _lib.EngineError(_lib.ERR_DEVICE, "LOCAL_RANK %d but only %d GPUs are visible" % (local, n))
            _lib.check(_lib.load().advntr_set_device(local % n))
    rdzv = FileRendezvous(rank, world)
    if backend == "host":
        return HostComm(rdzv)
    # RCCL, agreed on by all ranks: if the communicator cannot be created on some rank (driver / IPC configuration), every
    # rank drops to the host communicator together instead of leaving the others waiting in a collective.  The choice is
    # visible in `.backend` and in `.fallback_reason`; ADVNTR_COMM_FALLBACK=0 turns the fallback into an error.
    c, err = None, b""
    try:
        c = RcclComm(rdzv)
    except Exception as e:          # noqa: BLE001 -- whatever went wrong is reported, not swallowed
        err = ("rank %d: %s" % (rank, e)).encode("utf-8", "replace")
    failures = [x for x in rdzv.allgather(err) if x]
    if not failures:
        return c
    if c is not None:
        c._lib.load().advntr_comm_destroy(c._h)
        c._h = None
    reason = failures[0].decode("utf-8", "replace")
    if os.environ.get("ADVNTR_COMM_FALLBACK", "1") == "0":
        raise RuntimeError("RCCL communicator could not be created: " + reason)
    import sys
    if rank == 0:
        sys.stderr.write("advntr_amd.comm: RCCL unavailable (%s); the result gather goes through the host communicator\n" % reason)
    h = HostComm(rdzv)
    h.fallback_reason = reason
    return h
