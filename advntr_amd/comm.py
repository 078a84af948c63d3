"""Multi-GPU plumbing of the scoring path without PyTorch: one process per GPU of ONE node, RCCL for the data.

The path shards by locus (sharding.py) and has a single exchange step, the gather of the result records to rank 0
(SURVEY.md section 8e).  That gather runs device to device over xGMI through the library's C ABI
(`advntr_comm_*`, csrc/abi_comm.h: grouped ncclSend / ncclRecv).  What RCCL needs from the host is the 128-byte
unique id handed from rank 0 to the others; `FileRendezvous` does that through a directory under /tmp (the ranks of
one node share a file system), named after the launch (launcher process and its start time, MASTER_PORT, run id, restart count) and closed to other users, so it works the same under
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

SUMMARY_INTS = 8            # int32 words of a result record's summary (include/advntr_hip.h: ADVNTR_SUMMARY_INTS)


def _launcher_identity():
    """What tells one launch of the ranks from another on this host: the launcher's port and run id, its restart count
    (a torchrun agent that restarts its workers keeps port, run id and its own process), and the launcher PROCESS --
    its pid together with its start time, so that a recycled pid is somebody else."""
    ppid = os.getppid()
    started = "x"
    try:
        with open("/proc/%d/stat" % ppid) as fh:
            started = fh.read().rsplit(")", 1)[1].split()[19]            # field 22: starttime (clock ticks since boot)
    except (OSError, IndexError):
        pass
    return "%s_%s_%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), ppid, started,
                               os.environ.get("TORCHELASTIC_RUN_ID", "x"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))


class FileRendezvous(object):
    """Small blobs between the ranks of one node through files: put / get by name, all-gather, barrier.

    The directory is this user's alone (mode 0700; a directory that exists already must be a real directory owned by
    this user and closed to others, anything else is refused), and its name carries the launch's identity, so the ranks
    of a restarted or later launch never meet the files a crashed one left behind."""

    def __init__(self, rank, world, directory=None, timeout=900.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if directory is None:
            directory = os.environ.get("ADVNTR_RDZV_DIR")
        if directory is None:
            directory = os.path.join("/tmp", "advntr_rdzv_%d_%s" % (os.getuid(), _launcher_identity()))
        self.dir = directory
        try:
            os.makedirs(self.dir, mode=0o700)
        except FileExistsError:
            pass
        st = os.lstat(self.dir)
        import stat as stat_mod
        if not stat_mod.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise PermissionError("rendezvous directory %s is not a directory of this user alone (uid %d, mode %o): refusing "
                                  "to exchange data through it" % (self.dir, st.st_uid, st.st_mode & 0o7777))
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
            sms.append(np.frombuffer(p[8 * c:], np.int32).reshape(c, SUMMARY_INTS))
        return np.concatenate(lps), np.concatenate(sms)

    def close(self):
        self.rdzv.close()


class InitWatchdog(object):
    """Stands beside a rank while it is inside ncclCommInitRank -- a collective that returns when ALL ranks have entered it
    and cannot be called back.  Ends the PROCESS (exit status 70) when the call has not returned after `timeout` seconds, and
    at once (status 71) when a peer has left an abort marker in the rendezvous directory: a rank whose own ncclCommInitRank
    failed (init_from_env writes `abort.<rank>`) -- its peers would otherwise sit in the collective until their timeouts.
    `on_end(status, message)` replaces the exit in tests."""

    def __init__(self, directory, rank, timeout, on_end=None, poll=0.25):
        import threading
        self._dir, self._rank, self._timeout, self._poll = directory, rank, float(timeout), float(poll)
        self._on_end = on_end or self._exit
        self._born = time.time()
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="advntr-init-watchdog", daemon=True)
        self._thread.start()

    @staticmethod
    def _exit(status, message):
        import sys
        sys.stderr.write(message + "\n")
        sys.stderr.flush()
        os._exit(status)

    def _fresh_markers(self):
        """Abort markers written since this watchdog started.  A marker left behind by an EARLIER job in a reused rendezvous
        directory (ADVNTR_RDZV_DIR: the crashed job never reached close()) must not end a healthy later one."""
        out = []
        try:
            for f in sorted(os.listdir(self._dir)):
                if not f.startswith("abort."):
                    continue
                try:
                    if os.stat(os.path.join(self._dir, f)).st_mtime >= self._born - 1.0:
                        out.append(f)
                except OSError:
                    pass
        except OSError:
            pass
        return out

    def _run(self):
        t0 = time.monotonic()
        while not self._stop.wait(self._poll):
            gone = self._fresh_markers()
            if gone:
                self._on_end(71, "advntr_amd.comm: rank %d: a peer could not create its communicator (%s); ending this process"
                             % (self._rank, ", ".join(gone)))
                return
            if time.monotonic() - t0 > self._timeout:
                self._on_end(70, "advntr_amd.comm: rank %d: ncclCommInitRank did not return within %.0f s (a peer is gone?); "
                             "ending this process" % (self._rank, self._timeout))
                return

    def cancel(self):
        self._stop.set()


class CommSetupError(RuntimeError):
    """RCCL cannot be used by this job; raised on every rank alike, before any of them entered a collective."""


class RcclComm(HostComm):
    """RCCL communicator of the library (advntr_comm_*), one rank per GPU; the current device must be set before."""
    backend = "rccl"

    def __init__(self, rdzv, init_timeout=None):
        """Collective.  Every rank makes the same rendezvous calls whatever fails where, and no rank enters
        ncclCommInitRank -- a collective that only returns when ALL ranks have called it -- before every rank has said
        that it can: (1) each rank loads librccl (advntr_comm_available) and the ranks exchange the outcome; (2) rank 0
        makes the unique id and broadcasts it, an empty blob if that failed; (3) the ranks create their communicators, each
        under a watchdog that ends the PROCESS if the call has not returned after `init_timeout` seconds (default
        ADVNTR_COMM_INIT_TIMEOUT or 180): a peer that died between (2) and (3) must not leave the job hanging.
        Raises CommSetupError on every rank alike when any of them reported a failure in (1) or (2)."""
        from . import _lib
        HostComm.__init__(self, rdzv)
        self._lib = _lib
        self._h = None
        L = _lib.load()
        try:                                                 # this rank's marker of an earlier job in a reused directory
            os.unlink(os.path.join(rdzv.dir, "abort.%d" % self.rank))
        except OSError:
            pass
        err = b""
        if L.advntr_comm_available() != _lib.OK:
            err = ("rank %d: %s" % (self.rank, _lib.last_error())).encode("utf-8", "replace")
        failures = [x for x in rdzv.allgather(err) if x]
        uid = ctypes.create_string_buffer(128)
        blob = b""
        if self.rank == 0 and not failures:
            if L.advntr_comm_unique_id(ctypes.addressof(uid)) == _lib.OK:
                blob = uid.raw
            else:
                failures.append(("rank 0: %s" % _lib.last_error()).encode("utf-8", "replace"))
        blob = rdzv.broadcast(blob, 0)                       # always: every rank takes this step
        if failures or len(blob) != 128:
            reason = failures[0].decode("utf-8", "replace") if failures else "rank 0 could not make the RCCL unique id"
            raise CommSetupError(reason)
        if init_timeout is None:
            init_timeout = float(os.environ.get("ADVNTR_COMM_INIT_TIMEOUT", "180"))
        watchdog = InitWatchdog(rdzv.dir, self.rank, init_timeout)
        try:
            self._h = L.advntr_comm_create(self.rank, self.world, blob)
        finally:
            watchdog.cancel()
        if not self._h:
            msg = _lib.last_error()
            # the peers are inside the collective and cannot be called back: their watchdogs end them on this marker (written
            # here, so that a direct user of RcclComm gets the fast abort too, not only init_from_env)
            try:
                rdzv.put("abort.%d" % self.rank, msg.encode("utf-8", "replace"))
            except OSError:
                pass
            raise _lib.EngineError(_lib.ERR_DEVICE, msg)

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

    def last_gather_ms(self):
        """Milliseconds the last finished result gather took on the communicator's stream, from the moment the records of
        its pass were staged (HIP events): about the transfer itself when it ran beside the next pass, about a whole pass
        when it had to wait for that pass's kernels to leave the compute units."""
        ms = ctypes.c_float(0)
        self._lib.check(self._lib.load().advntr_comm_last_gather_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.load().advntr_comm_destroy(self._h)
            self._h = None
        HostComm.close(self)


def env_world():
    """(rank, local_rank, world) from the launcher's environment; (0, 0, 1) for a plain single process."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


def frugal_host_for_ranks(world):
    """N ranks share one host (and, in a container, one CPU quota).  Before this rank touches its GPU: its waits on the device
    sleep instead of spinning (ADVNTR_BLOCKING_SYNC, read by advntr_set_device), and the bulk host-side calls of the library use
    1 / world of the CPUs the process group may use (ADVNTR_HOST_THREADS, read by advntr_host_threads at every call).  Values
    already in the environment win.  Returns what was set."""
    from . import _lib
    out = {}
    if "ADVNTR_BLOCKING_SYNC" not in os.environ:
        os.environ["ADVNTR_BLOCKING_SYNC"] = out["ADVNTR_BLOCKING_SYNC"] = "1"
    if "ADVNTR_HOST_THREADS" not in os.environ:
        share = max(1, int(_lib.load().advntr_host_threads()) // max(1, int(world)))
        os.environ["ADVNTR_HOST_THREADS"] = out["ADVNTR_HOST_THREADS"] = str(share)
    return out


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
    if world > 1:
        frugal_host_for_ranks(world)
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
    # RCCL, agreed on by all ranks.  When it cannot be used at the SET-UP stage (librccl missing or without the entry points,
    # rank 0 cannot make the unique id: CommSetupError) every rank learns so at the same point of the same sequence of
    # rendezvous steps, and the job ENDS with an error on every rank -- a gather through host files is not the path this
    # package is about.  ADVNTR_COMM_FALLBACK=1 (explicitly) lets the ranks drop to the host communicator together instead;
    # the choice is then visible in `.backend` and `.fallback_reason`.  The fallback covers that stage ONLY: a rank whose
    # ncclCommInitRank itself fails (driver / IPC configuration, a device error) has peers that are already inside the
    # collective and cannot be called back, so it ends at once with the error -- the launcher (bench.py's, torchrun) then
    # ends the job, and a peer left waiting is ended by its own watchdog (ADVNTR_COMM_INIT_TIMEOUT).
    c, err = None, b""
    try:
        c = RcclComm(rdzv)
    except CommSetupError as e:     # raised on every rank alike: nobody is inside a collective
        err = str(e).encode("utf-8", "replace")
    except Exception as e:          # noqa: BLE001 -- past the agreed stage: no collective fallback, no waiting for the others
        try:
            rdzv.put("abort.%d" % rank, str(e).encode("utf-8", "replace"))      # the peers' watchdogs end them at once
        except OSError:
            pass
        raise RuntimeError("rank %d: RCCL communicator could not be created after every rank had agreed to (%s); the other "
                           "ranks are inside ncclCommInitRank and end with the job" % (rank, e)) from e
    failures = [x for x in rdzv.allgather(err) if x]
    if not failures:
        return c
    if c is not None and c._h:
        c._lib.load().advntr_comm_destroy(c._h)
        c._h = None
    reason = failures[0].decode("utf-8", "replace")
    if os.environ.get("ADVNTR_COMM_FALLBACK", "0") != "1":
        rdzv.close()
        raise RuntimeError("RCCL communicator could not be created (%s); set ADVNTR_COMM_FALLBACK=1 to gather through the "
                           "host communicator instead" % reason)
    import sys
    if rank == 0:
        sys.stderr.write("advntr_amd.comm: RCCL unavailable (%s); ADVNTR_COMM_FALLBACK=1: the result gather goes through "
                         "the host communicator\n" % reason)
    h = HostComm(rdzv)
    h.fallback_reason = reason
    return h
