"""The N>1 path on CPU: locus->rank partition and the final gather of result records, world_size 2.
(The HIP kernels cannot run here; ranks fill their records with a deterministic function of the global read
id, which is exactly what rank 0 must receive back, ordered by global read id.)

The product's sharding code is written against comm.py's communicator interface and imports no torch; here it is driven
(a) over a gloo process group through the small adapter below and (b) over comm.HostComm, the file-rendezvous
communicator that the product itself uses when ranks share a GPU.  The RCCL communicator has the same methods and is
exercised on the GPU box (tests/test_gpu_parity.py)."""
import os
import socket

import numpy as np
import pytest


class GlooComm(object):
    """comm.py's interface over a torch.distributed (gloo) process group."""
    backend = "gloo"

    def __init__(self):
        import torch.distributed as dist
        self.dist, self.rank, self.world = dist, dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def gather_bytes(self, data, root=0):
        out = [None] * self.world if self.rank == root else None
        self.dist.gather_object(bytes(data), out, dst=root)
        return out

    def allgather_i64(self, x):
        out = [None] * self.world
        self.dist.all_gather_object(out, int(x))
        return out

    def allreduce_max(self, x):
        out = [None] * self.world
        self.dist.all_gather_object(out, float(x))
        return max(out)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from advntr_amd import sharding
    rng = np.random.default_rng(5)                      # same on every rank
    n_loci = 11
    reads_per_locus = rng.integers(1, 9, n_loci)
    edges = rng.integers(100, 5000, n_loci)
    work = [sharding.locus_work(np.full(reads_per_locus[i], 150), edges[i]) for i in range(n_loci)]
    parts = sharding.partition_loci(work, world)
    first = np.concatenate([[0], np.cumsum(reads_per_locus)])
    ids = np.concatenate([np.arange(first[i], first[i + 1]) for i in parts[rank]]) if len(parts[rank]) else np.zeros(0, np.int64)
    logp = -ids.astype(np.float64) * 1.5 - 0.25
    summ = np.stack([ids * 8 + k for k in range(8)], axis=1).astype(np.int32) if len(ids) else np.zeros((0, 8), np.int32)
    res = sharding.gather_records(GlooComm(), ids, logp, summ, dst=0)
    if rank == 0:
        all_ids, all_lp, all_sm = res
        total = int(reads_per_locus.sum())
        ok = (np.array_equal(all_ids, np.arange(total)) and np.array_equal(all_lp, -np.arange(total) * 1.5 - 0.25)
              and np.array_equal(all_sm[:, 3], np.arange(total) * 8 + 3))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    from advntr_amd import sharding
    rng = np.random.default_rng(1)
    work = rng.integers(1, 1000, 200)
    for world in (1, 2, 4, 8):
        parts = sharding.partition_loci(work, world)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(200))
        loads = [int(work[p].sum()) for p in parts]
        assert max(loads) - min(loads) <= int(work.max())


def test_partition_capacity_gives_the_gather_root_a_smaller_share():
    """partition_loci(work, world, capacity): rank 0 at capacity 0.9 ends with ~0.9 of the others' load; every locus once;
    capacity None == all ones; bad capacities are refused."""
    import numpy as np
    from advntr_amd import sharding
    rng = np.random.default_rng(4)
    work = rng.integers(1000, 20000, 4000)
    parts = sharding.partition_loci(work, 8, [0.9] + [1.0] * 7)
    assert sorted(int(k) for p in parts for k in p) == list(range(4000))
    loads = np.array([work[p].sum() for p in parts], np.float64)
    assert abs(loads[0] / loads[1:].mean() - 0.9) < 0.005 and loads[1:].max() / loads[1:].min() < 1.002
    same = sharding.partition_loci(work, 8, [1.0] * 8)
    assert all(np.array_equal(a, b) for a, b in zip(same, sharding.partition_loci(work, 8)))
    for bad in ([1.0] * 7, [0.0] + [1.0] * 7):
        with pytest.raises(ValueError):
            sharding.partition_loci(work, 8, bad)


def test_gather_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_c3_plan_matches_the_generated_share():
    """bench.py --workload c3: every rank derives the same plan (calls, states per locus) without generating reads,
    partitions it, and generates exactly its share; the shares cover the locus set once."""
    from advntr_amd import sharding, workloads
    plan = workloads.c2_plan(24, seed=99)
    parts = sharding.partition_loci([c * 151 * m for c, m in plan], 3)
    assert sorted(int(k) for p in parts for k in p) == list(range(24))
    loads = [sum(plan[int(k)][0] * plan[int(k)][1] for k in p) for p in parts]
    assert max(loads) < 1.25 * min(loads)
    loci, reads, which = workloads.make_c2_parallel(24, seed=99, only=parts[1], workers=2)
    assert len(loci) == len(parts[1]) and len(reads) == sum(plan[int(k)][0] for k in parts[1])
    for locus, k in zip(loci, parts[1]):
        assert locus.model.n_states == plan[int(k)][1]
    # the same locus comes out identical whichever rank asks for it
    again, reads2, _ = workloads.make_c2_parallel(24, seed=99, only=[int(parts[1][0])], workers=1, build=False)
    assert (again[0].left, again[0].units) == (loci[0].left, loci[0].units)
    assert reads2 == reads[:len(reads2)]


def test_c4_plan_prices_a_locus_before_its_reads_exist():
    """bench.py --workload c4 --gpus N: the plan replays the two draws _c4_locus makes first (pattern length, reference VNTR
    length), so every rank prices every locus without generating a read; the generated share is exactly the planned loci, and
    the estimate tracks the actual work (trellis cells of the 20 noisy reads on the model sized for the longest) closely
    enough for the LPT split to balance it."""
    import numpy as np
    from advntr_amd import sharding, workloads
    plan = workloads.c4_plan(48, seed=20240603)
    assert all(c == 20 for c, _, _ in plan)
    est = np.array([c * (n + 1) * m for c, n, m in plan], np.float64)
    parts = sharding.partition_loci(est, 4)
    assert sorted(int(k) for p in parts for k in p) == list(range(48))
    loci, reads, which = workloads.make_c4(48, seed=20240603, workers=2)
    workloads.build_models(loci)
    lens = np.array([len(r) for r in reads]).reshape(48, 20)
    actual = (lens + 1).sum(1) * np.array([l.model.n_states for l in loci], np.float64)
    assert np.all(np.abs(actual / est - 1.0) < 0.25) and np.corrcoef(actual, est)[0, 1] > 0.99
    loads = [actual[p].sum() for p in parts]
    assert max(loads) / (sum(loads) / 4) < 1.08
    mine, reads1, which1 = workloads.make_c4(48, seed=20240603, workers=1, only=parts[1])
    assert len(mine) == len(parts[1]) and len(reads1) == 20 * len(parts[1])
    for j, k in enumerate(parts[1]):
        assert (mine[j].left, mine[j].units, mine[j].copies) == (loci[int(k)].left, loci[int(k)].units, loci[int(k)].copies)
        assert reads1[20 * j:20 * j + 20] == reads[20 * int(k):20 * int(k) + 20]


def _sharded_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from advntr_amd import sharding
    work = [(i * 37) % 11 + 1 for i in range(23)]
    seen = []

    def job(indices):
        seen.extend(indices)
        return ["row %d from rank %d" % (i, rank) for i in indices]
    res = sharding.run_sharded(work, job, GlooComm())
    if rank == 0:
        ok = res is not None and len(res) == 23 and all(r.startswith("row %d from rank" % i) for i, r in enumerate(res))
        ok = ok and len({r.split()[-1] for r in res}) == world and 0 < len(seen) < 23
        q.put(bool(ok))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_run_sharded_world_size_2_gloo():
    """Per-locus jobs split over two ranks, rows back on rank 0 in locus order; and the single-process fallback."""
    import torch.multiprocessing as mp
    from advntr_amd import sharding
    assert sharding.run_sharded([3, 1, 2], lambda idx: [i * i for i in idx]) == [0, 1, 4]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _host_worker(rank, world, directory, q):
    from advntr_amd import comm, sharding
    c = comm.HostComm(comm.FileRendezvous(rank, world, directory, timeout=120))
    assert (c.rank, c.world) == (rank, world)
    c.barrier()
    assert c.allreduce_max(1.5 + rank) == 1.5 + (world - 1)
    assert c.allgather_i64(10 * rank + 7) == [10 * r + 7 for r in range(world)]
    parts = c.gather_bytes(b"x" * rank + b"|%d" % rank, root=0)           # ragged, incl. a short one
    if rank == 0:
        assert parts == [b"x" * r + b"|%d" % r for r in range(world)]
    else:
        assert parts is None
    ids = np.arange(rank, 40, world, dtype=np.int64)                       # interleaved global ids
    res = sharding.gather_records(c, ids, -ids * 0.5, np.stack([ids + k for k in range(8)], axis=1), dst=0)
    rows = sharding.run_sharded([5, 1, 1, 7, 2, 2, 9], lambda idx: ["r%d@%d" % (i, rank) for i in idx], c)
    if rank == 0:
        ok = (np.array_equal(res[0], np.arange(40)) and np.array_equal(res[1], -np.arange(40) * 0.5)
              and np.array_equal(res[2][:, 5], np.arange(40) + 5))
        ok = ok and [r.split("@")[0] for r in rows] == ["r%d" % i for i in range(7)] and len({r.split("@")[1] for r in rows}) == world
        q.put(bool(ok))
    else:
        assert res is None and rows is None
    c.close()


def test_host_comm_world_size_3(tmp_path):
    """comm.HostComm (file rendezvous, no torch, no GPU): small collectives, ragged gather, gather_records, run_sharded."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = str(tmp_path / "rdzv")
    procs = [ctx.Process(target=_host_worker, args=(r, 3, d, q)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
    assert not os.path.exists(d)                          # rank 0 removed the rendezvous directory


def test_product_imports_no_torch():
    """north_star: no PyTorch on the path -- nothing under advntr_amd/, and neither bench.py nor its package benchlib/."""
    import re
    from conftest import ROOT
    offenders = []
    files = [os.path.join(ROOT, "bench.py")]
    for pkg in ("advntr_amd", "benchlib"):
        for dirpath, _, names in os.walk(os.path.join(ROOT, pkg)):
            files += [os.path.join(dirpath, n) for n in names if n.endswith(".py")]
    for path in files:
        if re.search(r"^\s*(import|from)\s+torch\b", open(path).read(), re.M):
            offenders.append(path)
    assert offenders == []


def _rccl_missing_worker(rank, world, directory, fallback, q):
    """init_from_env with an RCCL library that cannot be loaded: no GPU is touched before the ranks have agreed."""
    import time
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT="29999", ADVNTR_RDZV_DIR=directory, ADVNTR_RCCL_LIB="/nonexistent/librccl.so")
    if fallback:
        os.environ["ADVNTR_COMM_FALLBACK"] = "1"
    else:
        os.environ.pop("ADVNTR_COMM_FALLBACK", None)
    from advntr_amd import comm
    t0 = time.time()
    try:
        c = comm.init_from_env(set_device=False)
    except RuntimeError as e:
        q.put((rank, "error", "librccl" in str(e) or "dlopen" in str(e), time.time() - t0))
        return
    ok = c.backend == "host" and c.fallback_reason is not None and c.allgather_i64(rank) == list(range(world))
    c.close()
    q.put((rank, "host", bool(ok), time.time() - t0))


@pytest.mark.parametrize("fallback", [False, True])
def test_missing_rccl_is_agreed_on_by_all_ranks_without_waiting(tmp_path, fallback):
    """rank 0 used to raise before its broadcast and leave the others in a 900 s wait.  Every rank now makes
    the same rendezvous calls; without ADVNTR_COMM_FALLBACK=1 all of them end with an error (in seconds), with it they drop
    to the host communicator together."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = str(tmp_path / "rdzv")
    procs = [ctx.Process(target=_rccl_missing_worker, args=(r, 2, d, fallback, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert [g[0] for g in got] == [0, 1]
    assert all(g[1] == ("host" if fallback else "error") and g[2] for g in got), got
    assert all(g[3] < 60 for g in got), got


def test_rendezvous_directory_must_be_this_users_alone(tmp_path):
    from advntr_amd import comm
    d = tmp_path / "open_to_all"
    d.mkdir(mode=0o755)
    os.chmod(str(d), 0o755)
    with pytest.raises(PermissionError):
        comm.FileRendezvous(0, 1, str(d))
    f = tmp_path / "not_a_directory"
    f.write_text("x")
    with pytest.raises((PermissionError, FileExistsError, NotADirectoryError)):
        comm.FileRendezvous(0, 1, str(f))
    fresh = comm.FileRendezvous(0, 1, str(tmp_path / "fresh"))
    assert (os.stat(fresh.dir).st_mode & 0o777) == 0o700
    # the default name tells one launch from the next: launcher pid AND its start time, port, run id, restart count
    os.environ["TORCHELASTIC_RESTART_COUNT"] = "0"
    a = comm._launcher_identity()
    os.environ["TORCHELASTIC_RESTART_COUNT"] = "1"
    b = comm._launcher_identity()
    os.environ.pop("TORCHELASTIC_RESTART_COUNT")
    assert a != b and str(os.getppid()) in a
    fresh.close()


def test_host_gather_results_with_a_rank_that_holds_no_reads(tmp_path):
    """reshape(0, -1) of an empty summary block raised on the root."""
    from advntr_amd import comm

    class FakeBatch(object):
        def __init__(self, n):
            self.n = n

        def fetch(self):
            return np.arange(self.n, dtype=np.float64), np.arange(8 * self.n, dtype=np.int32).reshape(self.n, 8)

    c = comm.HostComm(comm.FileRendezvous(0, 1, str(tmp_path / "r")))
    logp, summ = c.gather_results(FakeBatch(0), [0])
    assert logp.shape == (0,) and summ.shape == (0, 8)
    logp, summ = c.gather_results(FakeBatch(3), [3])
    assert logp.tolist() == [0.0, 1.0, 2.0] and summ.shape == (3, 8)
    c.close()


def test_run_sharded_carries_rows_as_json_not_pickle():
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "advntr_amd", "sharding.py")).read()
    assert "pickle" not in src


def test_init_watchdog_ends_a_rank_whose_peer_left_an_abort_marker(tmp_path):
    """A rank inside ncclCommInitRank cannot be called back: its watchdog ends the process when the call outlives its timeout
    (status 70) and -- at once -- when a peer whose own communicator failed has left `abort.<rank>` in the rendezvous directory
    (status 71), instead of after the timeout."""
    import time
    from advntr_amd import comm
    d = tmp_path / "rdzv"
    d.mkdir(mode=0o700)
    got = []
    w = comm.InitWatchdog(str(d), 1, timeout=30.0, on_end=lambda status, msg: got.append((status, msg)), poll=0.02)
    time.sleep(0.1)
    assert got == []                                            # nothing wrong yet
    comm.FileRendezvous(0, 2, str(d)).put("abort.0", b"hipErrorInvalidDevice")
    t0 = time.monotonic()
    while not got and time.monotonic() - t0 < 5:
        time.sleep(0.01)
    w.cancel()
    assert got and got[0][0] == 71 and "abort.0" in got[0][1] and time.monotonic() - t0 < 2
    got2 = []
    d2 = tmp_path / "rdzv2"
    d2.mkdir(mode=0o700)
    w2 = comm.InitWatchdog(str(d2), 0, timeout=0.1, on_end=lambda status, msg: got2.append(status), poll=0.02)
    time.sleep(0.5)
    w2.cancel()
    assert got2 == [70]
    quiet = []
    w3 = comm.InitWatchdog(str(d2), 0, timeout=30.0, on_end=lambda status, msg: quiet.append(status), poll=0.02)
    w3.cancel()                                                 # the call returned in time: nobody is ended
    time.sleep(0.1)
    assert quiet == []
    # a marker an EARLIER job left in a reused rendezvous directory (its mtime lies before this watchdog's start) ends nobody
    d3 = tmp_path / "rdzv3"
    d3.mkdir(mode=0o700)
    stale = d3 / "abort.1"
    stale.write_bytes(b"left by a job that crashed yesterday")
    os.utime(str(stale), (time.time() - 3600, time.time() - 3600))
    old = []
    w4 = comm.InitWatchdog(str(d3), 0, timeout=30.0, on_end=lambda status, msg: old.append(status), poll=0.02)
    time.sleep(0.2)
    assert old == []
    (d3 / "abort.2").write_bytes(b"this job's")                 # ... a fresh one does
    t0 = time.monotonic()
    while not old and time.monotonic() - t0 < 5:
        time.sleep(0.01)
    w4.cancel()
    assert old == [71]


def test_ranks_of_one_host_are_frugal_with_its_cpus(monkeypatch):
    """N ranks on one host: each rank's waits on the device sleep (ADVNTR_BLOCKING_SYNC) and its bulk host calls use 1 / N of the
    CPUs the job may use (ADVNTR_HOST_THREADS); values the caller set win."""
    from advntr_amd import _lib, comm
    # (set before deleted: monkeypatch then restores whatever the environment held, also for the values the call under test sets)
    for name in ("ADVNTR_BLOCKING_SYNC", "ADVNTR_HOST_THREADS"):
        monkeypatch.setenv(name, os.environ.get(name, ""))
        monkeypatch.delenv(name)
    total = int(_lib.load().advntr_host_threads())
    got = comm.frugal_host_for_ranks(4)
    assert got["ADVNTR_BLOCKING_SYNC"] == "1" and int(got["ADVNTR_HOST_THREADS"]) == max(1, total // 4)
    assert int(_lib.load().advntr_host_threads()) == max(1, total // 4)      # (read at every call)
    monkeypatch.setenv("ADVNTR_HOST_THREADS", "3")
    monkeypatch.setenv("ADVNTR_BLOCKING_SYNC", "0")
    assert comm.frugal_host_for_ranks(4) == {}
    assert int(_lib.load().advntr_host_threads()) == 3
