"""The N>1 path on CPU: locus->rank partition and the final gather of result records, world_size 2, gloo.
(The HIP kernels cannot run here; ranks fill their records with a deterministic function of the global read
id, which is exactly what rank 0 must receive back, ordered by global read id.)"""
import os
import socket

import numpy as np


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
    res = sharding.gather_records(ids, logp, summ, dst=0)
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
    res = sharding.run_sharded(work, job)
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
