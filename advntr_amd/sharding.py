"""Multi-GPU layout of the scoring path: one process per GPU, loci (with all their reads) assigned to
ranks, no data-path collective, one gather of the per-read result records to rank 0 at the end
(SURVEY.md section 8e).  The functions here are written against the small communicator interface of comm.py
(rank, world, gather_bytes, ...): RcclComm (RCCL over xGMI, through the library's C ABI) on a multi-GPU node,
HostComm for ranks that share a GPU, and any adapter with the same methods in tests (tests/test_sharding_gloo.py
drives them over a gloo process group).  Nothing here imports torch.  The reference has no counterpart (it scores
loci serially, /root/reference/advntr/genome_analyzer.py:280)."""
import json

import numpy as np


def locus_work(n_bases_per_read, n_edges):
    """Edge relaxations of one locus: sum over its calls of (n+1)*E  (SURVEY 8d)."""
    return int((np.asarray(n_bases_per_read, dtype=np.int64) + 1).sum() * int(n_edges))


def partition_loci(work, world_size, capacity=None):
    """Longest-processing-time greedy: heaviest locus first onto the least loaded rank.  Returns a list of
    index arrays (one per rank); every locus appears exactly once; ties broken by locus index so that
    every rank computes the same partition without communicating.  capacity (optional, one positive number per rank):
    rank r's load counts as load / capacity[r] -- the root of the result gather also hosts the receive side of every
    peer's records, so a job may give it a slightly smaller share (bench.py --root-capacity)."""
    work = np.asarray(work, dtype=np.int64)
    cap = [1.0] * world_size if capacity is None else [float(c) for c in capacity]
    if len(cap) != world_size or min(cap) <= 0:
        raise ValueError("partition_loci: capacity needs one positive number per rank")
    order = sorted(range(len(work)), key=lambda i: (-int(work[i]), i))
    load = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k] / cap[k], k))
        parts[r].append(i)
        load[r] += int(work[i])
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def gather_records(comm, read_ids, logp, summary, dst=0):
    """Gather ragged per-read records (global read id, fp64 logp, 8 x int32 summary) to rank `dst`: one ragged gather of
    the packed records (48 B each).  Returns on dst the arrays sorted by global read id, elsewhere None."""
    ids = np.ascontiguousarray(read_ids, dtype=np.int64)
    lp = np.ascontiguousarray(logp, dtype=np.float64)
    sm = np.ascontiguousarray(summary, dtype=np.int32).reshape(-1, 8)
    if not (len(ids) == len(lp) == len(sm)):
        raise ValueError("gather_records: %d ids, %d log-probabilities, %d summaries" % (len(ids), len(lp), len(sm)))
    rec = np.zeros(len(ids), dtype=[("id", "<i8"), ("logp", "<f8"), ("summary", "<i4", (8,))])
    rec["id"], rec["logp"], rec["summary"] = ids, lp, sm
    parts = comm.gather_bytes(rec.tobytes(), dst)
    if comm.rank != dst:
        return None
    everything = np.concatenate([np.frombuffer(p, dtype=rec.dtype) for p in parts]) if parts else rec
    order = np.argsort(everything["id"], kind="stable")
    everything = everything[order]
    return everything["id"].copy(), everything["logp"].copy(), everything["summary"].copy()


def run_sharded(work, fn, comm=None, dst=0):
    """Locus-sharded execution of a per-locus job: every rank takes its LPT share of range(len(work)), calls
    fn(indices) -> list of results (one per index, same order; text rows or anything else JSON can carry) and the
    results come back to rank `dst` in index order (None elsewhere).  The rows travel as JSON text -- a blob handed over
    by another process is parsed, never executed.  One ragged gather at the end; with no communicator (single process) it just runs fn
    over everything.  This is how `python -m advntr_amd genotype` runs with one process per GPU."""
    n = len(work)
    if comm is None or comm.world <= 1:
        return list(fn(list(range(n))))
    mine = [int(i) for i in partition_loci(work, comm.world)[comm.rank]]
    results = list(fn(mine))
    if len(results) != len(mine):
        raise ValueError("run_sharded: fn returned %d results for %d indices" % (len(results), len(mine)))
    parts = comm.gather_bytes(json.dumps([mine, results]).encode("utf-8"), dst)
    if comm.rank != dst:
        return None
    out = [None] * n
    for blob in parts:
        idx, res = json.loads(bytes(blob).decode("utf-8"))
        if len(idx) != len(res):
            raise ValueError("run_sharded: a rank sent %d results for %d indices" % (len(res), len(idx)))
        for i, r in zip(idx, res):
            out[int(i)] = r
    return out
