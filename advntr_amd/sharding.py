"""Multi-GPU layout of the scoring path: one process per GPU, loci (with all their reads) assigned to
ranks, no data-path collective, one gather of the per-read result records to rank 0 at the end
(SURVEY.md section 8e).  Backend-agnostic torch.distributed code: `nccl` (= RCCL over xGMI) on the GPU
box, `gloo` in the CPU tests.  The reference has no counterpart (it scores loci serially,
/root/reference/advntr/genome_analyzer.py:280)."""
import numpy as np


def locus_work(n_bases_per_read, n_edges):
    """Edge relaxations of one locus: sum over its calls of (n+1)*E  (SURVEY 8d)."""
    return int((np.asarray(n_bases_per_read, dtype=np.int64) + 1).sum() * int(n_edges))


def partition_loci(work, world_size):
    """Longest-processing-time greedy: heaviest locus first onto the least loaded rank.  Returns a list of
    index arrays (one per rank); every locus appears exactly once; ties broken by locus index so that
    every rank computes the same partition without communicating."""
    work = np.asarray(work, dtype=np.int64)
    order = sorted(range(len(work)), key=lambda i: (-int(work[i]), i))
    load = [0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(work[i])
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def gather_records(read_ids, logp, summary, dst=0, device=None):
    """Gather ragged per-read records (global read id, fp64 logp, 8 x int32 summary) to rank `dst`.
    One all_gather of the per-rank counts, then one padded gather per array.  Returns on dst the arrays
    sorted by global read id, elsewhere None."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    ids = torch.as_tensor(np.asarray(read_ids, dtype=np.int64), device=dev)
    lp = torch.as_tensor(np.asarray(logp, dtype=np.float64), device=dev)
    sm = torch.as_tensor(np.asarray(summary, dtype=np.int32).reshape(-1, 8), device=dev)
    count = torch.tensor([ids.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)

    def padded(t, width=None):
        shape = (cap,) if width is None else (cap, width)
        out = torch.zeros(shape, dtype=t.dtype, device=dev)
        out[:t.shape[0]] = t
        return out

    bufs = []
    for t, width in ((ids, None), (lp, None), (sm, 8)):
        mine = padded(t, width)
        recv = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, recv, dst=dst)
        bufs.append(recv)
    if rank != dst:
        return None
    all_ids = torch.cat([bufs[0][k][:counts[k]] for k in range(world)]).cpu().numpy()
    all_lp = torch.cat([bufs[1][k][:counts[k]] for k in range(world)]).cpu().numpy()
    all_sm = torch.cat([bufs[2][k][:counts[k]] for k in range(world)]).cpu().numpy()
    order = np.argsort(all_ids, kind="stable")
    return all_ids[order], all_lp[order], all_sm[order]


def run_sharded(work, fn, dst=0):
    """Locus-sharded execution of a per-locus job: every rank takes its LPT share of range(len(work)), calls
    fn(indices) -> list of picklable results (one per index, same order) and the results come back to rank `dst` in
    index order (None elsewhere).  One gather of Python objects at the end; with no process group (single process) it
    just runs fn over everything.  This is how `python -m advntr_amd genotype` runs under torch.distributed.run."""
    n = len(work)
    try:
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized()
    except ImportError:
        active = False
    if not active:
        return list(fn(list(range(n))))
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = [int(i) for i in partition_loci(work, world)[rank]]
    results = list(fn(mine))
    if len(results) != len(mine):
        raise ValueError("run_sharded: fn returned %d results for %d indices" % (len(results), len(mine)))
    gathered = [None] * world if rank == dst else None
    dist.gather_object(list(zip(mine, results)), gathered, dst=dst)
    if rank != dst:
        return None
    out = [None] * n
    for part in gathered:
        for i, r in part:
            out[i] = r
    return out
