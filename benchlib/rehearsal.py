"""The strong-scaling lines rehearsed on one GPU: every rank's share as its own resident batch (a PROJECTION)."""
import numpy as np

from .passes import Passes, two_in_flight_ms


def scale_rehearsal(_lib, sharding, loci, dms, bases, off, which, n_ranks, whole, flags, steps, planned_work=None,
                    root_capacity=0.99):
    """What a 1-GPU lease can say about the north star's "strong scaling to 8 GPUs": the C3 locus set partitioned for
    n_ranks ranks exactly as `--gpus N` partitions it (whole loci, LPT on calls x (n+1) x states, sharding.partition_loci), and
    every rank's share run on THIS GPU as its own resident batch with the launch parameters of the multi-GPU job (the slots
    the gather asks each pass to leave free: advntr_batch_reserve_next(8)).  projected_efficiency = T(whole set, 1 rank) /
    (n_ranks x slowest share): what load balance, the per-launch costs that do not shrink with the batch and the partial last
    round of resident wavefronts leave of perfect strong scaling, BEFORE the gather (43 MB over xGMI per pass, overlapped with
    the next pass by design) and before any difference between GPUs.  A projection, labelled as such; the measured curve is
    the driver's SCALE run."""
    lens = np.diff(off)
    ms = np.array([d.m for d in dms])
    calls = np.bincount(which, minlength=len(dms))
    # planned_work: the per-locus estimates the multi-GPU job partitions by when it cannot know a locus's calls exactly (C4:
    # workloads.c4_plan); otherwise the plan is exact (C3: calls x 151 x states)
    work = list(planned_work) if planned_work is not None else [int(calls[k]) * 151 * int(ms[k]) for k in range(len(dms))]
    parts = sharding.partition_loci(work, n_ranks, [root_capacity] + [1.0] * (n_ranks - 1))
    loads = [float(sum(work[int(k)] for k in p)) for p in parts]
    cells = np.bincount(which, weights=(lens + 1) * ms[which], minlength=len(dms))       # actual work: trellis cells per locus
    actual = [float(cells[p].sum()) for p in parts]
    uniform = bool(len(lens) and lens.min() == lens.max())
    shares = []
    for r, mine in enumerate(parts):
        remap = np.full(len(dms), -1, np.int32)
        remap[mine] = np.arange(len(mine), dtype=np.int32)
        sel = remap[which] >= 0
        if uniform:
            sub_bases = bases.reshape(len(lens), -1)[sel].reshape(-1)
        else:
            sub_bases = bases[np.repeat(sel, lens)]
        sub_off = np.zeros(int(sel.sum()) + 1, np.int64)
        np.cumsum(lens[sel], out=sub_off[1:])
        make = lambda extra=0: _lib.DeviceBatch([dms[int(k)] for k in mine], sub_bases, sub_off, remap[which[sel]],      # noqa: E731
                                                flags=flags | extra)
        one = Passes(make, 1)
        batch = one.batches[0]
        loop_ms = one.ms_per_pass(steps, reserve=8)
        loop2_ms = two_in_flight_ms(batch, make, steps, reserve=8)
        kernel_ms = batch.run_timed(steps)                  # (no reservation: the kernel alone)
        shares.append({"rank": r, "loci": int(len(mine)), "calls": int(sel.sum()), "loop_ms": loop_ms,
                       "loop_ms_two_passes_in_flight": loop2_ms, "kernel_ms": kernel_ms})
        one.close()
    worst_loop = max(x["loop_ms"] for x in shares)
    worst_loop2 = max(x["loop_ms_two_passes_in_flight"] for x in shares)
    worst_kernel = max(x["kernel_ms"] for x in shares)
    whole2 = whole.get("loop_ms_two_passes_in_flight")
    return {"projection": True, "ranks": n_ranks, "whole_set": whole, "shares": shares,
            "sum_of_shares_loop_ms": sum(x["loop_ms"] for x in shares), "slowest_share_loop_ms": worst_loop,
            "load_imbalance_max_over_mean": max(loads) / (sum(loads) / n_ranks),
            "root_capacity": root_capacity, "root_load_over_mean": loads[0] / (sum(loads) / n_ranks),
            "root_share_loop_ms_over_slowest": shares[0]["loop_ms"] / worst_loop,
            "actual_cells_imbalance_max_over_mean": max(actual) / (sum(actual) / n_ranks),
            "per_locus_work_max_over_min": float(max(work)) / max(float(min(work)), 1.0),
            # as the strong-scaling lines run (bench.py --workload c3|c4: two passes in flight, class Passes) ...
            "projected_efficiency": (whole2 / (n_ranks * worst_loop2)) if whole2 else whole["loop_ms"] / (n_ranks * worst_loop),
            "passes_in_flight": 2 if whole2 else 1,
            # ... and with one pass at a time (rounds 3-5: a share's launch ends on single sweeps that nothing overlaps)
            "projected_efficiency_one_pass_in_flight": whole["loop_ms"] / (n_ranks * worst_loop),
            "projected_efficiency_kernels_only": whole["kernel_ms"] / (n_ranks * worst_kernel),
            "projected_value_calls_per_s": float(len(lens)) / ((worst_loop2 if whole2 else worst_loop) * 1e-3),
            "partitioned_by": ("estimated work per locus (calls x (reference VNTR length + 201) x expected states, workloads.c4_plan)"
                               if planned_work is not None else "exact work per locus (calls x 151 x states)"),
            "excludes": "the RCCL gather of the result records (40 B per call to rank 0, queued behind pass i and overlapped with "
                        "pass i + 1) and differences between the GPUs of a node",
            "note": "ONE GPU ran the %d shares one after the other; each share is a rank's whole batch (its models, its calls), "
                    "launched as the multi-GPU job launches it" % n_ranks}
