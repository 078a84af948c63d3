#!/usr/bin/env python3
"""bench.py -- reads/s Viterbi-scored on the BASELINE.json workloads, one process per GPU, no PyTorch.

A "step" is one pass of the hot path (Viterbi DP + traceback + path summaries) over one resident batch of synthetic
reads; inputs are in HBM before the timed region.  Prints ONE JSON line on rank 0.

  --gpus 1 (default): config C1 of BASELINE.json / SURVEY 8d -- one REF150 locus (flank 150, 14-bp pattern, 11 copies:
      1413 states / 921 emitting / 4626 edges), 100 000 synthetic 150-bp reads.  The line also carries the roofline object,
      the VALU bound that actually binds (nominal, at the measured clock, at the measured issue rate), the CPU baseline (the C
      oracle on the host cores, with its calibration against the vendored pomegranate), and as sub-records everything else that
      has a number: `s300` (the "~300-state" label of the metric, Viterbi and log_probability), `log_probability`, `c2` and
      `end_to_end` (BASELINE config 2: 6 719 loci, kernel alone and candidate reads -> genotypes), `scale_rehearsal` (the 8-rank
      strong-scaling line projected from this one GPU), `c4`, `c4_scale_rehearsal` and `pacbio_end_to_end` (BASELINE config 5:
      8 960 PacBio loci, its 8-rank split rehearsed; whole 5-15 kb reads -> genotypes), `prefilter` and `flank_align` (the two
      kernels upstream of the scoring path), `two_passes_in_flight` (the C1 batch with the next pass queued on a second copy's
      stream).  About 40 s on the GPU box.
  --gpus N > 1: config C3 -- ONE set of 6 719 synthetic Illumina loci (~1.07 M calls) partitioned over the N GPUs by
      estimated work (strong scaling; whole loci per rank, LPT), every rank scores its share with no exchange, and the
      per-call result records are gathered to rank 0 over RCCL inside the timed region (the gather of pass i overlaps
      the kernels of pass i+1; all gathers complete before the clock stops).  `value` = calls of the WHOLE set per second.
      The strong-scaling lines keep TWO passes in flight (--in-flight, class Passes): consecutive passes alternate between two
      copies of the rank's device batch (scratch, results and stream of their own), so pass i+1 starts while the last
      workgroups of pass i drain -- the end of a launch is most of what separates an 8-rank share from an eighth of the set.
  --workload c4 --gpus N: BASELINE config 5 the same way -- ONE set of 8 960 PacBio loci (179 200 calls), whole loci to ranks
      by LPT on what is known of a locus before its reads exist (workloads.c4_plan), records gathered over RCCL, "strong".

Launching: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (the driver's way; only the
launcher is torch, this file imports none of it) the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; started
plainly with --gpus N > 1 this process starts the N ranks itself as CHILD processes, before anything touches a GPU,
forwards rank 0's line and exits with their status (it never replaces itself by another program).
--dry-run does the planning and the rendezvous without any GPU work (CPU-side check of the multi-rank plumbing).
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.main import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
