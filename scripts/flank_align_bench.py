"""Throughput of the flank alignment (advntr_flank_align) on PacBio-sized input: reads of 5-15 kb, two 100-base flanks,
both strands = 4 alignments per read.  One JSON line: alignments/s, DP cells/s, HBM roofline by algorithmic bytes (each
alignment streams its read once: n bytes), CPU restatement on a bounded sample.
    python scripts/flank_align_bench.py [n_reads]"""
import json, sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads, vntr_finder
from oracle import oracle as O

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(11)
left, right, pattern = workloads.rand_seq(rng, 100), workloads.rand_seq(rng, 100), workloads.rand_seq(rng, 40)
reads = []
for k in range(n_reads):
    n = int(rng.integers(5000, 15001))
    s = workloads.rand_seq(rng, n)
    if k % 3 == 0:
        core = workloads.noisy_copy(rng, left + pattern * int(rng.integers(3, 20)) + right, 0.12)
        at = int(rng.integers(0, n - len(core)))
        s = s[:at] + core + s[at + len(core):]
    reads.append(s)
# both strands of every read: strand 2 r + 1 = the reverse complement of read r, which the device makes from the uploaded read
# (pair_read = n_reads + r); the strings below are only what the CPU restatement is given
strands = []
for s in reads:
    strands += [s, vntr_finder.reverse_complement(s)]
strand_read = np.arange(2 * n_reads, dtype=np.int32) // 2 + (np.arange(2 * n_reads, dtype=np.int32) & 1) * n_reads
pr_dev = np.repeat(strand_read, 2)
pr = np.repeat(np.arange(len(strands), dtype=np.int32), 2)
pf = np.tile(np.array([0, 1], np.int32), len(strands))
_lib.flank_align(reads[:8], [left, right], np.arange(16, dtype=np.int32), pf[:16])          # warm-up
t0 = time.perf_counter()
score, begin, end, ms = _lib.flank_align(reads, [left, right], pr_dev, pf)
wall = time.perf_counter() - t0
cells = float(sum(len(strands[r]) for r in pr)) * 100
bytes_alg = float(sum(len(strands[r]) for r in pr))
# steps of the sweep: read length + flank length - 1 anti-diagonals, two 64-column chunks each (a 100-base flank fills 100 of 128)
steps = float(sum(len(strands[r]) + 99 for r in pr))           # steps of 128 cells (two 64-column chunks packed in one register)
n_cpu = 24
t0 = time.perf_counter()
for p in range(n_cpu):
    got = O.flank_align(strands[pr[p]], [left, right][pf[p]])
    assert got == (int(score[p]), int(begin[p]), int(end[p])), p
cpu = n_cpu / (time.perf_counter() - t0)
print(json.dumps({"metric": "flank alignments/s (100-base flank vs 5-15 kb read, Smith-Waterman 1/-1/-1)",
                  "value": len(pr) / (ms * 1e-3), "unit": "alignments/s", "n_gpus": 1, "dtype": "i32", "data": "synthetic",
                  "config": {"workload": "%d reads of 5-15 kb x 2 strands x 2 flanks" % n_reads, "alignments": int(len(pr)),
                             "cells_per_s": cells / (ms * 1e-3), "kernel_ms": ms, "call_ms_incl_pcie_and_host": wall * 1e3,
                             "spanning_found": int(((score[0::2] >= 70) & (score[1::2] >= 70) & (begin[1::2] >= begin[0::2])).sum())},
                  "roofline": {"bound": "hbm", "achieved": bytes_alg / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": bytes_alg / (ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                               "note": "tier rule (HBM) only: the binding roof is int32 VALU issue, see bound_actual",
                               "bound_actual": {"bound": "valu_int", "unit": "DP cells/s",
                                                "achieved": cells / (ms * 1e-3),
                                                # 18 wave64 VALU instructions per step of 128 cells (round 3: 28 per 64 cells), at the
                                                # nominal 2 cycles each on 1024 SIMDs at 2.4 GHz; 15 of them are DPP / VOP3P / VOP3
                                                # encodings, which this part issues at ~4.5 cycles (profiles/r01_valu_ubench.txt)
                                                "valu_per_step_of_128_cells": 18,
                                                "peak": 128.0 / (18 * 2) * 1024 * 2.4e9,
                                                "frac": cells / (ms * 1e-3) / (128.0 / (18 * 2) * 1024 * 2.4e9),
                                                "cycles_per_step_measured": ms * 1e-3 * 2.4e9 * 1024 / steps,
                                                "cycles_per_step_at_measured_issue_rates": 15 * 4.5 + 3 * 2.6}},
                  "cpu_baseline": {"value": cpu, "unit": "alignments/s", "cores": 1, "kind": "port",
                                   "sample": "first %d alignments, oracle/flank_align_oracle.c (biopython itself is absent: parity "
                                             "unpinned); results equal to the GPU's" % n_cpu}}))
