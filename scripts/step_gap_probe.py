"""Run on the GPU box: where does the difference between a bench step (enqueue K passes, one sync) and the kernel-only time
(HIP events around the same K passes) come from?  Prints both for K = 1, 3, 20 on the bench workload."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from advntr_amd import workloads, _lib

locus = workloads.ref150() if hasattr(workloads, "ref150") else workloads.c1_locus()
n_reads, n = 100000, 150
reads = workloads.make_reads(np.random.default_rng(20240601), locus, n_reads, n)
bases, off = _lib.encode_reads(reads)
batch = _lib.DeviceBatch([locus.model.device_model()], bases, off, np.zeros(n_reads, np.int32), flags=0)
batch.run(); batch.sync()
for K in (1, 3, 20, 20, 60):
    t0 = time.perf_counter()
    for _ in range(K):
        batch.run()
    batch.sync()
    wall = (time.perf_counter() - t0) / K * 1e3
    ev = batch.run_timed(K)
    print("K=%d  wall ms/pass %.3f   events ms/pass %.3f" % (K, wall, ev))
