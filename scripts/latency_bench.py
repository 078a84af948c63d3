"""Latency of the pomegranate-shaped single calls (model.viterbi(seq)) and of small one-shot batches."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads
loc = workloads.ref150()
reads = workloads.make_reads(np.random.default_rng(2), loc, 2000, 150)
m = loc.model
m.viterbi(reads[0])
t = time.perf_counter()
for r in reads[:200]:
    m.viterbi(r)
dt = (time.perf_counter() - t) / 200
print("model.viterbi(seq) single call (with path): %.3f ms" % (dt * 1e3))
for nb in (1, 16, 160, 2000):
    t = time.perf_counter()
    for _ in range(20):
        m.viterbi_batch(reads[:nb])
    dt = (time.perf_counter() - t) / 20
    print("viterbi_batch of %4d reads: %.3f ms  (%.0f reads/s)" % (nb, dt * 1e3, nb / dt))
