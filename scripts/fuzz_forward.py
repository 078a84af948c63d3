"""Differential fuzz of Model.log_probability: the linear-domain column kernel vs the generic kernel (pair_lse in the
reference's order) on random loci and reads; relative difference must stay below 1e-9 (the north star allows 1e-4).
Usage: python scripts/fuzz_forward.py [n_loci] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
worst, total = 0.0, 0
for k in range(n_loci):
    loc = workloads.make_locus(rng, int(rng.integers(3, 160)), int(rng.integers(2, 80)), int(rng.integers(1, 14)),
                               float(rng.choice([0.05, 0.3, 0.1])), n_units=int(rng.integers(1, 8)))
    dm = loc.model.device_model()
    reads = []
    for _ in range(int(rng.integers(30, 200))):
        n = int(rng.integers(1, 700))
        r = workloads.make_reads(rng, loc, 1, n, locus_fraction=0.6, sub_rate=float(rng.choice([0.0, 0.01, 0.1])))[0]
        if rng.random() < 0.2:
            r = ("ACGT"[int(rng.integers(0, 4))] * n)
        reads.append(r)
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    a = _lib.forward_batch([dm], bases, off, which)
    b = _lib.forward_batch([dm], bases, off, which, flags=_lib.FLAG_FORCE_GENERIC)
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), ("finite", k)
    rel = np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin]))
    worst = max(worst, float(rel.max()) if rel.size else 0.0)
    assert worst < 1e-9, (k, worst)
    total += len(reads)
print("forward fuzz ok: %d loci, %d reads, max relative difference %.3g, %.1f s" % (n_loci, total, worst, time.time() - t0))
