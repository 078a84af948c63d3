import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads
loc = workloads.ref150()
dm = loc.model.device_model()
for nreads in (16, 64, 160, 500, 1000, 2000, 4000):
    reads = workloads.make_reads(np.random.default_rng(nreads), loc, nreads, 150)
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(nreads, np.int32)
    out = []
    for flags in (0, _lib.FLAG_ANTIDIAGONAL):
        B = _lib.DeviceBatch([dm], bases, off, which, flags=flags)
        B.run()
        ms = B.run_timed(20)
        B.close()
        out.append(ms)
    print(nreads, "default %.3f ms  antidiagonal %.3f ms" % tuple(out))
