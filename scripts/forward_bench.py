import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e; e.build()
from advntr_amd import _lib, workloads
loc = workloads.ref150()
reads = workloads.make_reads(np.random.default_rng(1), loc, 100000, 150)
bases, off = _lib.encode_reads(reads)
dm = loc.model.device_model()
_lib.forward_batch([dm], bases[:off[100]], off[:101], np.zeros(100, np.int32))
t = time.perf_counter(); lp = _lib.forward_batch([dm], bases, off, np.zeros(len(reads), np.int32)); dt = time.perf_counter() - t
print("forward, column program (one-shot call incl. PCIe/alloc): %d reads in %.1f ms -> %.0f reads/s" % (len(reads), dt * 1e3, len(reads) / dt))
sub = 20000
t = time.perf_counter(); lg = _lib.forward_batch([dm], bases[:off[sub]], off[:sub + 1], np.zeros(sub, np.int32), flags=_lib.FLAG_FORCE_GENERIC); dt = time.perf_counter() - t
print("forward, generic kernel: %d reads in %.1f ms -> %.0f reads/s; max rel diff vs column %.2e" % (sub, dt * 1e3, sub / dt, float(np.max(np.abs(lg - lp[:sub]) / np.abs(lg)))))
