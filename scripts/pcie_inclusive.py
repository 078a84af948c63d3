"""PCIe-inclusive rate of the one-shot C-ABI call (host buffers in, host results out), for DESIGN.md section 8."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads
loc = workloads.ref150()
reads = workloads.make_reads(np.random.default_rng(20240601), loc, 100000, 150)
bases, off = _lib.encode_reads(reads)
dm = loc.model.device_model()
which = np.zeros(len(reads), np.int32)
_lib.viterbi_batch([dm], bases[:1500], off[:11], which[:10])           # warm up (module load, first hipMalloc)
for _ in range(8):
    t = time.perf_counter()
    logp, summ, _ = _lib.viterbi_batch([dm], bases, off, which)
    dt = time.perf_counter() - t
    print("one-shot advntr_viterbi_batch, 100k reads from host buffers: %.1f ms -> %.2f M reads/s" % (dt * 1e3, 1e-1 / dt))
B = _lib.DeviceBatch([dm], bases, off, which)
t = time.perf_counter(); B.run(); B.sync(); dt = time.perf_counter() - t
print("resident batch, run+sync: %.1f ms" % (dt * 1e3))
for _ in range(2):
    t0 = time.perf_counter(); B2 = _lib.DeviceBatch([dm], bases, off, which); t1 = time.perf_counter()
    B2.run(); B2.sync(); t2 = time.perf_counter(); r = B2.fetch(); t3 = time.perf_counter(); B2.close(); t4 = time.perf_counter()
    print("create %.2f ms, run %.2f ms, fetch %.2f ms, destroy %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
