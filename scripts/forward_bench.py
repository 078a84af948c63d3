"""Model.log_probability on the bench batches (100 000 reads of 150 bases on the REF150 and S300 models), resident in HBM:
advntr_batch_forward_timed (HIP events on the launch stream), the one-shot call from host buffers, and the agreement of the
two.  Run under `rocprofv3 --kernel-trace --stats` this is the kernel-trace summary of forward_rows_kernel<5, 2> alone.
    python scripts/forward_bench.py [n_reads]"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
shapes = [x for x in os.environ.get("FWD_SHAPES", "REF150,S300").split(",") if x]          # (one shape per profiler run)
for name, locus in (("REF150", workloads.ref150()), ("S300", workloads.s300())):
    if name not in shapes:
        continue
    reads = workloads.make_reads(np.random.default_rng(20240601), locus, n, 150)
    bases, off = _lib.encode_reads(reads)
    dm = locus.model.device_model()
    which = np.zeros(n, np.int32)
    B = _lib.DeviceBatch([dm], bases, off, which, flags=_lib.FLAG_NO_SUMMARY)
    B.forward(); B.sync()
    ms = B.forward_timed(10)
    lp, _ = B.fetch()
    t0 = time.perf_counter()
    one = _lib.forward_batch([dm], bases, off, which)
    dt = time.perf_counter() - t0
    exp = bool(os.environ.get('ADVNTR_EXP'))          # experiment builds (timing only): results are not checked
    assert exp or np.array_equal(lp, one)
    # deep tiles (full back-to-back depth everywhere) and the generic kernel's pair_lse order as a cross-check on a sample
    k = 0 if os.environ.get("FWD_NO_GENERIC") else min(n, 3000)
    gen = _lib.forward_batch([dm], bases[:off[k]], off[:k + 1], which[:k], flags=_lib.FLAG_FORCE_GENERIC) if k else np.zeros(0)
    worst = float(np.max(np.abs(lp[:k] - gen) / np.maximum(1.0, np.abs(gen)))) if k else 0.0
    cells = float(n) * 150 * dm.n_columns()
    print("%s: forward_rows kernel %.3f ms per %d reads = %.2f M reads/s; %.1f TFLOP/s at 11 FMA per cell; one-shot call %.2f ms; "
          "max rel diff vs generic kernel on %d reads %.2e" % (name, ms, n, n / ms / 1e3, cells * 22 / (ms * 1e-3) / 1e12, dt * 1e3, k, worst))
    assert exp or worst < 1e-9
    B.close()
