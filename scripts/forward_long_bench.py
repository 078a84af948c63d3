"""Run on the GPU box: log_probability of long reads (row-tiled anti-diagonal sum-product kernel), one-shot calls."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advntr_amd import _lib, workloads
rng = np.random.default_rng(3)
loc = workloads.make_locus(rng, 100, 30, 40, error_rate=0.3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reads = [workloads.make_reads(rng, loc, 1, int(k), locus_fraction=1.0, sub_rate=0.1)[0] for k in rng.integers(900, 1500, n)]
bases, off = _lib.encode_reads(reads)
dm = loc.model.device_model()
which = np.zeros(len(reads), np.int32)
_lib.forward_batch([dm], bases[:off[50]], off[:51], which[:50])
for _ in range(3):
    t = time.perf_counter(); lp = _lib.forward_batch([dm], bases, off, which); dt = time.perf_counter() - t
    print("forward, %d reads of 900-1500 bases: %.1f ms -> %.0f reads/s" % (len(reads), dt * 1e3, len(reads) / dt))
sub = 300
lg = _lib.forward_batch([dm], bases[:off[sub]], off[:sub + 1], which[:sub], flags=_lib.FLAG_FORCE_GENERIC)
print("max rel diff vs generic kernel on %d reads: %.2e" % (sub, float(np.max(np.abs(lg - lp[:sub]) / np.abs(lg)))))
