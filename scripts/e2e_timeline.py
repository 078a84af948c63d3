"""Timeline of vntr_finder.genotype_loci_pipelined on the C2 set: when every stage call of every piece started and ended
(ms from the start of the run), for several piece plans.   python scripts/e2e_timeline.py [n_loci]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
e.build()
from advntr_amd import workloads, vntr_finder, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6719
loci, reads, which, counts = workloads.make_c2_parallel(n, seed=20240602, build=False, return_counts=True)
desc = [(l.left, l.right, l.units, l.copies) for l in loci]
first = np.concatenate([[0], np.cumsum([nm + 2 * nu for nm, nu in counts])])
cand = [reads[first[k]:first[k] + nm + nu] for k, (nm, nu) in enumerate(counts)]
import gc
gc.collect(); gc.freeze()
_lib.require_gpu()
vntr_finder.genotype_loci_pipelined(desc[:64], cand[:64], chunks=2)
plans = [dict(), dict(chunks=16, ramp=4), dict(), dict(chunks=4), dict(chunks=16)]
for i, plan in enumerate(plans):
    T = {"trace": None}
    t0 = time.perf_counter()
    vntr_finder.genotype_loci_pipelined(desc, cand, timings=T, **plan)
    print("%s total %.3f s  %s" % (plan, T["total"], {k: round(v, 3) for k, v in T.items() if k not in ("trace", "total")}))
    if i in (0, 2):
        for st, k, a, b in sorted(T["trace"], key=lambda x: x[2]):
            print("   %-14s piece %2d  %7.1f -> %7.1f ms  (%.1f)" % (st, k, (a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3))
