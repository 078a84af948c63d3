"""genotype_loci_pipelined on the C2 set under different piece plans and per-stage thread counts (one process, one box):
python scripts/e2e_sweep.py [n_loci]"""
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
plans = [dict(chunks=16, ramp=4), dict(chunks=16, ramp=4, stage_threads=(8, 4, 4)), dict(chunks=16, ramp=4, stage_threads=(10, 4, 2)),
         dict(chunks=16, ramp=4, stage_threads=(12, 4, 4)), dict(chunks=16, ramp=4, stage_threads=(12, 8, 4)),
         dict(chunks=16, ramp=4, stage_threads=(16, 8, 4)), dict(chunks=24, ramp=4), dict(chunks=24, ramp=4, stage_threads=(12, 4, 4))]
import collections
res = collections.defaultdict(list)
for rep in range(6):
    for plan in plans:
        T = {}
        vntr_finder.genotype_loci_pipelined(desc, cand, timings=T, **plan)
        res[str(plan)].append(T["total"])
for k, v in res.items():
    print("%-70s median %.3f  min %.3f  max %.3f" % (k, float(np.median(v)), min(v), max(v)))
