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
g = [1 / 64, 1 / 64, 1 / 32, 1 / 16]
plans = [dict(), dict(chunks=16, ramp=4),
         dict(piece_fractions=[1 / 32, 1 / 32, 1 / 16] + [1 / 8] * 7), dict(piece_fractions=g + [1 / 8, 1 / 8] + [1 / 4] * 3),
         dict(chunks=12, ramp=4), dict(chunks=10, ramp=6)]
import collections
res = collections.defaultdict(list)
cpu = collections.defaultdict(list)
for rep in range(8):
    for plan in plans:
        T = {}
        c0 = time.process_time()
        vntr_finder.genotype_loci_pipelined(desc, cand, timings=T, **plan)
        cpu[str(plan)].append(time.process_time() - c0)
        res[str(plan)].append(T["total"])
for k, v in res.items():
    print("%-50s median %.3f  min %.3f  max %.3f   cpu %.2f core-s" % (k, float(np.median(v)), min(v), max(v), float(np.median(cpu[k]))))
