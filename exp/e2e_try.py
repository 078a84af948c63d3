import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads, vntr_finder, hmm_utils
n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 6719
loci, reads, which, counts = workloads.make_c2_parallel(n_loci, seed=20240602, build=False, return_counts=True)
desc = [(l.left, l.right, l.units, l.copies) for l in loci]
first = np.concatenate([[0], np.cumsum([nm + 2 * nu for nm, nu in counts])])
cand = [reads[first[k]:first[k] + nm + nu] for k, (nm, nu) in enumerate(counts)]
hmm_utils.build_read_matcher_models(desc[:4])
vntr_finder.score_reads_arrays(hmm_utils.build_read_matcher_models(desc[:1]), [cand[0][:8]])
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
for chunks in (8, 8, 12, 16, 24, 12):
    P = {}
    vntr_finder.genotype_loci_pipelined(desc, cand, chunks=chunks, timings=P)
    print(chunks, json.dumps({k: round(v, 4) for k, v in P.items()}))
