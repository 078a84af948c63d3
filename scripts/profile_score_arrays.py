"""cProfile of vntr_finder.score_reads_arrays on the C2-size set (where the host time of the end-to-end run goes)."""
import cProfile, pstats, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads, vntr_finder, hmm_utils
n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 6719
loci, reads, which = workloads.make_c2_parallel(n_loci, build=False, unmapped_mean=20)
per_locus = [[] for _ in loci]
for r, k in zip(reads, which):
    per_locus[int(k)].append(r)
models = hmm_utils.build_read_matcher_models([(l.left, l.right, l.units, l.copies) for l in loci])
vntr_finder.score_reads_arrays(models[:4], per_locus[:4])
pr = cProfile.Profile()
pr.enable()
res = vntr_finder.score_reads_arrays(models, per_locus, None, compute_reverse=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
