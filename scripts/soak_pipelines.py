"""Both host pipelines run over and over in one process: same genotypes every pass, no thread left behind, resident memory
levelling off (the page-locked upload segments and the allocator's per-thread arenas fill up over the first passes).
    python scripts/soak_pipelines.py [illumina passes] [pacbio passes]"""
import gc
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                                   # noqa: E402
from advntr_amd import _lib, settings, vntr_finder, workloads       # noqa: E402


def rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * 4096 / 1e6


n_a = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_b = int(sys.argv[2]) if len(sys.argv) > 2 else 12
loci, reads, which, counts = workloads.make_c2_parallel(2000, seed=20240602, build=False, return_counts=True)
desc = [(l.left, l.right, l.units, l.copies) for l in loci]
first = np.concatenate([[0], np.cumsum([nm + 2 * nu for nm, nu in counts])])
cand = [reads[first[k]:first[k] + nm + nu] for k, (nm, nu) in enumerate(counts)]
_lib.require_gpu()
ref = None
for it in range(n_a):
    out = vntr_finder.genotype_loci_pipelined(desc, cand)
    key = [(g.copy_numbers, g.recruited_reads_count, g.maximum_likelihood) for g in out]
    ref = ref or key
    assert key == ref, "pass %d differs" % it
    if it % 20 == 0:
        gc.collect()
        print("genotype_loci_pipelined pass %d: rss %.0f MB, threads %d" % (it, rss_mb(), threading.active_count()), flush=True)
ploci, pread = workloads.make_pacbio_whole_reads(600, seed=20240603, workers=8)
settings.MAX_ERROR_RATE = 0.3
ref = None
for it in range(n_b):
    out = vntr_finder.genotype_pacbio_loci(ploci, pread)
    key = [(g.copy_numbers, g.spanning_reads_count, g.maximum_likelihood) for g in out]
    ref = ref or key
    assert key == ref, "pass %d differs" % it
    if it % 4 == 0:
        gc.collect()
        print("genotype_pacbio_loci pass %d: rss %.0f MB, threads %d" % (it, rss_mb(), threading.active_count()), flush=True)
assert threading.active_count() == 1, [t.name for t in threading.enumerate()]
print("soak ok: %d + %d passes, every pass the same genotypes, no thread left" % (n_a, n_b))
