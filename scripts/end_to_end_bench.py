"""End-to-end wall time of a genome-scale Illumina genotyping pass on already-selected reads (SURVEY 8d config C2:
6719 synthetic loci, ~160 candidate reads each): model build (native builder) -> encode + both-strand scoring in one
engine batch -> recruit rule -> per-locus aggregation and maximum-likelihood genotype.  Everything between "reads in
Python lists" and "genotypes" is timed; generating the synthetic reads is not.
    python scripts/end_to_end_bench.py [n_loci]"""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads, vntr_finder, hmm_utils

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 6719
_lib.require_gpu()
rng = np.random.default_rng(5)
t = time.perf_counter()
loci, reads, which = workloads.make_c2_parallel(n_loci, build=False, unmapped_mean=20)
# candidate reads per locus (forward strand only; score_reads_arrays adds the reverse complements itself)
per_locus = [[] for _ in loci]
for r, k in zip(reads, which):
    per_locus[int(k)].append(r)
print("synthetic input: %d loci, %d candidate reads (%.1f s, not timed)" % (n_loci, len(reads), time.perf_counter() - t))
hmm_utils.build_read_matcher_models([(l.left, l.right, l.units, l.copies) for l in loci[:4]])      # warm-up
vntr_finder.score_reads_arrays([loci[0].model], [per_locus[0][:8]])

T = {}
t0 = time.perf_counter()
models = hmm_utils.build_read_matcher_models([(l.left, l.right, l.units, l.copies) for l in loci])
T["build_models"] = time.perf_counter() - t0
t1 = time.perf_counter()
from advntr_amd.pomegranate import device_models
device_models(models)                                   # bulk upload (otherwise done inside the scoring call)
T["upload_models"] = time.perf_counter() - t1
t1 = time.perf_counter()
res = vntr_finder.score_reads_arrays(models, per_locus, None, compute_reverse=True)
T["encode_score_recruit"] = time.perf_counter() - t1
t2 = time.perf_counter()
keep = res["recruited"] & (res["summary"][:, _lib.SUM_REPEAT_BP] > 2)
locus, summ = res["locus"][keep], res["summary"][keep]
bounds = np.searchsorted(locus, np.arange(n_loci + 1)).astype(np.int64)
results = vntr_finder.find_repeat_counts_of_loci(summ, bounds)                   # advntr_genotype_illumina, host threads
genotypes = [g.copy_numbers for g in results]
T["aggregate_genotype"] = time.perf_counter() - t2
total = time.perf_counter() - t0
calls = 2 * len(res["logp"])
called = sum(g is not None for g in genotypes)
# the same through the one-call driver (vntr_finder.genotype_loci) must give the same genotypes
again = [g.copy_numbers for g in vntr_finder.genotype_loci(models, per_locus)]
assert again == genotypes
print(json.dumps({"loci": n_loci, "viterbi_calls": calls, "recruited_reads": int(keep.sum()), "loci_with_genotype": called,
                  "seconds": {k: round(v, 3) for k, v in T.items()}, "total_s": round(total, 3),
                  "calls_per_s_end_to_end": round(calls / total)}))
