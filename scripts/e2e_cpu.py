"""How much CPU the pipelined end-to-end run burns (process CPU seconds, all threads) next to its wall time, per piece plan:
python scripts/e2e_cpu.py [n_loci]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as e
e.build()
from advntr_amd import workloads, vntr_finder, _lib, hmm_utils
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6719
loci, reads, which, counts = workloads.make_c2_parallel(n, seed=20240602, build=False, return_counts=True)
desc = [(l.left, l.right, l.units, l.copies) for l in loci]
first = np.concatenate([[0], np.cumsum([nm + 2 * nu for nm, nu in counts])])
cand = [reads[first[k]:first[k] + nm + nu] for k, (nm, nu) in enumerate(counts)]
import gc
gc.collect(); gc.freeze()
_lib.require_gpu()
vntr_finder.genotype_loci_pipelined(desc[:64], cand[:64], chunks=2)
def cpu_wall(f):
    c0, w0 = time.process_time(), time.perf_counter()
    r = f()
    return time.process_time() - c0, time.perf_counter() - w0, r
print("host threads", _lib.load().advntr_host_threads())
for rep in range(3):
    c, w, models = cpu_wall(lambda: hmm_utils.build_read_matcher_models(desc))
    print("build all models: cpu %.2f s wall %.3f s" % (c, w))
    c, w, _ = cpu_wall(lambda: vntr_finder.device_models(models))
    print("upload all models: cpu %.2f s wall %.3f s" % (c, w))
    c, w, prep = cpu_wall(lambda: vntr_finder._prepare_reads(cand))
    print("encode all reads: cpu %.2f s wall %.3f s" % (c, w))
    c, w, batch = cpu_wall(lambda: _lib.DeviceBatch(vntr_finder.device_models(models), prep["bases"], prep["off"], prep["locus"], flags=_lib.FLAG_BOTH_STRANDS))
    print("bind: cpu %.2f s wall %.3f s" % (c, w))
    def score():
        batch.run(); r = batch.recruit(None, 2); batch.close(); return r
    c, w, _ = cpu_wall(score)
    print("score+recruit: cpu %.2f s wall %.3f s" % (c, w))
    del models, prep, batch
    for plan in (dict(chunks=8), dict(chunks=16)):
        c, w, _ = cpu_wall(lambda: vntr_finder.genotype_loci_pipelined(desc, cand, **plan))
        print("%-30s cpu %.2f s  wall %.3f s  -> %.1f cores" % (plan, c, w, c / w), flush=True)
