"""Where the host time of the end-to-end run goes, stage by stage (scripts/end_to_end_bench.py measures the totals):
model build, model upload, read flattening / encoding / reverse complements, batch creation (validation, routing sort,
uploads), kernels, result download, recruit rule, per-locus aggregation + genotypes.
    python scripts/host_profile.py [n_loci]"""
import itertools, json, sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads, vntr_finder, hmm_utils
from advntr_amd.pomegranate import device_models

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 6719
_lib.require_gpu()
loci, reads, which = workloads.make_c2_parallel(n_loci, build=False, unmapped_mean=20)
per_locus = [[] for _ in loci]
for r, k in zip(reads, which):
    per_locus[int(k)].append(r)
hmm_utils.build_read_matcher_models([(l.left, l.right, l.units, l.copies) for l in loci[:4]])
vntr_finder.score_reads_arrays([loci[0].model], [per_locus[0][:8]])
T = {}
def lap(name, t0):
    T[name] = round(time.perf_counter() - t0, 4)
    return time.perf_counter()
t = time.perf_counter()
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
built = _lib.build_read_matchers([l.left for l in loci], [l.right for l in loci], [list(l.units) for l in loci],
                                 [int(l.copies) for l in loci], 0.05)
pr.disable()
t = lap("build_native", t)
sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(8); sys.stderr.write(sio.getvalue())
from advntr_amd.pomegranate import HiddenMarkovModel
models = [HiddenMarkovModel._from_built(b, 'Read Matcher') for b in built]
t = lap("wrap_models_python", t)
dms = device_models(models)
t = lap("upload_models", t)
flat = list(itertools.chain.from_iterable(per_locus))
t = lap("flatten_lists", t)
all_len = np.fromiter(map(len, flat), dtype=np.int64, count=len(flat))
off = np.zeros(len(flat) + 1, np.int64); np.cumsum(all_len, out=off[1:])
t = lap("lengths_offsets", t)
raw = "".join(flat).encode("latin-1", "replace")
t = lap("join_encode_bytes", t)
bases = np.empty(len(raw), np.uint8); bad = np.zeros(len(flat), np.uint8)
_lib.check(_lib.load().advntr_encode_ascii(raw, off.ctypes.data, len(flat), 0, bases.ctypes.data, bad.ctypes.data))
t = lap("encode_ascii_native", t)
rc = (3 - bases)[::-1]
both = np.concatenate([bases, rc])
off2 = np.concatenate([off, off[-1] + np.cumsum(all_len[::-1])])
counts = np.array([len(x) for x in per_locus])
wl = np.repeat(np.arange(n_loci, dtype=np.int32), counts)
which2 = np.concatenate([wl, wl[::-1]])
t = lap("revcomp_concat", t)
B = _lib.DeviceBatch(dms, both, off2, which2)
t = lap("batch_create", t)
B.run(); B.sync()
t = lap("batch_run_sync", t)
logp, summ = B.fetch()
t = lap("batch_fetch", t)
kernel_ms = B.run_timed(2)
B.close()
t = time.perf_counter()
nf = len(flat)
rlogp, rsumm = logp[nf:][::-1], summ[nf:][::-1]
use_rev = logp[:nf] < rlogp
lp = np.where(use_rev, rlogp, logp[:nf]); sm = np.where(use_rev[:, None], rsumm, summ[:nf])
t = lap("strand_choice", t)
rec = vntr_finder.recruit_mask(lp, sm, all_len, np.full(nf, np.nan))
t = lap("recruit_mask", t)
keep = rec & (sm[:, _lib.SUM_REPEAT_BP] > 2)
order = np.argsort(wl[keep], kind="stable")
s2 = sm[keep][order]
bounds = np.searchsorted(wl[keep][order], np.arange(n_loci + 1))
t = lap("group_by_locus", t)
g = vntr_finder.find_repeat_counts_of_loci(s2, bounds.astype(np.int64))
t = lap("aggregate_genotype_native", t)
print(json.dumps({"loci": n_loci, "reads": nf, "calls": 2 * nf, "kernel_ms": kernel_ms, "seconds": T, "sum_s": round(sum(T.values()), 3)}))
