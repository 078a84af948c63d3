"""BASELINE config 5 end to end at FULL size, once: 8 960 PacBio loci x 20 whole reads of 5-15 kb (1.8 GB of read text) from the
reads to the RU-count genotypes (vntr_finder.genotype_pacbio_loci = find_repeat_count_from_pacbio_reads, vntr_finder.py:652-665,
for many loci), with the stage split and the HBM high-water mark (rocm-smi sampled beside the run).  Too long for the default
bench line (its `pacbio_end_to_end` record runs a tenth of the loci); the record goes to profiles/.
    python scripts/pacbio_full_size.py [n_loci] [out.json]"""
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e  # noqa: E402
e.build()
from advntr_amd import _lib, settings, vntr_finder, workloads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8960
out_path = sys.argv[2] if len(sys.argv) > 2 else None
t = time.perf_counter()
loci, read_lists = workloads.make_pacbio_whole_reads(n, seed=20240603, workers=max(1, min(32, (os.cpu_count() or 2) - 1)))
gen_s = time.perf_counter() - t
n_reads = sum(len(r) for r in read_lists)
n_bases = sum(len(s) for r in read_lists for s in r)
_lib.require_gpu()
settings.MAX_ERROR_RATE = 0.3


def vram_used():
    try:
        txt = subprocess.run(["rocm-smi", "--showmeminfo", "vram"], stdout=subprocess.PIPE, timeout=10).stdout.decode()
        m = re.search(r"Used Memory \(B\):\s*(\d+)", txt)
        return int(m.group(1)) if m else None
    except Exception:      # noqa: BLE001 -- the sampler is best effort
        return None


peak = {"v": vram_used() or 0, "base": vram_used() or 0, "stop": False}


def sample():
    while not peak["stop"]:
        v = vram_used()
        if v:
            peak["v"] = max(peak["v"], v)
        time.sleep(0.05)


vntr_finder.genotype_pacbio_loci(loci[:8], read_lists[:8], chunks=2)           # warm-up
th = threading.Thread(target=sample)
th.start()
passes = []
res = None
for _ in range(2):
    T = {}
    res = vntr_finder.genotype_pacbio_loci(loci, read_lists, timings=T)
    passes.append(T)
peak["stop"] = True
th.join()
best = min(passes, key=lambda T: T["total"])
# a sample of loci the way the reference walks them, one at a time: same spanning reads, same genotype
import numpy as np  # noqa: E402
same = 0
sample_loci = np.linspace(0, n - 1, 8).astype(int)
for k in sample_loci:
    left, right, segments, pattern = loci[k]
    spanning, _ = vntr_finder.extract_spanning_reads(left, right, read_lists[k])
    want, prob = vntr_finder.get_dominant_copy_numbers_from_spanning_reads(left, right, segments, pattern, [s[0] for s in spanning])
    same += int(res[k].copy_numbers == want and res[k].maximum_likelihood == prob and res[k].spanning_reads_count == len(spanning))
rec = {"workload": "BASELINE config 5 at full size: %d PacBio loci x 20 whole reads of 5-15 kb (the C4 recipe, seed 20240603), one GPU" % n,
       "loci": n, "whole_reads": n_reads, "read_bases": n_bases, "flank_alignments": 4 * n_reads,
       "spanning_reads_scored": int(sum(g.spanning_reads_count for g in res)),
       "loci_with_genotype": sum(g.copy_numbers is not None for g in res),
       "total_s": best["total"], "total_s_of_each_pass": [T["total"] for T in passes],
       "stage_s_overlapped": {k: v for k, v in best.items() if k != "total"},
       "whole_reads_per_s": n_reads / best["total"], "loci_per_s": n / best["total"],
       "hbm_used_bytes_before": peak["base"], "hbm_high_water_bytes": peak["v"],
       "hbm_high_water_source": "rocm-smi --showmeminfo vram sampled every 50 ms beside the run (includes the engine's cached blocks)",
       "per_locus_route_identical_on_sample": "%d of %d loci" % (same, len(sample_loci)),
       "synthetic_input_generated_in_s": gen_s,
       "note": "extraction parity with biopython's pairwise2 is unpinned (absent from the image); kernel == restatement in tests/test_flank_align.py"}
line = json.dumps(rec)
print(line)
if out_path:
    with open(out_path, "w") as fh:
        fh.write(json.dumps(rec, indent=1) + "\n")
assert same == len(sample_loci), "pipelined PacBio route differs from the per-locus route"
