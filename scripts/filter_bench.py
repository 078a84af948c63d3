#!/usr/bin/env python3
"""Throughput of the keyword prefilter (the stage upstream of the scoring path) on one MI355X.

Workload: the 6719-locus Illumina keyword set shape (15-mers every 5 bases over 15+VNTR+15, vntr_finder.py:140-153)
against synthetic 150-base unmapped reads (1 % locus-derived).  Kernel time from HIP events inside
advntr_kwfilter_scan; roofline = HBM read of 1 byte per base.  CPU baseline: the reference binary itself
(oracle/_ref/adVNTR-Filtering = filtering/main.cc, compiled by oracle/Makefile) on a bounded sample of the same
reads, 1 thread, its ~10 s start-up memset of the static automaton tables reported separately."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
e.build()
from advntr_amd import filtering

n_loci = int(os.environ.get("LOCI", 6719)); n_reads = int(os.environ.get("READS", 2000000))
LOCUS_EVERY = int(os.environ.get("LOCUS_EVERY", 100))        # every how many reads one is cut from a locus (0: none)
rng = np.random.default_rng(20240604)
ACGT = np.frombuffer(b"ACGT", np.uint8)
rs = lambda n: ACGT[rng.integers(0, 4, n)].tobytes().decode()
lines, loci = [], []
for v in range(n_loci):
    plen = int(rng.integers(6, 101)); pat = rs(plen); reps = [pat] * int(rng.integers(2, 21))
    left, right = rs(15), rs(15)
    lines.append((v + 1, filtering.get_keywords_for_filtering(left, reps, right, pat, True, 15)))
    if v < 200: loci.append(left + "".join(reps) + right)
n_kw = sum(len(k) for _, k in lines)
seqs = []
big = rs(150 * 50000)
for r in range(n_reads):
    if LOCUS_EVERY and r % LOCUS_EVERY == 0:
        s = loci[r % len(loci)]; s = (s * (150 // len(s) + 1))[:150]
    else:
        o = (r * 137) % (len(big) - 150); s = big[o:o + 150]
    seqs.append(s)
names = ["r%d" % i for i in range(n_reads)]
t = time.perf_counter(); f = filtering.KeywordFilter(lines); t_build = time.perf_counter() - t
f.count_matches(seqs[:1000])
t = time.perf_counter(); counts = f.count_matches(seqs); t_call = time.perf_counter() - t; kernel_ms = f.kernel_ms
# the same reads as the bytes of a FASTA file (what the reference binary is given): line index + encoding + upload + scan +
# aggregation of the hit records, everything but the final text formatting
fasta = "".join(">%s\n%s\n" % (nm, sq) for nm, sq in zip(names, seqs)).encode()
from advntr_amd import _lib
def scan_fasta(text):
    starts = _lib.line_index(text)
    ends = starts[1:] - 1
    k = (len(starts) - 1) // 2
    return f.scan_text(text, starts[1:2 * k:2], ends[1:2 * k:2])
scan_fasta(fasta[:200000])
t = time.perf_counter(); recs = scan_fasta(fasta); t_fasta = time.perf_counter() - t; ms_text = f.kernel_ms
f.count_matches(seqs[:1000])
assert len(set(recs[0].tolist())) == len(counts)
bases = sum(len(s) for s in seqs)
gbps = bases / (kernel_ms * 1e-3) / 1e9
out = {"metric": "bases/s keyword-prefiltered (15-mer keyword sets of %d loci, 150-base reads)" % n_loci,
       "value": bases / (kernel_ms * 1e-3), "unit": "bases/s", "n_gpus": 1, "dtype": "u8", "data": "synthetic",
       "config": {"workload": "prefilter: %d keywords of %d loci x %d reads of 150 bases" % (n_kw, n_loci, n_reads),
                  "reads_with_hits": len(counts), "filter_build_s": t_build, "call_ms_incl_pcie_and_host": t_call * 1e3,
                  "call_ms_from_fasta_bytes_incl_pcie_and_host": t_fasta * 1e3, "fasta_bytes": len(fasta), "kernel_ms_on_text": ms_text},
       "roofline": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                    "traffic": None, "kernel": "keyword_filter_short_kernel", "kernel_ms": kernel_ms, "bytes_per_base": 1}}
binary = os.path.join(ROOT, "oracle", "_ref", "adVNTR-Filtering")
if os.path.exists(binary) and not os.environ.get("NO_CPU"):
    sample = int(os.environ.get("CPU_READS", 200000))
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "s.fa"); kw = os.path.join(d, "kw.txt")
        open(fa, "w").write("".join(">%s\n%s\n" % (names[i], seqs[i]) for i in range(sample)))
        open(kw, "w").write("".join("%d %s\n" % (v, " ".join(sorted(k))) for v, k in lines))
        open(os.path.join(d, "e.fa"), "w").write(">x\nACGT\n")
        t = time.perf_counter(); subprocess.run([binary, os.path.join(d, "e.fa")], stdin=open(kw), stdout=subprocess.DEVNULL, check=True); t0 = time.perf_counter() - t
        t = time.perf_counter(); ref = subprocess.run([binary, fa], stdin=open(kw), stdout=subprocess.PIPE, check=True).stdout.decode(); t1 = time.perf_counter() - t
    mine = f.select(names[:sample], seqs[:sample])
    out["cpu_baseline"] = {"value": sample * 150 / max(t1 - t0, 1e-9), "unit": "bases/s", "cores": 1, "kind": "reference",
                           "sample": "first %d reads through oracle/_ref/adVNTR-Filtering; start-up (automaton build + 1.9 GB memset) %.1f s "
                                     "subtracted from %.1f s; stdout identical to the GPU path: %s" % (sample, t0, t1, mine == ref)}
print(json.dumps(out))
