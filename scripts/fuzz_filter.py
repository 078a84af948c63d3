"""Differential fuzz of the keyword prefilter against the REFERENCE binary (oracle/_ref/adVNTR-Filtering = the reference's
filtering/main.cc compiled untouched; it ships to the GPU box with the other built files) and the CPU restatement:
random keyword sets (lengths 5-100 mixed, strings shared between VNTRs), reads with N, lower case, reads shorter than the
keywords, several min_matches.  stdout must be identical byte for byte.
Usage: python scripts/fuzz_filter.py [n_cases] [seed]"""
import os, subprocess, sys, tempfile, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from advntr_amd import filtering
from oracle import filter_oracle as F

BIN = os.path.join(REPO, "oracle", "_ref", "adVNTR-Filtering")
have_ref = os.path.exists(BIN)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
seq = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def run_ref(fasta, keywords, min_matches):
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "reads.fa")
        open(fa, "w").write(fasta)
        cmd = [BIN, fa] + (["--min_matches", str(min_matches)] if min_matches is not None else [])
        return subprocess.run(cmd, input=keywords.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()


t0 = time.time()
n_ref = 0
for case in range(n_cases):
    mode = int(rng.integers(0, 3))                    # 0: one short length, 1: mixed lengths, 2: long (flank) keywords
    n_loci = int(rng.integers(1, 60))
    loci, lines = [], []
    pool = []
    lens1 = rng.choice([5, 9, 15, 21, 29, 30, 31, 40, 64], int(rng.integers(2, 9)), replace=False)      # the filter takes up to 8 distinct lengths
    # one length of at most 16 bases runs on keyword_filter_short_kernel (both its instantiations: 15 / 16 bases and shorter)
    short_len = int(rng.choice([15, 15, 15, 16, 14, 12, 9, 5, 3, 1]))
    for v in range(n_loci):
        full = seq(int(rng.integers(120, 400)))
        kws = set()
        for _ in range(int(rng.integers(1, 12))):
            if mode == 0: L = short_len
            elif mode == 1: L = int(rng.choice(lens1))
            else: L = int(rng.choice([80, 100, 33]))
            if L > len(full): continue
            p = int(rng.integers(0, len(full) - L + 1))
            kws.add(full[p:p + L])
        if pool and rng.random() < 0.2:               # a string owned by two VNTRs
            kws.add(pool[int(rng.integers(0, len(pool)))])
        if not kws: kws.add(full[:int(lens1[0]) if mode == 1 else (short_len if mode == 0 else 33)])
        pool.extend(kws)
        loci.append(full)
        lines.append("%d %s" % (1000 + v * 3, " ".join(sorted(kws))))
    keywords = "\n".join(lines) + "\n"
    fasta = []
    for r in range(int(rng.integers(1, 1500))):
        u = rng.random()
        if u < 0.5:
            full = loci[int(rng.integers(0, n_loci))]
            st = int(rng.integers(0, max(1, len(full) - 30)))
            s = full[st:st + int(rng.integers(10, 260))]
        else:
            s = seq(int(rng.integers(1, 260)))
        if rng.random() < 0.15 and s:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + "N" + s[p + 1:]
        if rng.random() < 0.05:
            s = s.lower()
        fasta.append(">r%d\n%s\n" % (r, s))
    fasta = "".join(fasta)
    mm = [None, 1, 2, 5, 9][int(rng.integers(0, 5))]
    got = filtering.run(fasta, keywords, min_matches=5 if mm is None else mm)
    want = F.run_filter(fasta, keywords, min_matches=5 if mm is None else mm)
    assert got == want, ("vs restatement", case, mode, mm)
    if case % 3 == 0:             # the same through the encoded-reads entry (advntr_kwfilter_scan instead of _scan_text)
        f = filtering.KeywordFilter.from_text(keywords)
        recs = [l for l in fasta.split("\n") if l]
        via_codes = f.select([l[1:] for l in recs[0::2]], recs[1::2], min_matches=5 if mm is None else mm)
        f.close()
        assert via_codes == want, ("encoded reads vs restatement", case, mode, mm)
    if have_ref:
        assert got == run_ref(fasta, keywords, mm), ("vs reference binary", case, mode, mm)
        n_ref += 1
print("filter fuzz ok: %d cases (%d against the reference binary), %.1f s" % (n_cases, n_ref, time.time() - t0))
