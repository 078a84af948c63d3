"""Differential fuzz of the flank-alignment kernel (csrc/flank_align.h) against its CPU restatement
(oracle/flank_align_oracle.c): random reads of 0-20 000 bases with noisy copies of the flanks planted, flanks of 1-128 bases (the kernel's limit; the reference aligns 100-base flanks),
low-complexity sequences (ties in score and in the walk-back), N; score, begin and end must agree.  (Parity with biopython's
pairwise2 itself stays unpinned: it is not in the image.)  Usage: python scripts/fuzz_flank_align.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from advntr_amd import _lib, workloads
from oracle import oracle as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)


def noisy(s, rate):
    out = []
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append("ACGT"[int(rng.integers(0, 4))])
        if 2 * rate / 3 <= x < rate:
            ch = "ACGT"[int(rng.integers(0, 4))]
        out.append(ch)
    return "".join(out)


def low_complexity(n):
    unit = workloads.rand_seq(rng, int(rng.integers(1, 5)))
    return (unit * (n // len(unit) + 1))[:n]


t0 = time.time()
total = 0
for case in range(n_cases):
    flanks = [workloads.rand_seq(rng, int(rng.integers(1, 129))) if rng.random() < 0.8 else low_complexity(int(rng.integers(1, 120)))
              for _ in range(int(rng.integers(1, 10)))]
    reads = []
    for _ in range(int(rng.integers(1, 40))):
        n = int(rng.choice([0, 1, 3, 63, 64, 65, 300, 2000, 9000, 20000]))
        body = workloads.rand_seq(rng, n) if rng.random() < 0.85 else low_complexity(n)
        if n >= 300:
            for _ in range(int(rng.integers(0, 4))):
                f = flanks[int(rng.integers(0, len(flanks)))]
                at = int(rng.integers(0, max(1, n - len(f))))
                c = noisy(f, float(rng.choice([0.0, 0.05, 0.15, 0.3])))
                body = (body[:at] + c + body[at + len(c):])[:n]
        if n > 10 and rng.random() < 0.2:
            p = int(rng.integers(0, n))
            body = body[:p] + "N" + body[p + 1:]
        reads.append(body)
    pr, pf = [], []
    for r in range(len(reads)):
        for f in range(len(flanks)):
            if rng.random() < 0.7:
                pr.append(r); pf.append(f)
    if not pr:
        continue
    score, begin, end, ms = _lib.flank_align(reads, flanks, pr, pf)
    for p in range(len(pr)):
        want = O.flank_align(reads[pr[p]], flanks[pf[p]])
        assert (int(score[p]), int(begin[p]), int(end[p])) == want, (case, p, len(reads[pr[p]]), len(flanks[pf[p]]), want,
                                                                    (int(score[p]), int(begin[p]), int(end[p])))
    total += len(pr)
print("flank-align fuzz ok: %d cases, %d alignments, %.1f s" % (n_cases, total, time.time() - t0))
