#!/usr/bin/env python3
"""Golden vectors of the keyword prefilter: inputs + the stdout of the REFERENCE binary itself.

Build container only:   make -C oracle _ref/adVNTR-Filtering && python tests/golden/make_filter_golden.py
(oracle/_ref/adVNTR-Filtering is /root/reference/filtering/main.cc compiled with g++ -O2, untouched.)
Keywords are produced the way the reference does for short reads (vntr_finder.py:140-153 with keyword_size=15 as
genome_analyzer.py:180 calls it): 15-mers every 5 bases (6 when the pattern has length 5) over
left_flank[-15:] + repeats + right_flank[:15]."""
import gzip
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
BIN = os.path.join(REPO, "oracle", "_ref", "adVNTR-Filtering")


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def keywords_for(left, repeats, right, pattern_len, k=15):
    vntr = "".join(repeats)
    if len(vntr) < k:
        vntr = vntr * (int(k / len(vntr)) + 1)
    locus = left[-15:] + vntr + right[:15]
    step = 5 if pattern_len != 5 else 6
    return sorted(set(locus[i:i + k] for i in range(0, len(locus) - k + 1, step)))


def run_ref(fasta, keywords, min_matches=None):
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "reads.fa")
        open(fa, "w").write(fasta)
        cmd = [BIN, fa] + (["--min_matches", str(min_matches)] if min_matches is not None else [])
        return subprocess.run(cmd, input=keywords.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()


def make_case(seed, n_loci, n_reads, read_len, min_matches, with_n=True, dup_id=False):
    rng = np.random.default_rng(seed)
    loci, kw_lines = [], []
    for v in range(n_loci):
        plen = int(rng.integers(5, 40))
        pat = rand_seq(rng, plen)
        reps = [pat] * int(rng.integers(2, 8))
        left, right = rand_seq(rng, 60), rand_seq(rng, 60)
        vid = 1000 + 7 * v
        loci.append((vid, left, reps, right))
        kws = keywords_for(left, reps, right, plen)
        if v == 1 and n_loci > 2:
            kws = kws + keywords_for(loci[0][1], loci[0][2], loci[0][3], len(loci[0][2][0]))[:3]   # shared keywords
        kw_lines.append("%d %s" % (vid, " ".join(kws)))
    if dup_id:
        kw_lines.append(kw_lines[0])
    fasta = []
    for r in range(n_reads):
        u = rng.random()
        if u < 0.5:
            vid, left, reps, right = loci[int(rng.integers(0, n_loci))]
            full = left + "".join(reps) + right
            st = int(rng.integers(0, max(1, len(full) - read_len // 2)))
            s = full[st:st + read_len]
            s = s + rand_seq(rng, read_len - len(s))
            s = "".join(("ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.01 else ch) for ch in s)
        else:
            s = rand_seq(rng, read_len)
        if with_n and rng.random() < 0.15:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + rng.choice(["N", "n", "a"]) + s[p + 1:]
        fasta.append(">read_%05d\n%s\n" % (r, s))
    fasta, keywords = "".join(fasta), "\n".join(kw_lines) + "\n"
    return {"fasta": fasta, "keywords": keywords, "min_matches": min_matches,
            "stdout": run_ref(fasta, keywords, min_matches)}


def make_long_case(seed, n_loci, n_reads, min_matches, mixed=False):
    """The long-read mode of get_keywords_for_filtering (vntr_finder.py:151-152): the keywords of a VNTR are its two
    80-base flanks.  Reads of 200-2500 bases carrying whole flanks, flanks with one substitution, flanks cut off by the
    read end, an N inside a flank, several copies of a flank (to reach min_matches), lower-case stretches (no match in the
    reference: char_to_num is case sensitive).  mixed: some VNTRs also get 15-mers, a 29-mer and a 30-mer; two VNTRs share
    a flank; two flanks share their first 29 bases."""
    rng = np.random.default_rng(seed)
    loci, kw_lines = [], []
    for v in range(n_loci):
        left, right = rand_seq(rng, 120), rand_seq(rng, 120)
        if v == 2 and n_loci > 3:
            left = loci[0][1]                                            # a flank shared by two VNTRs
        if v == 3 and n_loci > 3:
            right = loci[1][2][:29] + rand_seq(rng, 91)                  # same 29-base prefix, different continuation
        kws = [left[-80:], right[:80]]
        if mixed and v % 3 == 0:
            kws += keywords_for(left, [rand_seq(rng, 12)] * 3, right, 12)[:4] + [left[-29:], right[:30]]
        vid = 300 + 11 * v
        loci.append((vid, left, right))
        kw_lines.append("%d %s" % (vid, " ".join(kws)))
    fasta = []
    for r in range(n_reads):
        n = int(rng.integers(200, 2500))
        parts, have = [], 0
        while have < n:
            u = rng.random()
            vid, left, right = loci[int(rng.integers(0, n_loci))]
            if u < 0.35:
                piece = rand_seq(rng, int(rng.integers(20, 300)))
            elif u < 0.7:
                piece = (left[-80:] if rng.random() < 0.5 else right[:80])
                if rng.random() < 0.4:
                    piece = piece * int(rng.integers(2, 5))             # several occurrences in one read
            elif u < 0.8:
                piece = list(left[-80:])
                q = int(rng.integers(0, 80))
                piece[q] = "ACGT"[("ACGT".index(piece[q]) + 1) % 4]    # one substitution: no match
                piece = "".join(piece)
            elif u < 0.9:
                piece = left[-80:]
                q = int(rng.integers(0, 80))
                piece = piece[:q] + rng.choice(["N", "a", "c"]) + piece[q + 1:]
            else:
                piece = left[-100:] + rand_seq(rng, 5) + right[:100]
            parts.append(piece)
            have += len(piece)
        s = "".join(parts)[:n]
        if rng.random() < 0.1:
            s = s + loci[r % n_loci][1][-80:][:int(rng.integers(29, 80))]   # a flank cut off by the end of the read
        fasta.append(">lr_%04d\n%s\n" % (r, s))
    fasta, keywords = "".join(fasta), "\n".join(kw_lines) + "\n"
    return {"fasta": fasta, "keywords": keywords, "min_matches": min_matches,
            "stdout": run_ref(fasta, keywords, min_matches)}


def main():
    cases = {
        "filter_long80": make_long_case(11, 6, 120, 1),
        "filter_long_mixed": make_long_case(12, 9, 150, 2, mixed=True),
        "filter_small": make_case(1, 5, 300, 100, None),
        "filter_min2": make_case(2, 12, 400, 150, 2),
        "filter_dup_id": make_case(3, 3, 150, 80, 3, dup_id=True),
    }
    for name, c in cases.items():
        path = os.path.join(HERE, name + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(json.dumps(c, separators=(",", ":")).encode())
        n_hit = sum(1 for l in c["stdout"].split("\n") if l and not l.split()[1].isdigit())
        print("wrote", path, os.path.getsize(path), "bytes;", n_hit, "reads selected")


if __name__ == "__main__":
    if not os.path.exists(BIN):
        sys.exit("build the reference binary first: make -C oracle _ref/adVNTR-Filtering")
    main()
