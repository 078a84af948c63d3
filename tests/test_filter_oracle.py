"""Keyword prefilter: the CPU restatement (oracle/filter_oracle.py) against the stdout of the reference binary
(tests/golden/filter_*.json.gz, produced by oracle/_ref/adVNTR-Filtering = filtering/main.cc compiled as is)."""
import os
import subprocess
import tempfile

import pytest

from conftest import ROOT, load_golden
from oracle import filter_oracle as F

CASES = ["filter_small", "filter_min2", "filter_dup_id", "filter_long80", "filter_long_mixed"]


@pytest.mark.parametrize("name", CASES)
def test_restatement_reproduces_reference_stdout(name):
    g = load_golden(name)
    mm = g["min_matches"] if g["min_matches"] is not None else 5
    assert F.run_filter(g["fasta"], g["keywords"], min_matches=mm) == g["stdout"]


def test_against_live_reference_binary_when_present():
    """Only in the build container: fresh random case through oracle/_ref (skipped on the GPU box)."""
    binary = os.path.join(ROOT, "oracle", "_ref", "adVNTR-Filtering")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref not built here")
    import numpy as np
    rng = np.random.default_rng(99)
    seq = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    kws = {5: [seq(15) for _ in range(6)], 6: [seq(15) for _ in range(4)]}
    kws[6].append(kws[5][0])                                            # a keyword shared by two VNTRs
    keywords = "".join("%d %s\n" % (v, " ".join(k)) for v, k in kws.items())
    fasta = ""
    for r in range(120):
        s = seq(90)
        for _ in range(int(rng.integers(0, 8))):
            p = int(rng.integers(0, 75))
            k = kws[5 + int(rng.integers(0, 2))]
            s = s[:p] + k[int(rng.integers(0, len(k)))] + s[p + 15:]
        fasta += ">r%d\n%s\n" % (r, s)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "x.fa")
        open(fa, "w").write(fasta)
        ref = subprocess.run([binary, fa, "--min_matches", "2"], input=keywords.encode(), stdout=subprocess.PIPE,
                             check=True).stdout.decode()
    assert F.run_filter(fasta, keywords, min_matches=2) == ref
    ids, reads = F.parse_output(ref)
    assert set(ids) == {5, 6} and len(reads) > 0
