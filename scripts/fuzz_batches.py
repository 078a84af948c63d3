"""Differential fuzz of the batch engine: several models in one batch, reads of mixed lengths (so that one batch goes through
the three short-read configurations, the row-tiled kernel and, now and then, a model without a column program), random
read -> model assignment, both-strands flag; default routing against the generic-CSR kernel, bit for bit.
Usage: python scripts/fuzz_batches.py [n_batches] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
COMP = str.maketrans("ACGT", "TGCA")
t0 = time.time()
total = 0
for b in range(n_batches):
    n_models = int(rng.integers(1, 13))
    loci = [workloads.make_locus(rng, int(rng.integers(3, 160)), int(rng.integers(2, 60)), int(rng.integers(1, 10)),
                                 float(rng.choice([0.05, 0.3])), n_units=int(rng.integers(1, 5))) for _ in range(n_models)]
    dms = [l.model.device_model() for l in loci]
    n_reads = int(rng.integers(1, 1500))
    which = rng.integers(0, n_models, n_reads).astype(np.int32)
    max_len = int(rng.choice([64, 124, 155, 400, 1200]))
    reads = []
    for i in range(n_reads):
        n = int(rng.integers(1, max_len + 1))
        src = loci[int(which[i])] if rng.random() < 0.8 else loci[int(rng.integers(0, n_models))]
        reads.append(workloads.make_reads(rng, src, 1, n, locus_fraction=0.7, sub_rate=0.02)[0])
    bases, off = _lib.encode_reads(reads)
    a = _lib.viterbi_batch(dms, bases, off, which)
    g = _lib.viterbi_batch(dms, bases, off, which, flags=_lib.FLAG_FORCE_GENERIC)
    same_lp = (a[0] == g[0]) | (np.isnan(a[0]) & np.isnan(g[0]))
    assert same_lp.all(), ("logp", b, np.flatnonzero(~same_lp)[:5])
    assert np.array_equal(a[1], g[1]), ("summary", b, np.flatnonzero((a[1] != g[1]).any(1))[:5])
    if b % 4 == 0:              # the other routes: one read per wavefront on anti-diagonals, reads packed back to back
        for name, fl in (("anti-diagonal", _lib.FLAG_ANTIDIAGONAL), ("stream", _lib.FLAG_STREAM)):
            o = _lib.viterbi_batch(dms, bases, off, which, flags=fl)
            assert np.array_equal(o[0], a[0]) and np.array_equal(o[1], a[1]), (name, b)
    # both strands in one call = the forward calls followed by the calls on the reverse complements
    s2 = _lib.viterbi_batch(dms, bases, off, which, flags=_lib.FLAG_BOTH_STRANDS)
    rc = [r.translate(COMP)[::-1] for r in reads]
    rb, ro = _lib.encode_reads(rc)
    r2 = _lib.viterbi_batch(dms, rb, ro, which)
    assert np.array_equal(s2[0][:n_reads], a[0]) and np.array_equal(s2[1][:n_reads], a[1]), ("both strands, forward half", b)
    assert np.array_equal(s2[0][n_reads:], r2[0]) and np.array_equal(s2[1][n_reads:], r2[1]), ("both strands, reverse half", b)
    if b % 10 == 0 and n_reads <= 400 and max_len <= 400:      # log_probability on the same mixed batch (the generic kernel is slow)
        fa = _lib.forward_batch(dms, bases, off, which)
        fg = _lib.forward_batch(dms, bases, off, which, flags=_lib.FLAG_FORCE_GENERIC)
        okf = (np.abs(fa - fg) <= 1e-9 * np.maximum(1.0, np.abs(fg))) | (np.isinf(fa) & np.isinf(fg))
        assert okf.all(), ("forward", b, np.flatnonzero(~okf)[:5])
        assert np.all(fa >= a[0] - 1e-9), ("forward below viterbi", b)
    total += n_reads
print("batch fuzz ok: %d batches, %d reads, %.1f s" % (n_batches, total, time.time() - t0))
