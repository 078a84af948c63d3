"""Differential fuzz: the row-blocked and anti-diagonal column kernels vs the generic-CSR kernel (independent implementations of
the reference's _viterbi) on random loci and reads; log-probs, all 8 summary ints and (on a sample) paths must agree
bit for bit.  Usage: python scripts/fuzz_kernels.py [n_loci] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
e.build()
from advntr_amd import _lib, workloads

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 100
MAX_LEN = int(os.environ.get("FUZZ_MAX_LEN", "400"))            # longest read (longer: the row-tiled kernels)
MAX_READS = int(os.environ.get("FUZZ_MAX_READS", "400"))
FLANK_MAX = int(os.environ.get("FUZZ_FLANK_MAX", "160"))         # wide models: column tables beyond the LDS (staging levels 1 and 0)
FLANK_MIN = int(os.environ.get("FUZZ_FLANK_MIN", "3"))
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t0 = time.time()
total = 0
for k in range(n_loci):
    flank = int(rng.integers(FLANK_MIN, FLANK_MAX))
    plen = int(rng.integers(2, 80))
    copies = int(rng.integers(1, 14))
    err = float(rng.choice([0.05, 0.3, 0.1]))
    loc = workloads.make_locus(rng, flank, plen, copies, err, n_units=int(rng.integers(1, 8)))
    dm = loc.model.device_model()
    if not dm.has_column_program():
        print("locus %d has no column program" % k)
        continue
    reads = []
    for _ in range(int(rng.integers(min(50, MAX_READS - 1), MAX_READS))):
        n = int(rng.integers(1, MAX_LEN))
        r = workloads.make_reads(rng, loc, 1, n, locus_fraction=0.6, sub_rate=float(rng.choice([0.0, 0.01, 0.1])))[0]
        if rng.random() < 0.3:      # homopolymer / low-complexity stretches provoke exact ties
            p = int(rng.integers(0, max(1, n - 5)))
            r = r[:p] + "A" * min(12, n - p) + r[p + 12:]
            r = r[:n]
        reads.append(r)
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    want_paths = (k % 5 == 0)
    a = _lib.viterbi_batch([dm], bases, off, which, want_paths=want_paths)
    arr = loc.model.baked_arrays()
    wide = arr["m"] > 10240                     # beyond the generic kernel (its trellis rows sit in LDS): the CPU oracle on a few reads
    if wide:
        from oracle.oracle import OracleModel
        edges = [(int(arr["in_src"][e2]), l, float(arr["in_logp"][e2])) for l in range(arr["m"]) for e2 in range(arr["in_ptr"][l], arr["in_ptr"][l + 1])]
        O = OracleModel(arr["m"], arr["silent_start"], arr["start_index"], arr["end_index"], edges, arr["emis_logp"])
        ap = _lib.viterbi_batch([dm], bases, off, which, want_paths=True)
        for i in range(min(3, len(reads))):
            olp, opath = O.viterbi(reads[i])
            assert ap[0][i] == olp and ap[2][i] == opath, ("wide model vs oracle", k, i, arr["m"])
        b = a
    else:
        b = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_FORCE_GENERIC, want_paths=want_paths)
    c = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_STREAM)
    d = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_ANTIDIAGONAL, want_paths=want_paths)
    # the row-blocked kernels once more with tiles of full back-to-back depth (a batch of this size gets single sweeps by default)
    dd = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_DEEP_TILES, want_paths=want_paths)
    assert np.array_equal(a[0], dd[0]), ("rows vs back-to-back rows logp", k, flank, plen, copies, err,
                                         np.flatnonzero(~((a[0] == dd[0]) | (np.isnan(a[0]) & np.isnan(dd[0]))))[:5])
    assert np.array_equal(a[1], dd[1]), ("rows vs back-to-back rows summary", k, np.flatnonzero((a[1] != dd[1]).any(1))[:5])
    if want_paths:
        assert a[2] == dd[2], ("rows vs back-to-back rows paths", k)
    assert np.array_equal(a[0], d[0]), ("rows vs anti-diagonal logp", k, flank, plen, copies, err,
                                        np.flatnonzero(~((a[0] == d[0]) | (np.isnan(a[0]) & np.isnan(d[0]))))[:5])
    assert np.array_equal(a[1], d[1]), ("rows vs anti-diagonal summary", k, np.flatnonzero((a[1] != d[1]).any(1))[:5])
    if want_paths:
        assert a[2] == d[2], ("rows vs anti-diagonal paths", k)
    assert np.array_equal(a[0], b[0]), ("logp", k, flank, plen, copies, err)
    assert np.array_equal(a[1], b[1]), ("summary", k, flank, plen, copies, err, np.flatnonzero((a[1] != b[1]).any(1))[:5])
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]), ("stream", k)
    if want_paths:
        assert a[2] == b[2], ("paths", k)
    total += len(reads)
print("fuzz ok: %d loci, %d reads, %.1f s" % (n_loci, total, time.time() - t0))
