"""Replay one locus of scripts/fuzz_kernels.py (same seed, same draw order) and print where the kernels and the oracle differ.
Usage: python scripts/fuzz_repro.py n_loci seed k"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
from advntr_amd import _lib, workloads
from oracle.oracle import OracleModel

n_loci, seed, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for k in range(n_loci):
    flank = int(rng.integers(3, 160)); plen = int(rng.integers(2, 80)); copies = int(rng.integers(1, 14))
    err = float(rng.choice([0.05, 0.3, 0.1]))
    loc = workloads.make_locus(rng, flank, plen, copies, err, n_units=int(rng.integers(1, 8)))
    reads = []
    for _ in range(int(rng.integers(50, 400))):
        n = int(rng.integers(1, 400))
        r = workloads.make_reads(rng, loc, 1, n, locus_fraction=0.6, sub_rate=float(rng.choice([0.0, 0.01, 0.1])))[0]
        if rng.random() < 0.3:
            p = int(rng.integers(0, max(1, n - 5)))
            r = r[:p] + "A" * min(12, n - p) + r[p + 12:]
            r = r[:n]
        reads.append(r)
    if k != want:
        continue
    dm = loc.model.device_model()
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    a = _lib.viterbi_batch([dm], bases, off, which)
    b = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_FORCE_GENERIC)
    arr = loc.model.baked_arrays()
    edges = [(int(arr["in_src"][e]), l, float(arr["in_logp"][e])) for l in range(arr["m"]) for e in range(arr["in_ptr"][l], arr["in_ptr"][l + 1])]
    O = OracleModel(arr["m"], arr["silent_start"], arr["start_index"], arr["end_index"], edges, arr["emis_logp"])
    names = [s.name for s in loc.model.states]
    print("locus", k, "flank", flank, "plen", plen, "copies", copies, "err", err, "reads", len(reads), "m", arr["m"])
    bad = np.flatnonzero((a[1] != b[1]).any(1))
    print("summary mismatches rows vs generic:", bad[:20])
    for i in bad[:4]:
        olp, opath = O.viterbi(reads[i])
        print("read", i, "len", len(reads[i]), "logp rows/generic/oracle", a[0][i], b[0][i], olp)
        print("  oracle path length", None if opath is None else len(opath), "n + m =", len(reads[i]) + arr["m"])
        print("  summaries rows", a[1][i], "generic", b[1][i])
    break
