#!/usr/bin/env python3
"""Build container only: the C oracle AND the product's host-side model builder against the reference itself on
random loci and reads (beyond the committed goldens).  Nothing is written.

    python oracle/tools/build_reference.py && python oracle/tools/fuzz_oracle_vs_reference.py [n_loci] [seed]
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "nx111"), os.path.join(HERE, "stubs"), os.environ.get("ADVNTR_REF_BUILD", "/tmp/advntr_ref_build"), REPO]

import numpy as np                                   # noqa: E402
from advntr import settings as ref_settings, hmm_utils as ref_hmm_utils      # noqa: E402  (the reference)
from advntr_amd import settings as my_settings, hmm_utils as my_hmm_utils    # noqa: E402  (the product's builder)
from oracle.oracle import OracleModel                # noqa: E402

n_loci = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
seq = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
n_reads = n_exact = n_native_exact = 0
for k in range(n_loci):
    flank, plen, copies = int(rng.integers(3, 120)), int(rng.integers(2, 60)), int(rng.integers(1, 9))
    err = float(rng.choice([0.05, 0.3]))
    left, right, pat = seq(flank), seq(flank), seq(plen)
    ref_settings.MAX_ERROR_RATE = err
    my_settings.MAX_ERROR_RATE = err
    m = ref_hmm_utils.get_read_matcher_model(left, right, [pat], copies)
    from oracle import stepwise_builder
    mine = stepwise_builder.get_read_matcher_model(left, right, [pat], copies)
    native = my_hmm_utils.get_read_matcher_model(left, right, [pat], copies)          # the library's C++ builder
    idx = {s: i for i, s in enumerate(m.states)}
    edges = [(idx[a], idx[b], d["probability"]) for a, b, d in m.graph.edges_iter(data=True)]
    assert [s.name for s in m.states] == [s.name for s in mine.states]
    midx = {s: i for i, s in enumerate(mine.states)}
    mine_edges = [(midx[a], midx[b], lp) for a, b, lp in mine.graph.edges()]
    assert [(a, b) for a, b, _ in edges] == [(a, b) for a, b, _ in mine_edges], "edge order differs"
    n_exact += int(all(x[2] == y[2] for x, y in zip(edges, mine_edges)))
    emis = [[s.distribution.log_probability(c) for c in "ACGT"] for s in m.states[:m.silent_start]]
    O = OracleModel(len(m.states), m.silent_start, m.start_index, m.end_index, edges, emis)
    in_ptr, in_src, in_logp, _ = O.csr()
    na = native.baked_arrays()
    assert [s.name for s in m.states] == [s.name for s in native.states]
    assert (na["silent_start"], na["start_index"], na["end_index"]) == (m.silent_start, m.start_index, m.end_index)
    assert np.array_equal(na["in_ptr"], in_ptr) and np.array_equal(na["in_src"], in_src), "native builder: CSR order differs"
    assert np.array_equal(na["emis_logp"], np.array(emis))
    n_native_exact += int(np.array_equal(na["in_logp"], in_logp))
    for _ in range(40):
        n = int(rng.integers(1, 160))
        if rng.random() < 0.6:
            full = left + pat * int(rng.integers(1, copies + 1)) + right
            st = int(rng.integers(0, max(1, len(full) - 10)))
            r = (full[st:st + n] + seq(n))[:n]
            r = "".join(("ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.03 else ch) for ch in r)
        else:
            r = seq(n)
        if rng.random() < 0.3:
            p = int(rng.integers(0, max(1, n - 3)))
            r = (r[:p] + "AAAAAAAAAA" + r[p + 10:])[:n]
        logp, vpath = m.viterbi(r)
        olp, opath = O.viterbi(r)
        assert logp == olp, (k, r)
        assert (None if vpath is None else [i for i, _ in vpath]) == opath, (k, r)
        assert abs(m.log_probability(r) - O.forward(r)) == 0.0, (k, r)
        n_reads += 1
print("oracle == reference on %d loci / %d reads (logp, path, forward all bit-equal); product builder: structure identical on all, "
      "log-probs bit-identical on %d/%d loci (stepwise) and %d/%d loci (native C++ builder)"
      % (n_loci, n_reads, n_exact, n_loci, n_native_exact, n_loci))
