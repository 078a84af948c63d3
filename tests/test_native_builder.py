"""The library's C++ model builder (csrc/model_builder.h, advntr_build_read_matchers) against the call-by-call
assembly through the pomegranate mirror, which tests/test_builder_golden.py pins on the reference's own baked models.
Host-only: no GPU is needed to build a model."""
import numpy as np
import pytest

from advntr_amd import _lib, hmm_utils
from oracle import stepwise_builder


def _random_locus(rng, rows_max=6):
    L = int(rng.integers(1, 40))
    flank_l, flank_r = int(rng.integers(1, 60)), int(rng.integers(1, 60))
    dna = lambda n: "".join(rng.choice(list("ACGT"), n))
    base = dna(L)
    rows = []
    for _ in range(int(rng.integers(1, rows_max + 1))):
        r = list(base)
        for _ in range(int(rng.integers(0, 3))):
            r[int(rng.integers(0, L))] = str(rng.choice(list("ACGT-")))
        rows.append("".join(r))
    if all(set(r) == {"-"} for r in rows):
        rows[0] = base
    copies = int(rng.integers(1, 6))
    return dna(flank_l), dna(flank_r), rows, copies


def _same(a, b, exact):
    A, B = a.baked_arrays(), b.baked_arrays()
    assert [s.name for s in a.states] == [s.name for s in b.states]
    for k in ("m", "silent_start", "start_index", "end_index"):
        assert A[k] == B[k], k
    for k in ("in_ptr", "in_src", "state_class"):
        assert np.array_equal(A[k], B[k]), k
    assert np.array_equal(A["emis_logp"], B["emis_logp"])
    if exact:
        assert np.array_equal(A["in_logp"], B["in_logp"])
    else:
        fin = np.isfinite(B["in_logp"])
        assert np.array_equal(np.isfinite(A["in_logp"]), fin)
        assert np.all(np.abs(A["in_logp"][fin] - B["in_logp"][fin]) <= 4 * np.spacing(np.abs(B["in_logp"][fin])))
    assert a.finite == b.finite


def _usable(rows):
    # the reference (and both builders) need at least one match column
    gaps = [sum(r[c] == "-" for r in rows) for c in range(len(rows[0]))]
    return any(g < 0.5 * len(rows) for g in gaps)


def test_native_equals_stepwise_on_random_loci():
    rng = np.random.default_rng(99)
    loci = []
    while len(loci) < 40:
        l = _random_locus(rng)
        if _usable(l[2]):
            loci.append(l)
    native = hmm_utils.build_read_matcher_models(loci, threads=4)               # numpy.exp through the callback
    libm = hmm_utils.build_read_matcher_models(loci, threads=4, exp="libm")
    for locus, n, lm in zip(loci, native, libm):
        ref = stepwise_builder.get_read_matcher_model(*locus)
        _same(n, ref, exact=True)
        _same(lm, ref, exact=False)


def test_illumina_shapes_and_thread_counts_agree():
    rng = np.random.default_rng(5)
    dna = lambda n: "".join(rng.choice(list("ACGT"), n))
    loci = [(dna(150), dna(150), [dna(L)], int(round(150.0 / L + 0.5))) for L in (6, 14, 57, 100)]
    one = hmm_utils.build_read_matcher_models(loci, threads=1)
    many = hmm_utils.build_read_matcher_models(loci, threads=0)
    for locus, a, b in zip(loci, one, many):
        _same(a, b, exact=True)
        _same(a, stepwise_builder.get_read_matcher_model(*locus), exact=True)
    assert (one[1].n_states, one[1].silent_start, one[1].n_edges) == (1413, 921, 4626)      # REF150


def test_builder_errors():
    with pytest.raises(NotImplementedError):
        hmm_utils.get_read_matcher_model("ACGTNACGT", "TTGACCAA", ["ACGTT"], 2)
    with pytest.raises(NotImplementedError):
        hmm_utils.get_read_matcher_model("ACGTACGT", "TTGACCAA", ["ACGTT", "ACG"], 2)
    with pytest.raises(_lib.EngineError):
        hmm_utils.get_read_matcher_model("ACGTACGT", "TTGACCAA", ["ACGTT"], 0)
    with pytest.raises(_lib.EngineError):
        hmm_utils.get_read_matcher_model("", "TTGACCAA", ["ACGTT"], 1)
    # one bad locus in a batch: the call fails as a whole and names it
    with pytest.raises(NotImplementedError, match="locus 1"):
        hmm_utils.build_read_matcher_models([("ACGT", "ACGT", ["ACG"], 1), ("ACXT", "ACGT", ["ACG"], 1)])


def test_built_model_host_surface():
    m = hmm_utils.get_read_matcher_model("ACGTACGTAC", "TTGACCAATG", ["ACGTT"], 2)
    assert m.states[m.start_index].name == "Read Matcher-start" and m.states[m.end_index].name == "Read Matcher-end"
    assert m.start is m.states[m.start_index]
    assert all(s.is_silent() == (i >= m.silent_start) for i, s in enumerate(m.states))
    ref = stepwise_builder.get_read_matcher_model("ACGTACGTAC", "TTGACCAATG", ["ACGTT"], 2)
    assert np.array_equal(m.dense_transition_matrix(), ref.dense_transition_matrix())
    i = m.silent_start - 1
    assert m.states[i].distribution.log_probability("A") == ref.states[i].distribution.log_probability("A")


# ---- the built-in repeat aligner (stands in for muscle; parity with muscle is not claimed) -----------------------
def _check_alignment(units, rows):
    assert len(rows) == len(units)
    assert len(set(len(r) for r in rows)) == 1
    for u, r in zip(units, rows):
        assert r.replace("-", "") == u
    width = len(rows[0])
    assert all(any(r[c] != "-" for r in rows) for c in range(width))        # no all-gap column


def test_aligner_properties():
    rng = np.random.default_rng(3)
    dna = lambda n: "".join(rng.choice(list("ACGT"), n))
    # equal lengths, substitutions only: gap-free, i.e. the input itself
    base = dna(30)
    units = [base] * 3
    for _ in range(4):
        u = list(base)
        u[int(rng.integers(0, 30))] = "ACGT"[int(rng.integers(0, 4))]
        units.append("".join(u))
    assert _lib.align_repeats(units) == units
    # planted indels: every row still spells its unit, identical units get identical rows
    for trial in range(30):
        base = dna(int(rng.integers(4, 50)))
        units = []
        for _ in range(int(rng.integers(2, 12))):
            u = list(base)
            for _ in range(int(rng.integers(0, 3))):
                p = int(rng.integers(0, len(u)))
                kind = rng.random()
                if kind < 0.4 and len(u) > 2:
                    u.pop(p)
                elif kind < 0.8:
                    u.insert(p, "ACGT"[int(rng.integers(0, 4))])
                else:
                    u[p] = "ACGT"[int(rng.integers(0, 4))]
            units.append("".join(u))
        rows = _lib.align_repeats(units)
        _check_alignment(units, rows)
        assert rows == _lib.align_repeats(units)                               # deterministic
        seen = {}
        for u, r in zip(units, rows):
            assert seen.setdefault(u, r) == r
    # one deleted base against a majority: one gap in that row, no other row changes
    rows = _lib.align_repeats(["ACGTTGCAACC", "ACGTGCAACC", "ACGTTGCAACC"])
    assert rows[0] == rows[2] == "ACGTTGCAACC" and rows[1].count("-") == 1 and len(rows[1]) == 11


def test_builder_aligns_ragged_units_only_on_request():
    from advntr_amd import settings
    left, right = "ACGTACGTACGGTCA", "TTGACCAATGCATGC"
    units = ["ACGTTGCAACC", "ACGTGCAACC", "ACGTTGCAACC", "ACGTTGGCAACC"]
    with pytest.raises(NotImplementedError):
        hmm_utils.get_read_matcher_model(left, right, units, 3)
    rows = _lib.align_repeats(units)
    want = hmm_utils.get_read_matcher_model(left, right, rows, 3)
    got = hmm_utils.build_read_matcher_models([(left, right, units, 3)], align=True)[0]
    _same(got, want, exact=True)
    settings.ALIGN_REPEATS = True
    try:
        _same(hmm_utils.get_read_matcher_model(left, right, units, 3), want, exact=True)
        _same(stepwise_builder.get_read_matcher_model(left, right, units, 3), want, exact=True)
    finally:
        settings.ALIGN_REPEATS = False


def test_kept_flank_blocks_do_not_leak_between_loci():
    """The builder keeps the flank blocks it has built per thread and per flank length (model_builder.h, FlankBlocks) and fills
    in a locus's bases: a locus built after fifty others -- more flank lengths than the threads keep, flanks longer than what
    they keep at all, the same lengths with other bases -- equals the same locus built alone, and the stepwise assembly."""
    rng = np.random.default_rng(404)
    dna = lambda n: "".join(rng.choice(list("ACGT"), n))
    lengths = [int(x) for x in rng.integers(1, 70, 50)] + [150, 150, 1100, 30]
    loci = [(dna(n), dna(lengths[-1 - i]), [dna(12)], 3) for i, n in enumerate(lengths)]
    together = hmm_utils.build_read_matcher_models(loci, threads=1)
    again = hmm_utils.build_read_matcher_models(loci[::-1], threads=2)[::-1]
    for k in (0, 17, 49, 50, 51, 52, 53):
        alone = hmm_utils.build_read_matcher_models([loci[k]], threads=1)[0]
        _same(together[k], alone, exact=True)
        _same(again[k], alone, exact=True)
    for k in (3, 51, 53):
        _same(together[k], stepwise_builder.get_read_matcher_model(*loci[k]), exact=True)
