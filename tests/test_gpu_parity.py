"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(1) the golden vectors captured from the reference and (2) the CPU oracle on seeded synthetic batches.

Bars: log-probabilities bit-exact (==; the north star allows 1e-4, asserted as well), Viterbi paths
identical, RU counts and the other integer summaries exact, recruit verdicts exact.
"""
import math

import numpy as np
import pytest

from conftest import ALL_MODEL_GOLDENS, READ_MATCHER_GOLDENS, load_golden

pytestmark = pytest.mark.gpu


def device_model_from_golden(g):
    from advntr_amd import _lib
    from advntr_amd.pomegranate import state_class_from_name
    from oracle.oracle import OracleModel
    gm = g["model"]
    O = OracleModel.from_golden(g)
    in_ptr, in_src, in_logp, _ = O.csr()
    bases = {}
    if g.get("kind") == "read_matcher":
        F = len([n for n in gm["state_names"] if n.startswith("M") and n.endswith("_suffix")])
        for i, ch in enumerate(g["left"][-F:]):
            bases["M%d_suffix" % (i + 1)] = "ACGT".index(ch)
        for i, ch in enumerate(g["right"][:F]):
            bases["M%d_prefix" % (i + 1)] = "ACGT".index(ch)
    cls = np.array([state_class_from_name(n, bases.get(n)) for n in gm["state_names"]], np.uint16)
    dm = _lib.DeviceModel(len(gm["state_names"]), gm["silent_start"], gm["start_index"], gm["end_index"],
                          in_ptr, in_src, in_logp, O.emis, cls)
    return dm, O


@pytest.mark.parametrize("force_generic", [True, False, "antidiagonal"])
@pytest.mark.parametrize("name", ALL_MODEL_GOLDENS)
def test_golden_logp_paths_summaries(name, force_generic):
    from advntr_amd import _lib
    from advntr_amd.hmm_utils import flanking_rate_from_counts
    g = load_golden(name)
    dm, _ = device_model_from_golden(g)
    if force_generic is not True and not dm.has_column_program():
        pytest.skip("model has no column program; generic kernel covered by the other parametrisation")
    reads = [r["seq"] for r in g["reads"]]
    bases, off = _lib.encode_reads(reads)
    # False = the default routing (row-blocked kernels), "antidiagonal" = one read per wavefront
    flags = {True: _lib.FLAG_FORCE_GENERIC, False: 0, "antidiagonal": _lib.FLAG_ANTIDIAGONAL}[force_generic]
    logp, summ, paths = _lib.viterbi_batch([dm], bases, off, np.zeros(len(reads), np.int32), flags=flags,
                                           want_paths=True)
    for i, r in enumerate(g["reads"]):
        want = r["logp"]
        assert logp[i] == want or (math.isinf(want) and math.isinf(logp[i])), (name, i, logp[i], want)
        assert math.isinf(want) or abs(logp[i] - want) <= 1e-4
        assert paths[i] == r["path"], (name, i)
        if r["path"] is None or g.get("kind") != "read_matcher":
            continue
        s = summ[i]
        assert s[_lib.SUM_PATH_LEN] == len(r["path"])
        assert s[_lib.SUM_RU] == r["ru"], (name, i)
        assert s[_lib.SUM_MATCHES] == r["matches"]
        assert s[_lib.SUM_REPEAT_BP] == r["repeat_bp"]
        assert s[_lib.SUM_LEFT_BP] == r["left_bp"]
        assert s[_lib.SUM_RIGHT_BP] == r["right_bp"]
        if "flank_rate" in r:
            rate = flanking_rate_from_counts(s[_lib.SUM_LEFT_MATCH], s[_lib.SUM_LEFT_BP],
                                             s[_lib.SUM_RIGHT_MATCH], s[_lib.SUM_RIGHT_BP])
            assert rate == r["flank_rate"], (name, i)
            rate_acc = flanking_rate_from_counts(s[_lib.SUM_LEFT_MATCH], s[_lib.SUM_LEFT_BP],
                                                 s[_lib.SUM_RIGHT_MATCH], s[_lib.SUM_RIGHT_BP], True)
            assert rate_acc == r["flank_rate_acc"]


@pytest.mark.parametrize("name", ALL_MODEL_GOLDENS)
def test_golden_forward(name):
    """log_probability: device libm + chunked fold order => rounding-level agreement (1e-9 relative;
    the north star asks for 1e-4 absolute)."""
    from advntr_amd import _lib
    g = load_golden(name)
    rs = [r for r in g["reads"] if "forward_logp" in r]
    if not rs:
        pytest.skip("no forward values in this golden")
    dm, _ = device_model_from_golden(g)
    bases, off = _lib.encode_reads([r["seq"] for r in rs])
    got = _lib.forward_batch([dm], bases, off, np.zeros(len(rs), np.int32))
    for v, r in zip(got, rs):
        want = r["forward_logp"]
        if math.isinf(want):
            assert math.isinf(v)
        else:
            assert abs(v - want) <= 1e-9 * max(1.0, abs(want)), (name, v, want)
            assert abs(v - want) <= 1e-4


def test_mirror_model_viterbi_api():
    """The pomegranate-shaped call: model.viterbi(seq) -> (logp, [(idx, State)...]); errors as the reference."""
    from advntr_amd import hmm_utils, settings
    g = load_golden("s300_f30_l12_c3")
    settings.MAX_ERROR_RATE = g["error_rate"]
    m = hmm_utils.get_read_matcher_model(g["left"], g["right"], g["aligned_repeats"], g["copies"])
    for r in g["reads"][:12]:
        logp, vpath = m.viterbi(r["seq"])
        assert abs(logp - r["logp"]) <= 1e-4
        assert [i for i, _ in vpath] == r["path"]
        assert vpath[0][1].name == "Read Matcher-start" and vpath[-1][1].name.endswith("-end")
        assert hmm_utils.get_number_of_repeats_in_vpath(vpath) == r["ru"]
    with pytest.raises(ValueError):
        m.viterbi("ACGTNACGT")                      # hmm.pyx:72,79
    lp = m.log_probability(g["reads"][0]["seq"])
    assert abs(lp - g["reads"][0]["forward_logp"]) <= 1e-4
    from advntr_amd import HiddenMarkovModel
    with pytest.raises(ValueError):
        HiddenMarkovModel("unbaked").viterbi("ACGT")  # hmm.pyx:1944-1945


@pytest.mark.parametrize("shape", ["s300", "ref150"])
@pytest.mark.parametrize("force_generic", [True, False, "stream", "rows", "deep"])
def test_synthetic_batch_vs_oracle(shape, force_generic):
    """Seeded C1-style batch at a size the oracle finishes in seconds.  "rows": the default routing
    (row-blocked kernels); "deep": the same kernels with tiles of full back-to-back depth (what a large batch gets);
    False: the anti-diagonal kernel, one read per wavefront."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    locus = getattr(workloads, shape)()
    n_reads = 1500 if shape == "s300" else 400
    reads = workloads.make_reads(np.random.default_rng(99), locus, n_reads, 150)
    # ragged + edge cases: empty read, 1 base, long read
    reads += ["", "A", workloads.rand_seq(np.random.default_rng(3), 301)]
    m = locus.model
    dm = m.device_model()
    if force_generic is not True and not dm.has_column_program():
        pytest.skip("no column program")
    flags = {True: _lib.FLAG_FORCE_GENERIC, False: _lib.FLAG_ANTIDIAGONAL, "stream": _lib.FLAG_STREAM, "rows": 0,
             "deep": _lib.FLAG_DEEP_TILES}[force_generic]
    bases, off = _lib.encode_reads(reads)
    logp, summ, paths = _lib.viterbi_batch([dm], bases, off, np.zeros(len(reads), np.int32), flags=flags,
                                           want_paths=True)
    a = m.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
             for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    want, _ = O.viterbi_many(bases, off)
    assert np.array_equal(logp, want)
    names = [s.name for s in m.states]
    from oracle import oracle as Or
    for i in range(0, len(reads), 7):
        olp, opath = O.viterbi(reads[i])
        assert paths[i] == opath
        inner = [names[j] for j in opath][1:-1]
        assert summ[i][_lib.SUM_RU] == Or.number_of_repeats(inner)
        assert summ[i][_lib.SUM_MATCHES] == Or.number_of_matches(inner)
        assert summ[i][_lib.SUM_REPEAT_BP] == Or.repeat_bp_matches(inner)
        if len(reads[i]) > 0:
            lm, lb, rm, rb = Or.flanking_counts(inner, reads[i], locus.left, locus.right)
            assert tuple(summ[i][[_lib.SUM_LEFT_MATCH, _lib.SUM_LEFT_BP, _lib.SUM_RIGHT_MATCH, _lib.SUM_RIGHT_BP]]) == (lm, lb, rm, rb)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_back_to_back_sweeps_ragged_multi_model_vs_oracle(seed):
    """Back-to-back sweeps (ADVNTR_FLAG_DEEP_TILES: up to four reads per lane group, one behind the other along the step
    axis): reads of 1..155 bases -- all three row-blocked instantiations, tiles that are not full, groups without a read,
    reads of different lengths behind each other -- of several loci incl. one too narrow for the scheme (< 64 columns),
    against the oracle (logp ==, paths identical, summaries equal to the default routing's)."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(seed)
    loci = [workloads.make_locus(rng, int(rng.integers(20, 150)), int(rng.integers(4, 40)), int(rng.integers(1, 6)),
                                 float(rng.choice([0.05, 0.3])), n_units=int(rng.integers(1, 4))) for _ in range(3)]
    loci.append(workloads.make_locus(rng, 6, 5, 2, 0.05))                 # 6 + 2 * 5 + 6 + connectors: a few dozen columns
    reads, which = [], []
    for k, loc in enumerate(loci):
        for _ in range(int(rng.integers(70, 140))):
            n = int(rng.choice([int(rng.integers(1, 156)), 150, 64, 65, 124, 125, 155, 1]))
            r = workloads.make_reads(rng, loc, 1, n, locus_fraction=0.7, sub_rate=0.02)[0]
            if rng.random() < 0.3 and n > 20:
                p = int(rng.integers(0, n - 12))
                r = (r[:p] + "T" * 12 + r[p + 12:])[:n]
            reads.append(r)
            which.append(k)
    perm = rng.permutation(len(reads))
    reads = [reads[i] for i in perm]
    which = np.array([which[i] for i in perm], np.int32)
    dms = [loc.model.device_model() for loc in loci]
    assert all(dm.has_column_program() for dm in dms)
    bases, off = _lib.encode_reads(reads)
    deep = _lib.viterbi_batch(dms, bases, off, which, flags=_lib.FLAG_DEEP_TILES, want_paths=True)
    plain = _lib.viterbi_batch(dms, bases, off, which, want_paths=True)
    assert np.array_equal(deep[0], plain[0]) and np.array_equal(deep[1], plain[1]) and deep[2] == plain[2]
    for k, loc in enumerate(loci):
        a = loc.model.baked_arrays()
        edges = [(int(a["in_src"][q]), l, float(a["in_logp"][q]))
                 for l in range(a["m"]) for q in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
        O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
        for i in np.flatnonzero(which == k):
            olp, opath = O.viterbi(reads[i])
            assert deep[0][i] == olp, (seed, k, i, len(reads[i]))
            assert deep[2][i] == opath, (seed, k, i, len(reads[i]))


def test_multi_model_batch_and_device_resident_api():
    """Several loci in one batch (reads interleaved), through the device-resident batch API."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(2024)
    loci = [workloads.make_locus(rng, 40, int(rng.integers(6, 30)), 4) for _ in range(5)]
    reads, which = [], []
    for k, loc in enumerate(loci):
        rs = workloads.make_reads(rng, loc, 30, int(rng.integers(50, 120)))
        reads += rs
        which += [k] * len(rs)
    perm = rng.permutation(len(reads))
    reads = [reads[i] for i in perm]
    which = np.array([which[i] for i in perm], np.int32)
    dms = [loc.model.device_model() for loc in loci]
    bases, off = _lib.encode_reads(reads)
    B = _lib.DeviceBatch(dms, bases, off, which)
    B.run()
    logp, summ = B.fetch()
    ms = B.run_timed(2)
    assert ms > 0
    logp2, _ = B.fetch()
    assert np.array_equal(logp, logp2)          # idempotent re-run on resident inputs
    for k, loc in enumerate(loci):
        a = loc.model.baked_arrays()
        edges = [(int(a["in_src"][q]), l, float(a["in_logp"][q]))
                 for l in range(a["m"]) for q in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
        O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
        for i in np.flatnonzero(which == k):
            assert logp[i] == O.viterbi(reads[i])[0]
    B.close()


def test_bad_symbol_is_rejected_before_launch():
    from advntr_amd import _lib, workloads
    loc = workloads.make_locus(np.random.default_rng(1), 8, 5, 2)
    with pytest.raises(ValueError):
        loc.model.viterbi_batch(["ACGT", "ACNT"])


def test_score_reads_strand_choice_and_recruit():
    """The caller mirror (process_unmapped_read: forward + reverse complement, keep the better strand,
    then recruit_read) on the GPU vs the same rule applied to oracle scores."""
    from advntr_amd import hmm_utils, settings, vntr_finder
    from oracle import oracle as Or
    g = load_golden("s300_f30_l12_c3")
    settings.MAX_ERROR_RATE = g["error_rate"]
    m = hmm_utils.get_read_matcher_model(g["left"], g["right"], g["aligned_repeats"], g["copies"])
    O = Or.OracleModel.from_golden(g)
    names = g["model"]["state_names"]
    reads = [r["seq"] for r in g["reads"] if len(r["seq"]) > 3][:30] + ["ACGTNNACGT"]
    out = vntr_finder.score_reads(m, reads, scaled_score=g["scaled_score"], compute_reverse=True)
    assert out[-1] is None                                  # reads with N are skipped (vntr_finder.py:237)
    n_rev = 0
    for s, sr in zip(reads[:-1], out[:-1]):
        lp_f, path_f = O.viterbi(s)
        rc = vntr_finder.reverse_complement(s)
        lp_r, path_r = O.viterbi(rc)
        use_rev = lp_f < lp_r
        n_rev += use_rev
        seq, lp, path = (rc, lp_r, path_r) if use_rev else (s, lp_f, path_f)
        assert sr.reversed == use_rev and sr.logp == lp and sr.sequence == seq
        inner = [names[i] for i in path][1:-1]
        assert sr.repeats == Or.number_of_repeats(inner)
        ms = g["scaled_score"] * len(seq)
        assert sr.recruited == Or.recruit_read(lp, inner, ms, seq, g["left"], g["right"])
    assert n_rev > 0


@pytest.mark.parametrize("kernel", ["antidiagonal", "rows"])
def test_long_reads_row_tiled_vs_oracle(kernel, monkeypatch):
    """Reads longer than one 256-row tile go through the row-tiled column kernel (seam rows in HBM): PacBio-like
    locus (error 0.3, flank 100), reads of 257..900 bases incl. exact tile multiples."""
    from advntr_amd import _lib as _l
    kflags = 0 if kernel == "rows" else _l.FLAG_ANTIDIAGONAL          # rows = the default routing
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    from oracle import oracle as Or
    rng = np.random.default_rng(31)
    loc = workloads.make_locus(rng, 100, 30, 12, error_rate=0.3)
    lens = [257, 300, 511, 512, 513, 640, 768, 900]
    reads = [workloads.make_reads(rng, loc, 1, n, locus_fraction=1.0, sub_rate=0.08)[0] for n in lens]
    reads += [workloads.rand_seq(rng, 700)]
    m = loc.model
    dm = m.device_model()
    assert dm.has_column_program()
    bases, off = _lib.encode_reads(reads)
    logp, summ, paths = _lib.viterbi_batch([dm], bases, off, np.zeros(len(reads), np.int32), want_paths=True, flags=kflags)
    a = m.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
             for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    names = [s.name for s in m.states]
    for i, r in enumerate(reads):
        olp, opath = O.viterbi(r)
        assert logp[i] == olp, (i, len(r))
        assert paths[i] == opath, (i, len(r))
        assert summ[i][_lib.SUM_RU] == Or.number_of_repeats([names[j] for j in opath][1:-1])


@pytest.mark.parametrize("kernel", ["antidiagonal", "rows"])
def test_pacbio_c4_style_loci_vs_oracle(kernel, monkeypatch):
    """Config C4 of BASELINE.json at test scale: flank 100, error 0.3, copies = round((max_len-100)/len(pattern))
    (vntr_finder.py:538-549), trimmed spanning reads of VNTR +-20 % + 200 bases with 12 % indel/substitution
    noise; RU counts exact, log-probs bit-equal."""
    from advntr_amd import _lib as _l
    kflags = 0 if kernel == "rows" else _l.FLAG_ANTIDIAGONAL          # rows = the default routing
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    from oracle import oracle as Or
    rng = np.random.default_rng(20240603)
    loci, reads, which = [], [], []
    for k in range(3):
        plen = int(rng.integers(10, 40))
        vntr_len = int(rng.integers(100, 400))
        true_copies = max(1, vntr_len // plen)
        lens = [int(vntr_len * rng.uniform(0.8, 1.2)) + 200 for _ in range(6)]
        max_copies = int(round((max(lens) - 100) / float(plen)))
        loc = workloads.make_locus(rng, 100, plen, max_copies, error_rate=0.3)
        loci.append(loc)
        for n in lens:
            s = loc.left + loc.units[0] * true_copies + loc.right
            out = []
            for ch in s:
                u = rng.random()
                if u < 0.04:
                    continue
                if u < 0.08:
                    out.append("ACGT"[int(rng.integers(0, 4))])
                out.append("ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.04 else ch)
            reads.append("".join(out)[:n])
            which.append(k)
    dms = [l.model.device_model() for l in loci]
    bases, off = _lib.encode_reads(reads)
    logp, summ, paths = _lib.viterbi_batch(dms, bases, off, np.asarray(which, np.int32), want_paths=True, flags=kflags)
    for k, loc in enumerate(loci):
        a = loc.model.baked_arrays()
        edges = [(int(a["in_src"][q]), l, float(a["in_logp"][q]))
                 for l in range(a["m"]) for q in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
        O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
        names = [s.name for s in loc.model.states]
        for i in [i for i, w in enumerate(which) if w == k]:
            olp, opath = O.viterbi(reads[i])
            assert logp[i] == olp
            assert paths[i] == opath
            assert summ[i][_lib.SUM_RU] == Or.number_of_repeats([names[j] for j in opath][1:-1])


def test_empty_batch_and_argument_errors():
    from advntr_amd import _lib, workloads
    loc = workloads.make_locus(np.random.default_rng(1), 8, 5, 2)
    dm = loc.model.device_model()
    logp, summ, paths = _lib.viterbi_batch([dm], np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(0, np.int32),
                                           want_paths=True)
    assert len(logp) == 0 and paths == []
    with pytest.raises(_lib.EngineError):                       # read_model out of range
        _lib.viterbi_batch([dm], np.zeros(4, np.uint8), np.array([0, 4], np.int64), np.array([3], np.int32))
    with pytest.raises(_lib.EngineError):                       # CSR that does not span its edges
        _lib.DeviceModel(3, 1, 1, 2, np.array([0, 1, 1, 5], np.int32), np.array([1], np.int32), np.array([0.0]),
                         np.zeros((1, 4)))


def test_full_size_c1_properties():
    """BASELINE config C1 at full size (REF150 x 100 000 reads): size-independent properties instead of an oracle
    replay -- run-to-run determinism, the two kernels agree on a subsample, permutation invariance (results do not
    depend on batch order / tiling), every log-prob finite and <= 0, path length consistent with the summaries, and a
    checksum of the RU counts that must not depend on how the batch is split."""
    from advntr_amd import _lib, workloads
    loc = workloads.ref150()
    reads = workloads.make_reads(np.random.default_rng(20240601), loc, 100000, 150)
    bases, off = _lib.encode_reads(reads)
    dm = loc.model.device_model()
    which = np.zeros(len(reads), np.int32)
    B = _lib.DeviceBatch([dm], bases, off, which)
    B.run()
    logp, summ = B.fetch()
    B.run()
    logp2, summ2 = B.fetch()
    B.close()
    assert np.array_equal(logp, logp2) and np.array_equal(summ, summ2)              # idempotent / deterministic
    assert np.all(np.isfinite(logp)) and np.all(logp <= 0)
    assert np.all(summ[:, _lib.SUM_PATH_LEN] >= 150 + 2)                             # >= n emitting states + start/end
    assert np.all(summ[:, _lib.SUM_MATCHES] <= 150) and np.all(summ[:, _lib.SUM_REPEAT_BP] <= 150)
    assert np.all(summ[:, _lib.SUM_LEFT_BP] + summ[:, _lib.SUM_RIGHT_BP] + summ[:, _lib.SUM_REPEAT_BP] == 150)
    # permutation + split invariance: shuffle, score in two halves, un-shuffle
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(reads))
    halves = [perm[:37123], perm[37123:]]
    got = np.zeros_like(logp)
    got_ru = np.zeros(len(reads), np.int64)
    for h in halves:
        hb, ho = _lib.encode_reads([reads[i] for i in h])
        lp, sm, _ = _lib.viterbi_batch([dm], hb, ho, np.zeros(len(h), np.int32))
        got[h] = lp
        got_ru[h] = sm[:, _lib.SUM_RU]
    assert np.array_equal(got, logp)
    assert int(got_ru.sum()) == int(summ[:, _lib.SUM_RU].sum())                      # checksum of checksums
    # generic kernel == column kernel on a subsample
    idx = rng.choice(len(reads), 3000, replace=False)
    sb, so = _lib.encode_reads([reads[i] for i in idx])
    lp_g, sm_g, _ = _lib.viterbi_batch([dm], sb, so, np.zeros(len(idx), np.int32), flags=_lib.FLAG_FORCE_GENERIC)
    assert np.array_equal(lp_g, logp[idx]) and np.array_equal(sm_g, summ[idx])
    # the batch above went through the row-blocked kernel (two reads per wavefront); one read per wavefront gives the same
    lp_a, sm_a, _ = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_ANTIDIAGONAL)
    assert np.array_equal(lp_a, logp) and np.array_equal(sm_a, summ)


@pytest.mark.parametrize("mode", ["columns", "stream", "rows"])
def test_random_locus_shapes_vs_oracle(mode, monkeypatch):
    """Shape sweep: random flank / pattern / copies (incl. a single copy: no fan-in state) / error rates / multi-row
    profiles, ragged read lengths 1..320 (1-4 chunks, row tiles, tiny models that use the range-checked sweep)."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(4242)
    flags = {"stream": _lib.FLAG_STREAM, "columns": _lib.FLAG_ANTIDIAGONAL, "rows": 0}[mode]     # rows = default routing
    for trial in range(10):
        flank = int(rng.integers(3, 60))
        plen = int(rng.integers(2, 30))
        copies = int(rng.integers(1, 8))
        loc = workloads.make_locus(rng, flank, plen, copies, float(rng.choice([0.05, 0.3])), n_units=int(rng.integers(1, 5)))
        dm = loc.model.device_model()
        assert dm.has_column_program()
        reads = [workloads.make_reads(rng, loc, 1, int(n))[0] for n in rng.integers(1, 321, 24)]
        bases, off = _lib.encode_reads(reads)
        logp, summ, paths = _lib.viterbi_batch([dm], bases, off, np.zeros(len(reads), np.int32), flags=flags, want_paths=True)
        a = loc.model.baked_arrays()
        edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
                 for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
        O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
        for i, r in enumerate(reads):
            olp, opath = O.viterbi(r)
            assert logp[i] == olp, (trial, flank, plen, copies, len(r))
            assert paths[i] == opath, (trial, flank, plen, copies, len(r))


@pytest.mark.parametrize("kernel", ["antidiagonal", "rows"])
def test_forward_column_kernel_vs_generic_and_oracle(kernel, monkeypatch):
    """log_probability on the column program (sum-product sweep, row tiles for long reads) against the generic
    forward kernel and the CPU oracle: rounding-level agreement (1e-9 relative; north-star bar 1e-4).  "rows": short
    reads through the row-blocked sum-product kernels (forward_rows.h), as in a large batch."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(8)
    loc = workloads.make_locus(rng, 60, 17, 5, n_units=3)
    reads = workloads.make_reads(rng, loc, 200, 150) + [workloads.rand_seq(rng, n) for n in (1, 2, 63, 64, 65, 257, 400, 700)]
    dm = loc.model.device_model()
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    col = _lib.forward_batch([dm], bases, off, which, flags=0 if kernel == "rows" else _lib.FLAG_ANTIDIAGONAL)
    gen = _lib.forward_batch([dm], bases, off, which, flags=_lib.FLAG_FORCE_GENERIC)
    assert np.all(np.abs(col - gen) <= 1e-9 * np.maximum(1.0, np.abs(gen)))
    a = loc.model.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
             for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    for i in list(range(0, 200, 9)) + list(range(200, len(reads))):
        want = O.forward(reads[i])
        assert abs(col[i] - want) <= 1e-9 * max(1.0, abs(want)), (i, len(reads[i]))
        assert col[i] >= _lib.viterbi_batch([dm], *_lib.encode_reads([reads[i]]), np.zeros(1, np.int32))[0][0] - 1e-9


def test_end_to_end_genotype_concordance():
    """prefilter -> scoring (both strands) -> recruit -> RU counts of spanning reads -> genotype, GPU pipeline vs the
    same pipeline on oracle scores: identical recruited sets, RU lists and genotype; and the genotype is the planted
    diploid one (3/5 copies)."""
    from advntr_amd import filtering, hmm_utils, settings, vntr_finder, workloads, _lib
    from oracle import oracle as Or
    rng = np.random.default_rng(77)
    pattern = workloads.rand_seq(rng, 14)
    left, right = workloads.rand_seq(rng, 150), workloads.rand_seq(rng, 150)
    reads = []
    for copies in (3, 5):
        allele = left + pattern * copies + right
        for _ in range(40):
            st = int(rng.integers(60, 120))
            s = allele[st:st + 150]
            s = "".join(("ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.005 else ch) for ch in s)
            reads.append(s if rng.random() < 0.5 else vntr_finder.reverse_complement(s))
    reads += [workloads.rand_seq(rng, 150) for _ in range(300)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    # stage 1: keyword prefilter (both strands of every read, as unmapped reads come in either orientation)
    kws = filtering.get_keywords_for_filtering(left, [pattern] * 4, right, pattern, True, 15)
    fasta = "".join(">r%d\n%s\n>r%d_rc\n%s\n" % (i, s, i, vntr_finder.reverse_complement(s)) for i, s in enumerate(reads))
    _, ids = filtering.get_filtered_read_ids(fasta, {1: kws}, min_matches=3)
    picked = sorted(set(int(n.split("_")[0][1:]) for n in ids[1]))
    assert 60 <= len(picked) <= 120
    cand = [reads[i] for i in picked]
    # stage 2: scoring + recruit on the GPU
    settings.MAX_ERROR_RATE = 0.05
    model = hmm_utils.get_read_matcher_model(left, right, [pattern], vntr_finder.get_copies_for_hmm(150, 14))
    scored = vntr_finder.score_reads(model, cand, scaled_score=None, compute_reverse=True)
    gpu_ru = [s.repeats for s in scored if s.recruited and s.summary[_lib.SUM_LEFT_BP] >= 5 and s.summary[_lib.SUM_RIGHT_BP] >= 5]
    # the same on oracle scores
    a = model.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
             for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = Or.OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    names = [s.name for s in model.states]
    cpu_ru = []
    for s in cand:
        lf, pf = O.viterbi(s)
        rc = vntr_finder.reverse_complement(s)
        lr, pr = O.viterbi(rc)
        seq, lp, path = (rc, lr, pr) if lf < lr else (s, lf, pf)
        inner = [names[i] for i in path][1:-1]
        if not Or.recruit_read(lp, inner, None, seq, left, right):
            continue
        if Or.left_flank_size(inner) >= 5 and Or.right_flank_size(inner) >= 5:
            cpu_ru.append(Or.number_of_repeats(inner))
    assert gpu_ru == cpu_ru and len(gpu_ru) >= 20
    geno, prob = vntr_finder.find_genotype_based_on_observed_repeats(gpu_ru)
    assert sorted(geno) == [3, 5]


def test_illumina_aggregation_on_gpu_summaries():
    """Reads of the aggregation golden scored on the GPU; the genotype results must equal what the reference's own
    find_repeat_count_from_alignment_file returned for them (tests/golden/illumina_aggregation.json.gz)."""
    from advntr_amd import _lib, vntr_finder
    g = load_golden("illumina_aggregation")
    dm, _ = device_model_from_golden(dict(g, kind="read_matcher"))
    for c in g["cases"]:
        reads = g["reads_by_case"][str(c["reads_ref"])]
        bases, off = _lib.encode_reads([r["seq"] for r in reads])
        logp, summ, _ = _lib.viterbi_batch([dm], bases, off, np.zeros(len(reads), np.int32))
        assert [float(x) for x in logp] == [r["logp"] for r in reads]
        res = vntr_finder.find_repeat_count_from_selected_reads(list(summ), accuracy_filter=c["accuracy_filter"],
                                                                average_coverage=c["average_coverage"])
        assert (None if res.copy_numbers is None else list(res.copy_numbers)) == c["copy_numbers"]
        assert (res.spanning_reads_count, res.flanking_reads_count) == (c["spanning"], c["flanking"])
        assert res.maximum_likelihood == c["max_likelihood"]


def test_pacbio_dominant_copy_numbers_match_reference():
    """get_dominant_copy_numbers_from_spanning_reads (vntr_finder.py:534-585): the golden genotypes/probabilities were
    returned by the reference's own method (tests/golden/make_golden.py); here the reads go through the row-tiled GPU
    kernel at the PacBio error setting."""
    from advntr_amd import settings, vntr_finder
    g = load_golden("pacbio_dominant_copy_numbers")
    settings.MAX_ERROR_RATE = 0.3
    try:
        for c in g["cases"]:
            geno, prob = vntr_finder.get_dominant_copy_numbers_from_spanning_reads(
                c["left"], c["right"], c["repeat_segments"], c["pattern"], c["reads"], accuracy_filter=c["accuracy_filter"])
            assert (None if geno is None else list(geno)) == c["copy_numbers"], c["case"]
            assert prob == c["max_prob"], c["case"]
    finally:
        settings.MAX_ERROR_RATE = 0.05


def test_cli_genotype_text_output(tmp_path):
    """python -m advntr_amd genotype: two loci, planted 3/5 and 4/4 genotypes, reference text format."""
    import json
    import subprocess
    import sys
    from advntr_amd import workloads, vntr_finder
    rng = np.random.default_rng(2718)
    loci, reads = [], []
    for vid, alleles in ((11, (3, 5)), (12, (4, 4))):
        pattern = workloads.rand_seq(rng, 16)
        left, right = workloads.rand_seq(rng, 150), workloads.rand_seq(rng, 150)
        loci.append({"id": vid, "left": left, "right": right, "pattern": pattern, "repeat_segments": [pattern], "scaled_score": None})
        for copies in alleles:
            allele = left + pattern * copies + right
            for _ in range(30):
                st = int(rng.integers(50, 110))
                s = allele[st:st + 150]
                reads.append(s if rng.random() < 0.5 else vntr_finder.reverse_complement(s))
    reads += [workloads.rand_seq(rng, 150) for _ in range(200)]
    (tmp_path / "loci.json").write_text(json.dumps(loci))
    (tmp_path / "reads.fa").write_text("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(reads)))
    from conftest import ROOT
    out = subprocess.run([sys.executable, "-m", "advntr_amd", "genotype", "--loci", str(tmp_path / "loci.json"),
                          "--reads", str(tmp_path / "reads.fa"), "--prefilter-both-strands"], cwd=ROOT,
                         stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out == "11\n3/5\n12\n4/4\n"


@pytest.mark.gpu
def test_cli_genotype_from_model_database(tmp_path):
    """The same pipeline fed from a sqlite model database in the reference's format, one locus with repeat segments of
    unequal length (needs --align-repeats; refused without it)."""
    import subprocess
    import sys
    from advntr_amd import models, workloads, vntr_finder
    from conftest import ROOT
    rng = np.random.default_rng(1618)
    db = str(tmp_path / "models.db")
    models.create_vntrs_database(db)
    reads = []
    for vid, alleles, ragged in ((21, (2, 4), False), (22, (3, 6), True)):
        pattern = workloads.rand_seq(rng, 18)
        left, right = workloads.rand_seq(rng, 500), workloads.rand_seq(rng, 500)
        segs = [pattern, pattern, pattern[:7] + pattern[8:]] if ragged else [pattern, pattern]
        v = models.ReferenceVNTR(vid, pattern, 1000 * vid, "chr1", None, None, len(segs))
        v.init_from_xml(segs, left, right)
        models.save_reference_vntr_to_database(v, db)
        for copies in alleles:
            allele = left + pattern * copies + right
            for _ in range(30):
                st = int(rng.integers(400, 460))
                s = allele[st:st + 150]
                reads.append(s if rng.random() < 0.5 else vntr_finder.reverse_complement(s))
    reads += [workloads.rand_seq(rng, 150) for _ in range(100)]
    (tmp_path / "reads.fa").write_text("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(reads)))
    cmd = [sys.executable, "-m", "advntr_amd", "genotype", "--models", db, "--reads", str(tmp_path / "reads.fa"),
           "--prefilter-both-strands"]
    out = subprocess.run(cmd + ["--align-repeats"], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out == "21\n2/4\n22\n3/6\n"
    out = subprocess.run(cmd + ["--vntr-id", "21"], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out == "21\n2/4\n"
    bad = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode != 0 and b"alignment" in bad.stderr
    # BED and VCF rows (the writers are pinned on the reference's output in tests/test_output_rows.py)
    bed = subprocess.run(cmd + ["--vntr-id", "21", "--outfmt", "bed"], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert bed.splitlines()[0] == "#CHROM\tStart\tEnd\tVNTR_ID\tGene\tMotif\tRefCopy\tR1\tR2"
    assert bed.splitlines()[1].split("\t")[:4] == ["chr1", "21000", str(21000 + 36), "21"] and bed.splitlines()[1].endswith("\t2\t4")
    vcf = subprocess.run(cmd + ["--vntr-id", "21", "--outfmt", "vcf"], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode()
    rows = [l for l in vcf.splitlines() if not l.startswith("#")]
    assert vcf.startswith("##fileformat=VCFv4.2\n") and "##contig=<ID=1>" in vcf and len(rows) == 1
    f = rows[0].split("\t")
    assert f[0] == "chr1" and f[7].startswith("END=%d;VID=21;RU=" % (21000 + 36)) and f[8] == "GT:DP:SR:FR:ML"
    assert f[9].startswith("0/1:")                      # 2 copies = the reference count, 4 = the one alternative allele


@pytest.mark.gpu
def test_score_reads_multi_equals_per_locus_calls():
    """One engine batch over several loci == the per-locus calls (same strand choice, logp, summaries, verdicts)."""
    from advntr_amd import workloads, vntr_finder
    rng = np.random.default_rng(4242)
    loci = [workloads.make_locus(rng, 150, int(L), vntr_finder.get_copies_for_hmm(150, int(L))) for L in (9, 14, 40)]
    workloads.build_models(loci)
    reads = [workloads.make_reads(rng, loc, 60, 150, locus_fraction=0.7) for loc in loci]
    reads[1][3] = reads[1][3][:20] + "N" + reads[1][3][21:]
    scores = [None, -1.2, None]
    multi = vntr_finder.score_reads_multi([l.model for l in loci], reads, scores)
    for loc, rs, sc, got in zip(loci, reads, scores, multi):
        want = vntr_finder.score_reads(loc.model, rs, sc)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            if b is None:
                assert a is None
                continue
            assert (a.sequence, a.logp, a.reversed, a.recruited) == (b.sequence, b.logp, b.reversed, b.recruited)
            assert np.array_equal(a.summary, b.summary)
    assert multi[1][3] is None and sum(s.recruited for s in multi[0] if s) > 10


@pytest.mark.gpu
def test_score_reads_arrays_equals_object_path():
    """The array form (no Python object per read, reverse complement on the code array, vectorised recruit rule)
    gives the same strand, logp, summaries and verdicts as score_reads_multi."""
    from advntr_amd import workloads, vntr_finder
    rng = np.random.default_rng(777)
    loci = [workloads.make_locus(rng, 150, int(L), vntr_finder.get_copies_for_hmm(150, int(L))) for L in (7, 22, 61)]
    workloads.build_models(loci)
    reads = [workloads.make_reads(rng, loc, 80, int(n), locus_fraction=0.7) for loc, n in zip(loci, (150, 150, 120))]
    reads[2][5] = "N" + reads[2][5][1:]
    scores = [None, -1.15, 0]
    objs = vntr_finder.score_reads_multi([l.model for l in loci], reads, scores)
    arr = vntr_finder.score_reads_arrays([l.model for l in loci], reads, scores)
    assert len(arr["logp"]) == sum(1 for rs in objs for s in rs if s is not None)
    for k in range(len(arr["logp"])):
        o = objs[int(arr["locus"][k])][int(arr["index"][k])]
        assert (o.logp, o.reversed, o.recruited, len(o.sequence)) == (float(arr["logp"][k]), bool(arr["reversed"][k]),
                                                                      bool(arr["recruited"][k]), int(arr["length"][k]))
        assert np.array_equal(o.summary, arr["summary"][k])
    assert arr["recruited"].sum() > 30 and (~arr["recruited"]).sum() > 30 and arr["reversed"].sum() > 5


@pytest.mark.gpu
def test_bulk_uploaded_models_equal_single_uploads():
    """advntr_built_upload_many (one slab for many models) vs one advntr_built_upload per model: same results; the slab
    survives until its last model is destroyed."""
    import gc
    from advntr_amd import _lib, workloads, hmm_utils
    from advntr_amd.pomegranate import device_models
    rng = np.random.default_rng(31)
    specs = []
    for L in (6, 11, 30, 57):
        loc = workloads.make_locus(rng, 150, L, 3, n_units=3)
        specs.append((loc.left, loc.right, loc.units, loc.copies))
    bulk = hmm_utils.build_read_matcher_models(specs)
    single = hmm_utils.build_read_matcher_models(specs)
    dms = device_models(bulk)
    assert all(d.has_column_program() for d in dms)
    reads = [workloads.rand_seq(rng, 150) for _ in range(40)] + [s[2][0] * 4 for s in specs]
    bases, off = _lib.encode_reads(reads * 4)
    which = np.repeat(np.arange(4, dtype=np.int32), len(reads))
    a = _lib.viterbi_batch(dms, bases, off, which)
    b = _lib.viterbi_batch([m.device_model() for m in single], bases, off, which)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # drop three of the four slab models; the fourth must still score
    keep = bulk[2]
    del dms, bulk
    gc.collect()
    one, off1 = _lib.encode_reads(reads)
    c = _lib.viterbi_batch([keep.device_model()], one, off1, np.zeros(len(reads), np.int32))
    assert np.array_equal(c[0], a[0][2 * len(reads):3 * len(reads)])


@pytest.mark.gpu
def test_cli_sharded_over_two_ranks_equals_single_process(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 -m advntr_amd genotype ...`: loci split over two ranks (the
    host communicator here, so that both can share the one GPU of the test box; RCCL on a multi-GPU node), rows gathered
    to rank 0 -- identical stdout to the single-process run.  torch is only the launcher; the product imports none of it."""
    import json
    import os
    import subprocess
    import sys
    from advntr_amd import workloads, vntr_finder
    from conftest import ROOT
    rng = np.random.default_rng(99)
    loci, reads = [], []
    for vid, alleles in ((31, (3, 5)), (32, (4, 4)), (33, (2, 6)), (34, (5, 7)), (35, (3, 3))):
        pattern = workloads.rand_seq(rng, int(rng.integers(10, 30)))
        left, right = workloads.rand_seq(rng, 150), workloads.rand_seq(rng, 150)
        loci.append({"id": vid, "left": left, "right": right, "pattern": pattern, "repeat_segments": [pattern],
                     "scaled_score": None, "chromosome": "chr2", "start_point": 1000 * vid})
        for copies in alleles:
            allele = left + pattern * copies + right
            for _ in range(25):
                st = int(rng.integers(40, 100))
                s = allele[st:st + 150]
                reads.append(s if rng.random() < 0.5 else vntr_finder.reverse_complement(s))
    (tmp_path / "loci.json").write_text(json.dumps(loci))
    (tmp_path / "reads.fa").write_text("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(reads)))
    args = ["genotype", "--loci", str(tmp_path / "loci.json"), "--reads", str(tmp_path / "reads.fa"), "--outfmt", "bed",
            "--prefilter-both-strands"]
    single = subprocess.run([sys.executable, "-m", "advntr_amd"] + args, cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout
    env = dict(os.environ, ADVNTR_DIST_BACKEND="host")
    multi = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", "29577", "-m", "advntr_amd"] + args,
                           cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert single.decode().count("\n") == 6 and b"\t5\t7\n" in single
    rows = lambda out: [l for l in out.split(b"\n") if l.startswith((b"#CHROM", b"chr"))]
    assert len(rows(single)) == 6 and rows(multi) == rows(single)


@pytest.mark.gpu
def test_model_update_from_reads_equals_the_reference_model():
    """vntr_finder.update_model_from_reads: the GPU's paths for the golden's reads give the alignment, and therefore the
    re-estimated model, the reference produced (tests/golden/model_update.json.gz)."""
    from advntr_amd import hmm_utils, vntr_finder
    from oracle.oracle import OracleModel
    g = load_golden("model_update")
    n_ref = 4
    seqs = [s for s, _ in g["vpaths"]]
    pattern = seqs[-1]
    model = hmm_utils.get_read_matcher_model(g["left"], g["right"], [pattern] * n_ref, g["copies"])
    new = vntr_finder.update_model_from_reads(model, g["left"], g["right"], seqs[-n_ref:], pattern, seqs[:-n_ref], read_length=100)
    gm = g["model"]
    assert [s.name for s in new.states] == gm["state_names"]
    a = new.baked_arrays()
    in_ptr, in_src, in_logp, _ = OracleModel.from_golden(g).csr()
    assert np.array_equal(a["in_ptr"], in_ptr) and np.array_equal(a["in_src"], in_src)
    assert np.all(np.abs(a["in_logp"] - in_logp)[np.isfinite(in_logp)] <= 4 * np.spacing(np.abs(in_logp[np.isfinite(in_logp)])))
    assert np.array_equal(a["emis_logp"], np.array([e["logp"] for e in gm["emissions"]]))


@pytest.mark.gpu
def test_sum_product_linear_domain_ranges():
    """Model.log_probability on the column program works in the linear domain (probabilities times 16^row, row tiles
    renormalised at their seams): long reads on a PacBio-parameter model, reads at every bucket / tile boundary and
    worst-case reads that match nothing, against the oracle's log-domain forward (hmm.pyx:1371-1484)."""
    from advntr_amd import settings, workloads
    from oracle.oracle import OracleModel

    def oracle_of(model):
        a = model.baked_arrays()
        edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k])) for l in range(a["m"])
                 for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
        return OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])

    rng = np.random.default_rng(3)
    loc = workloads.make_locus(rng, 100, 25, 30, error_rate=0.3)
    O = oracle_of(loc.model)
    reads = [workloads.noisy_copy(rng, loc.left[-100:] + loc.units[0] * int(rng.integers(2, 28)) + loc.right[:100], 0.12)
             for _ in range(8)]
    reads += [workloads.rand_seq(rng, n) for n in (1, 2, 63, 64, 65, 150, 192, 193, 200, 256, 257, 384, 385, 700, 1500)]
    got = loc.model.log_probability_batch(reads)
    for r, g in zip(reads, got):
        want = O.forward(r)
        assert abs(g - want) <= 1e-9 * max(1.0, abs(want)), (len(r), g, want)
    ref = workloads.ref150()
    O = oracle_of(ref.model)
    hard = ["A" * 150, "ACGT" * 48, "T" * 192, "G" * 256, "C" * 400]
    got = ref.model.log_probability_batch(hard)
    for r, g in zip(hard, got):
        want = O.forward(r)
        assert np.isfinite(g) and abs(g - want) <= 1e-9 * abs(want), (len(r), g, want)
    assert min(got) < -800                                     # far below exp(-709): the scaling keeps it representable


@pytest.mark.gpu
def test_repeat_finder_segmentation_equals_the_reference():
    """ReferenceVNTR.find_repeat_segments (reference_vntr.py:80-87): the repeat-finder HMM (default bake with merging,
    emitting random-match states => generic-CSR kernel) segments a reference region; log-prob, path and segments are
    the reference's (tests/golden/bake_merge.json.gz)."""
    from advntr_amd import hmm_utils, models
    g = load_golden("bake_merge")
    for f in g["repeat_finder"]:
        m = hmm_utils.build_reference_repeat_finder_hmm([f["pattern"]], copies=f["copies"])
        assert not m.device_model().has_column_program()
        for r in f["regions"]:
            logp, path = m.viterbi(r["region"])
            assert logp == r["logp"] and [i for i, _ in path] == r["path"]
            assert hmm_utils.find_repeat_segments(f["pattern"], f["copies"], r["region"]) == r["segments"]
    # init_from_vntrseek_data on a synthetic chromosome: the planted units come back as the repeat segments
    f = g["repeat_finder"][0]
    units = f["regions"][0]["segments"]
    rng = np.random.default_rng(1)
    left = "".join("ACGT"[i] for i in rng.integers(0, 4, 700))
    right = "".join("ACGT"[i] for i in rng.integers(0, 4, 700))
    chrom = left + "".join(units) + right
    v = models.ReferenceVNTR(1, f["pattern"], 700, "chr1", None, None, estimated_repeats=len(units))
    models.init_from_vntrseek_data(v, chrom)
    assert len(v.repeat_segments) == len(units) and "".join(v.repeat_segments) == chrom[700:700 + v.get_length()]
    assert v.repeat_segments[:3] == units[:3]
    assert v.left_flanking_region == left[-500:] and v.right_flanking_region == chrom[700 + v.get_length():][:500]


@pytest.mark.gpu
def test_models_loaded_from_json_score_like_the_reference():
    """The stored-HMM path (vntr_finder.py:124-137): a model loaded from the reference's JSON scores reads as the
    reference's loaded model does."""
    from advntr_amd import HiddenMarkovModel
    g = load_golden("model_json")
    for case in g["cases"]:
        m = HiddenMarkovModel.from_json(case["json"])
        for r, want in zip(case["reads"], case["logp"]):
            assert m.viterbi(r)[0] == want, (case["name"], r)


@pytest.mark.gpu
def test_row_blocked_kernels_at_their_length_boundaries():
    """A large mixed batch under the default routing: every row-blocked configuration at the edges
    of its length range (<4,4> 1-64, <4,2> 65-124, <5,2> 125-155, tiled 156+ incl. exact tile multiples and every number of
    rows per lane in the last tile) against the
    generic-CSR kernel and, on a sample, the oracle; log-probs, summaries and paths identical."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(77)
    loc = workloads.make_locus(rng, 60, 9, 9, 0.05, n_units=3)
    dm = loc.model.device_model()
    assert dm.has_column_program()
    # (a long read's last row tile of up to 320 rows runs with 1 .. 5 rows per lane: 64-row steps of its fill on either side of each boundary)
    lens = [1, 2, 4, 5, 63, 64, 65, 80, 123, 124, 125, 128, 129, 150, 154, 155, 156, 160, 192, 193, 255, 256, 257, 300, 320, 321,
            384, 385, 448, 449, 511, 512, 513, 576, 577, 640, 641, 705, 960, 961]
    reads = []
    for i in range(7000):
        n = lens[i % len(lens)]
        reads.append(workloads.make_reads(rng, loc, 1, n, locus_fraction=0.7, sub_rate=0.03)[0])
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    a = _lib.viterbi_batch([dm], bases, off, which, want_paths=True)
    b = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_FORCE_GENERIC, want_paths=True)
    c = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_ANTIDIAGONAL, want_paths=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and a[2] == c[2]
    m = loc.model.baked_arrays()
    edges = [(int(m["in_src"][k]), l, float(m["in_logp"][k])) for l in range(m["m"]) for k in range(m["in_ptr"][l], m["in_ptr"][l + 1])]
    O = OracleModel(m["m"], m["silent_start"], m["start_index"], m["end_index"], edges, m["emis_logp"])
    for i in range(0, 2 * len(lens)):
        olp, opath = O.viterbi(reads[i])
        assert a[0][i] == olp and a[2][i] == opath, (i, len(reads[i]))


@pytest.mark.parametrize("mode", ["rows", "columns"])
def test_long_deletion_runs_in_the_traceback(mode):
    """Reads with 9-40 consecutive bases of the locus missing (flank, inside a unit, across unit boundaries, whole units): the
    Viterbi path crosses them in runs of delete / silent b states longer than the traceback's horizontal gather (8 cells),
    some ending on a fan-in sink column; short reads (row-blocked kernels) and reads beyond 155 bases (tiled kernel).  Paths
    and scores against the oracle."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(777)
    flags = {"columns": _lib.FLAG_ANTIDIAGONAL, "rows": 0}[mode]
    n_checked = n_long = 0
    for trial in range(6):
        flank = int(rng.integers(40, 90))
        plen = int(rng.integers(8, 40))
        copies = int(rng.integers(3, 7))
        loc = workloads.make_locus(rng, flank, plen, copies, 0.05, n_units=int(rng.integers(1, 4)))
        dm = loc.model.device_model()
        assert dm.has_column_program()
        full = loc.left + "".join(loc.units[i % len(loc.units)] for i in range(copies)) + loc.right
        reads = []
        for _ in range(20):
            cut = int(rng.integers(9, 41))
            at = int(rng.integers(5, max(6, len(full) - cut - 5)))
            s = full[:at] + full[at + cut:]
            lo = int(rng.integers(0, max(1, len(s) - 60)))
            n = int(rng.integers(60, 260))
            r = s[lo:lo + n]
            if len(r) >= 30:
                reads.append(r)
        bases, off = _lib.encode_reads(reads)
        logp, summ, paths = _lib.viterbi_batch([dm], bases, off, np.zeros(len(reads), np.int32), flags=flags, want_paths=True)
        a = loc.model.baked_arrays()
        edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
                 for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
        O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
        names = [s.name for s in loc.model.states]
        for i, r in enumerate(reads):
            olp, opath = O.viterbi(r)
            assert logp[i] == olp, (trial, i, len(r))
            assert paths[i] == opath, (trial, i, len(r))
            run = longest = 0
            for j in opath:                      # longest run of non-emitting states on the path
                run = run + 1 if j >= a["silent_start"] else 0
                longest = max(longest, run)
            n_long += longest > 8
            n_checked += 1
    assert n_checked >= 60 and n_long >= 10, (n_checked, n_long)      # the case this test is about did occur


def test_paths_near_the_reference_capacity_on_a_tiny_model():
    """A 60-state model (flank 3, 7-bp unit) and reads of up to 400 bases: the Viterbi path loops through the unit's silent
    states once per copy and comes within a few entries of n + m, the size of the reference's own path buffer
    (hmm.pyx:1953).  Every kernel accepts paths of up to n + m entries and refuses longer ones the same way (length -2, no
    summary) -- found by scripts/fuzz_kernels.py: the column kernels used to stop 66 entries early."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(1459)
    loc = workloads.make_locus(rng, 3, 7, 1, 0.3, n_units=2)
    dm = loc.model.device_model()
    a = loc.model.baked_arrays()
    m = a["m"]
    reads = [workloads.make_reads(rng, loc, 1, int(n), locus_fraction=0.9, sub_rate=0.05)[0] for n in rng.integers(150, 400, 60)]
    reads += [(loc.units[0] * 60)[:n] for n in (120, 155, 156, 250, 399)]          # nothing but copies of the unit
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k])) for l in range(m) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = OracleModel(m, a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    res = {name: _lib.viterbi_batch([dm], bases, off, which, flags=fl)
           for name, fl in (("rows", 0), ("antidiagonal", _lib.FLAG_ANTIDIAGONAL), ("generic", _lib.FLAG_FORCE_GENERIC))}
    near = 0
    for i, r in enumerate(reads):
        olp, opath = O.viterbi(r, path_cap=4 * (len(r) + m))
        want_len = len(opath) if len(opath) <= len(r) + m else -2
        near += len(opath) > len(r) + m - 66
        for name, (logp, summ, _) in res.items():
            assert logp[i] == olp, (name, i)
            assert summ[i][_lib.SUM_PATH_LEN] == want_len, (name, i, len(r), len(opath), summ[i])
        assert np.array_equal(res["rows"][1][i], res["generic"][1][i]) and np.array_equal(res["rows"][1][i], res["antidiagonal"][1][i])
    assert near >= 3, near            # the case this test is about did occur


def test_very_wide_model_runs_from_the_lowest_lds_staging_level():
    """A locus with 2 000-base flanks: 4 000+ columns, 12 000+ states.  Only the class / emission tables and the sweep's
    padded info copy fit in LDS (staging level 0: the traceback reads the state and info tables from the model blob); the
    generic kernel cannot take the model at all (its trellis rows sit in LDS).  Scores and paths against the oracle, short
    reads (row-blocked kernels) and long ones (row tiles), and the anti-diagonal route against the default one."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(2000)
    loc = workloads.make_locus(rng, 2000, 25, 6, 0.05, n_units=2)
    dm = loc.model.device_model()
    assert dm.has_column_program()
    a = loc.model.baked_arrays()
    assert a["m"] > 10240
    reads = [workloads.make_reads(rng, loc, 1, int(n), locus_fraction=1.0, sub_rate=0.03)[0] for n in (1, 40, 100, 150, 155, 156, 300, 700)]
    bases, off = _lib.encode_reads(reads)
    which = np.zeros(len(reads), np.int32)
    logp, summ, paths = _lib.viterbi_batch([dm], bases, off, which, want_paths=True)
    alt = _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_ANTIDIAGONAL, want_paths=True)
    assert np.array_equal(alt[0], logp) and np.array_equal(alt[1], summ) and alt[2] == paths
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k])) for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    for i, r in enumerate(reads):
        olp, opath = O.viterbi(r)
        assert logp[i] == olp and paths[i] == opath, (i, len(r))
    with pytest.raises(_lib.EngineError):
        _lib.viterbi_batch([dm], bases, off, which, flags=_lib.FLAG_FORCE_GENERIC)


def test_reads_of_a_hundred_thousand_bases():
    """Reads far beyond the old 65 536-base bound of the row-tiled kernel (ultra-long nanopore reads): scores and paths
    against the oracle."""
    from advntr_amd import _lib, workloads
    from oracle.oracle import OracleModel
    rng = np.random.default_rng(5)
    loc = workloads.make_locus(rng, 60, 20, 5, 0.3, n_units=2)
    dm = loc.model.device_model()
    a = loc.model.baked_arrays()
    edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k])) for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
    O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
    unit = loc.units[0]
    reads = []
    for n in (65537, 100000):
        body = (loc.left + unit * (n // len(unit) + 1))[:n - 60] + loc.right
        reads.append("".join(c if rng.random() > 0.1 else "ACGT"[int(rng.integers(0, 4))] for c in body))
    bases, off = _lib.encode_reads(reads)
    logp, summ, paths = _lib.viterbi_batch([dm], bases, off, np.zeros(len(reads), np.int32), want_paths=True)
    for i, r in enumerate(reads):
        olp, opath = O.viterbi(r, path_cap=4 * (len(r) + a["m"]))
        assert logp[i] == olp and paths[i] == opath, (i, len(r))


def test_model_destroyed_before_its_batch_is_released_by_the_batch():
    """advntr_hmm_destroy on a model that a live batch is bound to (a C-ABI caller breaking the ownership contract) must
    not hand the model's memory to the next upload: the release is deferred to advntr_batch_destroy.  The batch keeps
    scoring correctly while other models are uploaded over the cache the memory would have gone to."""
    import gc
    from advntr_amd import _lib, workloads, hmm_utils
    from advntr_amd.pomegranate import device_models
    rng = np.random.default_rng(77)
    L = _lib.load()
    specs = []
    for k in (9, 14, 33):
        loc = workloads.make_locus(rng, 150, k, 4, n_units=2)
        specs.append((loc.left, loc.right, loc.units, loc.copies))
    reads = [workloads.rand_seq(rng, 150) for _ in range(64)] + [s[2][0] * 6 for s in specs]
    bases, off = _lib.encode_reads(reads * 3)
    which = np.repeat(np.arange(3, dtype=np.int32), len(reads))
    models = hmm_utils.build_read_matcher_models(specs)
    dms = device_models(models)                                     # one slab for the three
    want = _lib.viterbi_batch(dms, bases, off, which)
    batch = _lib.DeviceBatch(dms, bases, off, which)
    for d in dms:                                                   # the owner lets go while the batch is alive
        L.advntr_hmm_destroy(d._h)
        d._h = None
    del dms, models
    gc.collect()
    others = []
    for _ in range(3):                                              # uploads that would reuse a released slab
        more = hmm_utils.build_read_matcher_models([(workloads.rand_seq(rng, 150), workloads.rand_seq(rng, 150),
                                                     [workloads.rand_seq(rng, 21)], 5) for _ in range(3)])
        others.append((more, device_models(more)))
    batch.run()
    logp, summ = batch.fetch()
    assert np.array_equal(logp, want[0]) and np.array_equal(summ, want[1])
    batch.models = []                                               # (the wrapper's references: handles are gone already)
    batch.close()
