"""Host logic of the caller mirror (advntr_amd/vntr_finder.py) against values captured from the reference."""
import numpy as np

from conftest import READ_MATCHER_GOLDENS, load_golden
from advntr_amd import _lib, vntr_finder
from oracle import oracle as O


def test_genotype_cases_match_reference():
    g = load_golden("genotype_cases")
    for c in g["cases"]:
        geno, prob = vntr_finder.find_genotype_based_on_observed_repeats(list(c["observed"]), c["haploid"])
        assert (None if geno is None else list(geno)) == c["genotype"], c
        assert prob == c["max_prob"], c
    for c in g["copies_for_hmm"]:
        assert vntr_finder.get_copies_for_hmm(c["read_length"], c["pattern_len"]) == c["copies"]


def test_recruit_read_from_summaries_matches_reference_verdicts():
    """recruit_read fed with the 8-int summary layout the kernel produces (built here from the golden path)."""
    for name in READ_MATCHER_GOLDENS:
        g = load_golden(name)
        names = g["model"]["state_names"]
        for r in g["reads"]:
            if "recruit" not in r:
                continue
            inner = [names[i] for i in r["path"]][1:-1]
            lm, lb, rm, rb = O.flanking_counts(inner, r["seq"], g["left"], g["right"])
            s = np.zeros(8, np.int32)
            s[_lib.SUM_RU], s[_lib.SUM_MATCHES], s[_lib.SUM_REPEAT_BP] = r["ru"], r["matches"], r["repeat_bp"]
            s[_lib.SUM_LEFT_BP], s[_lib.SUM_RIGHT_BP], s[_lib.SUM_LEFT_MATCH], s[_lib.SUM_RIGHT_MATCH] = lb, rb, lm, rm
            s[_lib.SUM_PATH_LEN] = len(r["path"])
            ms = vntr_finder.get_min_score_to_select_a_read(g["scaled_score"], len(r["seq"]))
            assert vntr_finder.recruit_read(r["logp"], s, ms, len(r["seq"])) == r["recruit"], (name, r["seq"])
            assert vntr_finder.recruit_read(r["logp"], s, None, len(r["seq"])) == r["recruit_noscore"]


def test_reverse_complement():
    assert vntr_finder.reverse_complement("AACGT") == "ACGTT"


def _summary_from_path(g, r):
    names = g["model"]["state_names"]
    inner = [names[i] for i in r["path"]][1:-1]
    lm, lb, rm, rb = O.flanking_counts(inner, r["seq"], g["left"], g["right"])
    s = np.zeros(8, np.int32)
    s[_lib.SUM_RU] = O.number_of_repeats(inner)
    s[_lib.SUM_MATCHES] = O.number_of_matches(inner)
    s[_lib.SUM_REPEAT_BP] = O.repeat_bp_matches(inner)
    s[_lib.SUM_LEFT_BP], s[_lib.SUM_RIGHT_BP], s[_lib.SUM_LEFT_MATCH], s[_lib.SUM_RIGHT_MATCH] = lb, rb, lm, rm
    s[_lib.SUM_PATH_LEN] = len(r["path"])
    return s


def test_illumina_aggregation_matches_reference():
    """find_repeat_count_from_alignment_file after read selection (vntr_finder.py:807-887): the golden results were
    produced by the reference's own method with its BAM-reading selection stubbed (tests/golden/make_golden.py)."""
    g = load_golden("illumina_aggregation")
    for c in g["cases"]:
        reads = g["reads_by_case"][str(c["reads_ref"])]
        summaries = [_summary_from_path(g, r) for r in reads]
        res = vntr_finder.find_repeat_count_from_selected_reads(summaries, accuracy_filter=c["accuracy_filter"],
                                                                average_coverage=c["average_coverage"])
        got = None if res.copy_numbers is None else list(res.copy_numbers)
        assert got == c["copy_numbers"], c
        assert (res.recruited_reads_count, res.spanning_reads_count, res.flanking_reads_count) == \
            (c["recruited"], c["spanning"], c["flanking"]), c
        assert res.maximum_likelihood == c["max_likelihood"], c


def test_recruit_mask_equals_scalar_rule():
    """The vectorised keep/discard rule against recruit_read (pinned on the reference's verdicts by the goldens),
    on random summaries covering every branch (zero flank bp, with/without a trained score, ties at the bounds)."""
    import numpy as np
    from advntr_amd import _lib, vntr_finder
    rng = np.random.default_rng(12)
    n = 4000
    summ = np.zeros((n, 8), np.int32)
    summ[:, _lib.SUM_LEFT_BP] = rng.integers(0, 40, n) * (rng.random(n) < 0.8)
    summ[:, _lib.SUM_RIGHT_BP] = rng.integers(0, 40, n) * (rng.random(n) < 0.8)
    summ[:, _lib.SUM_LEFT_MATCH] = (summ[:, _lib.SUM_LEFT_BP] * rng.choice([1.0, 0.95, 0.9, 0.85], n)).astype(np.int32)
    summ[:, _lib.SUM_RIGHT_MATCH] = (summ[:, _lib.SUM_RIGHT_BP] * rng.choice([1.0, 0.95, 0.9, 0.5], n)).astype(np.int32)
    lens = rng.integers(100, 151, n)
    summ[:, _lib.SUM_MATCHES] = (lens * rng.choice([1.0, 0.9, 0.89, 0.5], n)).astype(np.int32)
    summ[:, _lib.SUM_PATH_LEN] = rng.choice([0, 2, 200], n, p=[0.05, 0.05, 0.9])
    logp = -rng.random(n) * 2.0 * lens
    scaled = rng.choice([np.nan, -1.0, -0.5], n)
    logp[::17] = (scaled * lens)[::17]                      # exact ties with the threshold
    logp[np.isnan(logp)] = -50.0
    got = vntr_finder.recruit_mask(logp, summ, lens, scaled * lens)
    for i in range(n):
        ms = None if np.isnan(scaled[i]) else float(scaled[i] * lens[i])
        want = summ[i, _lib.SUM_PATH_LEN] > 2 and vntr_finder.recruit_read(float(logp[i]), summ[i], ms, int(lens[i]))
        assert bool(got[i]) == bool(want), i
    assert 0.05 < got.mean() < 0.95


def test_reference_genotyping_unit_tests_replayed():
    """/root/reference/tests/test_genotyping.py, its five cases with their expected values (the reference's own asserts
    compare against the (genotype, probability) pair the function returns nowadays; the genotypes are what is pinned)."""
    import numpy as np
    from advntr_amd import vntr_finder

    def genotype(observed, haploid=False):
        g = vntr_finder.find_genotype_based_on_observed_repeats(observed, haploid)[0]
        return tuple(sorted(g))
    assert genotype([3, 3, 3, 3, 3]) == (3, 3)
    assert genotype([2, 3, 3, 3, 3], haploid=True) == (3, 3)
    assert genotype([2, 2, 3, 3, 3]) == (2, 3)
    assert genotype([4, 5, 5, 5, 7, 8, 8, 8, 9]) == (5, 8)
    # test_recruit_read_for_positive_read: an empty path has flank match rate 1, the score clears the threshold
    empty = np.zeros(8, np.int32)
    assert vntr_finder.recruit_read(-20, empty, -50, 100) is True
    assert vntr_finder.recruit_read(-60, empty, -50, 100) is False


def test_spanning_piece_equals_slicing_the_whole_strand():
    """extract_spanning_reads_multi upper-cases and reverse-complements only the trimmed piece of a long read: the same string
    as slicing the upper-cased (reverse-complemented) whole read, as the reference does (vntr_finder.py:362,367-371)."""
    import numpy as np
    from advntr_amd import vntr_finder as vf
    rng = np.random.default_rng(9)
    for _ in range(300):
        n = int(rng.integers(0, 400))
        read = "".join(rng.choice(list("ACGTacgtN"), n))
        b = int(rng.integers(0, n + 2))
        e = int(rng.integers(b, n + 120))
        fwd = read.upper()
        rev = fwd.translate(str.maketrans("ACGTN", "TGCAN"))[::-1]
        assert vf._spanning_piece(read, b, e, False) == fwd[b:e]
        assert vf._spanning_piece(read, b, e, True) == rev[b:e]


def test_round4_workload_generators_are_seeded_and_shaped():
    from advntr_amd import workloads
    loci, reads = workloads.make_pacbio_whole_reads(3, seed=5, n_reads=10, min_len=1200, max_len=2000, workers=1)
    again = workloads.make_pacbio_whole_reads(3, seed=5, n_reads=10, min_len=1200, max_len=2000, workers=1)
    assert reads == again[1] and len(loci) == 3 and all(len(r) == 10 for r in reads)
    assert all(len(l[0]) == 500 and len(l[1]) == 500 and l[2] == [l[3]] for l in loci)
    lines, fasta, rec = workloads.make_prefilter_workload(20, 300, read_len=50, locus_every=10)
    assert len(fasta) == 300 * rec and rec == 10 + 50 + 1 and len(lines) == 20
    assert fasta[:rec] == b">r0000000\n" + fasta[10:60] + b"\n" and fasta[rec:rec + 10] == b">r0000001\n"
    assert set(fasta[10:60]) <= set(b"ACGT") and all(len(k) == 15 for _, kws in lines for k in kws)
    left, right, long_reads = workloads.make_flank_align_workload(6, min_len=500, max_len=900)
    assert len(left) == len(right) == 100 and len(long_reads) == 6 and all(500 <= len(s) <= 900 for s in long_reads)
