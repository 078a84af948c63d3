"""The native genotype caller (csrc/genotype_caller.h behind advntr_genotype_illumina; host C++, no GPU) against
(1) the goldens the reference's own methods produced -- genotypes, read counts and probabilities bit-equal
    (/root/reference/advntr/vntr_finder.py:473-532, 807-887) -- and
(2) the Python mirror of the same functions (pinned on the same goldens) on random read sets covering every branch."""
import numpy as np

from conftest import load_golden
from advntr_amd import _lib, vntr_finder
from test_vntr_finder import _summary_from_path


def _spanning(ru):
    """a summary record that read_flanks_repeats_with_confidence accepts, with the given RU count"""
    s = np.zeros(8, np.int32)
    s[_lib.SUM_RU] = ru
    s[_lib.SUM_LEFT_BP] = s[_lib.SUM_LEFT_MATCH] = s[_lib.SUM_RIGHT_BP] = s[_lib.SUM_RIGHT_MATCH] = 20
    s[_lib.SUM_PATH_LEN] = 200
    return s


def test_genotype_cases_golden():
    g = load_golden("genotype_cases")
    for haploid in (False, True):
        cases = [c for c in g["cases"] if bool(c["haploid"]) == haploid]
        summ = np.array([_spanning(ru) for c in cases for ru in c["observed"]], np.int32).reshape(-1, 8)
        off = np.concatenate([[0], np.cumsum([len(c["observed"]) for c in cases])])
        res = vntr_finder.find_repeat_counts_of_loci(summ, off, is_haploid=haploid)
        for c, r in zip(cases, res):
            assert (None if r.copy_numbers is None else list(r.copy_numbers)) == c["genotype"], c
            assert r.maximum_likelihood == c["max_prob"], c                  # bit-equal


def test_illumina_aggregation_golden():
    g = load_golden("illumina_aggregation")
    for accuracy in (False, True):
        cases = [c for c in g["cases"] if bool(c["accuracy_filter"]) == accuracy and not c["average_coverage"]]
        assert cases
        groups = [[_summary_from_path(g, r) for r in g["reads_by_case"][str(c["reads_ref"])]] for c in cases]
        summ = np.array([s for grp in groups for s in grp], np.int32).reshape(-1, 8)
        off = np.concatenate([[0], np.cumsum([len(grp) for grp in groups])])
        for threads in (1, 4):
            res = vntr_finder.find_repeat_counts_of_loci(summ, off, accuracy_filter=accuracy, threads=threads)
            for c, r in zip(cases, res):
                assert (None if r.copy_numbers is None else list(r.copy_numbers)) == c["copy_numbers"], c
                assert (r.recruited_reads_count, r.spanning_reads_count, r.flanking_reads_count) == \
                    (c["recruited"], c["spanning"], c["flanking"]), c
                assert r.maximum_likelihood == c["max_likelihood"], c


def test_native_equals_python_mirror_on_random_loci():
    rng = np.random.default_rng(77)
    groups = []
    for k in range(600):
        n = int(rng.integers(0, 40))
        base = int(rng.integers(0, 12))
        rows = []
        for _ in range(n):
            s = np.zeros(8, np.int32)
            s[_lib.SUM_RU] = max(0, base + int(rng.choice([0, 0, 0, 0, 1, -1, 3, 5])))
            lb, rb = int(rng.choice([0, 3, 6, 20, 40])), int(rng.choice([0, 3, 6, 20, 40]))
            s[_lib.SUM_LEFT_BP], s[_lib.SUM_RIGHT_BP] = lb, rb
            s[_lib.SUM_LEFT_MATCH] = int(lb * rng.choice([1.0, 0.96, 0.95, 0.94, 0.5]))
            s[_lib.SUM_RIGHT_MATCH] = int(rb * rng.choice([1.0, 0.96, 0.95, 0.94, 0.5]))
            s[_lib.SUM_PATH_LEN] = 200
            rows.append(s)
        groups.append(rows)
    groups += [[], [_spanning(0)] * 3, [_spanning(4)], [_spanning(7)] * 30 + [_spanning(8)] * 29]
    summ = np.array([s for grp in groups for s in grp], np.int32).reshape(-1, 8)
    off = np.concatenate([[0], np.cumsum([len(grp) for grp in groups])])
    for accuracy in (False, True):
        for haploid in (False, True):
            got = vntr_finder.find_repeat_counts_of_loci(summ, off, accuracy_filter=accuracy, is_haploid=haploid)
            for grp, r in zip(groups, got):
                with np.errstate(all="ignore"):
                    want = vntr_finder.find_repeat_count_from_selected_reads(grp, accuracy_filter=accuracy, is_haploid=haploid)
                assert r.copy_numbers == want.copy_numbers, (accuracy, haploid, [int(s[0]) for s in grp])
                assert r.maximum_likelihood == want.maximum_likelihood
                assert (r.recruited_reads_count, r.spanning_reads_count, r.flanking_reads_count) == \
                    (want.recruited_reads_count, want.spanning_reads_count, want.flanking_reads_count)


def test_argument_errors():
    import pytest
    with pytest.raises(_lib.EngineError):
        _lib.genotype_illumina(np.zeros((2, 8), np.int32), np.array([0, 3, 2]))


def test_genotype_observed_on_the_reference_goldens_and_the_python_mirror():
    """advntr_genotype_observed = the tail of get_dominant_copy_numbers_from_spanning_reads (vntr_finder.py:568-580): the
    reference's own genotype cases bit-equal; with the >= 3-reads filter against the Python mirror on random loci."""
    from collections import Counter
    g = load_golden("genotype_cases")
    for haploid in (False, True):
        cases = [c for c in g["cases"] if bool(c["haploid"]) == haploid]
        ru = np.array([r for c in cases for r in c["observed"]], np.int32)
        off = np.concatenate([[0], np.cumsum([len(c["observed"]) for c in cases])])
        geno, prob = _lib.genotype_observed(ru, off, is_haploid=haploid)
        for c, ab, p in zip(cases, geno.tolist(), prob.tolist()):
            assert (None if ab[0] < 0 else ab) == c["genotype"], c
            assert p == c["max_prob"], c
    rng = np.random.default_rng(5)
    groups = [[int(v) for v in np.maximum(0, rng.integers(0, 12) + rng.choice([0, 0, 0, 1, -1, 4], int(rng.integers(0, 30))))]
              for _ in range(400)] + [[], [0, 0, 0], [4], [7] * 30 + [8] * 29]
    ru = np.array([v for grp in groups for v in grp], np.int32)
    off = np.concatenate([[0], np.cumsum([len(grp) for grp in groups])])
    for accuracy in (False, True):
        for haploid in (False, True):
            for threads in (1, 4):
                geno, prob = _lib.genotype_observed(ru, off, accuracy, haploid, threads)
                for grp, ab, p in zip(groups, geno.tolist(), prob.tolist()):
                    if not grp:
                        assert ab == [-1, -1] and p == 0.0        # "There is no spanning read": (None, 0)
                        continue
                    obs = list(grp)
                    if accuracy:
                        obs = [k for k, c in Counter(obs).most_common() if c >= 3 for _ in range(c)]
                    with np.errstate(all="ignore"):
                        want, wp = vntr_finder.find_genotype_based_on_observed_repeats(obs, haploid)
                    assert (None if ab[0] < 0 else tuple(ab)) == want, (accuracy, haploid, grp)
                    assert p == wp
    import pytest
    with pytest.raises(_lib.EngineError):
        _lib.genotype_observed(np.zeros(2, np.int32), np.array([0, 3, 2]))
