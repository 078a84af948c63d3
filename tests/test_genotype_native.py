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
