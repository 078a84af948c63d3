"""filtering.select_candidates -- the bookkeeping of the reference's filter binary (filtering/main.cc:286-331) on arrays, which the
one-timeline genotype flow uses instead of printing and re-parsing text -- against the same bookkeeping done record by record
(KeywordFilter._select_records, whose output is pinned byte for byte on the reference binary by tests/test_gpu_filter.py):
random hit records, random names (fixed and ragged widths), small caps so that the intake cap (3 x max_reads + 1 reads in file
order) and the output cap (max_reads + 1 names in descending (count, name) order) both bite.  No GPU."""
import numpy as np

from advntr_amd import filtering


class _NoDevice(filtering.KeywordFilter):
    def __init__(self, ids):                                   # the bookkeeping only: no keyword tables, no device
        self.vntr_ids = ids
        self.uniq_ids = sorted(set(ids))


def _lists_of(text):
    out = {}
    for line in text.split("\n"):
        p = line.split()
        if len(p) >= 2 and p[0].isdigit() and p[1].isdigit():
            out[int(p[0])] = sorted(p[2:])
    return out


def test_select_candidates_equals_the_record_by_record_bookkeeping():
    rng = np.random.default_rng(3)
    capped = 0
    for trial in range(300):
        nv = int(rng.integers(1, 6))
        ids = [int(x) for x in rng.permutation(np.arange(10, 10 + nv))]
        n_reads = int(rng.integers(1, 80))
        if trial % 2 == 0:
            names = ["r%04d" % i for i in range(n_reads)]
        else:
            names = ["q%d_%d" % (int(rng.integers(0, 10 ** int(rng.integers(1, 5)))), i) for i in range(n_reads)]
        names = [names[i] for i in rng.permutation(n_reads)]
        recs = [(r, vi, int(rng.integers(1, 9))) for r in range(n_reads) for vi in range(nv) if rng.random() < 0.5]
        f = _NoDevice(ids)
        max_reads, min_matches = int(rng.integers(1, 4)), int(rng.integers(1, 6))
        want = _lists_of(f._select_records([(r, f.uniq_ids[vi], c) for r, vi, c in recs], lambda r: names[r], lambda r: "ACGT",
                                           min_matches, max_reads))
        R, V, C = (np.array(x, np.int64) for x in zip(*recs)) if recs else (np.zeros(0, np.int64),) * 3
        keys = lambda idx: np.array([names[i].encode() for i in idx], dtype=bytes) if len(idx) else np.zeros(0, "S1")
        pr, pv = filtering.select_candidates(R, V, C, keys, min_matches, max_reads)
        got = {vid: [] for vid in ids}
        for r, vi in zip(pr.tolist(), pv.tolist()):
            got[f.uniq_ids[vi]].append(names[r])
        for vid in ids:
            assert got[vid] == want[vid], (trial, vid, min_matches, max_reads)          # same reads, ascending name
            capped += len(want[vid]) == max_reads + 1
    assert capped > 50                                        # the caps did bite


def test_name_keys_order_like_the_strings():
    text = b">r0000012\nACGT\n>r0000003\nACGT\n>zz\nAC\n"
    fixed = filtering._name_keys(text, np.array([1, 16]), np.array([9, 24]))
    assert fixed.tolist() == [b"r0000012", b"r0000003"] and fixed.dtype.kind == "S"
    ragged = filtering._name_keys(text, np.array([1, 16, 31]), np.array([9, 24, 33]))
    assert ragged.tolist() == [b"r0000012", b"r0000003", b"zz"]
    assert filtering._name_keys(text, np.zeros(0, np.int64), np.zeros(0, np.int64)).shape == (0,)
