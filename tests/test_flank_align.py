"""Flank alignment for long reads (csrc/flank_align.h) against its CPU restatement (oracle/flank_align_oracle.c).
PARITY UNPINNED with respect to biopython's pairwise2 (absent from the image): these tests pin the kernel on the
restatement -- full score matrix and explicit walk-back, a different formulation from the kernel's forward start
propagation -- and on hand-checked cases."""
import numpy as np
import pytest

from oracle import oracle as O


def test_restatement_hand_checked_cases():
    assert O.flank_align("TTTTACGTACGTTTTT", "ACGTACGT") == (8, 4, 11)
    assert O.flank_align("TTTTACGTTCGTTTTT", "ACGTACGT") == (6, 4, 11)            # one mismatch inside: 7 - 1
    assert O.flank_align("GGGG", "ACAC") == (0, -1, -1)
    assert O.flank_align("", "ACGT") == (0, -1, -1)
    # the flank occurs twice: the first alignment pairwise2 returns ends at the LAST best cell
    assert O.flank_align("ACGTACGTGGGGGACGTACGT", "ACGTACGT") == (8, 13, 20)
    # alignment starts inside the flank (flank index 3 > read index 0): begin = max of the two start indices
    assert O.flank_align("TACGT", "GGGTACGT") == (5, 3, 4)
    # one base missing from the read: 10 matches, one gap
    assert O.flank_align("CCCCACGTAGTACCCCC", "ACGTACGTAC")[0] == 8
    # N matches nothing
    assert O.flank_align("TTTTACGNACGTTTTT", "ACGTACGT")[0] == 6


def test_restatement_score_is_optimal_by_brute_force():
    """Property check for the unpinned row: on short inputs the restatement's score equals the best local-alignment score
    found by exhaustive search over all substring pairs (global alignment of every pair under +1 / -1 / gap -1), and the
    reported end is a cell where that score is reached."""
    rng = np.random.default_rng(31337)

    def global_score(a, b):
        prev = [-j for j in range(len(b) + 1)]
        for i in range(1, len(a) + 1):
            cur = [-i] + [0] * len(b)
            for j in range(1, len(b) + 1):
                cur[j] = max(prev[j - 1] + (1 if a[i - 1] == b[j - 1] and a[i - 1] != "N" else -1), prev[j] - 1, cur[j - 1] - 1)
            prev = cur
        return prev[len(b)]

    for _ in range(60):
        read = "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04]) for _ in range(int(rng.integers(1, 10))))
        flank = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(1, 8))))
        best, ends = 0, set()
        for i0 in range(len(read)):
            for i1 in range(i0 + 1, len(read) + 1):
                for j0 in range(len(flank)):
                    for j1 in range(j0 + 1, len(flank) + 1):
                        sc = global_score(read[i0:i1], flank[j0:j1])
                        if sc > best:
                            best, ends = sc, {i1 - 1}
                        elif sc == best and sc > 0:
                            ends.add(i1 - 1)
        score, begin, end = O.flank_align(read, flank)
        assert score == best, (read, flank, score, best)
        if best > 0:
            assert end in ends and 0 <= begin <= end + len(flank)
        else:
            assert (begin, end) == (-1, -1)


def _noisy(rng, s, rate):
    out = []
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append("ACGT"[int(rng.integers(0, 4))])
        if 2 * rate / 3 <= x < rate:
            ch = "ACGT"[int(rng.integers(0, 4))]
        out.append(ch)
    return "".join(out)


@pytest.mark.gpu
def test_kernel_equals_restatement_on_random_pairs():
    from advntr_amd import _lib, workloads
    rng = np.random.default_rng(2024)
    reads, flanks, pr, pf = [], [], [], []
    for f in range(12):
        flanks.append(workloads.rand_seq(rng, int(rng.choice([1, 7, 63, 64, 65, 100, 100, 100, 128]))))
    for r in range(60):
        n = int(rng.choice([0, 1, 5, 64, 200, 1500, 4000]))
        body = workloads.rand_seq(rng, n)
        if n >= 200:                                   # plant noisy copies of some flanks (once or twice)
            for _ in range(int(rng.integers(1, 4))):
                f = flanks[int(rng.integers(0, len(flanks)))]
                at = int(rng.integers(0, max(1, n - len(f))))
                c = _noisy(rng, f, float(rng.choice([0.0, 0.05, 0.15])))
                body = body[:at] + c + body[at + len(c):]
        if r % 9 == 0 and n > 10:
            body = body[:5] + "N" + body[6:]
        reads.append(body)
    for r in range(len(reads)):
        for f in range(len(flanks)):
            pr.append(r)
            pf.append(f)
    # low-complexity pairs provoke ties in score and in the walk-back
    reads += ["ACACACACACACACACACAC", "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", "ACGTACGTACGTACGTACGTACGT"]
    flanks += ["ACACAC", "AAAAAAAA", "ACGTACGT", "CACACACA"]
    for r in range(len(reads) - 3, len(reads)):
        for f in range(len(flanks) - 4, len(flanks)):
            pr.append(r)
            pf.append(f)
    score, begin, end, ms = _lib.flank_align(reads, flanks, pr, pf)
    for p in range(len(pr)):
        want = O.flank_align(reads[pr[p]], flanks[pf[p]])
        assert (int(score[p]), int(begin[p]), int(end[p])) == want, (p, pr[p], pf[p], len(reads[pr[p]]), len(flanks[pf[p]]))
    assert (score > 50).sum() > 20


@pytest.mark.gpu
def test_extract_spanning_reads():
    """Long reads that span, half-span or miss a VNTR, on either strand: the spanning ones come back trimmed to the
    flanks, as check_if_flanking_regions_align_to_str does (vntr_finder.py:324-365)."""
    from advntr_amd import settings, vntr_finder, workloads
    rng = np.random.default_rng(7)
    left, right, pattern = workloads.rand_seq(rng, 500), workloads.rand_seq(rng, 500), workloads.rand_seq(rng, 30)
    settings.MAX_ERROR_RATE = 0.3
    try:
        reads, truth = [], []
        for k in range(24):
            copies = int(rng.integers(3, 30))
            core = left[-100:] + pattern * copies + right[:100]
            kind = k % 4
            if kind == 0:                                   # spans, forward
                s = workloads.rand_seq(rng, 800) + _noisy(rng, left[:-100][-300:] + core + right[100:][:300], 0.1) + workloads.rand_seq(rng, 500)
            elif kind == 1:                                 # spans, reverse strand
                s = vntr_finder.reverse_complement(workloads.rand_seq(rng, 300) + _noisy(rng, core, 0.1) + workloads.rand_seq(rng, 900))
            elif kind == 2:                                 # only the left flank
                s = workloads.rand_seq(rng, 700) + _noisy(rng, left[-100:] + pattern * copies, 0.1)
            else:                                           # unrelated
                s = workloads.rand_seq(rng, 2500)
            reads.append(s)
            truth.append((kind, copies))
        spanning, lengths = vntr_finder.extract_spanning_reads(left, right, reads)
        # the same decisions from the CPU restatement, read by read and strand by strand
        want, want_len = [], []
        for idx, s in enumerate(reads):
            for rev, strand in ((False, s.upper()), (True, vntr_finder.reverse_complement(s.upper()))):
                ls, lb, _ = O.flank_align(strand, left[-100:])
                rs, rb, _ = O.flank_align(strand, right[:100])
                if ls >= 70 and rs >= 70 and rb >= lb:
                    want.append((strand[lb:rb + 100], idx, rev))
                    want_len.append(rb - (lb + 100))
        assert spanning == want and lengths == want_len
        found = {idx for _, idx, _ in spanning}
        planted = [k for k, (kind, _) in enumerate(truth) if kind in (0, 1)]
        assert found <= set(planted) and len(found) >= 8            # 10 % noise puts a few flanks under the 0.7 threshold
        for (seq, idx, rev), ln in zip(spanning, lengths):
            kind, copies = truth[idx]
            assert rev == (kind == 1)
            assert abs(ln - copies * 30) <= 0.25 * copies * 30 + 12            # the VNTR part, up to the planted noise
            assert abs(len(seq) - (ln + 200)) <= 1
    finally:
        settings.MAX_ERROR_RATE = 0.05


@pytest.mark.gpu
def test_cli_pacbio_from_whole_long_reads(tmp_path):
    """--pacbio --extract-spanning: whole long reads in, spanning reads found and trimmed on the GPU, RU genotype out."""
    import json
    import subprocess
    import sys
    from advntr_amd import workloads, vntr_finder
    from conftest import ROOT
    rng = np.random.default_rng(5150)
    left, right, pattern = workloads.rand_seq(rng, 300), workloads.rand_seq(rng, 300), workloads.rand_seq(rng, 25)
    reads = []
    for k in range(24):
        copies = 6 if k % 2 else 9
        s = workloads.rand_seq(rng, int(rng.integers(500, 2000))) + workloads.noisy_copy(rng, left + pattern * copies + right, 0.05) + \
            workloads.rand_seq(rng, int(rng.integers(500, 2000)))
        reads.append(s if k % 3 else vntr_finder.reverse_complement(s))
    reads += [workloads.rand_seq(rng, 3000) for _ in range(6)]
    loci = [{"id": 9, "left": left, "right": right, "pattern": pattern, "repeat_segments": [pattern], "scaled_score": None}]
    (tmp_path / "loci.json").write_text(json.dumps(loci))
    (tmp_path / "reads.fa").write_text("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(reads)))
    out = subprocess.run([sys.executable, "-m", "advntr_amd", "genotype", "--loci", str(tmp_path / "loci.json"), "--reads",
                          str(tmp_path / "reads.fa"), "--pacbio", "--extract-spanning"], cwd=ROOT, stdout=subprocess.PIPE,
                         check=True).stdout.decode()
    assert out == "9\n6/9\n"


@pytest.mark.gpu
def test_best_cells_tying_across_the_two_flank_chunks():
    """A low-complexity flank of more than 64 bases whose best cells tie in one row of the read, some in the first 64 columns
    and some beyond: the last one in row-major order ends the alignment (found by scripts/fuzz_flank_align.py -- the
    per-lane step of the reduction used to keep the first chunk's cell)."""
    from advntr_amd import _lib
    read = "ATGTGGTTGTGTGTATTTACCCTGGGACTAAAAACCCGGCTTTCTCCAGATCAAAAGAAGGATT"
    flank = "CTAG" * 18
    reads = [read, read + "ACGTTGCA", "GG" + read]
    flanks = [flank[:n] for n in (60, 64, 65, 67, 70, 71)]
    pr = [r for r in range(len(reads)) for _ in flanks]
    pf = [f for _ in reads for f in range(len(flanks))]
    score, begin, end, _ = _lib.flank_align(reads, flanks, pr, pf)
    for p in range(len(pr)):
        assert (int(score[p]), int(begin[p]), int(end[p])) == O.flank_align(reads[pr[p]], flanks[pf[p]]), (pr[p], len(flanks[pf[p]]))
    assert O.flank_align(read, flank[:71]) == (3, 55, 52)


@pytest.mark.gpu
def test_kernel_equals_restatement_at_baseline_config_4_read_lengths():
    """BASELINE config 4 names reads of 5-15 kb: PacBio-like reads of 5 000-15 000 bases (and one of 20 000) with noisy
    copies of the 100-base flanks planted once or several times, either strand, against the restatement
    (oracle/flank_align_oracle.c: score, begin, end equal).  Parity with biopython's pairwise2 stays unpinned (it is not in
    the image); this covers the kernel's long-read path -- windows of the read past 64 K steps never occur, 15 k rows do."""
    from advntr_amd import _lib, vntr_finder, workloads
    rng = np.random.default_rng(415)
    flanks = [workloads.rand_seq(rng, 100) for _ in range(6)] + [workloads.rand_seq(rng, 37), "ACGT" * 25]
    reads = []
    for n in [5000, 6111, 7500, 9000, 10240, 12345, 14999, 15000, 20000]:
        body = workloads.rand_seq(rng, n)
        for _ in range(int(rng.integers(2, 6))):
            f = flanks[int(rng.integers(0, len(flanks)))]
            c = _noisy(rng, f, float(rng.choice([0.0, 0.1, 0.2, 0.3])))
            if rng.random() < 0.3:
                c = vntr_finder.reverse_complement(c)
            at = int(rng.integers(0, n - len(c)))
            body = body[:at] + c + body[at + len(c):]
        # a flank cut off by either end of the read
        body = flanks[0][40:] + body[60:n - 50] + flanks[1][:50]
        reads.append(body)
    pr, pf = [], []
    for r in range(len(reads)):
        for f in range(len(flanks)):
            pr.append(r)
            pf.append(f)
    score, begin, end, _ = _lib.flank_align(reads, flanks, pr, pf)
    for p in range(len(pr)):
        want = O.flank_align(reads[pr[p]], flanks[pf[p]])
        assert (int(score[p]), int(begin[p]), int(end[p])) == want, (p, len(reads[pr[p]]), pf[p])
    assert (score >= 60).sum() >= 3                     # planted copies are found


@pytest.mark.gpu
def test_reverse_strands_made_on_the_device_equal_explicit_reverse_complements():
    """pair_read = n_reads + r stands for the reverse complement of read r (check_if_pacbio_read_spans_vntr tests both strands,
    vntr_finder.py:367-371): same (score, begin, end) as the alignment against the reverse-complemented string, N included."""
    from advntr_amd import _lib, vntr_finder, workloads
    rng = np.random.default_rng(99)
    flanks = [workloads.rand_seq(rng, int(k)) for k in (100, 100, 64, 65, 17, 128)]
    reads = []
    for n in (0, 1, 63, 64, 65, 300, 2000, 7001):
        body = workloads.rand_seq(rng, n)
        if n >= 300:
            for f in flanks[:3]:
                c = vntr_finder.reverse_complement(_noisy(rng, f, 0.1))
                at = int(rng.integers(0, n - len(c)))
                body = body[:at] + c + body[at + len(c):]
            body = body[:50] + "N" + body[51:]
        reads.append(body)
    n = len(reads)
    pr = np.repeat(np.arange(n, 2 * n, dtype=np.int32), len(flanks))
    pf = np.tile(np.arange(len(flanks), dtype=np.int32), n)
    score, begin, end, _ = _lib.flank_align(reads, flanks, pr, pf)
    explicit = _lib.flank_align([vntr_finder.reverse_complement(r) if "N" not in r else
                                 r.translate(str.maketrans("ACGTN", "TGCAN"))[::-1] for r in reads], flanks, pr - n, pf)
    assert np.array_equal(score, explicit[0]) and np.array_equal(begin, explicit[1]) and np.array_equal(end, explicit[2])
    for p in range(0, len(pr), 5):
        rc = reads[pr[p] - n].translate(str.maketrans("ACGTN", "TGCAN"))[::-1]
        assert (int(score[p]), int(begin[p]), int(end[p])) == O.flank_align(rc, flanks[pf[p]])
    assert (score >= 50).sum() >= 3
    with pytest.raises(Exception):
        _lib.flank_align(reads, flanks, np.array([2 * n], np.int32), np.array([0], np.int32))


@pytest.mark.gpu
def test_pacbio_loci_from_whole_reads_equal_the_per_locus_route():
    """vntr_finder.genotype_pacbio_loci (many loci, one flank-alignment call and one scoring batch per piece, stages
    overlapped) against find_repeat_count_from_pacbio_reads done locus by locus with extract_spanning_reads +
    get_dominant_copy_numbers_from_spanning_reads (vntr_finder.py:652-665): same spanning reads, same genotypes, bit-equal
    probabilities; and the planted copy numbers come out."""
    from advntr_amd import settings, vntr_finder, workloads
    loci, read_lists = workloads.make_pacbio_whole_reads(14, seed=77, n_reads=10, min_len=1500, max_len=4000, workers=1)
    read_lists[3] = []                                              # a locus without candidates
    read_lists[5] = [workloads.rand_seq(np.random.default_rng(1), 2000)]     # a locus whose only candidate does not span
    settings.MAX_ERROR_RATE = 0.3
    try:
        for accuracy in (False, True):
            T = {}
            got = vntr_finder.genotype_pacbio_loci(loci, read_lists, accuracy_filter=accuracy, chunks=4, timings=T)
            assert T["total"] > 0 and len(got) == len(loci)
            n_called = 0
            for (left, right, segments, pattern), reads, g in zip(loci, read_lists, got):
                spanning, _ = vntr_finder.extract_spanning_reads(left, right, reads)
                want, prob = vntr_finder.get_dominant_copy_numbers_from_spanning_reads(
                    left, right, segments, pattern, [s[0] for s in spanning], accuracy_filter=accuracy)
                assert g.copy_numbers == want and g.maximum_likelihood == prob
                assert g.recruited_reads_count == g.spanning_reads_count == len(spanning) and g.flanking_reads_count == 0
                n_called += want is not None
            assert got[3].copy_numbers is None and got[3].maximum_likelihood == 0
            assert got[5].copy_numbers is None and got[5].spanning_reads_count == 0
            assert n_called >= (6 if accuracy else 10)
    finally:
        settings.MAX_ERROR_RATE = 0.05


@pytest.mark.gpu
def test_a_read_list_shared_by_many_loci_is_uploaded_once(monkeypatch):
    """`python -m advntr_amd genotype --pacbio --extract-spanning` hands the SAME read list to every locus (every read is a
    candidate of every locus): extract_spanning_reads_multi uploads each distinct read once and lets the pairs of every locus
    index that copy -- same spanning reads per locus as with private copies of the list, and the flank-alignment call sees
    len(reads) strings, not len(reads) x loci; genotype_pacbio_loci cuts its pieces by the number of alignments."""
    from advntr_amd import _lib, settings, vntr_finder, workloads
    loci, read_lists = workloads.make_pacbio_whole_reads(6, seed=78, n_reads=5, min_len=1500, max_len=3000, workers=1)
    everything = [r for rl in read_lists for r in rl]                  # 30 reads, each a candidate of all 6 loci
    pairs = [(l[0], l[1]) for l in loci]
    seen = []
    real = _lib.flank_align

    def spy(reads, flanks, pair_read, pair_flank, **kw):
        seen.append((len(reads), len(pair_read)))
        return real(reads, flanks, pair_read, pair_flank, **kw)
    monkeypatch.setattr(_lib, "flank_align", spy)
    settings.MAX_ERROR_RATE = 0.3
    try:
        shared = vntr_finder.extract_spanning_reads_multi(pairs, [everything] * 6)
        assert seen[-1] == (30, 4 * 30 * 6)
        private = vntr_finder.extract_spanning_reads_multi(pairs, [[str(s[:1]) + s[1:] for s in everything] for _ in range(6)])
        assert seen[-1] == (180, 4 * 30 * 6)
        assert shared == private
        for i, (sp, lengths) in enumerate(shared):
            assert {k for _, k, _ in sp} <= set(range(5 * i, 5 * i + 5))       # its own reads, by position in the shared list
            assert len(sp) >= 3 and len(lengths) == len(sp)
        # pieces by alignments: 2 loci x 3 000 000 shared "reads" would be 24 M alignments -> at least 6 pieces
        calls = []
        monkeypatch.setattr(vntr_finder, "_spanning_prepare", lambda fp, rl, size=100: calls.append(len(fp)))
        big = [None] * 3000000
        res = vntr_finder.genotype_pacbio_loci(loci[:2] * 6, [big] * 12, chunks=1)
        assert len(res) == 12 and len(calls) >= 9 and all(g.copy_numbers is None for g in res)
    finally:
        settings.MAX_ERROR_RATE = 0.05
