"""Batch driver for the callers of the scoring path -- host-side mirror of the pieces of
/root/reference/advntr/vntr_finder.py that decide WHICH (read, strand, locus) calls are scored and what
is kept:

  get_copies_for_hmm                       vntr_finder.py:98-99
  get_min_score_to_select_a_read           vntr_finder.py:174-177
  recruit_read                             vntr_finder.py:179-190
  process_unmapped_read (fwd + revcomp)    vntr_finder.py:235-254   -> score_reads(..., compute_reverse=True)
  find_genotype_based_on_observed_repeats  vntr_finder.py:485-532   (+ get_conditional_likelihood :473-483)
  read_flanks_repeats_with_confidence      vntr_finder.py:311-322
  find_repeat_count_from_alignment_file    vntr_finder.py:807-887   -> find_repeat_count_from_selected_reads (after selection)
  build_vntr_matcher_hmm / get_dominant_copy_numbers_from_spanning_reads (PacBio)   vntr_finder.py:108-115, 534-585

The reference loops over reads in Python and calls hmm.viterbi twice per unmapped read; here the whole
locus batch (both strands) goes to the GPU in one advntr_viterbi_batch call and the keep/discard rule is
applied to the 8-int summaries the kernel returns (no Viterbi path leaves the device).
"""
import numpy as np

from . import _lib
from .pomegranate import device_models
from .hmm_utils import flanking_rate_from_counts

_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def reverse_complement(seq):
    return seq.encode("ascii").translate(_COMP)[::-1].decode("ascii")


def get_copies_for_hmm(read_length, pattern_length):
    return int(round(float(read_length) / pattern_length + 0.5))


def get_min_score_to_select_a_read(scaled_score, read_length):
    if scaled_score is None or scaled_score == 0:
        return None
    return scaled_score * read_length


def recruit_read(logp, summary, min_score_to_count_read, read_length):
    """summary = the 8 int32 of ADVNTR_SUM_* for this read."""
    rate = flanking_rate_from_counts(int(summary[_lib.SUM_LEFT_MATCH]), int(summary[_lib.SUM_LEFT_BP]),
                                     int(summary[_lib.SUM_RIGHT_MATCH]), int(summary[_lib.SUM_RIGHT_BP]))
    if rate < 0.90:
        return False
    if min_score_to_count_read is not None and logp > min_score_to_count_read:
        return True
    matches = int(summary[_lib.SUM_MATCHES])
    if min_score_to_count_read is None and matches >= 0.9 * read_length and logp > -read_length:
        return True
    return False


class ScoredRead(object):
    __slots__ = ("sequence", "logp", "summary", "reversed", "recruited")

    def __init__(self, sequence, logp, summary, reversed_, recruited):
        self.sequence, self.logp, self.summary, self.reversed, self.recruited = \
            sequence, logp, summary, reversed_, recruited

    @property
    def repeats(self):
        return int(self.summary[_lib.SUM_RU])

    @property
    def repeat_bp(self):
        return int(self.summary[_lib.SUM_REPEAT_BP])


def score_reads(model, sequences, scaled_score=None, compute_reverse=True):
    """Score reads against one locus model; with compute_reverse both strands are scored and the reverse
    strand replaces the forward one iff logp < rev_logp (vntr_finder.py:242-246).  Reads holding 'N' are
    skipped before scoring, as the reference does (vntr_finder.py:237).  Returns a list of ScoredRead
    (None for skipped reads)."""
    return score_reads_multi([model], [sequences], [scaled_score], compute_reverse)[0]


def score_reads_multi(models, read_lists, scaled_scores=None, compute_reverse=True):
    """score_reads for many loci in ONE engine call: models[i] scores read_lists[i].  The reference walks loci and
    reads in nested Python loops (genome_analyzer.py:280, vntr_finder.py:235-254); here the whole read x strand x
    locus batch is a single advntr_viterbi_batch launch set, and the keep/discard rule runs on the returned
    summaries.  Returns one list of ScoredRead (None for reads holding 'N') per locus."""
    n_loci = len(models)
    scaled_scores = list(scaled_scores) if scaled_scores is not None else [None] * n_loci
    batch, which, layout = [], [], []
    for i, seqs in enumerate(read_lists):
        keep = [j for j, s in enumerate(seqs) if s.count('N') <= 0]
        fwd = [seqs[j].upper() for j in keep]
        start = len(batch)
        batch += fwd
        if compute_reverse:
            batch += [reverse_complement(s) for s in fwd]
        which += [i] * (len(batch) - start)
        layout.append((keep, start, len(fwd)))
    out = [[None] * len(seqs) for seqs in read_lists]
    if not batch:
        return out
    bases, off = _lib.encode_reads(batch)
    logp, summ, _ = _lib.viterbi_batch(device_models(models), bases, off, np.asarray(which, np.int32),
                                       want_paths=False, want_summary=True)
    for i, (keep, start, nf) in enumerate(layout):
        for j, pos in enumerate(keep):
            a = start + j
            seq, lp, sm, rev = batch[a], float(logp[a]), summ[a], False
            if compute_reverse and lp < float(logp[a + nf]):
                b = a + nf
                seq, lp, sm, rev = batch[b], float(logp[b]), summ[b], True
            ok = False
            if sm[_lib.SUM_PATH_LEN] > 2:
                ok = recruit_read(lp, sm, get_min_score_to_select_a_read(scaled_scores[i], len(seq)), len(seq))
            out[i][pos] = ScoredRead(seq, lp, sm, rev, ok)
    return out


def recruit_mask(logp, summary, read_lengths, min_scores):
    """recruit_read (vntr_finder.py:179-190) on arrays: logp float64[n], summary int32[n][8], read_lengths int[n],
    min_scores float64[n] with NaN where the locus has no trained score.  Reads without a path are not recruited."""
    logp = np.asarray(logp, np.float64)
    s = np.asarray(summary).reshape(-1, _lib.SUMMARY_INTS)
    n = np.asarray(read_lengths, np.float64)
    ms = np.asarray(min_scores, np.float64)
    lb, rb = s[:, _lib.SUM_LEFT_BP].astype(np.float64), s[:, _lib.SUM_RIGHT_BP].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        left = np.where(lb != 0, s[:, _lib.SUM_LEFT_MATCH] / lb, 1.0)
        right = np.where(rb != 0, s[:, _lib.SUM_RIGHT_MATCH] / rb, 1.0)
    rate_ok = np.minimum(left, right) >= 0.90
    has_score = ~np.isnan(ms)
    with np.errstate(invalid="ignore"):
        by_score = has_score & (logp > ms)
    by_default = ~has_score & (s[:, _lib.SUM_MATCHES] >= 0.9 * n) & (logp > -n)
    return rate_ok & (by_score | by_default) & (s[:, _lib.SUM_PATH_LEN] > 2)


def _empty_scores():
    return dict(locus=np.zeros(0, np.int32), index=np.zeros(0, np.int32), logp=np.zeros(0),
                summary=np.zeros((0, _lib.SUMMARY_INTS), np.int32), reversed=np.zeros(0, bool),
                recruited=np.zeros(0, bool), length=np.zeros(0, np.int64))


def _prepare_reads(read_lists, threads=0):
    """The host half of score_reads_arrays: one code buffer for every read of every locus (case folding, encoding and the
    test for symbols outside ACGT on host threads in the library, advntr_encode_ascii).  Reads holding 'N' are dropped as
    the reference does (vntr_finder.py:237); any other foreign symbol raises, as the reference's viterbi does
    (hmm.pyx:72,79).  Returns None when nothing is left to score."""
    import itertools
    n_loci = len(read_lists)
    counts = np.fromiter((len(seqs) for seqs in read_lists), dtype=np.int64, count=n_loci)
    flat = list(itertools.chain.from_iterable(read_lists))
    n_all = len(flat)
    if n_all == 0:
        return None
    codes, all_off, bad = _lib.encode_ascii(flat, threads)
    if np.any(bad == 2):
        raise ValueError("Symbol is not defined in a distribution (read %d holds a symbol outside ACGTN)" % int(np.argmax(bad == 2)))
    all_len = np.diff(all_off)
    keep = bad == 0
    locus_all = np.repeat(np.arange(n_loci, dtype=np.int32), counts)
    index_all = (np.arange(n_all, dtype=np.int64) - np.repeat(np.cumsum(counts) - counts, counts)).astype(np.int32)
    nf = int(keep.sum())
    if nf == 0:
        return None
    if nf == n_all:
        return dict(locus=locus_all, index=index_all, lens=all_len, bases=codes, off=all_off)
    lens = all_len[keep]
    off = np.zeros(nf + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    return dict(locus=locus_all[keep], index=index_all[keep], lens=lens, bases=codes[np.repeat(keep, all_len)], off=off)


def _score_prepared(models, prep, scaled_scores=None, compute_reverse=True):
    """The device half: both strands in one engine batch -- the reverse complements are made on the device
    (ADVNTR_FLAG_BOTH_STRANDS), call nf + i = reverse complement of read i --, the strand choice and the recruit rule."""
    if prep is None:
        return _empty_scores()
    n_loci = len(models)
    scaled = [np.nan if (s is None or s == 0) else float(s) for s in (scaled_scores or [None] * n_loci)]
    lens, nf = prep["lens"], len(prep["lens"])
    logp, summ, _ = _lib.viterbi_batch(device_models(models), prep["bases"], prep["off"], prep["locus"], want_paths=False,
                                       want_summary=True, flags=_lib.FLAG_BOTH_STRANDS if compute_reverse else 0)
    if compute_reverse:
        rlogp, rsumm = logp[nf:], summ[nf:]
        use_rev = logp[:nf] < rlogp
        logp = np.where(use_rev, rlogp, logp[:nf])
        summ = np.where(use_rev[:, None], rsumm, summ[:nf])
    else:
        use_rev = np.zeros(nf, bool)
    min_scores = np.asarray(scaled, np.float64)[prep["locus"]] * lens
    return dict(locus=prep["locus"], index=prep["index"], logp=logp, summary=summ, reversed=use_rev, length=lens,
                recruited=recruit_mask(logp, summ, lens, min_scores))


def _select_prepared(models, prep, scaled_scores=None, compute_reverse=True, min_repeat_bp=2):
    """_score_prepared followed by the selection of process_unmapped_read (vntr_finder.py:246-254: recruit_read, then more
    than min_repeat_bp repeat bases), with strand choice, recruit rule and selection applied ON THE DEVICE
    (advntr_batch_recruit): only the selected reads' records come back, in read order.  Returns (position in the prepared
    read list, locus, summary[8], reversed) of the selected reads -- what _score_prepared + recruit_mask select."""
    if prep is None:
        return (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, _lib.SUMMARY_INTS), np.int32), np.zeros(0, bool))
    batch = _lib.DeviceBatch(device_models(models), prep["bases"], prep["off"], prep["locus"],
                             flags=_lib.FLAG_BOTH_STRANDS if compute_reverse else 0)
    try:
        batch.run()
        index, _, summ, rev = batch.recruit(scaled_scores, min_repeat_bp)
    finally:
        batch.close()
    return index, prep["locus"][index], summ, rev


def score_reads_arrays(models, read_lists, scaled_scores=None, compute_reverse=True):
    """score_reads_multi without a Python object per read, for genome-scale runs: returns a dict of arrays over all
    kept reads (reads holding 'N' are dropped): locus, index (position in read_lists[locus]), logp, summary,
    reversed, recruited, length -- the chosen strand per read (reverse iff logp < rev_logp, vntr_finder.py:242-246)."""
    return _score_prepared(models, _prepare_reads(read_lists), scaled_scores, compute_reverse)


def _genotypes_from_scores(res, n_loci, accuracy_filter, is_haploid, threads):
    keep = res["recruited"] & (res["summary"][:, _lib.SUM_REPEAT_BP] > 2)
    locus = res["locus"][keep]                       # ascending already: reads are laid out locus by locus
    summ = res["summary"][keep]
    if len(locus) and np.any(np.diff(locus) < 0):
        order = np.argsort(locus, kind="stable")
        locus, summ = locus[order], summ[order]
    bounds = np.searchsorted(locus, np.arange(n_loci + 1)).astype(np.int64)
    return find_repeat_counts_of_loci(summ, bounds, accuracy_filter, is_haploid, threads=threads)


def genotype_loci(models, read_lists, scaled_scores=None, accuracy_filter=False, is_haploid=False, compute_reverse=True,
                  threads=0):
    """The Illumina path after candidate selection, for many loci at once and without a Python object per read: both
    strands of every candidate read scored in one engine batch (process_unmapped_read, vntr_finder.py:235-254), the
    recruit rule on the summary records (:179-190), reads with more than two repeat bases selected (:251), and the
    per-locus aggregation + maximum-likelihood genotype (:807-887, :473-532) in the library's host threads
    (advntr_genotype_illumina).  Returns one GenotypeResult per locus."""
    res = score_reads_arrays(models, read_lists, scaled_scores, compute_reverse)
    return _genotypes_from_scores(res, len(models), accuracy_filter, is_haploid, threads)


def get_conditional_likelihood(ck, ci, cj, r, r_e):
    if ck == ci == cj:
        return 1 - r
    if cj == 0:
        return 0.5 * (1 - r)
    if ck == ci:
        return 0.5 * ((1 - r) + r_e ** abs(ck - cj))
    if ck == cj:
        return 0.5 * ((1 - r) + r_e ** abs(ck - ci))
    return 0.5 * (r_e ** abs(ck - ci) + r_e ** abs(ck - cj))


def find_genotype_based_on_observed_repeats(observed_copy_numbers, is_haploid=False):
    """Maximum-likelihood diploid (or haploid) RU genotype from the per-read RU counts; returns
    (genotype tuple | None, max_prob) like vntr_finder.py:532."""
    counts = {}
    for cn in observed_copy_numbers:
        counts[cn] = counts.get(cn, 0) + 1
    if len(counts) < 2:
        priors = 0.5
        counts[0] = 1
    else:
        priors = 1.0 / (len(counts) * (len(counts) - 1) / 2)
    ranked = sorted(counts.items(), key=lambda kv: kv[1], reverse=True)
    r = 0.03
    r_e = r / (2 + r)
    terms = {}
    for ck, occ in ranked:
        if ck == 0:
            continue
        for i, (ci, _) in enumerate(ranked):
            for j in range(i, len(ranked)):
                if is_haploid and i != j:
                    continue
                cj = ranked[j][0]
                terms.setdefault((ci, cj), []).append(get_conditional_likelihood(ck, ci, cj, r, r_e) ** occ)
    posteriors = {key: np.prod(np.array(vals)) * priors for key, vals in terms.items()}
    total = sum(posteriors.values())
    max_prob, result = 1e-20, None
    for key, value in posteriors.items():
        if value / total > max_prob:
            max_prob = value / total
            result = key
    return result, max_prob


class GenotypeResult(object):
    """Same fields as the reference's GenotypeResult (vntr_finder.py:28-34)."""

    def __init__(self, copy_numbers, recruited_reads_count, spanning_reads_count, flanking_reads_count, max_likelihood):
        self.copy_numbers = copy_numbers
        self.recruited_reads_count = recruited_reads_count
        self.spanning_reads_count = spanning_reads_count
        self.flanking_reads_count = flanking_reads_count
        self.maximum_likelihood = max_likelihood


def read_flanks_repeats_with_confidence(summary, minimum_left_flanking_size=5, minimum_right_flanking_size=5):
    """vntr_finder.py:311-322 on the kernel's summary: flank match rate >= 0.95 and more than the minimum number of
    bases on either flank => the read spans the VNTR."""
    rate = flanking_rate_from_counts(int(summary[_lib.SUM_LEFT_MATCH]), int(summary[_lib.SUM_LEFT_BP]),
                                     int(summary[_lib.SUM_RIGHT_MATCH]), int(summary[_lib.SUM_RIGHT_BP]))
    if rate < 0.95:
        return False
    return bool(summary[_lib.SUM_LEFT_BP] > minimum_left_flanking_size and
                summary[_lib.SUM_RIGHT_BP] > minimum_right_flanking_size)


def find_repeat_count_from_selected_reads(summaries, accuracy_filter=False, average_coverage=None, is_haploid=False,
                                          minimum_left_flanking_size=5, minimum_right_flanking_size=5):
    """The Illumina aggregation of find_repeat_count_from_alignment_file after read selection
    (vntr_finder.py:807-887): RU counts of spanning reads, plus the largest RU count of flanking reads when at least
    five of them agree on it and it is not below the largest spanning count; with the accuracy filter only RU counts
    supported by >= 3 spanning reads survive and flanking reads are ignored; optional coverage-based estimate.
    `summaries` = the 8-int records of the selected (recruited) reads."""
    from collections import Counter
    covered, flanking = [], []
    for s in summaries:
        repeats = int(s[_lib.SUM_RU])
        if read_flanks_repeats_with_confidence(s, minimum_left_flanking_size, minimum_right_flanking_size):
            covered.append(repeats)
        elif not accuracy_filter:
            flanking.append(repeats)
    flanking = sorted(flanking)
    min_valid_flanked = max(covered) if covered else 0
    max_flanking = [r for r in flanking if r == max(flanking) and r >= min_valid_flanked]
    if len(max_flanking) < 5:
        max_flanking = []
    if accuracy_filter:
        modified = []
        for key, count in Counter(covered).most_common():
            if count >= 3:
                modified.extend([key] * count)
        covered = modified
        max_flanking = []
    genotype, max_prob = find_genotype_based_on_observed_repeats(covered + max_flanking, is_haploid)
    if average_coverage is None or average_coverage is False or average_coverage == 0:
        return GenotypeResult(genotype, len(summaries), len(covered), len(flanking), max_prob)
    occurrences = sum(flanking) + sum(covered)
    haplotypes = 1 if is_haploid else 2
    estimate = [int(occurrences / (float(average_coverage) * haplotypes))] * 2
    return GenotypeResult(estimate, len(summaries), len(covered), len(flanking), 0)


def find_repeat_counts_of_loci(summaries, locus_off, accuracy_filter=False, is_haploid=False, minimum_left_flanking_size=5,
                               minimum_right_flanking_size=5, threads=0):
    """find_repeat_count_from_selected_reads for many loci at once (advntr_genotype_illumina: the same arithmetic in C++ on
    host threads): summaries = the 8-int records of the selected reads grouped by locus, locus i = rows
    locus_off[i]:locus_off[i+1].  Returns a list of GenotypeResult (copy_numbers None where the reference returns None)."""
    geno, prob, counts = _lib.genotype_illumina(summaries, locus_off, accuracy_filter, is_haploid,
                                                minimum_left_flanking_size, minimum_right_flanking_size, threads)
    out = []
    for (a, b), p, (rec, span, flank) in zip(geno.tolist(), prob.tolist(), counts.tolist()):
        out.append(GenotypeResult(None if a < 0 else (a, b), rec, span, flank, p))
    return out


def build_vntr_matcher_hmm(left_flanking_region, right_flanking_region, repeat_segments, copies, flanking_region_size=100):
    """vntr_finder.py:108-115: the read-matcher model over the last/first `flanking_region_size` flank bases."""
    from .hmm_utils import get_read_matcher_model
    return get_read_matcher_model(left_flanking_region[-flanking_region_size:],
                                  right_flanking_region[:flanking_region_size], repeat_segments, copies)


def pacbio_max_copies(spanning_read_lengths, pattern_length):
    """vntr_finder.py:538-543: copies of the model = round((longest trimmed read - 100) / len(pattern))."""
    max_length = 0
    for n in spanning_read_lengths:
        if n - 100 > max_length:
            max_length = n - 100
    return int(round(max_length / float(pattern_length)))


def get_dominant_copy_numbers_from_spanning_reads(left_flanking_region, right_flanking_region, repeat_segments, pattern,
                                                  spanning_reads, accuracy_filter=False, is_haploid=False):
    """PacBio RU-count genotyping (vntr_finder.py:534-585): one model sized for the longest spanning read, every
    spanning read scored on the forward strand (one batch on the GPU, row-tiled kernel), RU count per read, optional
    >= 3-reads support filter, maximum-likelihood genotype.  spanning_reads = trimmed read sequences."""
    from collections import Counter
    from . import settings
    if len(spanning_reads) < 1:
        return None, 0
    max_copies = pacbio_max_copies([len(s) for s in spanning_reads], len(pattern))
    model = build_vntr_matcher_hmm(left_flanking_region, right_flanking_region, repeat_segments, max_copies)
    _, summ, _ = model.viterbi_batch(list(spanning_reads), want_paths=False, want_summary=True)
    observed = [int(s[_lib.SUM_RU]) for s in summ]
    if accuracy_filter:
        modified = []
        for key, count in Counter(observed).most_common():
            if count >= 3:                                  # settings.ACCURACY_FILTER_SR_MIN_SUPPORT
                modified.extend([key] * count)
        observed = modified
    return find_genotype_based_on_observed_repeats(observed, is_haploid)


def get_vntr_matcher_hmm(reference_vntr, read_length):
    """vntr_finder.py:116-138: flanks of read_length bases, copies for that length.  With settings.USE_TRAINED_HMMS the
    model is loaded from / stored to `<TRAINED_HMMS_DIR><id>_<read_length>.json` in the reference's JSON format (a loaded
    model is baked with merging, as in the reference: hmm.pyx:3143)."""
    import os
    from . import settings
    from .pomegranate import HiddenMarkovModel
    stored = None
    if getattr(settings, "USE_TRAINED_HMMS", False):
        stored = settings.TRAINED_HMMS_DIR + str(reference_vntr.id) + '_' + str(read_length) + '.json'
        if os.path.isfile(stored):
            return HiddenMarkovModel.from_json(stored)
    copies = get_copies_for_hmm(read_length, len(reference_vntr.pattern))
    model = build_vntr_matcher_hmm(reference_vntr.left_flanking_region, reference_vntr.right_flanking_region,
                                   reference_vntr.get_repeat_segments(), copies, flanking_region_size=read_length)
    if stored is not None:
        with open(stored, 'w') as outfile:
            outfile.write(model.to_json())
    return model


# ------------------------------------------------------------------------------------------------
# PacBio spanning-read extraction (vntr_finder.py:324-371): which long reads cover the whole VNTR, and where
# ------------------------------------------------------------------------------------------------
def extract_spanning_reads(left_flanking_region, right_flanking_region, reads, flanking_region_size=100):
    """check_if_pacbio_read_spans_vntr for a batch of long reads: both strands of every read are tested with
    check_if_flanking_regions_align_to_str -- local alignment (1, -1, -1, -1) of the last `flanking_region_size` bases
    of the left flank and the first of the right flank; both must reach len(flank) * (1 - MAX_ERROR_RATE) and the right
    one must not begin before the left one.  The four alignments per read run as one advntr_flank_align call (the
    reference calls Bio.pairwise2 once per alignment, in Python).  Returns (spanning, length_distribution):
    spanning = [(trimmed sequence = read[left_begin : right_begin + flank size], read index, is_reverse_strand)],
    length_distribution = [right_begin - (left_begin + flank size)].  PARITY UNPINNED with respect to biopython
    (absent from the image): scores are plain Smith-Waterman; `begin` follows pairwise2's documented conventions."""
    return extract_spanning_reads_multi([(left_flanking_region, right_flanking_region)], [list(reads)], flanking_region_size)[0]


_COMP_STR = str.maketrans("ACGTN", "TGCAN")


def _spanning_piece(read, begin, end, reverse):
    """str(read).upper()[begin:end] -- or, for the reverse strand, reverse_complement(read).upper()[begin:end] -- touching only
    that piece of the read: the piece [begin, end) of the reverse complement is the reverse complement of [n - end, n - begin)."""
    if not reverse:
        return read[begin:end].upper()
    n = len(read)
    return read[max(n - end, 0):max(n - begin, 0)].upper().translate(_COMP_STR)[::-1]


def _spanning_hits(flank_pairs, read_lists, flanking_region_size=100):
    """The alignment half of extract_spanning_reads_multi: which (locus, read) uses span, on which strand and where.  Returns
    None when there are no reads, else a dict: reads (every distinct read once), codes / read_off (their encoding), uses (read of
    every (locus, read) use), use_locus, first (uses of locus i = first[i] .. first[i+1]), and per hit -- ordered by use, forward
    strand before reverse -- use, reverse, left_begin, right_begin."""
    return _spanning_align(_spanning_prepare(flank_pairs, read_lists, flanking_region_size))


def _spanning_prepare(flank_pairs, read_lists, flanking_region_size=100):
    """_spanning_hits up to the device call: the distinct reads, encoded, and the (read, flank) pairs to align (host only)."""
    import itertools
    n_loci = len(flank_pairs)
    flanks = []
    for lf, rf in flank_pairs:
        flanks += [lf[-flanking_region_size:], rf[:flanking_region_size]]
    first = np.zeros(n_loci + 1, np.int64)
    np.cumsum(np.fromiter(map(len, read_lists), dtype=np.int64, count=n_loci), out=first[1:])
    flat = list(itertools.chain.from_iterable(read_lists))
    n_uses = len(flat)
    if n_uses == 0:
        return None
    # a read that is a candidate of several loci (the same object in several lists, or one list handed over for every locus)
    # is encoded and uploaded ONCE; the pairs of every locus index that copy
    ids = np.fromiter(map(id, flat), dtype=np.int64, count=n_uses)
    _, where, uses = np.unique(ids, return_index=True, return_inverse=True)
    if len(where) == n_uses:
        reads, uses = flat, np.arange(n_uses, dtype=np.int32)
    else:
        reads, uses = [flat[i] for i in where.tolist()], uses.astype(np.int32)
    if not all(type(s) is str for s in reads):
        reads = [s if isinstance(s, str) else str(s) for s in reads]
    n = len(reads)
    codes, read_off, _ = _lib.encode_ascii(reads)
    total_bases = int(read_off[n])
    if 4 * n_uses >= 2 ** 31 or total_bases >= 2 ** 31:
        raise ValueError("extract_spanning_reads_multi: %d alignments over %d read bases in one call exceed the 32-bit indices of "
                         "advntr_flank_align; hand the loci over in pieces (genotype_pacbio_loci does)" % (4 * n_uses, total_bases))
    use_locus = np.repeat(np.arange(n_loci, dtype=np.int32), np.diff(first))
    # per (locus, read) use: forward strand then reverse strand (read index + n), each against the left then the right flank
    strand_read = np.repeat(uses, 2) + np.tile(np.array([0, n], np.int32), n_uses)
    pair_read = np.repeat(strand_read, 2)
    pair_flank = (2 * np.repeat(use_locus, 4) + np.tile(np.array([0, 1], np.int32), 2 * n_uses)).astype(np.int32)
    return dict(reads=reads, codes=codes, read_off=read_off, uses=uses, use_locus=use_locus, first=first, flanks=flanks,
                pair_read=pair_read, pair_flank=pair_flank)


def _spanning_align(H):
    """_spanning_hits from the device call on: the flank alignments and the spanning rule."""
    from . import settings
    if H is None:
        return None
    flanks, pair_flank = H.pop("flanks"), H.pop("pair_flank")
    score, begin, _, _ = _lib.flank_align(H["reads"], flanks, H.pop("pair_read"), pair_flank, encoded=(H["codes"], H["read_off"]))
    flen = np.fromiter(map(len, flanks), dtype=np.int64, count=len(flanks))
    need = flen * (1 - settings.MAX_ERROR_RATE)
    ok = ((score[0::2] > 0) & (score[0::2] >= need[pair_flank[0::2]]) & (score[1::2] > 0) &
          (score[1::2] >= need[pair_flank[1::2]]) & (begin[1::2] >= begin[0::2]))
    hits = np.flatnonzero(ok)                        # index = 2 * use + strand
    H.update(use=hits >> 1, reverse=(hits & 1).astype(np.uint8), left_begin=begin[2 * hits].astype(np.int64),
             right_begin=begin[2 * hits + 1].astype(np.int64))
    return H


def _spanning_pieces_encoded(H, flanking_region_size=100, threads=0):
    """The trimmed pieces of _spanning_hits' hits as ENCODED reads -- str(read).upper()[left_begin : right_begin + flank size] of
    the strand that spans, cut (and, for the reverse strand, reverse-complemented) out of the codes the alignment was fed with
    (advntr_cut_pieces): (codes, off, locus of every piece)."""
    rd = H["uses"][H["use"]]
    n = (H["read_off"][1:] - H["read_off"][:-1])[rd]
    begin, end = np.minimum(H["left_begin"], n), np.minimum(H["right_begin"] + flanking_region_size, n)
    end = np.maximum(end, begin)
    rev = H["reverse"] != 0
    # the piece [begin, end) of the reverse complement is the reverse complement of [n - end, n - begin) of the read as stored
    src_b, src_e = np.where(rev, n - end, begin), np.where(rev, n - begin, end)
    codes, off = _lib.cut_pieces(H["codes"], H["read_off"], rd, src_b, src_e, H["reverse"], threads)
    return codes, off, H["use_locus"][H["use"]]


def extract_spanning_reads_multi(flank_pairs, read_lists, flanking_region_size=100):
    """extract_spanning_reads for many loci in ONE advntr_flank_align call: flank_pairs[i] = (left_flanking_region,
    right_flanking_region) of locus i, read_lists[i] = its candidate long reads.  Returns one (spanning, length_distribution)
    pair per locus, each as extract_spanning_reads returns it (reads in input order, forward strand before reverse).  The
    reads go to the device as they are (the encoder folds case); only the trimmed piece of a spanning read -- a few hundred
    bases of its 5-15 kb -- is upper-cased and, for the reverse strand, reverse-complemented on the host."""
    out = [([], []) for _ in range(len(flank_pairs))]
    H = _spanning_hits(flank_pairs, read_lists, flanking_region_size)
    if H is None:
        return out
    reads, first = H["reads"], H["first"]
    use, rd, loc = H["use"].tolist(), H["uses"][H["use"]].tolist(), H["use_locus"][H["use"]].tolist()
    for r, u, i, rev, lb, rb in zip(use, rd, loc, H["reverse"].tolist(), H["left_begin"].tolist(), H["right_begin"].tolist()):
        spanning, lengths = out[i]
        spanning.append((_spanning_piece(reads[u], lb, rb + flanking_region_size, bool(rev)), r - int(first[i]), bool(rev)))
        lengths.append(rb - (lb + flanking_region_size))
    return out


# ------------------------------------------------------------------------------------------------
# Read selection from an alignment file (vntr_finder.py:701-767): mapped reads over the locus + filtered unmapped reads
# ------------------------------------------------------------------------------------------------
class SelectedRead(object):
    """vntr_finder.py:46-53 (the Viterbi path is replaced by the kernel's summary record; reference_start is kept, the
    reference only keeps whether there was one)."""
    __slots__ = ("sequence", "logp", "summary", "mapq", "reference_start", "query_name", "is_mapped")

    def __init__(self, sequence, logp, summary, mapq=None, reference_start=None, query_name=None):
        self.sequence, self.logp, self.summary = sequence, logp, summary
        self.mapq, self.reference_start, self.query_name = mapq, reference_start, query_name
        self.is_mapped = reference_start is not None


def select_illumina_reads(reference_vntr, samfile, unmapped_filtered_reads):
    """VNTRFinder.select_illumina_reads on a parsed alignment (advntr_amd.sam_utils.SamFile) and the sequences of the
    keyword-filtered unmapped reads.  Read length = median of the file's first five reads; mapped reads overlapping the
    VNTR (not unmapped/duplicate, at least 0.9 read lengths long, no N) are scored on the forward strand and kept if
    they are not low quality (utils.py:20-38) and recruit_read accepts them; unmapped reads of at least the read length
    go through process_unmapped_read (both strands, > 2 repeat bases).  All of it is ONE engine batch.  Returns
    (selected reads, the model): mapped reads first, in file order, then the unmapped ones, as the reference appends
    them."""
    selected, models = select_illumina_reads_multi([reference_vntr], samfile, [unmapped_filtered_reads])
    return selected[0], models[0]


def select_illumina_reads_multi(reference_vntrs, samfile, unmapped_lists, models=None):
    """select_illumina_reads for many loci over one alignment: every (read, strand, locus) call of all loci goes to the
    engine as one batch.  `models` (one per locus, built for the file's read length) are built here when not given.
    Returns (list of selected-read lists, models)."""
    from . import settings
    from .sam_utils import get_reference_genome_of_alignment_file, is_low_quality_read
    reference = get_reference_genome_of_alignment_file(samfile)
    lengths = sorted(len(r.seq) for r in samfile.head(5))
    read_length = lengths[len(lengths) // 2]
    min_read_length = int(read_length * 0.9) if settings.MIN_READ_LENGTH is None else settings.MIN_READ_LENGTH
    if models is None:
        models = [get_vntr_matcher_hmm(v, read_length) for v in reference_vntrs]
    batch, which, layout = [], [], []
    for i, (vntr, unmapped_reads) in enumerate(zip(reference_vntrs, unmapped_lists)):
        vntr_start = vntr.start_point
        vntr_end = vntr_start + vntr.get_length()
        chromosome = vntr.chromosome if reference == 'HG19' else vntr.chromosome[3:]
        mapped = []
        for read in samfile.fetch(chromosome, vntr_start, vntr_end):
            if read.is_unmapped or read.is_duplicate or len(read.seq) < min_read_length:
                continue
            read_end = read.reference_end if read.reference_end else read.reference_start + len(read.seq)
            if vntr_start - read_length < read.reference_start < vntr_end or vntr_start < read_end < vntr_end:
                if read.seq.count('N') <= 0:
                    mapped.append(read)
        unmapped = [str(s) for s in unmapped_reads if len(s) >= read_length and str(s).count('N') <= 0]
        start = len(batch)
        batch += [r.seq.upper() for r in mapped]
        fwd = [s.upper() for s in unmapped]
        batch += fwd
        batch += [reverse_complement(s) for s in fwd]
        which += [i] * (len(batch) - start)
        layout.append((start, mapped, len(unmapped)))
    out = [[] for _ in reference_vntrs]
    if not batch:
        return out, models
    bases, off = _lib.encode_reads(batch)
    logp, summ, _ = _lib.viterbi_batch(device_models(models), bases, off, np.asarray(which, np.int32),
                                       want_paths=False, want_summary=True)
    for i, (start, mapped, nu) in enumerate(layout):
        score = get_min_score_to_select_a_read(reference_vntrs[i].scaled_score, read_length)
        selected = out[i]
        for k, read in enumerate(mapped):
            a = start + k
            if summ[a][_lib.SUM_PATH_LEN] <= 2 or is_low_quality_read(read):
                continue
            if recruit_read(float(logp[a]), summ[a], score, len(batch[a])):
                selected.append(SelectedRead(batch[a], float(logp[a]), summ[a], read.mapq, read.reference_start, read.query_name))
        first = start + len(mapped)
        for j in range(nu):
            a = first + j
            if logp[a] < logp[first + nu + j]:
                a = first + nu + j
            if summ[a][_lib.SUM_PATH_LEN] <= 2:
                continue
            if recruit_read(float(logp[a]), summ[a], score, len(batch[a])) and summ[a][_lib.SUM_REPEAT_BP] > 2:
                selected.append(SelectedRead(batch[a], float(logp[a]), summ[a]))
    return out, models


# ------------------------------------------------------------------------------------------------
# Names that live in modules of their own and are looked up here on first use (no import cycle): the host pipelines
# (advntr_amd/pipelines.py) and the consumers outside the hot-path scope (advntr_amd/vntr_extras.py)
# ------------------------------------------------------------------------------------------------
_MOVED = {"pipelines": ("TextReads", "genotype_loci_pipelined", "genotype_pacbio_loci", "_Stage", "_Aborted"),
          "vntr_extras": ("simulate_true_reads", "simulate_false_filtered_reads",
                          "find_hmm_score_of_simulated_reads", "find_recruitment_score_threshold", "train_classifier_threshold",
                          "identify_frameshift", "find_frameshift_from_selected_reads", "find_frameshift", "update_model_from_reads",
                          "_off_length_unit_indels")}


def __getattr__(name):
    import importlib
    for module, names in _MOVED.items():
        if name in names:
            return getattr(importlib.import_module("." + module, __package__), name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
