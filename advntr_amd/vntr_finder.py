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


class TextReads(object):
    """The candidate reads of many loci as SPANS of one text -- the bytes of the FASTA file the keyword prefilter scanned
    (filtering.KeywordFilter.candidate_spans) -- instead of a list of str per locus: locus k's reads are spans
    locus_off[k] .. locus_off[k + 1].  genotype_loci_pipelined encodes a piece's reads straight out of the text
    (advntr_encode_spans, host threads); no Python object per read exists anywhere between the file and the genotypes."""

    def __init__(self, text, span_start, span_end, locus_off):
        self.text = text
        self._ready = None
        self._failed = None
        self.span_start = np.ascontiguousarray(span_start, np.int64)
        self.span_end = np.ascontiguousarray(span_end, np.int64)
        self.locus_off = np.ascontiguousarray(locus_off, np.int64)
        self.n_loci = len(self.locus_off) - 1

    @classmethod
    def pending(cls, text, n_loci):
        """The reads of n_loci loci that are still being selected (the prefilter runs on another thread): genotype_loci_pipelined
        can be started on this object at once -- its model building and upload stages need no reads -- and its read-encoding
        stage waits until fill() (or fail()) has been called."""
        import threading
        self = cls(text, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(int(n_loci) + 1, np.int64))
        self._ready = threading.Event()
        return self

    def fill(self, span_start, span_end, locus_off):
        if len(locus_off) != self.n_loci + 1:
            raise ValueError("TextReads.fill: %d loci were announced, %d are given" % (self.n_loci, len(locus_off) - 1))
        self.span_start = np.ascontiguousarray(span_start, np.int64)
        self.span_end = np.ascontiguousarray(span_end, np.int64)
        self.locus_off = np.ascontiguousarray(locus_off, np.int64)
        if self._ready is not None:
            self._ready.set()

    def fail(self, error):
        self._failed = error
        if self._ready is not None:
            self._ready.set()

    def _wait(self):
        if self._ready is not None:
            self._ready.wait()
        if self._failed is not None:
            raise self._failed

    def __len__(self):
        return self.n_loci

    def read_lists(self):
        """The same reads as lists of str (the stage-by-stage route and the tests)."""
        self._wait()
        t = self.text
        return [[t[a:b].decode("latin-1") for a, b in zip(self.span_start[lo:hi].tolist(), self.span_end[lo:hi].tolist())]
                for lo, hi in zip(self.locus_off[:-1].tolist(), self.locus_off[1:].tolist())]

    def prepare(self, lo, hi, threads=0):
        """_prepare_reads for loci lo .. hi - 1."""
        self._wait()
        a, b = int(self.locus_off[lo]), int(self.locus_off[hi])
        if a == b:
            return None
        counts = np.diff(self.locus_off[lo:hi + 1])
        codes, off, bad = _lib.encode_spans(self.text, self.span_start[a:b], self.span_end[a:b], threads=threads)
        if np.any(bad == 2):
            raise ValueError("Symbol is not defined in a distribution (read %d holds a symbol outside ACGTN)" % int(np.argmax(bad == 2)))
        lens = np.diff(off)
        locus_all = np.repeat(np.arange(hi - lo, dtype=np.int32), counts)
        index_all = (np.arange(b - a, dtype=np.int64) - np.repeat(np.cumsum(counts) - counts, counts)).astype(np.int32)
        keep = bad == 0
        nf = int(keep.sum())
        if nf == 0:
            return None
        if nf == b - a:
            return dict(locus=locus_all, index=index_all, lens=lens, bases=codes, off=off)
        klens = lens[keep]
        koff = np.zeros(nf + 1, np.int64)
        np.cumsum(klens, out=koff[1:])
        return dict(locus=locus_all[keep], index=index_all[keep], lens=klens, bases=codes[np.repeat(keep, lens)], off=koff)


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


class _Stage(object):
    """One stage of a host pipeline: a thread that takes one item from each input queue per piece, applies fn and puts the
    result on its output queue (bounded: a stage runs at most two pieces ahead of its consumer).  A failure travels downstream
    as the exception object and ends every stage it passes; `abort` (set by the consumer when it gives up) ends the rest."""

    def __init__(self, name, fn, n_pieces, inputs, abort, timer=None):
        import queue
        import threading
        self.out = queue.Queue(maxsize=2)
        self._fn, self._n, self._inputs, self._abort, self._timer = fn, n_pieces, inputs, abort, timer
        self.thread = threading.Thread(target=self._run, name=name)
        self.thread.start()

    def _get(self, q):
        import queue
        while not self._abort.is_set():
            try:
                return q.get(timeout=0.05)
            except queue.Empty:
                pass
        raise _Aborted()

    def _put(self, item):
        import queue
        while not self._abort.is_set():
            try:
                return self.out.put(item, timeout=0.05)
            except queue.Full:
                pass
        raise _Aborted()

    def _run(self):
        import time
        try:
            for k in range(self._n):
                args = [self._get(q.out) for q in self._inputs]
                for a in args:
                    if isinstance(a, BaseException):
                        self._put(a)
                        return
                t = time.perf_counter()
                res = self._fn(k, *args)
                if self._timer is not None:
                    self._timer[0][self._timer[1]] += time.perf_counter() - t
                    if "trace" in self._timer[0]:
                        self._timer[0]["trace"].append((self._timer[1], k, t, time.perf_counter()))
                self._put(res)
        except _Aborted:
            pass
        except BaseException as e:                  # handed to the consumer: a failure must not leave it waiting
            try:
                self._put(e)
            except _Aborted:
                pass


class _Aborted(Exception):
    pass


def genotype_loci_pipelined(loci, read_lists, scaled_scores=None, accuracy_filter=False, is_haploid=False,
                            compute_reverse=True, chunks=12, threads=0, timings=None, stage_threads=None, ramp=4,
                            piece_fractions=None):
    """genotype_loci from the locus DESCRIPTIONS -- loci = [(left_flank, right_flank, aligned_repeat_units, copies), ...], what
    the reference turns into a model per locus inside its serial loop (genome_analyzer.py:280-297 -> vntr_finder.py:117-138) --
    with the host stages overlapped with the device's.  The locus set is cut into `chunks` pieces that flow through a
    pipeline of host threads: build the models of a piece (native builder, host threads) -> upload them; encode the piece's
    reads; -> bind reads and models into a device batch (routing, tile lists, upload); the calling thread only launches a
    piece's kernels (both strands, reverse complements made on the device) and has the reads selected on the device (strand
    choice, recruit rule, more than two repeat bases: advntr_batch_recruit), so that only the selected reads' records come
    back.  The per-locus aggregation and the maximum-likelihood genotypes of a piece run on one more thread while the next
    piece is scored (that thread also drops what the piece leaves behind).  Same results
    as genotype_loci(build_read_matcher_models(loci), ...).  timings (a dict) receives wall seconds per stage.
    ramp: the first piece goes in parts of 1, 1, 2, 4 ... `ramp`-ths, so that the device starts after a small piece's host work
    instead of a full one's; piece_fractions: explicit piece sizes instead (shares of the locus set); stage_threads = (build, upload, encode) host threads of the three threaded stages."""
    import threading
    import time
    from . import hmm_utils
    n_loci = len(loci)
    chunks = max(1, min(int(chunks), n_loci)) if n_loci else 1
    cuts = [n_loci * i // chunks for i in range(chunks + 1)]
    if piece_fractions:
        # explicit piece sizes (shares of the locus set, in order; what they leave over is one more piece)
        acc, cuts = 0.0, [0]
        for f in piece_fractions:
            acc += float(f)
            at = min(n_loci, int(round(acc * n_loci)))
            if at > cuts[-1]:
                cuts.append(at)
        if cuts[-1] < n_loci:
            cuts.append(n_loci)
        chunks = len(cuts) - 1
    elif ramp and ramp > 1 and chunks > 1 and cuts[1] >= ramp:
        # the first piece in growing parts -- 1, 1, 2, 4, ... of `ramp` shares: the device starts on a small piece while the
        # host is still building the rest, and the pieces (a launch each, with its tail) do not stay small for long
        parts = [0, 1]
        while parts[-1] < ramp:
            parts.append(min(int(ramp), 2 * parts[-1]))
        cuts = [cuts[1] * q // int(ramp) for q in parts[:-1]] + cuts[1:]
        chunks = len(cuts) - 1
    T = dict(build_models=0.0, upload_models=0.0, encode_reads=0.0, bind_batch=0.0, score_recruit=0.0, aggregate_genotype=0.0)
    if timings is not None and "trace" in timings:
        T["trace"] = []                             # (stage, piece, start, end) of every stage call: scripts/host_profile.py
    abort = threading.Event()

    def upload(k, models):
        device_models(models, threads=t_other)
        return models

    def bind(k, models, prep):
        if prep is None:
            return models, None, None
        return models, prep, _lib.DeviceBatch(device_models(models), prep["bases"], prep["off"], prep["locus"],
                                              flags=_lib.FLAG_BOTH_STRANDS if compute_reverse else 0)

    # Host threads: the model builder may use the CPUs this process may use (advntr_host_threads: the hardware threads cut
    # down to the control group's quota -- 16 on the GPU boxes of this pool), the upload's table preparation half and the read
    # encoding (mostly the interpreter's own work on a million strings) a quarter of them.  The run burns ~2.3 core-seconds
    # (round 5: model building 1.4, table preparation 0.45, encoding 0.17, the device thread 0.15), in bursts that exhaust the
    # quota of a 100 ms accounting period early: the scheduler then stops EVERY thread of the process, the one that launches
    # kernels included, until the period ends (scripts/e2e_timeline.py shows such a run: a piece's kernels "take" 35 ms instead
    # of 9).  More threads per stage make that worse (32 per stage: +60 ms, round 4), fewer starve a stage.
    cpus = int(threads) if threads and threads > 0 else int(_lib.load().advntr_host_threads())
    t_build, t_other, t_enc = max(1, cpus), max(1, cpus // 2), max(1, cpus // 4)
    if stage_threads:
        t_build, t_other, t_enc = [max(1, int(x)) for x in stage_threads]
    t0 = time.perf_counter()
    built = _Stage("advntr-build", lambda k: hmm_utils.build_read_matcher_models(loci[cuts[k]:cuts[k + 1]], threads=t_build),
                   chunks, [], abort, (T, "build_models"))
    uploaded = _Stage("advntr-upload", upload, chunks, [built], abort, (T, "upload_models"))
    # (read_lists: a list of str per locus, or the loci's reads as spans of one text -- TextReads)
    prepare = (lambda k: (read_lists.prepare(cuts[k], cuts[k + 1], t_enc),)) if isinstance(read_lists, TextReads) else \
              (lambda k: (_prepare_reads(read_lists[cuts[k]:cuts[k + 1]], t_enc),))
    encoded = _Stage("advntr-encode", prepare, chunks, [], abort, (T, "encode_reads"))
    bound = _Stage("advntr-bind", lambda k, models, prep: bind(k, models, prep[0]), chunks, [uploaded, encoded], abort, (T, "bind_batch"))
    stages = [built, uploaded, encoded, bound]
    # What follows a piece's kernels runs on a thread of its own: the aggregation and the maximum-likelihood genotypes of the
    # piece's loci (pieces are whole loci), and the release of what the piece leaves behind -- 840 models are 840 destructor
    # calls, 3-20 ms that the calling thread would otherwise spend between two pieces' kernels
    import queue
    spent = queue.Queue()
    results, after = [None] * chunks, {"error": None}

    def finish_pieces():
        while True:
            item = spent.get()
            if item is None:
                return
            k, selected = item[0], item[1]
            del item                                # the piece's models leave the device with their last reference
            if after["error"] is not None:
                continue
            try:
                t = time.perf_counter()
                n_piece = cuts[k + 1] - cuts[k]
                if selected is None:
                    locus, summ = np.zeros(0, np.int64), np.zeros((0, _lib.SUMMARY_INTS), np.int32)
                else:
                    locus, summ = selected
                bounds = np.searchsorted(locus, np.arange(n_piece + 1)).astype(np.int64)
                results[k] = find_repeat_counts_of_loci(summ, bounds, accuracy_filter, is_haploid, threads=max(1, t_enc))
                T["aggregate_genotype"] += time.perf_counter() - t
            except BaseException as e:              # noqa: BLE001 -- handed to the caller after the join
                after["error"] = e

    finisher = threading.Thread(target=finish_pieces, name="advntr-aggregate-release", daemon=True)
    finisher.start()
    def finish(k, item):
        """Piece k's kernels are queued: wait for them, have the reads selected, hand the piece to the finisher."""
        models, prep, batch = item
        t = time.perf_counter()
        selected = None
        if batch is not None:
            try:
                index, _, summ, _ = batch.recruit(None if scaled_scores is None else scaled_scores[cuts[k]:cuts[k + 1]], 2)
            finally:
                batch.close()
            selected = (prep["locus"][index].astype(np.int64), summ)      # survivors in read order: grouped by locus
        T["score_recruit"] += time.perf_counter() - t
        if "trace" in T:
            T["trace"].append(("  dev:finish", k, t, time.perf_counter()))
        spent.put((k, selected, models, prep, batch, item))

    # The calling thread queues piece k + 1's kernels (every batch has a stream of its own) BEFORE it waits for piece k's: the
    # device starts on the next piece while the last workgroups of the previous one drain, and the selection, the download and
    # the interpreter's steps between two pieces are off the device's critical path
    pending = None
    try:
        for k in range(chunks):
            item = None
            if pending is not None:
                try:
                    item = bound.out.get_nowait()
                except queue.Empty:                 # nothing to launch yet: finish the piece in flight first
                    finish(*pending)
                    pending = None
            if item is None:
                tw = time.perf_counter()
                item = bound.out.get()
                if "trace" in T:
                    T["trace"].append(("  dev:wait", k, tw, time.perf_counter()))
            if isinstance(item, BaseException):
                raise item
            t = time.perf_counter()
            try:
                if item[2] is not None:
                    item[2].run()
            except BaseException:
                if item[2] is not None:
                    item[2].close()
                raise
            T["score_recruit"] += time.perf_counter() - t
            if "trace" in T:
                T["trace"].append(("  dev:launch", k, t, time.perf_counter()))
            if pending is not None:
                finish(*pending)
            pending = (k, item)
            del item
        if pending is not None:
            finish(*pending)
            pending = None
    finally:
        abort.set()                                 # (no stage is left waiting on a queue nobody serves any more)
        for st in stages:
            st.thread.join()
        # batches bound but never run (a failure upstream of them): release their device memory now
        while not bound.out.empty():
            left = bound.out.get_nowait()
            if isinstance(left, tuple) and left[2] is not None:
                left[2].close()
        if pending is not None and pending[1][2] is not None:      # launched, never finished (a failure in between)
            pending[1][2].close()
        spent.put(None)
    finisher.join()
    if after["error"] is not None:
        raise after["error"]
    out = [g for piece in results for g in piece]
    T["total"] = time.perf_counter() - t0
    if timings is not None:
        timings.update(T)
    return out


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


# ------------------------------------------------------------------------------------------------
# Recruitment-threshold training (the `addmodel` consumer of the scoring path): mirror of
# VNTRFinder.train_classifier_threshold and the methods it calls, /root/reference/advntr/vntr_finder.py:902-1021.
# reference_vntr: an object with the fields of advntr_amd.models.ReferenceVNTR.
# ------------------------------------------------------------------------------------------------
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


def simulate_true_reads(reference_vntr, read_length):
    """vntr_finder.py:975-1005: every read_length window of the locus, reads that enter/leave the VNTR with 1..10 flank
    bases for each prefix of the repeat segments, 40 reads from inside a long run of the VNTR; each gets one or two
    random substitutions.  Draws from Python's global `random` in the reference's order (the reference's stream starts
    at seed 0 because building the HMM just before -- bake(), hmm.pyx:858-859 -- seeds it; see
    train_classifier_threshold)."""
    from random import randint
    segments = reference_vntr.get_repeat_segments()
    vntr = ''.join(segments)
    left, right = reference_vntr.left_flanking_region, reference_vntr.right_flanking_region
    locus = left[-read_length:] + vntr + right[:read_length]
    templates = [locus[i:i + read_length].upper() for i in range(0, len(locus) - read_length + 1)]
    for copies in range(1, len(segments) - 1):
        section = ''.join(segments[:copies])
        for i in range(1, 11):
            templates.append((left[-i:] + section + right)[:read_length])
            templates.append((left + section + right[:i])[-read_length:])
    run = vntr * (int(read_length / len(vntr)) + 1)
    for i in range(1, 21):
        templates.append(run[i:read_length + i])
        templates.append(run[-read_length - i:-i])
    reads = []
    for read in templates:
        for _ in range(randint(1, 2)):
            chars = list(read)
            chars[randint(0, len(read) - 1)] = 'ACGT'[randint(0, 3)]
            read = ''.join(chars)
        reads.append(read)
    return reads


def simulate_false_filtered_reads(reference_vntr, sequences, min_match=3):
    """vntr_finder.py:927-973 on (name, sequence) pairs instead of a FASTA path: reads of 150 bases around places of the
    VNTR's chromosome, outside the VNTR, where >= min_match keyword 11-mers fall within 150 bases of each other -- what
    the keyword prefilter would wrongly let through.  The reference walks the chromosome with a rolling hash in
    Python; its effect, reproduced here with array operations, is: position i is examined iff i >= 1 and windows i-1
    and i both hold only A/C/G/T (the first clean window after the start or after an N only primes the hash), i stops
    one short of the last window, and a hash hit counts iff the 11-mer is a keyword.  Symbols other than ACGTN (which
    make the reference raise) are treated like N."""
    from .filtering import get_keywords_for_filtering
    keyword_size, read_size, max_false_reads = 11, 150, 10000
    keywords = get_keywords_for_filtering(reference_vntr.left_flanking_region, reference_vntr.get_repeat_segments(),
                                          reference_vntr.right_flanking_region, reference_vntr.pattern, True, keyword_size)
    table = np.zeros(4 ** keyword_size, dtype=bool)
    for kw in keywords:
        if len(kw) == keyword_size and all(ch in _BASE4 for ch in kw.upper()):
            v = 0
            for ch in kw.upper():
                v = v * 4 + _BASE4[ch]
            table[v] = True
    vntr_start = reference_vntr.start_point
    vntr_end = vntr_start + reference_vntr.get_length()
    false_reads, match_positions = [], []
    for name, sequence in sequences:
        if name != reference_vntr.chromosome:
            continue
        n = len(sequence)
        if n - keyword_size < 2:
            continue
        codes = _lib._CODE[np.frombuffer(sequence.upper().encode("latin-1", "replace"), dtype=np.uint8)]
        bad = np.concatenate([[0], np.cumsum(codes > 3)])
        n_win = n - keyword_size + 1
        clean = (bad[keyword_size:keyword_size + n_win] - bad[:n_win]) == 0
        value = np.zeros(n_win, dtype=np.int64)
        for t in range(keyword_size):
            value = value * 4 + np.minimum(codes[t:t + n_win], 3)
        i = np.arange(1, n - keyword_size)                      # the reference's loop stops at len - keyword_size - 1
        hit = clean[i] & clean[i - 1] & table[value[i]] & ~((vntr_start - read_size < i) & (i < vntr_end))
        for pos in i[hit].tolist():
            match_positions.append(pos)
            if len(match_positions) >= min_match and match_positions[-1] - match_positions[-min_match] < read_size:
                for j in range(match_positions[-1] - read_size, match_positions[-min_match], 5):
                    if 'N' not in sequence[j:j + read_size].upper():
                        false_reads.append(sequence[j:j + read_size])
            if len(false_reads) > max_false_reads:
                break
    return false_reads


_BASE4 = {'A': 0, 'C': 1, 'G': 2, 'T': 3}


def find_hmm_score_of_simulated_reads(model, reads):
    """vntr_finder.py:915-924: forward strand only, recruited against an absolute score of -10000, kept when more than
    two repeat bases are matched; returns the kept reads' log-probabilities (one GPU batch instead of a Python loop)."""
    kept = [r.upper() for r in reads if r.count('N') <= 0]
    if not kept:
        return []
    bases, off = _lib.encode_reads(kept)
    logp, summ, _ = _lib.viterbi_batch(device_models([model]), bases, off, np.zeros(len(kept), np.int32),
                                       want_paths=False, want_summary=True)
    lens = np.diff(off)
    ok = recruit_mask(logp, summ, lens, np.full(len(kept), -10000.0)) & (summ[:, _lib.SUM_REPEAT_BP] > 2)
    return [float(x) for x in logp[ok]]


def find_recruitment_score_threshold(true_scores, false_scores):
    """The score that separates a locus's own reads from the false positives of its keyword filter: a one-feature logistic
    classifier on the Viterbi scores, asked about every integer score -1, -2, ... -299 at once; the first one it calls
    "false" is the threshold, the best true score when it calls none (vntr_finder.py:1007-1021)."""
    from sklearn.linear_model import LogisticRegression
    pos = np.asarray(list(true_scores), np.float64)
    neg = np.asarray(list(false_scores), np.float64)
    if neg.size == 0:
        neg = np.array([pos.min() - 2])
    clf = LogisticRegression()
    clf.fit(np.concatenate([pos, neg]).reshape(-1, 1), np.concatenate([np.ones(pos.size, int), np.zeros(neg.size, int)]))
    grid = np.arange(-1, -300, -1)
    rejected = np.flatnonzero(clf.predict(grid.reshape(-1, 1).astype(np.float64)) == 0)
    return int(grid[rejected[0]]) if rejected.size else float(pos.max())


def train_classifier_threshold(reference_vntr, sequences, read_length=150):
    """vntr_finder.py:902-913: scaled recruitment score of a locus = threshold / read_length.  `sequences` = the
    reference genome as (name, sequence) pairs."""
    import random
    model = get_vntr_matcher_hmm(reference_vntr, read_length)
    random.seed(0)          # what baking the model does in the reference (hmm.pyx:858-859); the simulation below depends on it
    true_reads = simulate_true_reads(reference_vntr, read_length)
    false_reads = simulate_false_filtered_reads(reference_vntr, sequences)
    true_scores = find_hmm_score_of_simulated_reads(model, true_reads)
    false_scores = find_hmm_score_of_simulated_reads(model, false_reads)
    return find_recruitment_score_threshold(true_scores, false_scores) / float(read_length)


# ------------------------------------------------------------------------------------------------
# Frameshift identification from Viterbi paths (vntr_finder.py:256-309) -- a consumer of the engine's PATH output
# ------------------------------------------------------------------------------------------------
def identify_frameshift(location_coverage, observed_indel_transitions, expected_indels, error_rate=0.01):
    """Is an indel seen `observed_indel_transitions` times at a position covered `location_coverage` times a frameshift or a
    sequencing error?  Binomial likelihood of the count under either rate; a frameshift when the error explanation is a
    hundred times less likely (vntr_finder.py:256-263; the coverage may be fractional, as there)."""
    if observed_indel_transitions >= location_coverage:
        return True
    from scipy.stats import binom
    as_error, as_frameshift = binom.pmf(observed_indel_transitions, location_coverage, [error_rate, expected_indels])
    return bool(as_error / as_frameshift < 0.01)


def _off_length_unit_indels(sequence, visited_states, pattern_length):
    """The insert / delete states a read's path takes inside repeat units whose length is one or two bases off the pattern's
    (an insert state labelled with the base it emitted, e.g. 'I3A'), in path order."""
    from .hmm_utils import get_emitted_basepair_from_visited_states, get_repeating_pattern_lengths
    unit_lengths = get_repeating_pattern_lengths(visited_states)
    unit = -1
    for name in visited_states:
        if name.startswith('unit_start'):
            unit += 1
            continue
        if unit < 0 or unit >= len(unit_lengths) or name[0] not in 'ID' or name.endswith('fix'):
            continue
        off = abs(unit_lengths[unit] - pattern_length)
        if off == 0 or off > 2:
            continue
        label = name.split('_')[0]
        yield label + get_emitted_basepair_from_visited_states(name, visited_states, sequence) if label[0] == 'I' else label


def find_frameshift_from_selected_reads(pattern_length, vntr_length, selected_reads):
    """vntr_finder.py:265-309.  selected_reads = [(sequence, visited_state_names)] with the names of vpath[1:-1].  Tallies the
    indel states of off-length repeat units over the reads, takes the most frequent one (of equally frequent ones the one
    first seen last) and tests its count against the per-base coverage of the repeat region.  Returns the state label or None."""
    from .hmm_utils import state_class_from_name
    tally = {}
    repeat_bases = 0
    for sequence, visited_states in selected_reads:
        classes = np.fromiter((state_class_from_name(name) for name in visited_states), dtype=np.int64, count=len(visited_states))
        repeat_bases += int(np.count_nonzero((classes & _lib.SC_EMIT != 0) & (classes & _lib.SC_FIX == 0)))
        for label in _off_length_unit_indels(sequence, visited_states, pattern_length):
            tally[label] = tally.get(label, 0) + 1
    best, best_count = None, 0
    for label, count in tally.items():              # first-seen order; '>=' keeps the last of equally frequent labels
        if count >= best_count:
            best, best_count = label, count
    coverage = float(repeat_bases) / vntr_length / 2
    return best if identify_frameshift(coverage, best_count, 1 / coverage) else None


def find_frameshift(model, pattern_length, vntr_length, sequences, scaled_score=None):
    """find_frameshift_from_alignment_file (vntr_finder.py:776-780) on already extracted reads: both strands scored
    with PATH output in one batch, recruited reads with > 2 repeat bases selected (process_unmapped_read), then the
    test above."""
    keep = [s.upper() for s in sequences if s.count('N') <= 0]
    if not keep:
        return None
    batch = keep + [reverse_complement(s) for s in keep]
    logp, summ, paths = model.viterbi_batch(batch, want_paths=True, want_summary=True)
    names = [st.name for st in model.states]
    nf = len(keep)
    selected = []
    for j in range(nf):
        a = j + nf if logp[j] < logp[j + nf] else j
        if paths[a] is None:
            continue
        seq = batch[a]
        recruited = recruit_read(float(logp[a]), summ[a], get_min_score_to_select_a_read(scaled_score, len(seq)), len(seq))
        if recruited and summ[a][_lib.SUM_REPEAT_BP] > 2:
            selected.append((seq, [names[i] for i in paths[a][1:-1]]))
    if not selected:
        return None
    return find_frameshift_from_selected_reads(pattern_length, vntr_length, selected)


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


def genotype_pacbio_loci(loci, read_lists, accuracy_filter=False, is_haploid=False, chunks=16, threads=0, timings=None,
                         flanking_region_size=100):
    """VNTRFinder.find_repeat_count_from_pacbio_reads (vntr_finder.py:652-665) for many loci at once, from the WHOLE long
    reads to the RU-count genotypes: loci = [(left_flanking_region, right_flanking_region, repeat_segments, pattern), ...],
    read_lists[i] = the candidate reads of locus i (what the keyword filter hands over).  Per piece of the locus set:
    spanning-read extraction (both strands x two flanks of every read in one advntr_flank_align call, :324-371), one model
    per locus sized for its longest trimmed read (:538-549, native builder), every trimmed read scored on the forward strand
    in one engine batch (:550-555); the stages of a piece run on host threads of their own, a piece behind each other
    (extraction -> models -> upload and encoding), while the calling thread has the piece before scored, as
    genotype_loci_pipelined does.  The >= 3-reads support filter and the maximum-likelihood
    call (:568-580) run once at the end on host threads (advntr_genotype_observed).  settings.MAX_ERROR_RATE is the
    caller's (0.3 for PacBio, advntr_commands.py).  Returns one GenotypeResult per locus, as the reference builds it
    (:665): GenotypeResult(copy_numbers, n_spanning, n_spanning, 0, max_prob).  timings (a dict) receives wall seconds per
    stage.  Extraction is PARITY UNPINNED with respect to biopython (see extract_spanning_reads)."""
    import queue
    import threading
    import time
    from . import hmm_utils
    n_loci = len(loci)
    # a piece's extraction is ONE advntr_flank_align call: at most 2^22 alignments (4 per candidate read and locus; seconds
    # of kernel time, result arrays of 50 MB), whatever `chunks` asks for -- a locus set whose loci all share one long read
    # list (every read a candidate of every locus) otherwise overflows the call's 32-bit pair index
    n_pairs = 4 * sum(len(rl) for rl in read_lists)
    chunks = max(int(chunks), -(-n_pairs // (1 << 22)))
    chunks = max(1, min(chunks, n_loci)) if n_loci else 1
    cuts = [n_loci * i // chunks for i in range(chunks + 1)]
    T = dict(extract_spanning=0.0, build_models=0.0, upload_models=0.0, encode_reads=0.0, score=0.0, genotype=0.0)
    if timings is not None and "trace" in timings:
        T["trace"] = []
    abort = threading.Event()

    def encode_whole(k):
        lo, hi = cuts[k], cuts[k + 1]
        return _spanning_prepare([(l[0], l[1]) for l in loci[lo:hi]], read_lists[lo:hi], flanking_region_size)

    def extract(k, prep):
        # (locus, read) uses that span -> the trimmed pieces as encoded reads, grouped by locus (hits come ordered by use)
        H = _spanning_align(prep)
        if H is None or len(H["use"]) == 0:
            return None
        return _spanning_pieces_encoded(H, flanking_region_size, threads)

    def build(k, ext):
        if ext is None:
            return None
        lo = cuts[k]
        codes, off, piece_locus = ext
        have, start = np.unique(piece_locus, return_index=True)            # loci with a spanning read, ascending
        longest = np.maximum.reduceat(np.diff(off), start)
        desc = []
        for i, n_max in zip(have.tolist(), longest.tolist()):
            left, right, segments, pattern = loci[lo + i]
            desc.append((left[-flanking_region_size:], right[:flanking_region_size], segments, pacbio_max_copies([n_max], len(pattern))))
        models = hmm_utils.build_read_matcher_models(desc, threads=threads)
        which = np.searchsorted(have, piece_locus).astype(np.int32)
        return have, models, (codes, off), which

    def upload_encode(k, item):
        if item is None:
            return None
        have, models, enc, which = item
        t = time.perf_counter()
        dms = device_models(models)
        T["upload_models"] += time.perf_counter() - t
        # reads and models bound into a device batch here (routing, tile lists, upload): the calling thread only launches
        return have, models, _lib.DeviceBatch(dms, enc[0], enc[1], which), which

    # the stages of a piece run on threads of their own, a piece behind each other: encoding of the whole reads -> extraction
    # (the flank alignment kernel and the cutting of the spanning pieces out of those codes) -> models -> upload -> scoring (the
    # calling thread)
    t0 = time.perf_counter()
    whole = _Stage("advntr-pacbio-encode", encode_whole, chunks, [], abort, (T, "encode_reads"))
    extracted = _Stage("advntr-pacbio-extract", extract, chunks, [whole], abort, (T, "extract_spanning"))
    built = _Stage("advntr-pacbio-build", build, chunks, [extracted], abort, (T, "build_models"))
    ready = _Stage("advntr-pacbio-upload", upload_encode, chunks, [built], abort, None)
    ru_parts, count = [], np.zeros(n_loci, np.int64)

    def collect(k, item):
        """Piece k's kernels are queued: wait for them and take the RU counts."""
        t = time.perf_counter()
        have, models, batch, which = item
        try:
            _, summ = batch.fetch()
        finally:
            batch.close()
        ru_parts.append(summ[:, _lib.SUM_RU].astype(np.int32))
        np.add.at(count, cuts[k] + have.astype(np.int64)[which], 1)
        T["score"] += time.perf_counter() - t
        if "trace" in T:
            T["trace"].append(("score", k, t, time.perf_counter()))

    # (as in genotype_loci_pipelined: piece k + 1's kernels are queued before piece k's are waited for)
    pending = None
    try:
        for k in range(chunks):
            item = None
            if pending is not None:
                try:
                    item = ready.out.get_nowait()
                except queue.Empty:
                    collect(*pending)
                    pending = None
            if item is None:
                item = ready.out.get()
            if isinstance(item, BaseException):
                raise item
            if item is not None:
                t = time.perf_counter()
                try:
                    item[2].run()
                except BaseException:
                    item[2].close()
                    raise
                T["score"] += time.perf_counter() - t
            if pending is not None:
                collect(*pending)
            pending = (k, item) if item is not None else None
            del item
        if pending is not None:
            collect(*pending)
            pending = None
    finally:
        abort.set()
        for st in (whole, extracted, built, ready):
            st.thread.join()
        while not ready.out.empty():                # batches bound but never run (a failure upstream of them)
            left = ready.out.get_nowait()
            if isinstance(left, tuple):
                left[2].close()
        if pending is not None:                     # launched, never collected (a failure in between)
            pending[1][2].close()
    t = time.perf_counter()
    off = np.zeros(n_loci + 1, np.int64)
    np.cumsum(count, out=off[1:])
    ru = np.concatenate(ru_parts) if ru_parts else np.zeros(0, np.int32)       # pieces and loci in ascending order: grouped by locus
    geno, prob = _lib.genotype_observed(ru, off, accuracy_filter, is_haploid, threads)
    out = [GenotypeResult(None if a < 0 else (a, b), int(c), int(c), 0, p)
           for (a, b), p, c in zip(geno.tolist(), prob.tolist(), count.tolist())]
    T["genotype"] = time.perf_counter() - t
    T["total"] = time.perf_counter() - t0
    if timings is not None:
        timings.update(T)
    return out


# ------------------------------------------------------------------------------------------------
# Model update from the sample's own reads (vntr_finder.py:667-697, the `update` mode of genotyping)
# ------------------------------------------------------------------------------------------------
def update_model_from_reads(model, left_flanking_region, right_flanking_region, repeat_segments, pattern, selected_sequences,
                            read_length=None):
    """One re-estimation step of VNTRFinder.iteratively_update_model: the selected reads and the reference repeat units
    are scored with PATH output, the repeat units their paths cut out are aligned by profile position and a new
    read-matcher model is built from that alignment (hmm_utils.py:424-431).  The reference wraps this in a loop of up
    to 1000 steps that stops when the fitness improves by less than 1 -- and computes the fitness from the unchanged
    first selection (vntr_finder.py:692), so the loop always ends after this one step; re-selecting reads with the
    returned model (score_reads) is what the caller does next, as select_illumina_reads(..., hmm) does there."""
    from .hmm_utils import get_read_matcher_model
    selected_sequences = [s.upper() for s in selected_sequences]
    read_length = read_length or len(selected_sequences[0])
    sequences = selected_sequences + [str(r).upper() for r in repeat_segments]
    logp, _, paths = model.viterbi_batch(sequences, want_paths=True, want_summary=False)
    states = model.states
    vpaths = [(seq, [(i, states[i]) for i in path]) for seq, path in zip(sequences, paths) if path is not None]
    copies = get_copies_for_hmm(read_length, len(pattern))
    return get_read_matcher_model(left_flanking_region[-read_length:], right_flanking_region[:read_length], None, copies,
                                  vpaths)


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
