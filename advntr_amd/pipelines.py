"""Host pipelines around the scoring path for genome-scale runs -- many loci, a stage per thread, the device fed piece by piece:

  genotype_loci_pipelined   Illumina: locus descriptions + candidate reads -> genotypes (what the reference's per-locus loop does,
                            /root/reference/advntr/genome_analyzer.py:280-297 -> vntr_finder.py:117-138, 235-254, 807-887)
  TextReads                 the candidate reads as spans of one text (the bytes of the FASTA file the prefilter scanned)
  genotype_pacbio_loci      PacBio: whole long reads -> spanning reads -> genotypes (vntr_finder.py:534-585, 652-665)

The per-call pieces they string together live in vntr_finder.py, which re-exports these names."""
import numpy as np

from . import _lib
from .pomegranate import device_models
from . import vntr_finder as _vf      # (looked up at call time: the per-call pieces stay replaceable where they are defined)


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
        """vntr_finder._prepare_reads for loci lo .. hi - 1."""
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
              (lambda k: (_vf._prepare_reads(read_lists[cuts[k]:cuts[k + 1]], t_enc),))
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
                results[k] = _vf.find_repeat_counts_of_loci(summ, bounds, accuracy_filter, is_haploid, threads=max(1, t_enc))
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
        return _vf._spanning_prepare([(l[0], l[1]) for l in loci[lo:hi]], read_lists[lo:hi], flanking_region_size)

    def extract(k, prep):
        # (locus, read) uses that span -> the trimmed pieces as encoded reads, grouped by locus (hits come ordered by use)
        H = _vf._spanning_align(prep)
        if H is None or len(H["use"]) == 0:
            return None
        return _vf._spanning_pieces_encoded(H, flanking_region_size, threads)

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
            desc.append((left[-flanking_region_size:], right[:flanking_region_size], segments, _vf.pacbio_max_copies([n_max], len(pattern))))
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
    out = [_vf.GenotypeResult(None if a < 0 else (a, b), int(c), int(c), 0, p)
           for (a, b), p, c in zip(geno.tolist(), prob.tolist(), count.tolist())]
    T["genotype"] = time.perf_counter() - t
    T["total"] = time.perf_counter() - t0
    if timings is not None:
        timings.update(T)
    return out
