"""Synthetic, seeded workloads of the scoring path: loci and read batches shaped like
the ones adVNTR builds -- used by bench.py, smoke() and the size-scaled parity tests.  Nothing here reads
/root/reference."""
import numpy as np

from . import hmm_utils, settings

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rand_seq(rng, n):
    return _ACGT[rng.integers(0, 4, n)].tobytes().decode()


class Locus(object):
    """Flanks, aligned repeat units, copies; the read-matcher model is built on first use."""

    def __init__(self, left, right, units, copies, error_rate, model=None):
        self.left, self.right, self.units, self.copies, self.error_rate, self._model = \
            left, right, units, copies, error_rate, model

    @property
    def model(self):
        if self._model is None:
            old = settings.MAX_ERROR_RATE
            settings.MAX_ERROR_RATE = self.error_rate
            try:
                self._model = hmm_utils.get_read_matcher_model(self.left, self.right, self.units, self.copies)
            finally:
                settings.MAX_ERROR_RATE = old
        return self._model

    def device_model(self):
        return self.model.device_model()


def build_models(loci, threads=0):
    """Build the models of many loci at once on host threads (advntr_build_read_matchers)."""
    todo = [l for l in loci if l._model is None]
    for rate in sorted(set(l.error_rate for l in todo)):
        group = [l for l in todo if l.error_rate == rate]
        old = settings.MAX_ERROR_RATE
        settings.MAX_ERROR_RATE = rate
        try:
            models = hmm_utils.build_read_matcher_models([(l.left, l.right, l.units, l.copies) for l in group], threads)
        finally:
            settings.MAX_ERROR_RATE = old
        for l, m in zip(group, models):
            l._model = m


def make_locus(rng, flank, pattern_len, copies, error_rate=0.05, n_units=1, max_subs=2):
    """Flanks and a repeat pattern of random ACGT; n_units equal-length repeat units that differ from the
    pattern by <= max_subs substitutions (=> gap-free alignment, no MSA needed)."""
    left, right = rand_seq(rng, flank), rand_seq(rng, flank)
    pattern = rand_seq(rng, pattern_len)
    units = [pattern]
    for _ in range(n_units - 1):
        u = list(pattern)
        for _ in range(int(rng.integers(0, max_subs + 1))):
            u[int(rng.integers(0, pattern_len))] = "ACGT"[int(rng.integers(0, 4))]
        units.append("".join(u))
    return Locus(left, right, units, copies, error_rate)


def make_reads(rng, locus, n_reads, read_len, locus_fraction=0.4, sub_rate=0.01):
    """40 % locus-derived ((left[-U(0,100):] + unit*U(1,C) + right)[:read_len], padded, 1 % substitutions),
    60 % uniform random ACGT."""
    reads = []
    for _ in range(n_reads):
        if rng.random() < locus_fraction:
            k = int(rng.integers(1, locus.copies + 1))
            lf = int(rng.integers(0, min(len(locus.left), 100) + 1))
            body = "".join(locus.units[int(rng.integers(0, len(locus.units)))] for _ in range(k))
            s = (locus.left[len(locus.left) - lf:] + body + locus.right)[:read_len]
            s = s + rand_seq(rng, read_len - len(s))
            a = np.frombuffer(s.encode(), dtype=np.uint8).copy()
            hit = rng.random(len(a)) < sub_rate
            a[hit] = _ACGT[rng.integers(0, 4, int(hit.sum()))]
            s = a.tobytes().decode()
        else:
            s = rand_seq(rng, read_len)
        reads.append(s)
    return reads


def ref150(seed=20240601):
    """Config C1's model: flank 150, 14-bp pattern, copies 11 -> 1413 states / 921 emitting / 4626 edges."""
    return make_locus(np.random.default_rng(seed), 150, 14, 11, 0.05)


def s300(seed=20240601):
    """flank 30, 12-bp pattern, copies 3 -> 315 states / 197 emitting / 1004 edges."""
    return make_locus(np.random.default_rng(seed), 30, 12, 3, 0.05)


def _c2_plan(args):
    """Locus k of the C2 model set and its read counts (cheap: no reads are generated).  Seeded per locus, so any
    subset of the loci can be produced on any rank."""
    k, seed, read_len, mapped_mean, unmapped_mean = args
    from .vntr_finder import get_copies_for_hmm
    rng = np.random.default_rng([seed, k])
    plen = int(rng.integers(6, 101))
    loc = make_locus(rng, read_len, plen, get_copies_for_hmm(read_len, plen), 0.05, n_units=int(rng.integers(2, 21)))
    return loc, int(rng.poisson(mapped_mean)), int(rng.poisson(unmapped_mean))


def c2_plan(n_loci, seed=20240602, read_len=150, mapped_mean=80, unmapped_mean=40):
    """[(calls, states)] of every C2 locus -- what the multi-GPU partitioner needs (sharding.locus_work's inputs)
    without generating a read: calls = mapped + 2 * unmapped, states = 6F + 3C(L+1) + 18."""
    out = []
    for k in range(n_loci):
        loc, nm, nu = _c2_plan((k, seed, read_len, mapped_mean, unmapped_mean))
        L = len(loc.units[0])
        out.append((nm + 2 * nu, 6 * read_len + 3 * loc.copies * (L + 1) + 18))
    return out


def _c2_locus(args):
    """One C2 locus and its calls (worker of make_c2_parallel's process pool; the model is built by the parent)."""
    k, seed, read_len, mapped_mean, unmapped_mean = args
    from .vntr_finder import reverse_complement
    loc, n_mapped, n_unmapped = _c2_plan(args)
    rng = np.random.default_rng([seed, k, 1])
    mapped = make_reads(rng, loc, n_mapped, read_len, locus_fraction=0.9)
    unmapped = make_reads(rng, loc, n_unmapped, read_len, locus_fraction=0.5)
    calls = mapped + unmapped + [reverse_complement(s) for s in unmapped]
    return (loc.left, loc.right, loc.units, loc.copies, loc.error_rate), calls, (n_mapped, n_unmapped)


def make_c2_parallel(n_loci, seed=20240602, read_len=150, mapped_mean=80, unmapped_mean=40, workers=None, build=True,
                     only=None, return_counts=False):
    """make_c2 with the synthetic read generation spread over a process pool (per-locus seeds, independent of the
    worker count); the models are then built by the native builder on host threads.  only = the locus indices this
    rank owns (multi-GPU sharding: every rank can produce exactly its share).  Returns ([Locus], reads, read_locus)
    with read_locus indexing the returned list (+ the (mapped, unmapped) read counts per locus with return_counts)."""
    import multiprocessing as mp
    import os
    workers = workers or max(1, min(32, (os.cpu_count() or 2) - 1))
    ks = list(range(n_loci)) if only is None else [int(k) for k in only]
    jobs = [(k, seed, read_len, mapped_mean, unmapped_mean) for k in ks]
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_c2_locus, jobs, chunksize=8)
    loci, reads, which, counts = [], [], [], []
    for i, (params, calls, nmu) in enumerate(res):
        loci.append(Locus(*params))
        reads += calls
        which += [i] * len(calls)
        counts.append(nmu)
    if build:
        build_models(loci)
    if return_counts:           # (mapped, unmapped) per locus: a locus's calls are mapped + unmapped + the unmapped reads' reverse complements
        return loci, reads, np.asarray(which, dtype=np.int32), counts
    return loci, reads, np.asarray(which, dtype=np.int32)


def make_c2(n_loci, seed=20240602, read_len=150, mapped_mean=80, unmapped_mean=40):
    """Config C2/C3 of BASELINE.json at a chosen locus count: loci with pattern length U{6..100},
    2-20 reference repeat units (equal length, <= 2 substitutions => gap-free alignment), flank = read length,
    copies = round(read_len/len(pattern)+0.5) as adVNTR does (vntr_finder.py:98-99,131); per locus Poisson(80)
    mapped-like reads (forward only) + Poisson(40) filtered-unmapped-like reads scored on both strands.
    Returns (loci, reads, read_locus)."""
    from .vntr_finder import get_copies_for_hmm, reverse_complement
    rng = np.random.default_rng(seed)
    loci, reads, which = [], [], []
    for k in range(n_loci):
        plen = int(rng.integers(6, 101))
        loc = make_locus(rng, read_len, plen, get_copies_for_hmm(read_len, plen), 0.05, n_units=int(rng.integers(2, 21)))
        loci.append(loc)
        mapped = make_reads(rng, loc, int(rng.poisson(mapped_mean)), read_len, locus_fraction=0.9)
        unmapped = make_reads(rng, loc, int(rng.poisson(unmapped_mean)), read_len, locus_fraction=0.5)
        calls = mapped + unmapped + [reverse_complement(s) for s in unmapped]
        reads += calls
        which += [k] * len(calls)
    return loci, reads, np.asarray(which, dtype=np.int32)


# ------------------------------------------------------------------------------------------------
# Config C4 (BASELINE configs[4]): PacBio loci, flank 100, error rate 0.3, trimmed spanning reads
# ------------------------------------------------------------------------------------------------
def noisy_copy(rng, s, rate=0.12):
    """Substitutions, insertions and deletions at a total rate `rate` (a third each), PacBio-like."""
    a = np.frombuffer(s.encode(), dtype=np.uint8)
    r = rng.random(len(a))
    out = []
    third = rate / 3.0
    for ch, x in zip(a.tolist(), r.tolist()):
        if x < third:
            continue                                            # deletion
        if x < 2 * third:
            out.append(int(_ACGT[rng.integers(0, 4)]))          # insertion before the base
        if 2 * third <= x < rate:
            ch = int(_ACGT[rng.integers(0, 4)])                 # substitution
        out.append(ch)
    return bytes(out).decode()


def _c4_locus(args):
    """One PacBio locus and its 20 trimmed spanning reads: VNTR of U(100,1000) bp, reads = 100-bp flanks around the
    VNTR at +-20 % of the reference copy number, 12 % noise; the model is sized for the longest read as
    get_dominant_copy_numbers_from_spanning_reads does (vntr_finder.py:538-549)."""
    k, seed, n_reads = args
    from .vntr_finder import pacbio_max_copies
    rng = np.random.default_rng([seed, k])
    plen = int(rng.integers(10, 61))
    ref_copies = max(2, int(round(int(rng.integers(100, 1001)) / plen)))
    left, right, pattern = rand_seq(rng, 500), rand_seq(rng, 500), rand_seq(rng, plen)
    reads = []
    for _ in range(n_reads):
        c = max(1, int(round(ref_copies * (0.8 + 0.4 * rng.random()))))
        reads.append(noisy_copy(rng, left[-100:] + pattern * c + right[:100]))
    copies = pacbio_max_copies([len(r) for r in reads], plen)
    return (left[-100:], right[:100], [pattern], max(1, copies), 0.3), reads


def c4_plan(n_loci, seed=20240603, n_reads=20):
    """[(calls, expected read length, expected states)] of every C4 locus WITHOUT generating a read -- what the multi-GPU
    partitioner gets (sharding.partition_loci on calls x (length + 1) x states), and all a real run knows about a locus before
    its reads have been extracted: the pattern length and the reference VNTR length.  _c4_locus draws both FIRST from the locus's
    own generator, so this replays exactly those two draws; the reads then come out at 0.8-1.2 x the reference copy number
    with indel noise, and the model is sized for the longest of them (vntr_finder.py:538-549): expected read length =
    VNTR + 200 flank bases, expected copies = round((1.2 x VNTR + 100) / pattern), states = 6 F + 3 C (L + 1) + 18 (F = 100)."""
    out = []
    for k in range(n_loci):
        rng = np.random.default_rng([seed, k])
        plen = int(rng.integers(10, 61))
        ref_copies = max(2, int(round(int(rng.integers(100, 1001)) / plen)))
        vntr = ref_copies * plen
        copies = max(1, int(round((1.2 * vntr + 100) / float(plen))))
        out.append((n_reads, vntr + 200, 6 * 100 + 3 * copies * (plen + 1) + 18))
    return out


def make_c4(n_loci, seed=20240603, n_reads=20, workers=None, only=None):
    """([Locus], reads, read_locus) of the PacBio configuration; models built by the native builder (error 0.3)."""
    import multiprocessing as mp
    import os
    workers = workers or max(1, min(32, (os.cpu_count() or 2) - 1))
    ks = list(range(n_loci)) if only is None else [int(k) for k in only]
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(_c4_locus, [(k, seed, n_reads) for k in ks], chunksize=8)
    loci, reads, which = [], [], []
    for i, (params, rs) in enumerate(res):
        loci.append(Locus(*params))
        reads += rs
        which += [i] * len(rs)
    return loci, reads, np.asarray(which, dtype=np.int32)


def _pacbio_whole_locus(args):
    """One PacBio locus as _c4_locus draws it, with its reads as the sequencer delivers them: WHOLE reads of
    U(min_len, max_len) bases, either strand, that hold a noisy copy of (300 flank bases + VNTR at +-20 % of the reference
    copy number + 300 flank bases) somewhere inside; every tenth read is unrelated sequence (a false candidate of the
    keyword filter).  Returns ((left_flanking_region, right_flanking_region, repeat_segments, pattern), reads)."""
    k, seed, n_reads, min_len, max_len = args
    from .vntr_finder import reverse_complement
    rng = np.random.default_rng([seed, k])
    plen = int(rng.integers(10, 61))
    ref_copies = max(2, int(round(int(rng.integers(100, 1001)) / plen)))
    left, right, pattern = rand_seq(rng, 500), rand_seq(rng, 500), rand_seq(rng, plen)
    reads = []
    for j in range(n_reads):
        n = int(rng.integers(min_len, max_len + 1))
        if j % 10 == 9:
            reads.append(rand_seq(rng, n))
            continue
        c = max(1, int(round(ref_copies * (0.8 + 0.4 * rng.random()))))
        core = noisy_copy(rng, left[-300:] + pattern * c + right[:300])
        n = max(n, len(core) + 2)
        at = int(rng.integers(0, n - len(core)))
        s = rand_seq(rng, at) + core + rand_seq(rng, n - at - len(core))
        reads.append(s if rng.random() < 0.5 else reverse_complement(s))
    return (left, right, [pattern], pattern), reads


def make_pacbio_whole_reads(n_loci, seed=20240603, n_reads=20, min_len=5000, max_len=15000, workers=None):
    """(loci, read_lists) for vntr_finder.genotype_pacbio_loci: BASELINE config 5's loci (the C4 recipe) with
    whole 5-15 kb reads instead of already trimmed ones."""
    import multiprocessing as mp
    import os
    workers = workers or max(1, min(32, (os.cpu_count() or 2) - 1))
    jobs = [(k, seed, n_reads, min_len, max_len) for k in range(n_loci)]
    if workers == 1 or n_loci < 4:
        res = [_pacbio_whole_locus(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_pacbio_whole_locus, jobs, chunksize=4)
    return [r[0] for r in res], [r[1] for r in res]


# ------------------------------------------------------------------------------------------------
# Upstream stages: the keyword prefilter's and the flank alignment's bench inputs
# ------------------------------------------------------------------------------------------------
def make_prefilter_workload(n_loci=6719, n_reads=2000000, read_len=150, locus_every=100, seed=20240604):
    """The Illumina prefilter at model-database scale: the 15-mer keyword sets of n_loci synthetic loci (a keyword every 5
    bases over 15 + VNTR + 15, vntr_finder.py:140-153) and n_reads reads of read_len bases as the BYTES OF A TWO-LINE FASTA
    FILE -- what adVNTR-Filtering reads (filtering/main.cc:247-252); every locus_every-th read is cut from a locus, the rest
    are windows of random sequence.  Returns (keyword lines [(vntr id, set of keywords)], fasta bytes, record length): every
    record is b'>r%07d\\n' + bases + b'\\n', so record r starts at r * record length."""
    from .filtering import get_keywords_for_filtering
    rng = np.random.default_rng(seed)
    lines, loci = [], []
    for v in range(n_loci):
        plen = int(rng.integers(6, 101))
        pat = rand_seq(rng, plen)
        reps = [pat] * int(rng.integers(2, 21))
        left, right = rand_seq(rng, 15), rand_seq(rng, 15)
        lines.append((v + 1, get_keywords_for_filtering(left, reps, right, pat, True, 15)))
        if v < 200:
            loci.append(left + "".join(reps) + right)
    big = _ACGT[rng.integers(0, 4, read_len * 50000)]
    rec_len = 10 + read_len + 1
    r = np.arange(n_reads, dtype=np.int64)
    head = np.empty((n_reads, 10), np.uint8)
    head[:, 0], head[:, 1], head[:, 9] = ord(">"), ord("r"), 10
    for d in range(7):
        head[:, 8 - d] = 48 + (r // 10 ** d) % 10
    start = (r * 137) % (len(big) - read_len)
    fasta = np.concatenate([head, np.lib.stride_tricks.sliding_window_view(big, read_len)[start],
                            np.full((n_reads, 1), 10, np.uint8)], axis=1)
    if locus_every:
        for k in range(0, n_reads, locus_every):
            s = loci[k % len(loci)]
            s = (s * (read_len // len(s) + 1))[:read_len]
            fasta[k, 10:10 + read_len] = np.frombuffer(s.encode(), np.uint8)
    return lines, fasta.tobytes(), rec_len


def make_illumina_pipeline_workload(loci, candidates, n_reads=10000000, read_len=150, seed=20240605):
    """The `advntr genotype` flow's input at model-database scale, for ONE timeline from file bytes to genotype rows: the
    keyword lines of `loci` (15-mers, a keyword every 5 bases over 15 + VNTR + 15: vntr_finder.py:140-153, ids 1 .. n in locus
    order) and the bytes of a two-line FASTA file of n_reads reads in which the loci's candidate reads (candidates[k]: forward
    strands, read_len bases each) are planted at random record positions among windows of random sequence.  Returns (keyword
    lines, fasta bytes, record length, planted record index per candidate in locus order)."""
    from .filtering import get_keywords_for_filtering
    rng = np.random.default_rng(seed)
    lines = [(k + 1, get_keywords_for_filtering(l.left, l.units, l.right, l.units[0], True, 15)) for k, l in enumerate(loci)]
    flat = [s for c in candidates for s in c]
    if any(len(s) != read_len for s in flat):
        raise ValueError("make_illumina_pipeline_workload: every candidate read must have %d bases" % read_len)
    n_reads = max(int(n_reads), 2 * len(flat))
    big = _ACGT[rng.integers(0, 4, read_len * 50000)]
    r = np.arange(n_reads, dtype=np.int64)
    rec_len = 10 + read_len + 1
    fasta = np.empty((n_reads, rec_len), np.uint8)
    fasta[:, 0], fasta[:, 1], fasta[:, 9], fasta[:, rec_len - 1] = ord(">"), ord("r"), 10, 10
    for d in range(7):
        fasta[:, 8 - d] = 48 + (r // 10 ** d) % 10
    start = (r * 137) % (len(big) - read_len)
    windows = np.lib.stride_tricks.sliding_window_view(big, read_len)
    for lo in range(0, n_reads, 1 << 20):                               # (in slabs: the gather's temporaries stay small)
        fasta[lo:lo + (1 << 20), 10:10 + read_len] = windows[start[lo:lo + (1 << 20)]]
    at = np.sort(rng.choice(n_reads, len(flat), replace=False))
    at = at[rng.permutation(len(flat))]                                 # a locus's reads are scattered over the file
    if flat:
        fasta[at, 10:10 + read_len] = np.frombuffer("".join(flat).encode(), np.uint8).reshape(len(flat), read_len)
    return lines, fasta.tobytes(), rec_len, at


def make_flank_align_workload(n_reads=4000, min_len=5000, max_len=15000, seed=11):
    """PacBio-sized input of advntr_flank_align: two 100-base flanks and n_reads reads of 5-15 kb, a third of which hold a
    noisy copy (12 %) of flank + repeats + flank.  Returns (left, right, reads)."""
    rng = np.random.default_rng(seed)
    left, right, pattern = rand_seq(rng, 100), rand_seq(rng, 100), rand_seq(rng, 40)
    reads = []
    for k in range(n_reads):
        n = int(rng.integers(min_len, max_len + 1))
        s = rand_seq(rng, n)
        if k % 3 == 0:
            core = noisy_copy(rng, left + pattern * int(rng.integers(3, 20)) + right, 0.12)
            if len(core) < n:
                at = int(rng.integers(0, n - len(core)))
                s = s[:at] + core + s[at + len(core):]
        reads.append(s)
    return left, right, reads
