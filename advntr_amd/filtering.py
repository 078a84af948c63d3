"""Keyword prefilter of unmapped reads -- host side of the stage upstream of the scoring path.

Mirror of what `GenomeAnalyzer.get_filtered_read_ids` obtains from the external `adVNTR-Filtering` binary
(/root/reference/advntr/genome_analyzer.py:173-199, /root/reference/filtering/main.cc) and of the keyword
generator `VNTRFinder.get_keywords_for_filtering` (/root/reference/advntr/vntr_finder.py:140-153).

The per-read keyword matching (main.cc:247-283) runs on the GPU through `advntr_kwfilter_*`
(csrc/keyword_filter.h); this module keeps the order-dependent bookkeeping of main.cc:286-331 (per-VNTR read lists
capped at 3 x 2000 in read order, descending (count, name) order, the 2000 + 1 names quirk) so that `run()` yields
byte-for-byte the text the reference binary prints.  No CPU matching fallback: without the engine, scan() raises.
"""
import ctypes

import numpy as np

from . import _lib

_CODE5 = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE5[ord(_c)] = _i


def get_keywords_for_filtering(left_flank, repeat_segments, right_flank, pattern, short_reads=True, keyword_size=21):
    vntr = ''.join(repeat_segments)
    if len(vntr) < keyword_size:
        vntr = str(vntr) * (int(keyword_size / len(vntr)) + 1)
    locus = left_flank[-15:] + vntr + right_flank[:15]
    step_size = 5 if len(pattern) != 5 else 6
    queries = [locus[i:i + keyword_size] for i in range(0, len(locus) - keyword_size + 1, step_size)]
    if not short_reads:
        queries = [left_flank[-80:], right_flank[:80]]
    return set(queries)


def _atoi(s):
    n, i, sign = 0, 0, 1
    if i < len(s) and s[i] in "+-":
        sign = -1 if s[i] == "-" else 1
        i += 1
    while i < len(s) and s[i].isdigit():
        n = n * 10 + ord(s[i]) - 48
        i += 1
    return sign * n


class KeywordFilter(object):
    """Keyword sets of many VNTRs resident on the GPU.  `lines` = [(vntr_id, iterable of keywords), ...] in the
    order of the keywords file (main.cc:176-216: duplicate keywords on one line collapse, order within a line is
    the sorted order of std::set)."""

    def __init__(self, lines):
        self.vntr_ids = [int(v) for v, _ in lines]
        self.uniq_ids = sorted(set(self.vntr_ids))
        index = {v: i for i, v in enumerate(self.uniq_ids)}
        words, owners = [], []
        for vid, kws in lines:
            toks = sorted(set(kws))
            words += toks
            owners += [index[int(vid)]] * len(toks)
        off = np.zeros(len(words) + 1, np.int64)
        if words:
            np.cumsum(np.fromiter(map(len, words), dtype=np.int64, count=len(words)), out=off[1:])
        flat = "".join(words).encode("latin-1", "replace")
        codes = _CODE5[np.frombuffer(flat, dtype=np.uint8)] if flat else np.zeros(0, np.uint8)
        owners = np.asarray(owners, np.int32)
        L = _lib.load()
        _lib.require_gpu()
        self._h = L.advntr_kwfilter_create(_lib.ptr(np.ascontiguousarray(codes)), _lib.ptr(off), _lib.ptr(owners), len(words))
        if not self._h:
            raise _lib.EngineError(_lib.ERR_ARG, _lib.last_error())
        self.kernel_ms = 0.0

    @classmethod
    def from_text(cls, keywords_text):
        lines = []
        for line in keywords_text.split("\n"):
            tokens = line.split()
            if len(tokens) < 1:
                break
            lines.append((_atoi(tokens[0]), tokens[1:]))
        return cls(lines)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().advntr_kwfilter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scan_codes(self, bases, off):
        """One advntr_kwfilter_scan over encoded reads -> (read index, vntr index, occurrences) arrays, one row per
        (read, VNTR) pair with at least one hit, sorted by read then VNTR index (= VNTR id order)."""
        L = _lib.load()
        off = np.ascontiguousarray(off, np.int64)
        return self._scan(len(off) - 1, lambda o_r, o_v, o_c, cap, n_out, ms: L.advntr_kwfilter_scan(
            self._h, _lib.ptr(bases), _lib.ptr(off), len(off) - 1, _lib.ptr(o_r), _lib.ptr(o_v), _lib.ptr(o_c), cap, n_out, ms))

    def scan_text(self, text, span_start, span_end):
        """advntr_kwfilter_scan_text: the reads are spans of `text` (bytes of a FASTA file), uploaded as they are and mapped
        to base codes on the device (upper-case ACGT only, like the reference's char_to_num).  Same result as scan_codes."""
        L = _lib.load()
        span_start = np.ascontiguousarray(span_start, np.int64)
        span_end = np.ascontiguousarray(span_end, np.int64)
        n = len(span_start)
        return self._scan(n, lambda o_r, o_v, o_c, cap, n_out, ms: L.advntr_kwfilter_scan_text(
            self._h, text, len(text), _lib.ptr(span_start), _lib.ptr(span_end), n, _lib.ptr(o_r), _lib.ptr(o_v), _lib.ptr(o_c),
            cap, n_out, ms))

    def _scan(self, n_reads, call):
        cap = max(1024, n_reads // 4)
        while True:
            o_r, o_v, o_c = (np.empty(cap, np.int32) for _ in range(3))
            n_out = ctypes.c_int64(0)
            ms = ctypes.c_float(0)
            rc = call(o_r, o_v, o_c, cap, ctypes.byref(n_out), ctypes.byref(ms))
            if rc == _lib.ERR_TOO_LARGE and n_out.value > cap:
                cap = int(n_out.value) + 1024
                continue
            _lib.check(rc)
            break
        self.kernel_ms = ms.value
        n = int(n_out.value)
        if n == 0:
            z = np.zeros(0, np.int64)
            return z, z, z
        # the kernel may split a (read, VNTR) pair over several records: sum them
        key = o_r[:n].astype(np.int64) * len(self.uniq_ids) + o_v[:n]
        uniq, inverse = np.unique(key, return_inverse=True)
        counts = np.bincount(inverse, weights=o_c[:n], minlength=len(uniq)).astype(np.int64)
        return uniq // len(self.uniq_ids), uniq % len(self.uniq_ids), counts

    def count_matches(self, seqs):
        """{read index: {vntr_id: occurrences}} for reads with at least one keyword hit (GPU)."""
        seqs = list(seqs)
        off = np.zeros(len(seqs) + 1, np.int64)
        if seqs:
            np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs)), out=off[1:])
        flat = "".join(seqs).encode("latin-1", "replace")
        # (case sensitive like the reference's char_to_num, filtering/main.cc:44-55: anything but upper-case ACGT is "N")
        bases, _, _ = _lib.encode_spans(flat, off[:-1], off[1:], case_sensitive=True)
        reads, vntrs, counts = self.scan_codes(bases, off)
        out = {}
        for r, v, c in zip(reads.tolist(), vntrs.tolist(), counts.tolist()):
            out.setdefault(r, {})[self.uniq_ids[v]] = c
        return out

    def select(self, names, seqs, min_matches=5, max_reads=2000):
        """The bookkeeping of main.cc:284-331 on top of the GPU counts -> the reference's stdout text."""
        counts = self.count_matches(seqs)
        recs = [(r, vid, c) for r in sorted(counts) for vid, c in sorted(counts[r].items())]
        return self._select_records(recs, lambda r: names[r], lambda r: seqs[r], min_matches, max_reads)

    def select_fasta(self, text, min_matches=5, max_reads=2000):
        """select() straight from the bytes of a two-line FASTA file (what adVNTR-Filtering reads, main.cc:247-252): line
        index, encoding and scan on arrays; Python only touches the reads that have hits."""
        starts = _lib.line_index(text)
        n_lines = len(starts) - 1
        n_rec = n_lines // 2 if n_lines % 2 == 0 else (n_lines - 1) // 2
        ends = starts[1:] - 1                                           # exclusive end of each line (before its newline)
        if n_lines and not text.endswith(b"\n"):
            ends = ends.copy()
            ends[-1] = len(text)
        seq_lines = np.arange(n_rec) * 2 + 1
        reads, vntrs, counts = self.scan_text(text, starts[seq_lines], ends[seq_lines])
        recs = [(r, self.uniq_ids[v], c) for r, v, c in zip(reads.tolist(), vntrs.tolist(), counts.tolist())]
        name_of = lambda r: text[starts[2 * r] + 1:ends[2 * r]].decode("latin-1")
        seq_of = lambda r: text[starts[2 * r + 1]:ends[2 * r + 1]].decode("latin-1")
        return self._select_records(recs, name_of, seq_of, min_matches, max_reads)

    @staticmethod
    def fasta_spans(text):
        """(name start, name end, sequence start, sequence end) of the records of a two-line FASTA file, as arrays."""
        starts = _lib.line_index(text)
        n_lines = len(starts) - 1
        n_rec = n_lines // 2 if n_lines % 2 == 0 else (n_lines - 1) // 2
        ends = starts[1:] - 1                                           # exclusive end of each line (before its newline)
        if n_lines and not text.endswith(b"\n"):
            ends = ends.copy()
            ends[-1] = len(text)
        name_lines, seq_lines = np.arange(n_rec) * 2, np.arange(n_rec) * 2 + 1
        return starts[name_lines] + 1, ends[name_lines], starts[seq_lines], ends[seq_lines]

    def candidate_spans(self, text, min_matches=5, max_reads=2000, timings=None, spans=None):
        """The per-VNTR read lists of select_fasta() -- what `GenomeAnalyzer.get_filtered_read_ids` parses out of the binary's
        stdout (genome_analyzer.py:183-197) and the per-locus loop then looks up read by read (:283) -- as ARRAYS, without a Python
        object per read: (locus_off int64[n_vntr + 1], read index, sequence span start, sequence span end), VNTRs in the order of
        the keyword lines, a VNTR's reads in ascending name order (the order of the binary's read lines, main.cc:325-331).  Same
        bookkeeping as _select_records (main.cc:286-331: the 3 x max_reads intake cap in file order, descending (count, name),
        the max_reads + 1 names quirk); the keyword lines must name every VNTR once."""
        if len(set(self.vntr_ids)) != len(self.vntr_ids):
            raise ValueError("candidate_spans: a VNTR id occurs on several keyword lines")
        import time
        t0 = time.perf_counter()
        name_s, name_e, seq_s, seq_e = spans if spans is not None else self.fasta_spans(text)     # (spans: made by the caller already)
        t1 = time.perf_counter()
        reads, vntrs, counts = self.scan_text(text, seq_s, seq_e)
        t2 = time.perf_counter()
        if timings is not None:
            timings.update(line_index=t1 - t0, scan=t2 - t1, scan_kernel_ms=self.kernel_ms, hit_records=int(len(reads)))
        picked_r, picked_v = select_candidates(reads, vntrs, counts, lambda idx: _name_keys(text, name_s[idx], name_e[idx]),
                                               min_matches, max_reads)
        # scan indices follow the SORTED ids; the lists go out in keyword-line order
        line_of_sorted = np.argsort(np.asarray(self.vntr_ids), kind="stable")      # sorted-id index k -> line of that id
        v_line = line_of_sorted[picked_v] if len(picked_v) else np.zeros(0, np.int64)
        o = np.argsort(v_line, kind="stable")                                      # (name order inside a VNTR is kept)
        picked_r, v_line = picked_r[o], v_line[o]
        locus_off = np.searchsorted(v_line, np.arange(len(self.vntr_ids) + 1)).astype(np.int64)
        if timings is not None:
            timings.update(select=time.perf_counter() - t2, candidates=int(len(picked_r)))
        return locus_off, picked_r, seq_s[picked_r], seq_e[picked_r]

    def _select_records(self, recs, name_of, seq_of, min_matches, max_reads):
        """recs = (read index, vntr id, occurrences) in read order, VNTR ids ascending within a read."""
        vntr_read_list, read_sequences = {}, {}
        for r, vid, c in recs:                                         # reads in file order
            lst = vntr_read_list.setdefault(vid, {})
            if len(lst) > max_reads * 3:
                continue
            if c >= min_matches:
                nm = name_of(r)
                lst[nm] = c
                read_sequences[nm] = seq_of(r)
        out, filtered, acc = [], set(), {}
        for vid in self.vntr_ids:
            vec = acc.setdefault(vid, [])
            for name in sorted(vntr_read_list.get(vid, {})):
                vec.append((vntr_read_list[vid][name], name))
            line = "%d %d" % (vid, min(len(vec), max_reads))
            if vec:
                vec.sort(reverse=True)
                for j, (_, name) in enumerate(vec):
                    filtered.add(name)
                    line += " " + name
                    if j >= max_reads:
                        break
            out.append(line)
        for name in sorted(filtered):
            out.append("%s %s" % (name, read_sequences[name]))
        return "\n".join(out) + "\n"


def _name_keys(text, start, end):
    """Read names text[start[i]:end[i]] as a numpy bytes array that orders like the strings do (byte-wise)."""
    start, end = np.asarray(start, np.int64), np.asarray(end, np.int64)
    if len(start) == 0:
        return np.zeros(0, "S1")
    width = end - start
    if width.min() == width.max() and width[0] > 0:                    # fixed-width names: one gather, no Python object per read
        L = int(width[0])
        buf = np.frombuffer(text, np.uint8)
        return np.ascontiguousarray(buf[start[:, None] + np.arange(L)]).view("S%d" % L).reshape(-1)
    return np.array([text[a:b] for a, b in zip(start.tolist(), end.tolist())], dtype=bytes)


def _name_columns(names):
    """A bytes array as columns of big-endian 64-bit integers that order (first column most significant) like the bytes do."""
    width = names.dtype.itemsize
    pad = (-width) % 8
    raw = np.ascontiguousarray(names).view(np.uint8).reshape(len(names), width)
    if pad:
        raw = np.concatenate([raw, np.zeros((len(names), pad), np.uint8)], axis=1)
    return np.ascontiguousarray(raw).view(">u8").astype(np.uint64)          # (n, ceil(width / 8))


def select_candidates(reads, vntrs, counts, name_keys, min_matches=5, max_reads=2000):
    """The bookkeeping of filtering/main.cc:286-331 on arrays.  (reads, vntrs, counts): one record per (read, VNTR) pair with
    a keyword hit, in FILE order of the reads; name_keys(read indices) -> their names as a bytes array.  Returns (read index,
    VNTR index) of the reads the binary lists for each VNTR: reads with >= min_matches hits, at most 3 * max_reads + 1 of them
    taken in file order (the intake cap, main.cc:300), of those the max_reads + 1 first in descending (count, name) order (the
    loop that prints names breaks AFTER index max_reads); grouped by VNTR, ascending name inside a VNTR.  (Two reads of one
    name collapse in the binary's std::map; here they stay two -- read files do not repeat names.)"""
    reads, vntrs, counts = np.asarray(reads, np.int64), np.asarray(vntrs, np.int64), np.asarray(counts, np.int64)
    ok = counts >= min_matches
    r, v, c = reads[ok], vntrs[ok], counts[ok]
    if len(r) == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    if np.any(r[1:] < r[:-1]):                                          # file order
        o = np.argsort(r, kind="stable")
        r, v, c = r[o], v[o], c[o]

    def rank_in_group(group):                                           # position of every element inside its run of equal `group`
        first = np.concatenate([[True], group[1:] != group[:-1]])
        start = np.maximum.accumulate(np.where(first, np.arange(len(group)), 0))
        return np.arange(len(group)) - start
    def group_order(keys):                                              # stable order by a small non-negative integer key
        # (numpy sorts 16-bit keys with a radix sort: 9 ms per 600 000 records instead of 56 for 64-bit ones)
        return np.argsort(keys.astype(np.uint16) if keys.max() < 65536 else keys, kind="stable")
    o = group_order(v)                                                  # per VNTR, file order
    r, v, c = r[o], v[o], c[o]
    rank = rank_in_group(v)
    if rank.max() > 3 * max_reads:                                      # the intake cap bites
        keep = rank <= 3 * max_reads
        r, v, c, rank = r[keep], v[keep], c[keep], rank[keep]
    cols = _name_columns(name_keys(r))
    # per VNTR, ascending name: order by name (one 64-bit key when the names have up to 8 bytes), then stably by VNTR
    on = np.argsort(cols[:, 0]) if cols.shape[1] == 1 else np.lexsort(tuple(cols[:, j] for j in range(cols.shape[1] - 1, -1, -1)))
    by_name = on[group_order(v[on])]
    if rank.max() <= max_reads:                                         # no VNTR has more than max_reads + 1 reads: all are listed
        return r[by_name], v[by_name]
    # the output cap: per VNTR the max_reads + 1 first in descending (count, name) order; equal names share a rank
    r, v, c, cols = r[by_name], v[by_name], c[by_name], cols[by_name]
    fresh = np.concatenate([[True], np.any(cols[1:] != cols[:-1], axis=1) | (v[1:] != v[:-1])])
    name_rank = np.cumsum(fresh)                                        # ascending with (VNTR, name)
    o = np.lexsort((-name_rank, -c, v))
    keep = np.zeros(len(r), bool)
    keep[o[rank_in_group(v[o]) <= max_reads]] = True
    return r[keep], v[keep]                                             # (still per VNTR, ascending name)


def run(fasta_text, keywords_text, min_matches=5):
    """Drop-in for `adVNTR-Filtering reads.fa [--min_matches N] < keywords.txt`: returns its stdout text.  fasta_text: str
    or bytes of the two-line FASTA file."""
    text = fasta_text if isinstance(fasta_text, (bytes, bytearray)) else fasta_text.encode("latin-1", "replace")
    f = KeywordFilter.from_text(keywords_text)
    try:
        return f.select_fasta(bytes(text), min_matches=min_matches)
    finally:
        f.close()


def get_filtered_read_ids(fasta_text, vntr_keywords, min_matches=5):
    """genome_analyzer.py:173-199: (reads [(name, seq)], {vid: set(read names)}) from {vid: keywords}."""
    text = "".join("%s %s\n" % (vid, " ".join(sorted(k))) for vid, k in vntr_keywords.items())
    vntr_read_ids = {vid: [] for vid in vntr_keywords}
    reads = []
    for line in run(fasta_text, text, min_matches).split("\n"):
        parts = line.split()
        if len(parts) < 2:
            continue
        if parts[0].isdigit() and parts[1].isdigit():
            vntr_read_ids[int(parts[0])] = set(parts[2:])
        else:
            reads.append((parts[0], parts[1]))
    return reads, vntr_read_ids
