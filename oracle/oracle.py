"""ctypes front-end of oracle/viterbi_oracle.c plus a names-based restatement of the reference's
Viterbi-path summaries.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; advntr_amd/ never does.  Parity status: PINNED against tests/golden/*.json.gz (outputs of the
reference itself, see tests/golden/make_golden.py) by tests/test_oracle_golden.py.

Reference lines restated here (relative to /root/reference):
  number_of_repeats      advntr/hmm_utils.py:155-188
  number_of_matches      advntr/hmm_utils.py:191-197
  repeat_bp_matches      advntr/hmm_utils.py:200-206
  flanking_matching_rate advntr/hmm_utils.py:209-268
  left/right flank size  advntr/hmm_utils.py:271-286
  recruit_read           advntr/vntr_finder.py:179-190
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "viterbi_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_model_create.restype = ctypes.c_void_p
        L.oracle_model_create.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
        L.oracle_model_destroy.argtypes = [ctypes.c_void_p]
        L.oracle_model_csr.restype = ctypes.c_int
        L.oracle_model_csr.argtypes = [ctypes.c_void_p] * 4
        L.oracle_viterbi.restype = ctypes.c_double
        L.oracle_viterbi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_int, ctypes.c_void_p]
        L.oracle_forward.restype = ctypes.c_double
        L.oracle_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.oracle_viterbi_many.restype = None
        L.oracle_viterbi_many.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.oracle_viterbi_many_mt.restype = None
        L.oracle_viterbi_many_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _LIB = L
    return _LIB


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i


def encode(seq):
    a = _CODE[np.frombuffer(seq.encode("ascii"), dtype=np.uint8)] if len(seq) else np.zeros(0, np.uint8)
    if (a == 255).any():
        raise ValueError("Symbol not defined in a distribution")
    return np.ascontiguousarray(a)


class OracleModel(object):
    """A baked model: states emitting-first, edge list in graph.edges_iter() order (hmm.pyx:994)."""

    def __init__(self, m, silent_start, start_index, end_index, edges, emis_logp, state_names=None):
        self.m, self.silent_start, self.start_index, self.end_index = m, silent_start, start_index, end_index
        self.state_names = state_names
        src = np.ascontiguousarray([e[0] for e in edges], dtype=np.int32)
        dst = np.ascontiguousarray([e[1] for e in edges], dtype=np.int32)
        lp = np.ascontiguousarray([e[2] for e in edges], dtype=np.float64)
        em = np.ascontiguousarray(emis_logp, dtype=np.float64).reshape(silent_start, 4)
        self.n_edges = len(edges)
        self._h = lib().oracle_model_create(m, silent_start, start_index, end_index, self.n_edges,
                                            src.ctypes.data, dst.ctypes.data, lp.ctypes.data, em.ctypes.data)
        self.emis = em

    @classmethod
    def from_golden(cls, g):
        mod = g["model"]
        emis = [e["logp"] for e in mod["emissions"]]
        return cls(len(mod["state_names"]), mod["silent_start"], mod["start_index"], mod["end_index"],
                   mod["edges"], emis, mod["state_names"])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_model_destroy(self._h)
            self._h = None

    def csr(self):
        in_ptr = np.zeros(self.m + 1, np.int32)
        in_src = np.zeros(max(self.n_edges, 1), np.int32)
        in_logp = np.zeros(max(self.n_edges, 1), np.float64)
        finite = lib().oracle_model_csr(self._h, in_ptr.ctypes.data, in_src.ctypes.data, in_logp.ctypes.data)
        return in_ptr, in_src[:self.n_edges], in_logp[:self.n_edges], bool(finite)

    def viterbi(self, seq, path_cap=None):
        codes = encode(seq) if isinstance(seq, str) else np.ascontiguousarray(seq, dtype=np.uint8)
        n = len(codes)
        cap = path_cap if path_cap is not None else (n + 1) * (self.m - self.silent_start + 1) + 2
        path = np.zeros(cap, np.int32)
        plen = ctypes.c_int(0)
        logp = lib().oracle_viterbi(self._h, codes.ctypes.data, n, path.ctypes.data, cap, ctypes.byref(plen))
        if plen.value == -2:
            raise OverflowError("path longer than %d" % cap)
        return logp, (path[:plen.value].tolist() if plen.value > 0 else None)

    def forward(self, seq):
        codes = encode(seq) if isinstance(seq, str) else np.ascontiguousarray(seq, dtype=np.uint8)
        return lib().oracle_forward(self._h, codes.ctypes.data, len(codes))

    def viterbi_many(self, bases, off):
        bases = np.ascontiguousarray(bases, np.uint8)
        off = np.ascontiguousarray(off, np.int64)
        n_reads = len(off) - 1
        out = np.zeros(n_reads, np.float64)
        cap = int((off[1:] - off[:-1]).max()) + self.m + 2 if n_reads else 1
        scratch = np.zeros(cap, np.int32)
        lens = np.zeros(n_reads, np.int32)
        lib().oracle_viterbi_many(self._h, bases.ctypes.data, off.ctypes.data, n_reads, out.ctypes.data,
                                  scratch.ctypes.data, cap, lens.ctypes.data)
        return out, lens

    def viterbi_many_threads(self, bases, off, n_threads):
        """log-probs only, reads spread over n_threads host threads (bench.py's all-cores baseline)."""
        bases = np.ascontiguousarray(bases, np.uint8)
        off = np.ascontiguousarray(off, np.int64)
        n_reads = len(off) - 1
        out = np.zeros(n_reads, np.float64)
        cap = int((off[1:] - off[:-1]).max()) + self.m + 2 if n_reads else 1
        lib().oracle_viterbi_many_mt(self._h, bases.ctypes.data, off.ctypes.data, n_reads, out.ctypes.data, cap,
                                     int(n_threads))
        return out


# ---------------------------------------------------------------------------------------------
# Viterbi-path summaries, restated over state NAMES exactly as the reference does.
# `names` is the visited-state name list with the model start/end already stripped (vpath[1:-1]).
# ---------------------------------------------------------------------------------------------
def _emitting(name):                                  # hmm_utils.py:122-126
    return name.startswith(("M", "I", "start_random_matches", "end_random_matches"))


def number_of_repeats(names):                         # hmm_utils.py:155-188
    read_length = sum(1 for s in names if _emitting(s))
    starts = ends = 0
    current_bp = 0
    first_end = last_end = first_start = last_start = None
    for s in names:
        if _emitting(s):
            current_bp += 1
        if s.startswith("unit_start") and read_length - current_bp >= 3:
            if first_start is None:
                first_start = current_bp
            last_start = current_bp
            starts += 1
        if s.startswith("unit_end") and current_bp >= 3:
            if first_end is None:
                first_end = current_bp
            last_end = current_bp
            ends += 1
    delta = 0
    if None not in (last_start, first_start, last_end, first_end):
        if first_end < first_start and last_start > last_end:
            delta = 1
    return max(starts, ends) + delta


def number_of_matches(names):                         # hmm_utils.py:191-197
    return sum(1 for s in names if s.startswith("M"))


def repeat_bp_matches(names):                         # hmm_utils.py:200-206
    return sum(1 for s in names if _emitting(s) and not s.endswith("fix"))


def left_flank_size(names):                           # hmm_utils.py:271-277
    return sum(1 for s in names if _emitting(s) and s.endswith("suffix"))


def right_flank_size(names):                          # hmm_utils.py:280-286
    return sum(1 for s in names if _emitting(s) and s.endswith("prefix"))


def flanking_counts(names, sequence, left_flank, right_flank):
    """hmm_utils.py:209-251: (left_matches, left_bp, right_matches, right_bp)."""
    rm = rb = lm = lb = 0
    seq_index = 0
    max_hmm_index = -1
    prev = names[0]
    for s in names:
        if "suffix_end_suffix" in s:
            max_hmm_index = int(prev.split("_")[0][1:])
            break
        prev = s
    for s in names:
        if "start" in s or "end" in s:
            continue
        hmm_state = int(s.split("_")[0][1:])
        if s.endswith("prefix"):
            if s.startswith("M") and sequence[seq_index] == right_flank[hmm_state - 1]:
                rm += 1
            if _emitting(s):
                rb += 1
        if s.endswith("suffix"):
            if s.startswith("M") and sequence[seq_index] == left_flank[-(max_hmm_index - hmm_state + 1)]:
                lm += 1
            if _emitting(s):
                lb += 1
        if _emitting(s):
            seq_index += 1
    return lm, lb, rm, rb


def flanking_matching_rate(names, sequence, left_flank, right_flank, accuracy_filter=False):  # :252-268
    lm, lb, rm, rb = flanking_counts(names, sequence, left_flank, right_flank)
    dflt = 0.00001 if accuracy_filter else 1
    right_rate = float(rm) / rb if rb != 0 else dflt
    left_rate = float(lm) / lb if lb != 0 else dflt
    return min(right_rate, left_rate)


def recruit_read(logp, names, min_score, sequence, left_flank, right_flank):  # vntr_finder.py:179-190
    read_length = len(sequence)
    if flanking_matching_rate(names, sequence, left_flank, right_flank) < 0.90:
        return False
    if min_score is not None and logp > min_score:
        return True
    matches = number_of_matches(names)
    if min_score is None and matches >= 0.9 * read_length and logp > -read_length:
        return True
    return False


# ---------------------------------------------------------------------------------------------
# flank alignment (flank_align_oracle.c) -- PARITY UNPINNED with respect to biopython, see the C file
# ---------------------------------------------------------------------------------------------
def flank_align(read, flank):
    """(score, begin, end) of the local alignment (1, -1, -1, -1) the reference asks pairwise2 for."""
    L = lib()
    L.oracle_flank_align.restype = None
    L.oracle_flank_align.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 3
    code = np.full(256, 4, np.uint8)
    for i, c in enumerate("ACGT"):
        code[ord(c)] = i
    r = np.ascontiguousarray(code[np.frombuffer(read.upper().encode("latin-1", "replace"), np.uint8)])
    f = np.ascontiguousarray(code[np.frombuffer(flank.upper().encode("latin-1", "replace"), np.uint8)])
    out = (ctypes.c_int * 3)()
    L.oracle_flank_align(r.ctypes.data, len(r), f.ctypes.data, len(f), ctypes.byref(out, 0), ctypes.byref(out, 4), ctypes.byref(out, 8))
    return int(out[0]), int(out[1]), int(out[2])
