"""ctypes binding of libadvntr_hip.so (include/advntr_hip.h) -- the only way into the HIP engine.

There is no CPU fallback: if the shared library is missing or no MI355X is visible, every scoring call
raises.  `load()` only dlopens the library (no GPU needed), so symbol/ABI checks run anywhere.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADVNTR_HIP_LIB") or os.path.join(_HERE, "libadvntr_hip.so")   # override: kernel experiments

OK, ERR_ARG, ERR_SYMBOL, ERR_DEVICE, ERR_TOO_LARGE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
FLAG_PATH, FLAG_FORCE_GENERIC, FLAG_NO_SUMMARY, FLAG_STREAM, FLAG_ANTIDIAGONAL, FLAG_BOTH_STRANDS = 1, 2, 4, 8, 16, 32
FLAG_DEEP_TILES = 64
FLAG_SECOND_QUEUE = 128      # a stream class of its own: see include/advntr_hip.h (two copies of a batch whose passes overlap)
SUMMARY_INTS = 8
SUM_RU, SUM_MATCHES, SUM_REPEAT_BP, SUM_LEFT_BP, SUM_RIGHT_BP, SUM_LEFT_MATCH, SUM_RIGHT_MATCH, SUM_PATH_LEN = range(8)

SC_EMIT, SC_MATCH, SC_SUFFIX, SC_PREFIX = 0x1, 0x2, 0x4, 0x8
SC_UNIT_START, SC_UNIT_END, SC_SKIP, SC_FIX = 0x10, 0x20, 0x40, 0x80
SC_BASE_SHIFT, SC_BASE_VALID = 8, 0x400

# every symbol include/advntr_hip.h declares: (restype, argtypes)
_vp, _i32, _u32, _i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_int64
SYMBOLS = {
    "advntr_device_count": (ctypes.c_int, []),
    "advntr_set_device": (ctypes.c_int, [ctypes.c_int]),
    "advntr_last_error": (ctypes.c_char_p, []),
    "advntr_version": (ctypes.c_char_p, []),
    "advntr_trim": (None, []),
    "advntr_host_threads": (ctypes.c_int, []),
    "advntr_hmm_create": (_vp, [_i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "advntr_hmm_destroy": (None, [_vp]),
    "advntr_hmm_has_column_program": (ctypes.c_int, [_vp]),
    "advntr_hmm_info": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "advntr_viterbi_batch": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _u32]),
    "advntr_forward_batch": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _u32]),
    "advntr_batch_create": (_vp, [_vp, _i32, _vp, _vp, _vp, _i32, _u32]),
    "advntr_batch_destroy": (None, [_vp]),
    "advntr_batch_run": (ctypes.c_int, [_vp]),
    "advntr_batch_sync": (ctypes.c_int, [_vp]),
    "advntr_batch_reserve_next": (ctypes.c_int, [_vp, _i32]),
    "advntr_batch_run_timed": (ctypes.c_int, [_vp, _i32, _vp]),
    "advntr_batch_forward": (ctypes.c_int, [_vp]),
    "advntr_batch_forward_timed": (ctypes.c_int, [_vp, _i32, _vp]),
    "advntr_batch_fetch": (ctypes.c_int, [_vp, _vp, _vp]),
    "advntr_batch_fetch_paths": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "advntr_batch_recruit": (ctypes.c_int, [_vp, _vp, _i32, _vp]),
    "advntr_batch_fetch_recruited": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "advntr_batch_result_ptrs": (ctypes.c_int, [_vp, _vp, _vp]),
    "advntr_batch_device_bytes": (_i64, [_vp]),
    "advntr_batch_info": (ctypes.c_int, [_vp, _vp, _i32]),
    "advntr_kwfilter_create": (_vp, [_vp, _vp, _vp, _i32]),
    "advntr_kwfilter_destroy": (None, [_vp]),
    "advntr_kwfilter_scan": (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "advntr_kwfilter_scan_text": (ctypes.c_int, [_vp, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "advntr_build_read_matchers": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, _vp, _vp, _i32, _u32, _vp]),
    "advntr_align_repeats": (ctypes.c_int, [_vp, _i32, _vp, _i64, _vp]),
    "advntr_flank_align": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "advntr_built_info": (ctypes.c_int, [_vp, _vp]),
    "advntr_built_info_many": (ctypes.c_int, [_vp, _i32, _vp]),
    "advntr_built_export": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "advntr_built_upload": (_vp, [_vp]),
    "advntr_built_upload_many": (ctypes.c_int, [_vp, _i32, _i32, _vp]),
    "advntr_built_destroy": (None, [_vp]),
    "advntr_encode_ascii": (ctypes.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "advntr_encode_texts": (ctypes.c_int, [_vp, _i32, _u32, _i32, _vp, _vp, _vp]),
    "advntr_encode_spans": (ctypes.c_int, [_vp, _vp, _vp, _i32, _u32, _i32, _vp, _vp, _vp]),
    "advntr_cut_pieces": (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "advntr_pylist_texts": (_i64, [_vp, _vp, _vp, _i64]),          # (called through _list_texts: needs the interpreter lock)
    "advntr_line_index": (ctypes.c_int, [_vp, _i64, _i32, _vp, _i64, _vp]),
    "advntr_genotype_illumina": (ctypes.c_int, [_vp, _vp, _i32, _u32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "advntr_genotype_observed": (ctypes.c_int, [_vp, _vp, _i32, _u32, _i32, _vp, _vp]),
    "advntr_comm_available": (ctypes.c_int, []),
    "advntr_comm_unique_id": (ctypes.c_int, [_vp]),
    "advntr_comm_create": (_vp, [_i32, _i32, _vp]),
    "advntr_comm_destroy": (None, [_vp]),
    "advntr_comm_info": (ctypes.c_int, [_vp, _vp, _vp]),
    "advntr_comm_allgather_i64": (ctypes.c_int, [_vp, _i64, _vp]),
    "advntr_comm_allreduce_max_f64": (ctypes.c_int, [_vp, _vp]),
    "advntr_comm_barrier": (ctypes.c_int, [_vp]),
    "advntr_comm_gather_results_start": (ctypes.c_int, [_vp, _vp, _i32, _vp]),
    "advntr_comm_gather_results_finish": (ctypes.c_int, [_vp, _vp, _vp]),
    "advntr_comm_last_gather_ms": (ctypes.c_int, [_vp, _vp]),
    "advntr_comm_gather_bytes": (ctypes.c_int, [_vp, _i32, _vp, _vp, _vp]),
}

_lib = None


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "advntr_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    """dlopen the engine and bind every declared symbol.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return load().advntr_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        msg = last_error()
        if rc == ERR_SYMBOL:
            raise ValueError("Symbol is not defined in a distribution (%s)" % msg)  # reference: hmm.pyx:72,79
        raise EngineError(rc, msg)


def require_gpu():
    n = load().advntr_device_count()
    if n <= 0:
        raise EngineError(ERR_DEVICE, "no HIP device visible; the scoring path runs on MI355X only (no CPU fallback)")
    return n


def ptr(a):
    return None if a is None else a.ctypes.data


_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i


def encode_reads(seqs):
    """list of str/bytes/uint8 arrays -> (bases uint8, read_off int64).  Unknown symbols become code 255,
    which the engine rejects with ADVNTR_ERR_SYMBOL (the reference's ValueError)."""
    n = len(seqs)
    off = np.zeros(n + 1, dtype=np.int64)
    if n and all(type(s) is str for s in seqs):
        # the common case, done without a Python loop per base or per read: one join, one table lookup
        np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=n), out=off[1:])
        raw = np.frombuffer("".join(seqs).encode("latin-1", "replace"), dtype=np.uint8)
        return np.ascontiguousarray(_CODE[raw]), off
    parts = []
    for i, s in enumerate(seqs):
        if isinstance(s, str):
            a = _CODE[np.frombuffer(s.encode("latin-1", "replace"), dtype=np.uint8)]
        elif isinstance(s, (bytes, bytearray)):
            a = _CODE[np.frombuffer(bytes(s), dtype=np.uint8)]
        elif isinstance(s, np.ndarray) and s.dtype == np.uint8:
            a = s
        else:
            a = _CODE[np.frombuffer("".join(s).encode("latin-1", "replace"), dtype=np.uint8)]
        parts.append(a)
        off[i + 1] = off[i] + len(a)
    bases = np.concatenate(parts).astype(np.uint8) if parts and off[-1] else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases), off


_pylist_texts = None


def _list_texts(seqs):
    """The buffers of a list of ASCII str -- (pointers uint64[n], lengths int64[n]) -- through ONE library call that walks the
    list (advntr_pylist_texts of include/advntr_pyhost.h, entered with the interpreter lock held), or None when the list holds
    anything else.  The pointers are borrowed from the strings: the caller keeps `seqs` unchanged until they have been consumed
    (encode_ascii does so before it returns; str objects are immutable, so only removing items from the list could free one)."""
    global _pylist_texts
    if type(seqs) is not list:
        return None
    if _pylist_texts is None:
        load()
        fn = ctypes.PyDLL(LIB_PATH).advntr_pylist_texts
        fn.restype = _i64
        fn.argtypes = [ctypes.py_object, _vp, _vp, _i64]
        _pylist_texts = fn
    n = len(seqs)
    ptrs, lens = np.empty(n, np.uint64), np.empty(n, np.int64)
    return (ptrs, lens) if _pylist_texts(seqs, ptr(ptrs), ptr(lens), n) == n else None


def encode_ascii(seqs, threads=0):
    """advntr_encode_ascii: list of str -> (codes uint8 over the concatenation, read_off int64, bad uint8[n]) on host
    threads: upper/lower-case ACGT -> 0..3, N -> 254, anything else -> 255; bad[read] = 1 for a read holding N (the
    reference skips those before scoring, vntr_finder.py:237), 2 for one holding another symbol (its viterbi raises
    ValueError there, hmm.pyx:72,79), 0 otherwise."""
    n = len(seqs)
    off = np.zeros(n + 1, dtype=np.int64)
    texts = _list_texts(seqs) if n else None
    if texts is not None:
        # a list of ASCII str (every read file): encoded straight out of the strings' own buffers, the interpreter touched once
        np.cumsum(texts[1], out=off[1:])
        codes = np.empty(int(off[n]), np.uint8)
        bad = np.zeros(n, np.uint8)
        check(load().advntr_encode_texts(ptr(texts[0]), n, 0, int(threads), ptr(off), ptr(codes), ptr(bad)))
        return codes, off, bad
    if n:
        np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=n), out=off[1:])
    if n and off[n] >= LONG_TEXT_MEAN * n:
        return _encode_texts(seqs, off, threads)
    text = "".join(seqs)
    # ASCII text (every read file) is handed over as the joined string's own buffer; anything else goes through latin-1, where a
    # character outside it becomes '?' -- "another symbol" either way
    size = ctypes.c_ssize_t(0)
    try:
        raw = _as_utf8(text, ctypes.byref(size)) if text else None
    except UnicodeError:                             # (lone surrogates have no UTF-8 form)
        raw = None
    if raw is None or size.value != len(text):
        raw = text.encode("latin-1", "replace")
    codes = np.empty(len(text), np.uint8)
    bad = np.zeros(n, np.uint8)
    check(load().advntr_encode_ascii(raw, ptr(off), n, int(threads), ptr(codes), ptr(bad)))
    del text
    return codes, off, bad


LONG_TEXT_MEAN = 2048          # mean read length from which the reads are encoded out of their own buffers
_as_utf8 = ctypes.pythonapi.PyUnicode_AsUTF8AndSize          # (for an ASCII str: its own buffer, no copy)
_as_utf8.restype = ctypes.c_void_p
_as_utf8.argtypes = [ctypes.py_object, ctypes.POINTER(ctypes.c_ssize_t)]


def _encode_texts(seqs, off, threads=0):
    """advntr_encode_texts: long reads (PacBio) are encoded straight out of the strings' own buffers -- joining and
    re-encoding 40 MB of text in Python costs more than the alignment kernel that follows.  A str whose UTF-8 form is
    not one byte per character (so: not ASCII) goes through latin-1 'replace' like the joined path's text does."""
    n = len(seqs)
    as_utf8 = _as_utf8
    ptrs = np.zeros(n, np.uint64)
    keep = []                                   # bytes objects made here: alive until the call returns
    size = ctypes.c_ssize_t(0)
    for r, s in enumerate(seqs):
        if isinstance(s, str):
            try:
                p = as_utf8(s, ctypes.byref(size))
            except UnicodeError:                 # a lone surrogate has no UTF-8 form: the latin-1 route maps it to '?'
                p = None
            if not p or size.value != len(s):
                b = s.encode("latin-1", "replace")
                keep.append(b)
                p = ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value
        else:
            b = bytes(s)
            keep.append(b)
            p = ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value
        ptrs[r] = p or 0
    codes = np.empty(int(off[n]), np.uint8)
    bad = np.zeros(n, np.uint8)
    check(load().advntr_encode_texts(ptr(ptrs), n, 0, int(threads), ptr(off), ptr(codes), ptr(bad)))
    del keep
    return codes, off, bad


ENCODE_CASE_SENSITIVE = 1


def line_index(text_bytes, threads=0):
    """advntr_line_index: int64 start offsets of the lines of a bytes object, plus the total length as a last entry."""
    n = len(text_bytes)
    cap = max(16, text_bytes.count(b"\n", 0, min(n, 1 << 16)) * (n // max(1, min(n, 1 << 16)) + 1) + 16)
    n_lines = ctypes.c_int64(0)
    while True:
        starts = np.empty(cap + 1, np.int64)
        rc = load().advntr_line_index(text_bytes, n, int(threads), ptr(starts), cap, ctypes.byref(n_lines))
        if rc == ERR_TOO_LARGE and n_lines.value > cap:
            cap = int(n_lines.value)
            continue
        check(rc)
        break
    k = int(n_lines.value)
    starts[k] = n
    return starts[:k + 1]


def encode_spans(text_bytes, span_start, span_end, case_sensitive=False, threads=0):
    """advntr_encode_spans: (codes over the concatenated spans, read_off int64, bad uint8[n])."""
    span_start = np.ascontiguousarray(span_start, np.int64)
    span_end = np.ascontiguousarray(span_end, np.int64)
    n = len(span_start)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(span_end - span_start, out=off[1:])
    codes = np.empty(int(off[-1]), np.uint8)
    bad = np.zeros(n, np.uint8)
    check(load().advntr_encode_spans(text_bytes, ptr(span_start), ptr(span_end), n, ENCODE_CASE_SENSITIVE if case_sensitive else 0,
                                     int(threads), ptr(off), ptr(codes), ptr(bad)))
    return codes, off, bad


class DeviceModel(object):
    """Owns one advntr_hmm handle (a baked model resident in HBM)."""

    def __init__(self, m, silent_start, start_index, end_index, in_ptr, in_src, in_logp, emis_logp, state_class=None):
        L = load()
        require_gpu()
        self.m, self.silent_start = int(m), int(silent_start)
        in_ptr = np.ascontiguousarray(in_ptr, dtype=np.int32)
        in_src = np.ascontiguousarray(in_src, dtype=np.int32)
        in_logp = np.ascontiguousarray(in_logp, dtype=np.float64)
        emis = np.ascontiguousarray(emis_logp, dtype=np.float64)
        cls = None if state_class is None else np.ascontiguousarray(state_class, dtype=np.uint16)
        self._h = L.advntr_hmm_create(m, silent_start, start_index, end_index, len(in_src), ptr(in_ptr),
                                      ptr(in_src), ptr(in_logp), ptr(emis), ptr(cls))
        if not self._h:
            raise EngineError(ERR_ARG, last_error())

    @property
    def handle(self):
        return self._h

    def has_column_program(self):
        return bool(load().advntr_hmm_has_column_program(self._h))

    def n_columns(self):
        """Columns of the model's column program (0: none, the model runs on the generic-CSR kernel)."""
        nc = ctypes.c_int32(0)
        check(load().advntr_hmm_info(self._h, None, None, None, ctypes.byref(nc)))
        return int(nc.value)

    def close(self):
        if getattr(self, "_h", None):
            load().advntr_hmm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _UploadedModel(DeviceModel):
    """A DeviceModel around an advntr_hmm handle the library created itself (advntr_built_upload)."""

    def __init__(self, handle, m, silent_start):
        self._h = handle
        self.m, self.silent_start = int(m), int(silent_start)


EXP_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)


def _numpy_exp(src, dst, n, _user):
    a = np.ctypeslib.as_array(ctypes.cast(src, ctypes.POINTER(ctypes.c_double)), shape=(n,))
    o = np.ctypeslib.as_array(ctypes.cast(dst, ctypes.POINTER(ctypes.c_double)), shape=(n,))
    np.exp(a, out=o)


_NUMPY_EXP = EXP_FN(_numpy_exp)
_NUMPY_EXP_LOOP = False          # (address, data) of numpy's own fp64 exp inner loop once found, None if not usable


def _numpy_exp_loop():
    """numpy.exp's fp64 inner loop as a C function pointer, so that the builder's worker threads call it directly instead
    of queueing for a Python callback behind the interpreter lock.  Read from the ufunc object (PyUFuncObject, numpy's
    public C struct: ufuncobject.h) after checking that the object looks the way that header says; verified against
    numpy.exp itself on a test vector before it is trusted.  None -> the callback is used."""
    global _NUMPY_EXP_LOOP
    if _NUMPY_EXP_LOOP is not False:
        return _NUMPY_EXP_LOOP
    _NUMPY_EXP_LOOP = None
    try:
        import sys
        import sysconfig
        u = np.exp
        base, ptr_size = id(u), ctypes.sizeof(ctypes.c_void_p)
        # the struct offsets below are those of a release build of CPython (no Py_DEBUG object header, no free-threaded
        # header) with NumPy 1.x / 2.x on a 64-bit platform; anything else takes the callback
        if (ptr_size != 8 or not isinstance(u, np.ufunc) or sys.implementation.name != "cpython"
                or sysconfig.get_config_var("Py_DEBUG") or sysconfig.get_config_var("Py_GIL_DISABLED")
                or hasattr(sys, "gettotalrefcount") or int(np.__version__.split(".")[0]) not in (1, 2)
                or os.environ.get("ADVNTR_NUMPY_EXP_LOOP", "1") == "0"):
            return None
        nin, nout, nargs, _identity = (ctypes.c_int * 4).from_address(base + 16)
        ntypes = ctypes.c_int.from_address(base + 48).value
        if (nin, nout, nargs, ntypes) != (1, 1, 2, u.ntypes):        # the integers first: no pointer is followed before they fit
            return None
        name_ptr = ctypes.c_void_p.from_address(base + 56).value
        if not name_ptr:
            return None
        name = ctypes.string_at(name_ptr, 3)
        if name != b"exp":
            return None
        codes = {"e": 23, "f": 11, "d": 12, "g": 13, "F": 14, "D": 15, "G": 16, "O": 17}
        want = [(codes.get(t[0], -1), codes.get(t[-1], -1)) for t in u.types]
        types = (ctypes.c_char * (2 * ntypes)).from_address(ctypes.c_void_p.from_address(base + 64).value).raw
        got = [(types[2 * i], types[2 * i + 1]) for i in range(ntypes)]
        if got != want or (12, 12) not in got:
            return None
        i = got.index((12, 12))
        fn = ctypes.c_void_p.from_address(ctypes.c_void_p.from_address(base + 32).value + 8 * i).value
        data = ctypes.c_void_p.from_address(ctypes.c_void_p.from_address(base + 40).value + 8 * i).value
        if not fn:
            return None
        loop = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)(fn)
        x = np.concatenate([np.random.default_rng(3).uniform(-40.0, 0.0, 1001), [0.0, -np.inf, -1e-300, -745.2]])
        y = np.empty_like(x)
        args = (ctypes.c_void_p * 2)(x.ctypes.data, y.ctypes.data)
        dims, steps = (ctypes.c_ssize_t * 1)(len(x)), (ctypes.c_ssize_t * 2)(8, 8)
        loop(ctypes.addressof(args), ctypes.addressof(dims), ctypes.addressof(steps), data)
        if np.array_equal(y, np.exp(x)):
            _NUMPY_EXP_LOOP = (fn, data)
    except Exception:
        _NUMPY_EXP_LOOP = None
    return _NUMPY_EXP_LOOP


class BuiltModel(object):
    """One model made by the native builder (advntr_built): host arrays + upload to the current device."""

    def __init__(self, handle, info=None):
        self._h = handle
        if info is None:
            info = np.zeros(6, np.int32)
            check(load().advntr_built_info(self._h, ptr(info)))
            info = info.tolist()
        self.m, self.silent_start, self.start_index, self.end_index, self.n_edges, self._names_bytes = info

    def arrays(self):
        a = dict(m=self.m, silent_start=self.silent_start, start_index=self.start_index, end_index=self.end_index,
                 in_ptr=np.zeros(self.m + 1, np.int32), in_src=np.zeros(self.n_edges, np.int32),
                 in_logp=np.zeros(self.n_edges, np.float64), emis_logp=np.zeros((self.silent_start, 4), np.float64),
                 state_class=np.zeros(self.m, np.uint16))
        check(load().advntr_built_export(self._h, ptr(a["in_ptr"]), ptr(a["in_src"]), ptr(a["in_logp"]),
                                         ptr(a["emis_logp"]), ptr(a["state_class"]), None))
        return a

    def names(self):
        buf = ctypes.create_string_buffer(self._names_bytes + 1)
        check(load().advntr_built_export(self._h, None, None, None, None, None, ctypes.addressof(buf)))
        return buf.raw[:self._names_bytes].decode("ascii").split("\n")

    def upload(self):
        require_gpu()
        h = load().advntr_built_upload(self._h)
        if not h:
            raise EngineError(ERR_ARG, last_error())
        return _UploadedModel(h, self.m, self.silent_start)

    def close(self):
        if getattr(self, "_h", None):
            load().advntr_built_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


BUILD_ALIGN_REPEATS, BUILD_EXP_STRIDED_LOOP = 1, 2


def align_repeats(units):
    """advntr_align_repeats: the built-in progressive aligner (stands in for the reference's `muscle` call; parity
    with muscle is not claimed).  Returns the rows in input order."""
    L = load()
    n = len(units)
    arr = (ctypes.c_char_p * max(n, 1))(*[u.encode("ascii") for u in units])
    width = ctypes.c_int32(0)
    cap = n * (2 * max([len(u) for u in units] or [0]) + 8)
    buf = ctypes.create_string_buffer(max(cap, 1))
    rc = L.advntr_align_repeats(arr, n, buf, cap, ctypes.byref(width))
    if rc == ERR_TOO_LARGE:
        cap = n * width.value
        buf = ctypes.create_string_buffer(max(cap, 1))
        rc = L.advntr_align_repeats(arr, n, buf, cap, ctypes.byref(width))
    if rc != OK:
        msg = last_error()
        if "not one of ACGT" in msg:
            raise NotImplementedError(msg)
        raise EngineError(rc, msg)
    w = width.value
    return [buf.raw[i * w:(i + 1) * w].decode("ascii") for i in range(n)]


def upload_built_models(built, threads=0):
    """advntr_built_upload_many: one device allocation and one copy for a whole list of BuiltModel."""
    require_gpu()
    n = len(built)
    handles = (ctypes.c_void_p * max(n, 1))(*[b._h for b in built])
    out = (ctypes.c_void_p * max(n, 1))()
    check(load().advntr_built_upload_many(handles, n, int(threads), out))
    return [_UploadedModel(out[i], built[i].m, built[i].silent_start) for i in range(n)]


def build_read_matchers(lefts, rights, repeat_lists, copies, max_error_rate, exp="numpy", threads=0, align=False):
    """advntr_build_read_matchers over n loci -> list of BuiltModel.  exp: "numpy" uses numpy.exp for the two
    probability round trips (what the reference calls; bit-identical parameters on the same machine) -- numpy's own
    inner loop called straight from the worker threads when it can be located (_numpy_exp_loop), a Python callback
    otherwise ("numpy-callback" forces the callback); "libm" lets the library use its own exp (<= 1 ulp away)."""
    L = load()
    n = len(lefts)
    keep = []

    def enc(strs):
        """char*[len(strs)] over ONE encoded copy of the strings (NUL-separated): an encode call and a ctypes object per
        string were a third of the Python time of a 6 719-locus build."""
        k = max(len(strs), 1)
        text = "\0".join(strs).encode("ascii") + b"\0"
        lens = np.fromiter(map(len, strs), dtype=np.int64, count=len(strs))
        if int(lens.sum()) + len(strs) != len(text):
            raise ValueError("sequence with an embedded NUL")
        at = np.zeros(k, np.uint64)
        base = ctypes.cast(ctypes.c_char_p(text), ctypes.c_void_p).value
        if len(strs):
            at[:len(strs)] = base + np.concatenate([[0], np.cumsum(lens[:-1] + 1)]).astype(np.uint64)
        keep.append((text, at))                     # alive until the call has returned
        return ptr(at)
    flat, off = [], np.zeros(n + 1, np.int32)
    for i, rows in enumerate(repeat_lists):
        flat.extend(rows)
        off[i + 1] = len(flat)
    cp = np.ascontiguousarray(copies, np.int32)
    out = (ctypes.c_void_p * max(n, 1))()
    if exp not in ("numpy", "numpy-callback", "libm"):
        raise ValueError("exp must be 'numpy', 'numpy-callback' or 'libm'")
    fn, user, flags = None, None, BUILD_ALIGN_REPEATS if align else 0
    if exp != "libm":
        loop = _numpy_exp_loop() if exp == "numpy" else None
        if loop is not None:                        # numpy's own inner loop, called by the worker threads directly
            fn, user = ctypes.c_void_p(loop[0]), ctypes.c_void_p(loop[1])
            flags |= BUILD_EXP_STRIDED_LOOP
        else:
            fn = ctypes.cast(_NUMPY_EXP, ctypes.c_void_p)
    rc = L.advntr_build_read_matchers(n, enc(lefts), enc(rights), enc(flat), ptr(off), ptr(cp), float(max_error_rate),
                                      fn, user, int(threads), flags, out)
    handles = list(out)[:n]
    info = np.zeros((max(n, 1), 6), np.int32)
    check(L.advntr_built_info_many(out, n, ptr(info)))          # (one call: a call and an array per model were 10 us each)
    built = [BuiltModel(h, i) if h else None for h, i in zip(handles, info.tolist())]
    if rc != OK:
        msg = last_error()
        for b in built:
            if b is not None:
                b.close()
        if "not one of ACGT" in msg or "need a multiple alignment" in msg:
            raise NotImplementedError(msg)
        raise EngineError(rc, msg)
    return built


def _handles(models):
    arr = (ctypes.c_void_p * len(models))(*[m.handle for m in models])
    return arr


def viterbi_batch(models, bases, read_off, read_model, flags=0, want_paths=False, want_summary=True):
    """One-shot scoring from host buffers.  Returns (logp, summary|None, paths|None).  With FLAG_BOTH_STRANDS the arrays
    describe the forward reads and every result holds 2n entries: entry n + i = the reverse complement of read i."""
    L = load()
    n_in = len(read_off) - 1
    bases = np.ascontiguousarray(bases, np.uint8)
    read_off = np.ascontiguousarray(read_off, np.int64)
    read_model = np.ascontiguousarray(read_model, np.int32)
    n = 2 * n_in if (flags & FLAG_BOTH_STRANDS) else n_in
    logp = np.zeros(n, np.float64)
    summ = np.zeros((n, SUMMARY_INTS), np.int32) if want_summary else None
    out_path = out_off = out_len = None
    if want_paths:
        flags |= FLAG_PATH
        caps = (read_off[1:] - read_off[:-1]) + np.array([models[i].m for i in read_model], np.int64) + 2
        if flags & FLAG_BOTH_STRANDS:
            caps = np.concatenate([caps, caps])
        out_off = np.zeros(n + 1, np.int64)
        np.cumsum(caps, out=out_off[1:])
        out_path = np.zeros(max(int(out_off[-1]), 1), np.int32)
        out_len = np.zeros(n, np.int32)
    check(L.advntr_viterbi_batch(_handles(models), len(models), ptr(bases), ptr(read_off), ptr(read_model), n_in,
                                 ptr(logp), ptr(summ), ptr(out_path), ptr(out_off), ptr(out_len), flags))
    paths = None
    if want_paths:
        paths = []
        for r in range(n):
            ln = int(out_len[r])
            if ln == -2:
                raise EngineError(ERR_ARG, "path of read %d exceeds the reference capacity n+m" % r)
            paths.append(out_path[out_off[r]:out_off[r] + ln].tolist() if ln > 0 else None)
    return logp, summ, paths


def forward_batch(models, bases, read_off, read_model, flags=0):
    L = load()
    n = len(read_off) - 1
    bases = np.ascontiguousarray(bases, np.uint8)
    read_off = np.ascontiguousarray(read_off, np.int64)
    read_model = np.ascontiguousarray(read_model, np.int32)
    logp = np.zeros(n, np.float64)
    check(L.advntr_forward_batch(_handles(models), len(models), ptr(bases), ptr(read_off), ptr(read_model), n,
                                 ptr(logp), flags))
    return logp


class DeviceBatch(object):
    """Device-resident batch (advntr_batch_*): upload once, run many times, fetch."""

    def __init__(self, models, bases, read_off, read_model, flags=0):
        L = load()
        self.models = list(models)
        n_in = len(read_off) - 1
        self.n_reads = 2 * n_in if (flags & FLAG_BOTH_STRANDS) else n_in        # calls = entries of every result array
        bases = np.ascontiguousarray(bases, np.uint8)
        read_off = np.ascontiguousarray(read_off, np.int64)
        read_model = np.ascontiguousarray(read_model, np.int32)
        self.flags = flags
        self._h = L.advntr_batch_create(_handles(self.models), len(self.models), ptr(bases), ptr(read_off),
                                        ptr(read_model), n_in, flags)
        if not self._h:
            msg = last_error()
            if "base code" in msg:
                raise ValueError("Symbol is not defined in a distribution (%s)" % msg)
            raise EngineError(ERR_ARG, msg)

    def run(self):
        check(load().advntr_batch_run(self._h))

    def sync(self):
        check(load().advntr_batch_sync(self._h))

    def reserve_next(self, n_workgroups):
        """The next run() leaves n workgroup slots free (what the multi-GPU gather asks for; one pass)."""
        check(load().advntr_batch_reserve_next(self._h, int(n_workgroups)))

    def run_timed(self, iters):
        ms = ctypes.c_float(0)
        check(load().advntr_batch_run_timed(self._h, iters, ctypes.byref(ms)))
        return ms.value

    def forward(self):
        """Model.log_probability over the resident reads: results replace the batch's log-probabilities (fetch())."""
        check(load().advntr_batch_forward(self._h))

    def forward_timed(self, iters):
        ms = ctypes.c_float(0)
        check(load().advntr_batch_forward_timed(self._h, iters, ctypes.byref(ms)))
        return ms.value

    def fetch(self):
        logp = np.zeros(self.n_reads, np.float64)
        summ = None if (self.flags & FLAG_NO_SUMMARY) else np.zeros((self.n_reads, SUMMARY_INTS), np.int32)
        check(load().advntr_batch_fetch(self._h, ptr(logp), ptr(summ)))
        return logp, summ

    def recruit(self, scaled_scores=None, min_repeat_bp=2):
        """VNTRFinder's keep / discard rule on the device (advntr_batch_recruit: strand choice, recruit_read, repeat_bp >
        min_repeat_bp) on the results of the last run(); scaled_scores: one per model (NaN / None / 0: no trained score) or
        None.  Returns (index of the forward read, logp, summary[8], reversed) of the survivors, in read order."""
        sc = None
        if scaled_scores is not None:
            sc = np.array([np.nan if (s is None or s == 0) else float(s) for s in scaled_scores], np.float64)
            if len(sc) != len(self.models):
                raise ValueError("recruit: %d scaled scores for %d models" % (len(sc), len(self.models)))
        n = ctypes.c_int32(0)
        check(load().advntr_batch_recruit(self._h, ptr(sc), int(min_repeat_bp), ctypes.byref(n)))
        k = n.value
        index, logp = np.zeros(k, np.int32), np.zeros(k, np.float64)
        summ, rev = np.zeros((k, SUMMARY_INTS), np.int32), np.zeros(k, np.uint8)
        check(load().advntr_batch_fetch_recruited(self._h, ptr(index), ptr(logp), ptr(summ), ptr(rev)))
        return index, logp, summ, rev.astype(bool)

    def device_bytes(self):
        return int(load().advntr_batch_device_bytes(self._h))

    def kernels(self):
        """[(kernel name as rocprofv3 shows it, reads, tiles)] in launch order (advntr_batch_info)."""
        return [k[:3] for k in self.kernel_info()]

    def kernel_info(self):
        """[(kernel name, reads, tiles, useful share of the sweeps' lane-steps or None)] in launch order: trellis cells of the
        reads over the cell slots the row-blocked sweeps' lane-steps offer (pipeline fill / drain, padding rows and lanes)."""
        buf = ctypes.create_string_buffer(4096)
        check(load().advntr_batch_info(self._h, ctypes.addressof(buf), 4096))
        out = []
        for line in buf.value.decode().splitlines():
            name, reads, tiles, useful = line.rsplit(" ", 3)
            out.append((name, int(reads), int(tiles), None if int(useful) < 0 else int(useful) / 1000.0))
        return out

    def result_ptrs(self):
        a, b = ctypes.c_void_p(0), ctypes.c_void_p(0)
        check(load().advntr_batch_result_ptrs(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def close(self):
        if getattr(self, "_h", None):
            load().advntr_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


GENOTYPE_ACCURACY_FILTER, GENOTYPE_HAPLOID = 1, 2


def genotype_illumina(summaries, locus_off, accuracy_filter=False, is_haploid=False, min_left=5, min_right=5, threads=0):
    """advntr_genotype_illumina: per-locus genotypes from the summary records of the selected reads, grouped by locus.
    Returns (genotype int32[n_loci][2] with -1 for None, max_prob float64[n_loci], counts int32[n_loci][3])."""
    summaries = np.ascontiguousarray(summaries, np.int32).reshape(-1, SUMMARY_INTS)
    locus_off = np.ascontiguousarray(locus_off, np.int64)
    n = len(locus_off) - 1
    geno = np.zeros((max(n, 0), 2), np.int32)
    prob = np.zeros(max(n, 0), np.float64)
    counts = np.zeros((max(n, 0), 3), np.int32)
    flags = (GENOTYPE_ACCURACY_FILTER if accuracy_filter else 0) | (GENOTYPE_HAPLOID if is_haploid else 0)
    check(load().advntr_genotype_illumina(ptr(summaries), ptr(locus_off), n, flags, int(min_left), int(min_right), int(threads),
                                          ptr(geno), ptr(prob), ptr(counts)))
    return geno, prob, counts


def genotype_observed(ru_counts, locus_off, accuracy_filter=False, is_haploid=False, threads=0):
    """advntr_genotype_observed: the tail of get_dominant_copy_numbers_from_spanning_reads (vntr_finder.py:568-580) per locus:
    RU counts of the spanning reads in scoring order -> (genotype int32[n_loci][2] with -1 for None, max_prob float64[n_loci]);
    a locus without reads gets (-1, -1) and probability 0."""
    ru = np.ascontiguousarray(ru_counts, np.int32)
    locus_off = np.ascontiguousarray(locus_off, np.int64)
    n = len(locus_off) - 1
    geno = np.zeros((max(n, 0), 2), np.int32)
    prob = np.zeros(max(n, 0), np.float64)
    flags = (GENOTYPE_ACCURACY_FILTER if accuracy_filter else 0) | (GENOTYPE_HAPLOID if is_haploid else 0)
    check(load().advntr_genotype_observed(ptr(ru), ptr(locus_off), n, flags, int(threads), ptr(geno), ptr(prob)))
    return geno, prob


def cut_pieces(codes, read_off, piece_read, begin, end, reverse, threads=0):
    """advntr_cut_pieces: pieces [begin, end) of encoded reads (reverse-complemented where reverse is set) as reads of their
    own -> (codes uint8, off int64)."""
    n = len(piece_read)
    pr = np.ascontiguousarray(piece_read, np.int32)
    b = np.ascontiguousarray(begin, np.int64)
    e = np.ascontiguousarray(end, np.int64)
    rv = np.ascontiguousarray(reverse, np.uint8)
    off = np.zeros(n + 1, np.int64)
    np.cumsum(e - b, out=off[1:])
    out = np.empty(int(off[n]), np.uint8)
    check(load().advntr_cut_pieces(ptr(codes), ptr(read_off), len(read_off) - 1, ptr(pr), ptr(b), ptr(e), ptr(rv), n,
                                   int(threads), ptr(off), ptr(out)))
    return out, off


def flank_align(reads, flanks, pair_read, pair_flank, encoded=None):
    """advntr_flank_align: local alignment (1, -1, -1, -1) of flanks[pair_flank[p]] to reads[pair_read[p]]; a pair_read of
    len(reads) + r stands for the reverse complement of read r (made on the device).  Returns (score, begin, end) int32
    arrays and the kernel time in ms.  Symbols outside ACGT match nothing.  encoded = (codes, read_off) of `reads` as
    encode_ascii gives them, when the caller has (or wants to keep) them."""
    L = load()
    require_gpu()
    # reads: case folding + encoding on host threads (N -> 254, other symbols -> 255; the library clamps them to "matches
    # nothing"); flanks are a few hundred bytes
    rb, roff = encoded if encoded is not None else encode_ascii(list(reads))[:2]
    code = _CODE.copy()
    code[code == 255] = 4
    flanks = list(flanks)
    foff = np.zeros(len(flanks) + 1, np.int32)
    np.cumsum(np.fromiter(map(len, flanks), dtype=np.int64, count=len(flanks)), out=foff[1:])
    fb = np.ascontiguousarray(code[np.frombuffer("".join(flanks).upper().encode("latin-1", "replace"), dtype=np.uint8)])
    pr = np.ascontiguousarray(pair_read, np.int32)
    pf = np.ascontiguousarray(pair_flank, np.int32)
    n = len(pr)
    score, begin, end = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    ms = ctypes.c_float(0)
    check(L.advntr_flank_align(ptr(rb), ptr(roff), len(reads), ptr(fb), ptr(foff), len(flanks), ptr(pr), ptr(pf), n,
                               ptr(score), ptr(begin), ptr(end), ctypes.byref(ms)))
    return score, begin, end, ms.value
