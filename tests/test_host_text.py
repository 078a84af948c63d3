"""Host-side text helpers of the C ABI (no GPU): advntr_line_index, advntr_encode_ascii / advntr_encode_spans, and the
builder's exp routes (numpy's own inner loop called from the worker threads vs the Python callback)."""
import numpy as np
import pytest

from advntr_amd import _lib


def test_line_index_equals_split():
    rng = np.random.default_rng(4)
    cases = [b"", b"\n", b"A", b"A\n", b"\n\nA\n\n", b">r1\nACGT\n>r2\nAC\n", b">r1\nACGT\n>r2\nAC"]
    for _ in range(20):
        n = int(rng.integers(0, 3000))
        cases.append(bytes(rng.choice(np.frombuffer(b"ACGT\n\n>x", np.uint8), n).tolist()))
    cases.append((b">name\n" + b"ACGT" * 50 + b"\n") * 40000)             # several threads' worth
    for text in cases:
        starts = _lib.line_index(text)
        lines = text.split(b"\n")
        if lines and lines[-1] == b"":
            lines.pop()
        assert len(starts) - 1 == len(lines), text[:40]
        assert int(starts[-1]) == len(text)
        at = 0
        for i, l in enumerate(lines[:2000]):
            assert int(starts[i]) == at
            at += len(l) + 1


def test_encode_ascii_and_spans():
    seqs = ["ACGT", "acgt", "ACGNNT", "", "AC-T", "nnn", "TTTT" * 100]
    codes, off, bad = _lib.encode_ascii(seqs)
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).tolist()
    assert bad.tolist() == [0, 0, 1, 0, 2, 1, 0]
    assert codes[:8].tolist() == [0, 1, 2, 3, 0, 1, 2, 3] and codes[11] == 254 and codes[off[4] + 2] == 255
    text = b">a\nACGT\n>b\nacgN\n>c\nAC-T"
    starts = _lib.line_index(text)
    ends = (starts[1:] - 1).copy()
    ends[-1] = len(text)
    c, o, b = _lib.encode_spans(text, starts[1::2], ends[1::2], case_sensitive=True)
    assert o.tolist() == [0, 4, 8, 12] and b.tolist() == [0, 2, 2]            # lower case is "another symbol" for the filter
    assert c.tolist() == [0, 1, 2, 3, 255, 255, 255, 254, 0, 1, 255, 3]
    c2, _, b2 = _lib.encode_spans(text, starts[1::2], ends[1::2], case_sensitive=False)
    assert c2[4:8].tolist() == [0, 1, 2, 254] and b2.tolist() == [0, 1, 2]
    with pytest.raises(_lib.EngineError):
        _lib.check(_lib.load().advntr_encode_ascii(b"ACGT", np.array([0, 3, 2], np.int64).ctypes.data, 2, 1, None, None))


def test_long_reads_are_encoded_out_of_their_own_buffers():
    """advntr_encode_texts (one pointer per read: the strings' own buffers) == advntr_encode_ascii over the joined text; a str
    that is not ASCII and a bytes object take the same route; encode_ascii picks the route by mean read length."""
    rng = np.random.default_rng(8)
    alphabet = np.array(list("ACGTacgtNn-"))
    seqs = ["".join(rng.choice(alphabet[:11 if k % 7 == 0 else (10 if k % 3 == 0 else 8)], int(rng.integers(0, 9000)))) for k in range(120)]
    seqs[5] = "ACGT\u00e9" * 900
    seqs[9] = "ACGT" * 600 + "\ud800" + "ACGT" * 100            # a lone surrogate (no UTF-8 form at all) in a long read
    as_text = list(seqs)
    seqs[7] = seqs[7].encode()
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    c, o, b = _lib._encode_texts(seqs, off)
    limit, _lib.LONG_TEXT_MEAN = _lib.LONG_TEXT_MEAN, 1 << 60
    try:
        c2, o2, b2 = _lib.encode_ascii(as_text)
    finally:
        _lib.LONG_TEXT_MEAN = limit
    assert np.array_equal(c, c2) and np.array_equal(o, o2) and np.array_equal(b, b2)
    assert b[5] == 2 and b[9] == 2 and set(b.tolist()) == {0, 1, 2}
    c3, _, b3 = _lib.encode_ascii(as_text)                         # mean length above LONG_TEXT_MEAN: the pointer route
    assert off[-1] >= _lib.LONG_TEXT_MEAN * len(seqs) and np.array_equal(c3, c2) and np.array_equal(b3, b2)
    with pytest.raises(_lib.EngineError):
        _lib.check(_lib.load().advntr_encode_texts(None, 2, 0, 1, off.ctypes.data, c.ctypes.data, None))


def test_builder_exp_routes_agree():
    """numpy.exp's inner loop located in the ufunc object and called by the worker threads == the ctypes callback around
    numpy.exp (bit for bit); libm's exp differs in the last bit of some parameters (why the reference's route matters)."""
    from advntr_amd import workloads
    rng = np.random.default_rng(11)
    loci = [workloads.make_locus(rng, int(rng.integers(20, 160)), int(rng.integers(5, 60)), int(rng.integers(2, 8)),
                                 n_units=int(rng.integers(1, 6))) for _ in range(24)]
    args = ([l.left for l in loci], [l.right for l in loci], [list(l.units) for l in loci], [l.copies for l in loci], 0.05)
    by = {}
    for route in ("numpy", "numpy-callback", "libm"):
        built = _lib.build_read_matchers(*args, exp=route, threads=4)
        by[route] = [b.arrays()["in_logp"] for b in built]
    if _lib._numpy_exp_loop() is None:
        pytest.skip("numpy's exp loop could not be located in this numpy build; the callback route is in use")
    assert all(np.array_equal(a, b) for a, b in zip(by["numpy"], by["numpy-callback"]))
    worst = max(float(np.max(np.abs(a - b)[np.isfinite(a)] / np.spacing(np.abs(b[np.isfinite(a)])))) for a, b in zip(by["libm"], by["numpy"]))
    assert worst <= 4


def test_host_threads_follow_the_cpu_quota_and_the_override():
    """advntr_host_threads: what the bulk host-side calls use when asked for 0 threads -- the hardware threads cut down to the
    control group's CPU quota (a container on a big host is given a few cores; more worker threads than that have the whole
    process stopped by the scheduler for the rest of every accounting period), ADVNTR_HOST_THREADS overriding."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r)\nimport __graft_entry__ as e; e.build()\n"
            "from advntr_amd import _lib\nprint('threads', _lib.load().advntr_host_threads())\n" % ROOT)

    def ask(env):
        out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, check=True, timeout=600).stdout.decode()
        return int(out.split("threads")[-1])
    plain = {k: v for k, v in os.environ.items() if k != "ADVNTR_HOST_THREADS"}
    n = ask(plain)
    assert 1 <= n <= (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = -(-int(q) // int(period))
    except (OSError, ValueError):
        pass
    if quota:
        assert n <= quota
    assert ask(dict(plain, ADVNTR_HOST_THREADS="5")) == 5
    assert ask(dict(plain, ADVNTR_HOST_THREADS="junk")) == n


def test_cut_pieces_equals_slicing_strings():
    """advntr_cut_pieces on the codes of whole reads == encoding what vntr_finder._spanning_piece cuts out of the strings (upper
    case, reverse complement of the reverse strand, N and foreign symbols -> 255), incl. empty pieces, pieces running to the
    end of a read and lower-case reads; bad pieces are refused."""
    from advntr_amd import vntr_finder
    rng = np.random.default_rng(5)
    reads = ["".join(rng.choice(list("ACGTacgtNnX"), p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .01, .01], size=int(n)))
             for n in rng.integers(1, 400, 60)]
    codes, off, _ = _lib.encode_ascii(reads)
    n_pieces = 500
    rd = rng.integers(0, len(reads), n_pieces).astype(np.int32)
    lens = np.diff(off)[rd]
    lb = (rng.random(n_pieces) * (lens + 20)).astype(np.int64)
    rb_end = lb + (rng.random(n_pieces) * 150).astype(np.int64)
    lb[:5], rb_end[:5] = 0, 0                                            # empty pieces
    rev = rng.integers(0, 2, n_pieces).astype(np.uint8)
    want = [vntr_finder._spanning_piece(reads[r], int(b), int(e), bool(v)) for r, b, e, v in zip(rd, lb, rb_end, rev)]
    n = lens
    begin, end = np.minimum(lb, n), np.maximum(np.minimum(rb_end, n), np.minimum(lb, n))
    src_b, src_e = np.where(rev != 0, n - end, begin), np.where(rev != 0, n - begin, end)
    got, goff = _lib.cut_pieces(codes, off, rd, src_b, src_e, rev, threads=3)
    wc, woff = _lib.encode_reads(want)
    assert np.array_equal(goff, woff) and np.array_equal(got, wc)
    assert len(got) > 10000 and (got == 255).any() and (rev != 0).sum() > 100
    with pytest.raises(_lib.EngineError):
        _lib.cut_pieces(codes, off, np.array([0], np.int32), np.array([0]), np.array([int(lens.max()) + 500]), np.array([0], np.uint8))
    with pytest.raises(_lib.EngineError):
        _lib.cut_pieces(codes, off, np.array([len(reads)], np.int32), np.array([0]), np.array([1]), np.array([0], np.uint8))


def test_encode_ascii_list_walk_equals_general_route():
    """A list of ASCII str is encoded straight out of the strings' buffers (advntr_pylist_texts walks the list in one library
    call); a tuple of the same strings takes the general route (one joined text): same codes, offsets and flags -- empty strings,
    lower case, N, foreign symbols included; a list holding a non-ASCII str, a lone surrogate or a non-str falls back by itself."""
    rng = np.random.default_rng(8)
    reads = ["".join(rng.choice(list("ACGTacgtNnX-"), int(n))) for n in rng.integers(0, 300, 400)] + ["", "ACGT", "n"]
    assert _lib._list_texts(reads) is not None
    a, b = _lib.encode_ascii(reads, 3), _lib.encode_ascii(tuple(reads), 3)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert set(a[2].tolist()) == {0, 1, 2} and len(a[0]) == sum(map(len, reads))
    for odd in (["ACGT", "ACGéT"], ["AC\ud800GT"], ["ACGT", b"ACGT"], ["ACGT", None]):
        assert _lib._list_texts(odd) is None
    c, d = _lib.encode_ascii(["ACGT", "ACGéT", "AC\ud800GT"], 1), _lib.encode_ascii(("ACGT", "ACGéT", "AC\ud800GT"), 1)
    assert all(np.array_equal(x, y) for x, y in zip(c, d)) and c[2].tolist() == [0, 2, 2]
    assert _lib._list_texts((1, 2)) is None and len(_lib._list_texts([])[0]) == 0


def test_text_reads_drop_n_and_refuse_other_symbols():
    """Reads holding N are dropped as the reference drops them (vntr_finder.py:237), any other foreign symbol raises as its viterbi
    does (hmm.pyx:72,79) -- for reads taken as spans of a text exactly as for lists of str."""
    from advntr_amd import vntr_finder
    text = b">a\nACGTACGTAC\n>b\nACGTNCGTAC\n>c\nacgtacgtac\n"
    starts, ends = np.array([3, 17, 31]), np.array([13, 27, 41])
    tr = vntr_finder.TextReads(text, starts, ends, np.array([0, 2, 3]))
    prep = tr.prepare(0, 2)
    assert prep["locus"].tolist() == [0, 1] and prep["index"].tolist() == [0, 0] and prep["lens"].tolist() == [10, 10]
    assert np.array_equal(prep["bases"][:10], prep["bases"][10:])              # (case folded like the str route)
    assert tr.read_lists() == [["ACGTACGTAC", "ACGTNCGTAC"], ["acgtacgtac"]] and len(tr) == 2
    assert tr.prepare(1, 1) is None
    bad = vntr_finder.TextReads(b">x\nACGT-CGT\n", np.array([3]), np.array([11]), np.array([0, 1]))
    with pytest.raises(ValueError):
        bad.prepare(0, 1)
