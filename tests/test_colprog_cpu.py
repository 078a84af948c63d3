"""CPU check of the column-program compiler (advntr_amd/csrc/column_program.h): every read-matcher golden
must compile to a column program whose scalar evaluation (oracle/colprog_check.cpp, a lane-by-lane mirror
of the HIP kernel) reproduces the reference's log-probabilities bit-exactly and its Viterbi paths."""
import ctypes
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import GENERIC_GOLDENS, READ_MATCHER_GOLDENS, ROOT, load_golden
from oracle.oracle import OracleModel, encode


def checker():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libcolprog_check.so"])
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libcolprog_check.so"))
    L.colprog_create.restype = ctypes.c_void_p
    L.colprog_create.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p] * 4
    L.colprog_destroy.argtypes = [ctypes.c_void_p]
    L.colprog_valid.argtypes = [ctypes.c_void_p]
    L.colprog_why.restype = ctypes.c_char_p
    L.colprog_why.argtypes = [ctypes.c_void_p]
    L.colprog_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.colprog_forward.restype = ctypes.c_double
    L.colprog_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.colprog_viterbi.restype = ctypes.c_double
    L.colprog_viterbi.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                  ctypes.c_void_p]
    return L


def compile_golden(L, g):
    O = OracleModel.from_golden(g)
    in_ptr, in_src, in_logp, _ = O.csr()
    gm = g["model"]
    h = L.colprog_create(O.m, gm["silent_start"], gm["start_index"], gm["end_index"], len(in_src),
                         in_ptr.ctypes.data, in_src.ctypes.data, in_logp.ctypes.data, O.emis.ctypes.data)
    return h, O


@pytest.mark.parametrize("name", READ_MATCHER_GOLDENS)
def test_read_matchers_compile_and_match(name):
    L = checker()
    g = load_golden(name)
    h, O = compile_golden(L, g)
    assert L.colprog_valid(h), L.colprog_why(h)
    stats = np.zeros(8, np.int32)
    L.colprog_stats(h, stats.ctypes.data)
    assert stats[6] <= 96 * 1024
    cap = 4096
    path = np.zeros(cap, np.int32)
    n_checked = 0
    for r in g["reads"]:
        if len(r["seq"]) < 1:
            continue
        codes = encode(r["seq"])
        ln = ctypes.c_int(0)
        logp = L.colprog_viterbi(h, codes.ctypes.data, len(codes), path.ctypes.data, cap, ctypes.byref(ln))
        assert logp == r["logp"] or (math.isinf(logp) and math.isinf(r["logp"])), (name, r["seq"], logp, r["logp"])
        if r["path"] is not None:
            assert path[:ln.value][::-1].tolist() == r["path"], (name, r["seq"])
        n_checked += 1
    assert n_checked > 10
    L.colprog_destroy(h)


@pytest.mark.parametrize("name", GENERIC_GOLDENS)
def test_generic_models_are_refused_not_miscompiled(name):
    """A model that does not fit the stencil must simply have no column program (it runs on the generic
    kernel) -- or, if it happens to fit, must still reproduce the reference."""
    L = checker()
    g = load_golden(name)
    h, O = compile_golden(L, g)
    if not L.colprog_valid(h):
        assert len(L.colprog_why(h)) > 0
    L.colprog_destroy(h)


def test_random_loci_vs_oracle():
    """Builder robustness over shapes: random flank / pattern / copies / error rates, multi-row profiles."""
    from advntr_amd import workloads
    L = checker()
    rng = np.random.default_rng(77)
    for trial in range(12):
        flank = int(rng.integers(3, 40))
        plen = int(rng.integers(2, 25))
        copies = int(rng.integers(1, 7))
        loc = workloads.make_locus(rng, flank, plen, copies, float(rng.choice([0.05, 0.3])), n_units=int(rng.integers(1, 5)))
        a = loc.model.baked_arrays()
        h = L.colprog_create(a["m"], a["silent_start"], a["start_index"], a["end_index"], len(a["in_src"]),
                             a["in_ptr"].ctypes.data, a["in_src"].ctypes.data, a["in_logp"].ctypes.data,
                             a["emis_logp"].ctypes.data)
        assert L.colprog_valid(h), L.colprog_why(h)
        edges = [(int(a["in_src"][k]), l, float(a["in_logp"][k]))
                 for l in range(a["m"]) for k in range(a["in_ptr"][l], a["in_ptr"][l + 1])]
        O = OracleModel(a["m"], a["silent_start"], a["start_index"], a["end_index"], edges, a["emis_logp"])
        reads = workloads.make_reads(rng, loc, 25, int(rng.integers(1, 120)))
        path = np.zeros(8192, np.int32)
        for s in reads:
            codes = encode(s)
            ln = ctypes.c_int(0)
            logp = L.colprog_viterbi(h, codes.ctypes.data, len(codes), path.ctypes.data, 8192, ctypes.byref(ln))
            olp, opath = O.viterbi(s)
            assert logp == olp, (trial, s)
            assert path[:ln.value][::-1].tolist() == opath, (trial, s)
        L.colprog_destroy(h)


@pytest.mark.parametrize("name", ["toy_f8_l5_c2", "s300_f30_l12_c3", "msa8_f50_c4"])
def test_forward_on_column_program_matches_reference(name):
    """Sum-product evaluation of the column program (fold order of the forward kernel) vs the reference's
    log_probability values: same libm, so agreement is at rounding level (the fold order differs only for the
    fan-in states)."""
    L = checker()
    g = load_golden(name)
    h, O = compile_golden(L, g)
    assert L.colprog_valid(h)
    n_checked = 0
    for r in g["reads"]:
        if "forward_logp" not in r or len(r["seq"]) < 1:
            continue
        codes = encode(r["seq"])
        got = L.colprog_forward(h, codes.ctypes.data, len(codes))
        assert abs(got - r["forward_logp"]) <= 1e-10 * max(1.0, abs(r["forward_logp"])), (name, r["seq"])
        n_checked += 1
    assert n_checked > 10
    L.colprog_destroy(h)
