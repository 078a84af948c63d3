"""The C-ABI library loads and exports every symbol include/*.h declares (no GPU, no compute)."""
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from advntr_amd import _lib
    L = _lib.load()
    # advntr_hip.h = the drop-in C ABI; advntr_pyhost.h = the one optional helper for a CPython host
    header = open(os.path.join(ROOT, "include", "advntr_hip.h")).read()
    pyhost = open(os.path.join(ROOT, "include", "advntr_pyhost.h")).read()
    assert "PyObject" not in header and "pylist" not in header, "the C ABI header stays host-language neutral"
    declared = set(re.findall(r"\b(advntr_[a-z0-9_]+)\s*\(", header)) | set(re.findall(r"\b(advntr_[a-z0-9_]+)\s*\(", pyhost))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.advntr_version()


def test_header_constants_match_binding():
    from advntr_amd import _lib
    header = open(os.path.join(ROOT, "include", "advntr_hip.h")).read()
    consts = dict(re.findall(r"#define\s+(ADVNTR_[A-Z_]+)\s+(-?(?:0x[0-9A-Fa-f]+|\d+))u?\b", header))
    val = lambda k: int(consts[k], 0)
    assert val("ADVNTR_SUMMARY_INTS") == _lib.SUMMARY_INTS
    assert (val("ADVNTR_FLAG_PATH"), val("ADVNTR_FLAG_FORCE_GENERIC"), val("ADVNTR_FLAG_NO_SUMMARY")) == \
        (_lib.FLAG_PATH, _lib.FLAG_FORCE_GENERIC, _lib.FLAG_NO_SUMMARY)
    assert (val("ADVNTR_FLAG_STREAM"), val("ADVNTR_FLAG_ANTIDIAGONAL"), val("ADVNTR_FLAG_BOTH_STRANDS")) == \
        (_lib.FLAG_STREAM, _lib.FLAG_ANTIDIAGONAL, _lib.FLAG_BOTH_STRANDS)
    assert (val("ADVNTR_BUILD_ALIGN_REPEATS"), val("ADVNTR_BUILD_EXP_STRIDED_LOOP")) == (_lib.BUILD_ALIGN_REPEATS, _lib.BUILD_EXP_STRIDED_LOOP)
    assert (val("ADVNTR_GENOTYPE_ACCURACY_FILTER"), val("ADVNTR_GENOTYPE_HAPLOID")) == (_lib.GENOTYPE_ACCURACY_FILTER, _lib.GENOTYPE_HAPLOID)
    for k in ("EMIT", "MATCH", "SUFFIX", "PREFIX", "UNIT_START", "UNIT_END", "SKIP", "FIX", "BASE_VALID"):
        assert val("ADVNTR_SC_" + k) == getattr(_lib, "SC_" + k)
    assert val("ADVNTR_ERR_SYMBOL") == _lib.ERR_SYMBOL


def test_scoring_without_gpu_fails_loudly():
    """No CPU fallback: on a box without a HIP device the product path raises."""
    from advntr_amd import _lib, workloads
    import numpy as np
    if _lib.load().advntr_device_count() > 0:
        pytest.skip("a GPU is present")
    locus = workloads.make_locus(np.random.default_rng(1), 8, 5, 2)
    with pytest.raises(_lib.EngineError):
        locus.model.viterbi("ACGTACGT")


def test_product_never_imports_the_oracle():
    """oracle/ is the checker: nothing under advntr_amd/ (Python or HIP sources) may import, include or load it."""
    pkg = os.path.join(ROOT, "advntr_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|oracle/", text, re.M):
                    offenders.append(os.path.join(dirpath, fn))
    assert offenders == []
