"""The product's host-only C++ (native model builder, repeat aligner) under AddressSanitizer + UBSan on the CPU build:
GPU sanitizers are not available on the pool, so this is where memory errors of the host code would show."""
import os
import subprocess

from conftest import ROOT


import pytest


@pytest.mark.parametrize("arena", ["bump", "plain"])
def test_host_cpp_under_asan_ubsan(tmp_path, arena):
    """arena = "plain": the builder's arrays as blocks of their own from the general allocator (-DADVNTR_ARENA_PLAIN), where the
    sanitizer sees an access past the end of one; "bump": the shipped per-thread bump arena."""
    exe = str(tmp_path / "asan_host")
    src = os.path.join(ROOT, "tests", "native", "asan_host.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-pthread", "-ldl"] + (["-DADVNTR_ARENA_PLAIN"] if arena == "plain" else []) +
                          ["-o", exe, src])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert out.stdout.decode().startswith("ok ")
