#!/usr/bin/env python3
"""Build the reference's vendored pomegranate (Cython 0.6.1) in a scratch directory.

TEST INFRASTRUCTURE, build container only.  Nothing here runs on the GPU box and
nothing it produces is committed: the only artefacts that enter the repo are the
golden vectors written by tests/golden/make_golden.py.

What it does:
  * copies /root/reference/{pomegranate,advntr} to a scratch dir (default
    /tmp/advntr_ref_build) -- the reference tree itself is read-only;
  * applies six signature-only edits so the 2017-era .pyx files compile under
    Cython 3 (`int n` -> `SIZE_t n` to agree with the .pxd declarations, and int
    temporaries for one dgemm call).  None touches hmm.pyx's Viterbi/forward/bake
    code; the only hmm.pyx edit is the `_summarize` signature at hmm.pyx:2620;
  * cythonizes with language_level=2 (hmm.pyx:2129 relies on C integer `/`).

The reference needs networkx==1.11 (setup.py:19, requirements.txt:6) and
biopython, neither of which is in this image.  oracle/tools/nx111 restates the
networkx-1.11 DiGraph subset that hmm.pyx calls (published algorithm, see that
file); oracle/tools/stubs/Bio only satisfies two import lines of
advntr/profile_hmm.py:6-7 (muscle is never invoked: goldens use pre-aligned repeats).
"""
import os, re, shutil, subprocess, sys

REF = "/root/reference"
DST = sys.argv[1] if len(sys.argv) > 1 else "/tmp/advntr_ref_build"


def sub_once(path, pattern, repl, count=1, expect=None):
    src = open(path).read()
    new, n = re.subn(pattern, repl, src, count=count)
    if n == 0 or (expect is not None and n != expect):
        raise SystemExit("patch failed for %s: %r matched %d times" % (path, pattern, n))
    open(path, "w").write(new)


def main():
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    os.makedirs(DST)
    shutil.copytree(os.path.join(REF, "pomegranate"), os.path.join(DST, "pomegranate"))
    shutil.copytree(os.path.join(REF, "advntr"), os.path.join(DST, "advntr"))
    pg = os.path.join(DST, "pomegranate")
    # the vendored __init__ installs pyximport; replace by plain imports of the built modules
    open(os.path.join(pg, "__init__.py"), "w").write(
        "from .hmm import *\nfrom .distributions import *\nfrom .base import *\n__version__ = '0.6.1'\n")
    # 1. base.pyx:270-271  _summarize(..., int n) -> SIZE_t n  (matches base.pxd:20-21)
    sub_once(os.path.join(pg, "base.pyx"),
             r"(cdef double _summarize\( self, double\* items,\s*double\* weights, )int n( \) nogil:)",
             r"\1SIZE_t n\2")
    # 2. distributions.pyx:1140, 2632, 2918  same signature change
    sub_once(os.path.join(pg, "distributions.pyx"),
             r"(cdef double _summarize\(self, double\* items, double\* weights, )int n( ?\) nogil:)",
             r"\1SIZE_t n\2", count=0, expect=3)
    # 3. distributions.pyx:2226 dgemm takes int*; d and n are SIZE_t there
    sub_once(os.path.join(pg, "distributions.pyx"),
             r"dgemm\('N', 'T', &d, &d, &n, &alpha, y, &d, items, &d, &beta, pair_sum, &d\)",
             "cdef int d_i = d\n\t\tcdef int n_i = n\n\t\tdgemm('N', 'T', &d_i, &d_i, &n_i, &alpha, y, &d_i, items, &d_i, &beta, pair_sum, &d_i)")
    # 4. hmm.pyx:2620 _summarize(..., int n) -> numpy.npy_intp n
    sub_once(os.path.join(pg, "hmm.pyx"),
             r"(cdef double _summarize\(self, double\* sequence, double\* weight, )int n(\) nogil:)",
             r"\1numpy.npy_intp n\2")
    setup_py = os.path.join(DST, "setup_ref.py")
    open(setup_py, "w").write(
        "from setuptools import setup\nfrom Cython.Build import cythonize\nimport numpy\n"
        "setup(name='pomegranate_ref', ext_modules=cythonize(['pomegranate/*.pyx'], language_level=2),\n"
        "      include_dirs=[numpy.get_include()])\n")
    subprocess.check_call([sys.executable, "setup_ref.py", "build_ext", "--inplace"], cwd=DST)
    print("reference pomegranate built in", DST)


if __name__ == "__main__":
    main()
