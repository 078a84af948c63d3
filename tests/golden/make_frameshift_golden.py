#!/usr/bin/env python3
"""Generate tests/golden/frameshift.json.gz by RUNNING THE REFERENCE: reads of a locus with and without a planted
one-base insertion are scored by the vendored pomegranate build, selected by VNTRFinder.process_unmapped_read and
handed to VNTRFinder.find_frameshift_from_selected_reads (advntr/vntr_finder.py:235-309); method bodies compiled at
run time from the reference file with `ast`, nothing of them stored.  TEST INFRASTRUCTURE; only data is written.

    python oracle/tools/build_reference.py && python tests/golden/make_frameshift_golden.py
"""
import ast
import gzip
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.environ.get("ADVNTR_REF_BUILD", "/tmp/advntr_ref_build")
sys.path[:0] = [os.path.join(REPO, "oracle", "tools", "nx111"), os.path.join(REPO, "oracle", "tools", "stubs"), REF_BUILD]

import numpy as np                                    # noqa: E402
from advntr import settings, hmm_utils                # noqa: E402


class _Seq(object):
    def __init__(self, s):
        self.s = s

    def reverse_complement(self):
        return self.s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def load_methods():
    path = "/root/reference/advntr/vntr_finder.py"
    tree = ast.parse(open(path).read())
    wanted = {"process_unmapped_read", "recruit_read", "identify_frameshift", "find_frameshift_from_selected_reads",
              "get_min_score_to_select_a_read"}
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VNTRFinder"][0]
    body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    for fn in body:
        fn.decorator_list = []
    helpers = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SelectedRead"]
    mod = ast.Module(body=helpers + [ast.ClassDef(name="VNTRFinder", bases=[], keywords=[], body=body, decorator_list=[])],
                     type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(vars(hmm_utils))
    import logging
    ns.update(logging=logging, Seq=_Seq)
    exec(compile(mod, path, "exec"), ns)
    return ns["VNTRFinder"]


def main():
    rng = np.random.default_rng(808)
    dna = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    Finder = load_methods()
    settings.MAX_ERROR_RATE = 0.05
    hmm_utils.build_profile_hmm_for_repeats = \
        lambda repeats, error_rate: hmm_utils.build_profile_hmm_pseudocounts_for_alignment(error_rate, repeats)
    cases = []
    # "both": every read carries the insertion (count >= per-base coverage => frameshift); the half-coverage cases come
    # out None in the reference because scipy's binom.pmf is nan for the non-integer coverage it is given -- recorded as is
    for name, plen, ru, planted in (("insertion_both", 12, 6, "ins"), ("insertion_half", 12, 6, "ins"), ("clean", 12, 6, None),
                                    ("deletion_both", 15, 5, "del")):
        pattern, left, right = dna(plen), dna(300), dna(300)
        segs = [pattern] * ru
        copies = int(round(150.0 / plen + 0.5))
        model = hmm_utils.get_read_matcher_model(left[-150:], right[:150], segs, copies)
        units = list(segs)
        if planted == "ins":
            units[2] = pattern[:5] + "T" + pattern[5:]
        elif planted == "del":
            units[3] = pattern[:7] + pattern[8:]
        allele = left + "".join(units) + right
        normal = left + "".join(segs) + right
        reads = []
        for k in range(150 if name.endswith('both') else 90):
            src = allele if (planted and (k % 2 == 0 or name.endswith('both'))) else normal
            st = int(rng.integers(150, len(src) - 300))
            r = src[st:st + 150]
            r = "".join(("ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.004 else ch) for ch in r)
            reads.append(r if rng.random() < 0.5 else _Seq(r).reverse_complement())
        f = Finder.__new__(Finder)
        f.reference_vntr = types.SimpleNamespace(id=1, pattern=pattern, scaled_score=0, left_flanking_region=left, right_flanking_region=right, get_length=lambda n=plen * ru: n)
        f.min_repeat_bp_to_add_read = 2
        f.min_repeat_bp_to_count_repeats = 2
        selected = []
        vbp = types.SimpleNamespace(value=0)
        for r in reads:
            f.process_unmapped_read(None, r, model, None, vbp, selected, True)
        result = f.find_frameshift_from_selected_reads(selected)
        cases.append({"name": name, "pattern": pattern, "left": left, "right": right, "repeat_segments": segs,
                      "copies": copies, "reads": reads, "vntr_length": plen * ru,
                      "selected": [[s.sequence, [st.name for _, st in s.vpath[1:-1]]] for s in selected],
                      "frameshift": result})
        print(name, "selected", len(selected), "->", result)
    with gzip.open(os.path.join(HERE, "frameshift.json.gz"), "wt") as fh:
        json.dump({"cases": cases}, fh)


if __name__ == "__main__":
    main()
