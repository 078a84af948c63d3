#!/usr/bin/env python3
"""Generate tests/golden/read_selection.json.gz by RUNNING THE REFERENCE's VNTRFinder.select_illumina_reads
(advntr/vntr_finder.py:701-767) -- mapped reads over a locus + keyword-filtered unmapped reads -- with the alignment file
served by a stand-in for pysam.AlignmentFile that holds plain records (pysam is absent here): .references, .head(n),
.fetch(chromosome, start, end) = mapped records overlapping [start, end) in file order.  is_low_quality_read
(advntr/utils.py:20-38) and get_reference_genome_of_alignment_file (advntr/sam_utils.py:32-39) are compiled from the
reference files like the methods.  TEST INFRASTRUCTURE; only data is written: the SAM text, the unmapped reads and what
the reference selected.

    python oracle/tools/build_reference.py && python tests/golden/make_selection_golden.py
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
from pomegranate import HiddenMarkovModel             # noqa: E402


class _Seq(object):
    def __init__(self, s):
        self.s = s

    def reverse_complement(self):
        return self.s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def functions_of(path, names):
    tree = ast.parse(open(path).read())
    return [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]


def load(fake_file):
    path = "/root/reference/advntr/vntr_finder.py"
    tree = ast.parse(open(path).read())
    wanted = {"select_illumina_reads", "process_unmapped_read", "recruit_read", "get_min_score_to_select_a_read",
              "get_vntr_matcher_hmm", "build_vntr_matcher_hmm", "get_copies_for_hmm", "get_alignment_file_read_mode"}
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VNTRFinder"][0]
    body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    for fn in body:
        fn.decorator_list = [d for d in fn.decorator_list if isinstance(d, ast.Name) and d.id == "staticmethod"]
    helpers = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "SelectedRead"]
    helpers += functions_of("/root/reference/advntr/utils.py", {"is_low_quality_read"})
    helpers += functions_of("/root/reference/advntr/sam_utils.py", {"get_reference_genome_of_alignment_file"})
    mod = ast.Module(body=helpers + [ast.ClassDef(name="VNTRFinder", bases=[], keywords=[], body=body, decorator_list=[])],
                     type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(vars(hmm_utils))
    import logging
    from multiprocessing import Value
    ns.update(vars(settings))
    ns.update(logging=logging, Value=Value, Seq=_Seq, settings=settings, os=os, Model=HiddenMarkovModel,
              pysam=types.SimpleNamespace(AlignmentFile=lambda *a, **k: fake_file), load_model=None)
    exec(compile(mod, path, "exec"), ns)
    return ns["VNTRFinder"]


class FakeAlignmentFile(object):
    def __init__(self, references, reads):
        self.references, self.reads = references, reads

    def head(self, n):
        return self.reads[:n]

    def fetch(self, reference, start, end):
        for r in self.reads:
            if r.reference_name == reference and not r.is_unmapped and r.reference_start < end and r.reference_end > start:
                yield r


def main():
    rng = np.random.default_rng(90210)
    dna = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    settings.MAX_ERROR_RATE = 0.05
    settings.USE_TRAINED_HMMS = False
    hmm_utils.build_profile_hmm_for_repeats = \
        lambda repeats, error_rate: hmm_utils.build_profile_hmm_pseudocounts_for_alignment(error_rate, repeats)
    cases = []
    for name, scaled in (("default_rule", 0), ("trained_score", -1.1)):
        pattern, left, right = dna(17), dna(400), dna(400)
        segs = [pattern] * 4
        start = 5000
        ref_seq = dna(start - 400) + left + "".join(segs) + right + dna(2000)
        sample = ref_seq[:start] + pattern * 6 + ref_seq[start + 4 * 17:]            # the donor carries 6 copies
        records, sam_lines = [], ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:chr5\tLN:%d" % len(ref_seq), "@SQ\tSN:chrX\tLN:1000"]
        positions = sorted(int(x) for x in rng.integers(start - 260, start + 4 * 17 + 100, 70))
        for k, pos in enumerate(positions):
            seq = sample[pos:pos + 150]
            qual = [40] * 150
            flag, mapq = 0, 60
            kind = k % 14
            if kind == 3:
                seq = seq[:70] + "N" + seq[71:]
            if kind == 5:
                flag = 0x400
            if kind == 7:
                mapq = 0
            if kind == 9:
                for q in range(20, 40):
                    qual[q] = 5
            if kind == 11:
                seq, qual = seq[:120], qual[:120]
            if kind == 12:
                qual[100] = qual[101] = qual[102] = qual[103] = 3
            if kind == 13:
                seq = seq.lower()
            r = types.SimpleNamespace(query_name="m%d" % k, flag=flag, reference_name="chr5", reference_start=pos,
                                      reference_end=pos + len(seq), mapq=mapq, seq=seq, query_qualities=qual,
                                      is_unmapped=False, is_duplicate=bool(flag & 0x400))
            records.append(r)
            sam_lines.append("\t".join([r.query_name, str(flag), "chr5", str(pos + 1), str(mapq), "%dM" % len(seq), "*", "0", "0",
                                        seq, "".join(chr(q + 33) for q in qual)]))
        unmapped = []
        for k in range(30):
            st = int(rng.integers(start - 120, start + 20))
            s = sample[st:st + 150]
            if k % 2:
                s = _Seq(s).reverse_complement()
            if k % 7 == 6:
                s = s[:100]
            if k % 9 == 8:
                s = dna(150)
            unmapped.append(s)
        fake = FakeAlignmentFile(["chr5", "chrX"], records)
        Finder = load(fake)
        f = Finder.__new__(Finder)
        f.reference_vntr = types.SimpleNamespace(id=3, pattern=pattern, chromosome="chr5", start_point=start, scaled_score=scaled,
                                                 left_flanking_region=left, right_flanking_region=right,
                                                 get_repeat_segments=lambda s=segs: s, get_length=lambda s=segs: sum(len(x) for x in s))
        f.reference_filename = None
        f.min_repeat_bp_to_add_read = 2
        f.min_repeat_bp_to_count_repeats = 2
        selected = f.select_illumina_reads("sample.bam", [types.SimpleNamespace(seq=s) for s in unmapped])
        cases.append({"name": name, "pattern": pattern, "left": left, "right": right, "repeat_segments": segs, "start_point": start,
                      "scaled_score": scaled, "sam": "\n".join(sam_lines) + "\n", "unmapped": unmapped,
                      "selected": [{"sequence": s.sequence, "logp": s.logp, "query_name": getattr(s, "query_name", None),
                                    "mapq": s.mapq, "is_mapped": s.is_mapped} for s in selected]})
        print(name, "mapped records", len(records), "unmapped", len(unmapped), "-> selected", len(selected))
    with gzip.open(os.path.join(HERE, "read_selection.json.gz"), "wt") as fh:
        json.dump({"cases": cases}, fh)


if __name__ == "__main__":
    main()
