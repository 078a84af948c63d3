#!/usr/bin/env python3
"""Generate tests/golden/threshold_training.json.gz by RUNNING THE REFERENCE's recruitment-threshold training
(VNTRFinder.train_classifier_threshold and the methods it calls, advntr/vntr_finder.py:902-1021) in this container.
TEST INFRASTRUCTURE; only data is written (inputs + what the reference's methods return).

vntr_finder.py cannot be imported here (keras, pysam, biopython absent): the method bodies are compiled at run time
from the reference file with `ast` and executed unchanged against the vendored pomegranate build
(oracle/tools/build_reference.py); nothing of them is stored.  Bio.SeqIO.parse is served by a FASTA reader below
(records with .id = first word of the header and .seq).

    python oracle/tools/build_reference.py && python tests/golden/make_training_golden.py
"""
import ast
import gzip
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.environ.get("ADVNTR_REF_BUILD", "/tmp/advntr_ref_build")
sys.path[:0] = [os.path.join(REPO, "oracle", "tools", "nx111"), os.path.join(REPO, "oracle", "tools", "stubs"), REF_BUILD]

import numpy as np                                    # noqa: E402
from advntr import settings, hmm_utils                # noqa: E402
from pomegranate import HiddenMarkovModel             # noqa: E402


class _Rec(object):
    def __init__(self, rid, seq):
        self.id, self.seq = rid, seq


class _SeqIO(object):
    @staticmethod
    def parse(handle, fmt):
        assert fmt == 'fasta'
        rid, chunks = None, []
        for line in handle:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if rid is not None:
                    yield _Rec(rid, "".join(chunks))
                rid, chunks = line[1:].split()[0], []
            else:
                chunks.append(line)
        if rid is not None:
            yield _Rec(rid, "".join(chunks))


def load_methods():
    path = "/root/reference/advntr/vntr_finder.py"
    tree = ast.parse(open(path).read())
    wanted = {"train_classifier_threshold", "find_hmm_score_of_simulated_reads", "simulate_false_filtered_reads",
              "simulate_true_reads", "find_recruitment_score_threshold", "process_unmapped_read", "recruit_read",
              "get_keywords_for_filtering", "get_vntr_matcher_hmm", "build_vntr_matcher_hmm", "get_copies_for_hmm"}
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VNTRFinder"][0]
    body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    for fn in body:
        fn.decorator_list = [d for d in fn.decorator_list if isinstance(d, ast.Name) and d.id == "staticmethod"]
    helpers = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("SelectedRead",)]
    mod = ast.Module(body=helpers + [ast.ClassDef(name="VNTRFinder", bases=[], keywords=[], body=body, decorator_list=[])],
                     type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(vars(hmm_utils))
    import logging
    from multiprocessing import Value
    # in-process list instead of a Manager().list proxy: the harness runs the methods in one process, and the
    # exec'd SelectedRead class cannot be pickled to a manager server
    Manager = lambda: types.SimpleNamespace(list=lambda x=(): list(x))
    ns.update(logging=logging, Manager=Manager, Value=Value, SeqIO=_SeqIO, settings=settings, os=os, Model=HiddenMarkovModel)
    exec(compile(mod, path, "exec"), ns)
    return ns["VNTRFinder"]


def main():
    rng = np.random.default_rng(4711)
    dna = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    pattern = dna(24)
    segs = []
    for _ in range(5):
        s = list(pattern)
        if rng.random() < 0.6:
            s[int(rng.integers(0, 24))] = "ACGT"[int(rng.integers(0, 4))]
        segs.append("".join(s))
    left, right = dna(500), dna(500)
    vntr = "".join(segs)
    # a chromosome: random background, the locus itself, decoys (mutated copies of locus pieces that share 11-mers with
    # the keyword set), N runs, lower-case stretches
    chrom = list(dna(60000))
    start = 20000
    chrom[start - 500:start + len(vntr) + 500] = list(left + vntr + right)
    locus = left[-15:] + vntr + right[:15]
    for d in range(14):
        at = 2000 + 4000 * d + (0 if d < 4 else 1500)
        if start - 1200 < at < start + 1200:
            continue
        piece = list((vntr * 2)[int(rng.integers(0, 20)):][:int(rng.integers(60, 140))])
        for _ in range(int(rng.integers(0, 4))):
            piece[int(rng.integers(0, len(piece)))] = "ACGT"[int(rng.integers(0, 4))]
        chrom[at:at + len(piece)] = piece
        if d % 3 == 0:
            chrom[at + 30] = "N"
    for p in (5000, 33333, 41000):
        chrom[p:p + int(rng.integers(1, 40))] = "N" * 20
    for p in (10100, 26000):
        chrom[p:p + 300] = [c.lower() for c in chrom[p:p + 300]]
    chrom = "".join(chrom)[:60000]
    other = dna(3000)

    Finder = load_methods()
    f = Finder.__new__(Finder)
    f.reference_vntr = types.SimpleNamespace(id=77, pattern=pattern, chromosome="chr7", start_point=start,
                                             left_flanking_region=left, right_flanking_region=right, scaled_score=0,
                                             get_repeat_segments=lambda: segs, get_length=lambda: len(vntr))
    f.min_repeat_bp_to_add_read = 2
    f.min_repeat_bp_to_count_repeats = 2
    settings.USE_TRAINED_HMMS = False
    settings.MAX_ERROR_RATE = 0.05
    hmm_utils.build_profile_hmm_for_repeats = \
        lambda repeats, error_rate: hmm_utils.build_profile_hmm_pseudocounts_for_alignment(error_rate, repeats)
    with tempfile.NamedTemporaryFile("w", suffix=".fa", delete=False) as fa:
        fa.write(">chr1 something\n%s\n>chr7 the one\n" % other)
        for i in range(0, len(chrom), 70):
            fa.write(chrom[i:i + 70] + "\n")
        ref_file = fa.name
    read_length = 150
    # the reference's own sequence of calls, step by step so the intermediate values can be recorded
    hmm = f.get_vntr_matcher_hmm(read_length=read_length)          # bake() seeds `random` with 0 as a side effect
    true_reads = f.simulate_true_reads(read_length)
    false_reads = f.simulate_false_filtered_reads(ref_file)
    true_scored = f.find_hmm_score_of_simulated_reads(hmm, true_reads)
    false_scored = f.find_hmm_score_of_simulated_reads(hmm, false_reads)
    true_scores = [r.logp for r in true_scored]
    false_scores = [r.logp for r in false_scored]
    threshold = f.find_recruitment_score_threshold(true_scored, false_scored)
    whole = f.train_classifier_threshold(ref_file, read_length)     # and once more as one call (re-seeds through bake)
    os.unlink(ref_file)
    out = {"pattern": pattern, "repeat_segments": segs, "left": left, "right": right, "chromosome": "chr7",
           "start_point": start, "sequences": [["chr1", other], ["chr7", chrom]], "read_length": read_length,
           "true_reads": true_reads, "false_reads": false_reads, "true_scores": true_scores, "false_scores": false_scores,
           "threshold": threshold, "scaled_threshold": whole,
           "no_false_threshold": f.find_recruitment_score_threshold(true_scored, [])}
    with gzip.open(os.path.join(HERE, "threshold_training.json.gz"), "wt") as fh:
        json.dump(out, fh)
    print("true reads %d (scored %d), false reads %d (scored %d), threshold %s, scaled %s"
          % (len(true_reads), len(true_scores), len(false_reads), len(false_scores), threshold, whole))


if __name__ == "__main__":
    main()
