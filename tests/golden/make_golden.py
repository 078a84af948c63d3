#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE REFERENCE in this container.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (the build container):

    python oracle/tools/build_reference.py          # patched scratch build in /tmp/advntr_ref_build
    python tests/golden/make_golden.py              # writes tests/golden/*.json.gz

Everything recorded here is produced by the reference's own code:
  * models by advntr.hmm_utils.get_read_matcher_model (advntr/hmm_utils.py:553-595) on the vendored
    pomegranate (pomegranate/hmm.pyx) -- state order, CSR edge order, log-probs, emissions;
  * logp / vpath by HiddenMarkovModel.viterbi (hmm.pyx:1911-2136), log_probability (hmm.pyx:1258);
  * the path summaries by advntr/hmm_utils.py:155-286;
  * recruit verdicts / genotypes by VNTRFinder.recruit_read and find_genotype_based_on_observed_repeats
    (advntr/vntr_finder.py:179-190, 473-532).  vntr_finder.py cannot be imported here (keras, pysam,
    biopython are absent), so those method bodies are compiled at run time from the reference file
    with `ast` and executed unchanged; nothing of them is stored.
Only data (inputs + expected outputs) is written.  No reference source enters the repo.
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
sys.path[:0] = [os.path.join(REPO, "oracle", "tools", "nx111"),
                os.path.join(REPO, "oracle", "tools", "stubs"), REF_BUILD]

import numpy as np                                    # noqa: E402
import networkx                                       # noqa: E402
assert networkx.__version__ == "1.11-restated"
from advntr import settings, hmm_utils                # noqa: E402
from pomegranate import HiddenMarkovModel, State, DiscreteDistribution  # noqa: E402


def _load_vntr_finder_methods():
    """Compile selected VNTRFinder methods straight from the reference file (not stored)."""
    path = "/root/reference/advntr/vntr_finder.py"
    tree = ast.parse(open(path).read())
    wanted = {"recruit_read", "get_conditional_likelihood", "find_genotype_based_on_observed_repeats",
              "get_copies_for_hmm", "get_min_score_to_select_a_read", "find_repeat_count_from_alignment_file",
              "read_flanks_repeats_with_confidence", "get_ru_count_with_coverage_method",
              "get_dominant_copy_numbers_from_spanning_reads", "build_vntr_matcher_hmm"}
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "VNTRFinder"][0]
    body = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    for fn in body:
        fn.decorator_list = []
    helpers = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("GenotypeResult", "ReadSource", "LoggedRead")]
    mod = ast.Module(body=helpers + [ast.ClassDef(name="VNTRFinder", bases=[], keywords=[], body=body,
                                                  decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(vars(hmm_utils))
    import logging
    from collections import Counter
    from enum import Enum
    ns["logging"] = logging
    ns["Counter"] = Counter
    ns["Enum"] = Enum
    exec(compile(mod, path, "exec"), ns)
    return ns["VNTRFinder"]


RefVNTRFinder = _load_vntr_finder_methods()


def make_finder(left, right, pattern, scaled_score, haploid=False):
    f = RefVNTRFinder.__new__(RefVNTRFinder)
    f.reference_vntr = types.SimpleNamespace(left_flanking_region=left, right_flanking_region=right,
                                             pattern=pattern, scaled_score=scaled_score, id=1)
    f.is_haploid = haploid
    return f


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def mutate(rng, s, sub=0.01, ins=0.0, dele=0.0):
    out = []
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + ins:
            out.append("ACGT"[rng.integers(0, 4)])
        if rng.random() < sub:
            ch = "ACGT"[rng.integers(0, 4)]
        out.append(ch)
    return "".join(out)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def dump_model(m):
    idx = {s: i for i, s in enumerate(m.states)}
    edges = [[idx[a], idx[b], d["probability"]] for a, b, d in m.graph.edges_iter(data=True)]
    emis = []
    for s in m.states[:m.silent_start]:
        p = s.distribution.parameters[0]
        emis.append({"prob": [p[c] for c in "ACGT"],
                     "logp": [s.distribution.log_probability(c) for c in "ACGT"]})
    return {"state_names": [s.name for s in m.states], "silent_start": m.silent_start,
            "start_index": m.start_index, "end_index": m.end_index, "edges": edges, "emissions": emis}


def score_read(m, seq, finder, left, right, with_forward):
    logp, vpath = m.viterbi(seq)
    rec = {"seq": seq, "logp": logp}
    if vpath is None:
        rec["path"] = None
        return rec
    rec["path"] = [i for i, _ in vpath]
    rec["ru"] = hmm_utils.get_number_of_repeats_in_vpath(vpath)
    rec["matches"] = hmm_utils.get_number_of_matches_in_vpath(vpath)
    rec["repeat_bp"] = hmm_utils.get_number_of_repeat_bp_matches_in_vpath(vpath)
    rec["left_bp"] = hmm_utils.get_left_flanking_region_size_in_vpath(vpath)
    rec["right_bp"] = hmm_utils.get_right_flanking_region_size_in_vpath(vpath)
    if len(vpath) > 2:
        rec["flank_rate"] = hmm_utils.get_flanking_regions_matching_rate(vpath, seq, left, right)
        rec["flank_rate_acc"] = hmm_utils.get_flanking_regions_matching_rate(vpath, seq, left, right,
                                                                              accuracy_filter=True)
        min_score = finder.get_min_score_to_select_a_read(len(seq))
        rec["recruit"] = bool(finder.recruit_read(logp, vpath, min_score, seq))
        nofinder = make_finder(left, right, finder.reference_vntr.pattern, None)
        rec["recruit_noscore"] = bool(nofinder.recruit_read(logp, vpath, None, seq))
    if with_forward:
        rec["forward_logp"] = m.log_probability(seq)
    return rec


def locus_reads(rng, left, right, units, n_reads, read_len, copies, sub, ins=0.0, dele=0.0, extra=()):
    """40 % locus-derived, 60 % random, plus reverse complements of a few and edge cases."""
    reads = []
    for _ in range(n_reads):
        if rng.random() < 0.4:
            k = int(rng.integers(1, copies + 1))
            lf = int(rng.integers(0, min(len(left), 100) + 1))
            body = "".join(units[int(rng.integers(0, len(units)))] for _ in range(k))
            s = (left[len(left) - lf:] + body + right)[:read_len]
            s = s + rand_seq(rng, read_len - len(s))
            s = mutate(rng, s, sub, ins, dele)
        else:
            s = rand_seq(rng, read_len)
        if not s:
            s = "A"
        reads.append(s)
    reads += [revcomp(r) for r in reads[:4]]
    reads += list(extra)
    return reads


def write(name, obj):
    path = os.path.join(HERE, name + ".json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(obj, separators=(",", ":")).encode())
    print("wrote", path, os.path.getsize(path), "bytes;", len(obj.get("reads", [])), "reads")


def adv_locus(name, seed, flank, units, aligned, copies, err, n_reads, read_len, scaled_score,
              sub=0.01, ins=0.0, dele=0.0, with_forward=False, extra=()):
    """One read-matcher locus.  `aligned` = pre-aligned repeat rows (no muscle in the image): a single
    row goes through build_profile_hmm_for_repeats unchanged (profile_hmm.py:165-175 else-branch); several
    rows are fed to build_profile_hmm_pseudocounts_for_alignment exactly as the muscle branch would."""
    rng = np.random.default_rng(seed)
    settings.MAX_ERROR_RATE = err
    left, right = rand_seq(rng, flank), rand_seq(rng, flank)
    orig = hmm_utils.build_profile_hmm_for_repeats
    if len(aligned) > 1:
        hmm_utils.build_profile_hmm_for_repeats = \
            lambda repeats, e: hmm_utils.build_profile_hmm_pseudocounts_for_alignment(e, list(aligned))
    try:
        m = hmm_utils.get_read_matcher_model(left[-flank:], right[:flank], list(aligned), copies)
    finally:
        hmm_utils.build_profile_hmm_for_repeats = orig
    pattern = units[0]
    finder = make_finder(left, right, pattern, scaled_score)
    reads = locus_reads(rng, left, right, units, n_reads, read_len, copies, sub, ins, dele, extra)
    recs = [score_read(m, r, finder, left, right, with_forward) for r in reads]
    write(name, {"kind": "read_matcher", "left": left, "right": right, "aligned_repeats": list(aligned),
                 "copies": copies, "error_rate": err, "scaled_score": scaled_score, "pattern": pattern,
                 "model": dump_model(m), "reads": recs})


def generic_model(name, seed, n_emit, n_silent, finite, n_reads):
    """A random pomegranate HMM built through the public API with bake(merge=None): loops between emitting
    states, a silent DAG, optional end state -- exercises the generic CSR path, not only adVNTR topologies."""
    rng = np.random.default_rng(seed)
    m = HiddenMarkovModel(name="generic%d" % seed)
    emit = []
    for i in range(n_emit):
        p = rng.dirichlet(np.ones(4))
        emit.append(State(DiscreteDistribution(dict(zip("ACGT", p.tolist()))), name="e%02d" % i))
    sil = [State(None, name="s%02d" % i) for i in range(n_silent)]
    m.add_states(emit + sil)
    for s in emit[:3]:
        m.add_transition(m.start, s, float(rng.random()))
    m.add_transition(m.start, sil[0], 0.5)
    for a in emit:
        for b in emit:
            if rng.random() < 0.3:
                m.add_transition(a, b, float(rng.random()))
        for b in sil:
            if rng.random() < 0.2:
                m.add_transition(a, b, float(rng.random()))
    for i, a in enumerate(sil):
        for b in sil[i + 1:]:
            if rng.random() < 0.4:
                m.add_transition(a, b, float(rng.random()))
        for b in emit:
            if rng.random() < 0.25:
                m.add_transition(a, b, float(rng.random()))
    if finite:
        for a in emit[-3:] + sil[-2:]:
            m.add_transition(a, m.end, float(rng.random()))
    m.bake(merge=None)
    reads = [rand_seq(rng, int(rng.integers(1, 40))) for _ in range(n_reads)] + [""]
    recs = []
    # The reference writes the path into a fixed n+m int buffer (hmm.pyx:1953) and overruns it when a
    # path revisits silent states often enough (heap corruption, seen on random silent DAGs).  Guard only:
    # skip reads whose path would not fit; the recorded values all come from the reference calls below.
    sys.path.insert(0, REPO)
    from oracle.oracle import OracleModel
    guard = OracleModel.from_golden({"model": dump_model(m)})
    for r in reads:
        _, gpath = guard.viterbi(r)
        if gpath is not None and len(gpath) >= len(r) + len(m.states) - 1:
            print("  skipping a read whose path (%d) would overrun the reference buffer" % len(gpath))
            continue
        logp, vpath = m.viterbi(r)
        recs.append({"seq": r, "logp": logp, "path": None if vpath is None else [i for i, _ in vpath],
                     "forward_logp": m.log_probability(r)})
    write(name, {"kind": "generic", "finite": bool(finite), "model": dump_model(m), "reads": recs})


def illumina_aggregation_cases():
    """VNTRFinder.find_repeat_count_from_alignment_file (vntr_finder.py:789-887) with its read selection stubbed out:
    select_illumina_reads needs a BAM (pysam); here it returns prepared SelectedRead-like records whose vpaths come
    from the reference's own viterbi, so everything after the selection is the reference's code."""
    rng = np.random.default_rng(31)
    settings.MAX_ERROR_RATE = 0.05
    pattern = rand_seq(rng, 14)
    left, right = rand_seq(rng, 150), rand_seq(rng, 150)
    m = hmm_utils.get_read_matcher_model(left, right, [pattern], 11)
    cases, all_reads = [], {}
    for case, (alleles, n_per, noise) in enumerate([((3, 5), 30, 0.005), ((4, 4), 25, 0.01), ((2, 9), 30, 0.005),
                                                    ((6, 7), 12, 0.02)]):
        finder = make_finder(left, right, pattern, None)
        finder.minimum_left_flanking_size = 5
        finder.minimum_right_flanking_size = 5
        recs, selected = [], []
        for copies in alleles:
            allele = left + pattern * copies + right
            for _ in range(n_per):
                st = int(rng.integers(40, 150))
                s = mutate(rng, allele[st:st + 150], noise)
                logp, vpath = m.viterbi(s)
                if vpath is None or not finder.recruit_read(logp, vpath, None, s):
                    continue
                if hmm_utils.get_number_of_repeat_bp_matches_in_vpath(vpath) <= 2:
                    continue
                mapped = bool(rng.random() < 0.5)
                selected.append(types.SimpleNamespace(sequence=s, logp=logp, vpath=vpath, is_mapped=mapped,
                                                      query_name="q%d" % len(selected)))
                recs.append({"seq": s, "logp": logp, "path": [i for i, _ in vpath], "is_mapped": mapped})
        finder.select_illumina_reads = lambda *a, **k: selected
        for acc, cov in ((False, None), (True, None), (False, 30.0)):
            res = finder.find_repeat_count_from_alignment_file(None, None, accuracy_filter=acc, average_coverage=cov)
            cn = res.copy_numbers
            cases.append({"case": case, "accuracy_filter": acc, "average_coverage": cov, "haploid": False,
                          "copy_numbers": None if cn is None else list(cn),
                          "recruited": res.recruited_reads_count, "spanning": res.spanning_reads_count,
                          "flanking": res.flanking_reads_count, "max_likelihood": res.maximum_likelihood})
        for c in cases[-3:]:
            c["reads_ref"] = case
        all_reads[case] = recs
    write("illumina_aggregation", {"kind": "illumina_aggregation", "left": left, "right": right, "pattern": pattern,
                                   "copies": 11, "error_rate": 0.05, "model": dump_model(m),
                                   "reads_by_case": {str(k): v for k, v in all_reads.items()}, "cases": cases})


def pacbio_cases():
    """VNTRFinder.get_dominant_copy_numbers_from_spanning_reads (vntr_finder.py:534-585) on trimmed spanning reads
    (flank 100 + VNTR + flank 100 with PacBio-like noise), error rate 0.3 as advntr_commands.py:66-71 sets it."""
    rng = np.random.default_rng(41)
    settings.MAX_ERROR_RATE = 0.3
    cases = []
    for case, (plen, alleles, n_per) in enumerate([(20, (4, 7), 8), (33, (3, 3), 6), (12, (10, 14), 7)]):
        pattern = rand_seq(rng, plen)
        left, right = rand_seq(rng, 120), rand_seq(rng, 120)
        finder = make_finder(left, right, pattern, None)
        finder.reference_vntr.get_repeat_segments = lambda pattern=pattern: [pattern]      # one segment: no muscle
        finder.minimum_left_flanking_size = 5
        finder.minimum_right_flanking_size = 5
        reads = []
        for copies in alleles:
            for _ in range(n_per):
                s = mutate(rng, left[-100:] + pattern * copies + right[:100], 0.03, 0.04, 0.03)
                reads.append(types.SimpleNamespace(sequence=s, read_id="p%d" % len(reads),
                                                   source=types.SimpleNamespace(name="MAPPED")))
        for acc in (False, True):
            finder.minimum_left_flanking_size = finder.minimum_right_flanking_size = 5
            cn, prob = finder.get_dominant_copy_numbers_from_spanning_reads(reads, False, acc)
            cases.append({"case": case, "pattern": pattern, "left": left, "right": right, "repeat_segments": [pattern],
                          "reads": [r.sequence for r in reads], "accuracy_filter": acc,
                          "copy_numbers": None if cn is None else list(cn), "max_prob": prob})
    settings.MAX_ERROR_RATE = 0.05
    write("pacbio_dominant_copy_numbers", {"kind": "pacbio_genotype", "cases": cases})


def genotype_cases():
    f = make_finder("A", "A", "A", None)
    cases = [[2, 2, 2, 5, 5], [3], [], [4, 4, 4, 4], [1, 2, 3], [7, 7, 8, 8, 8, 9], [10, 10, 2],
             [5] * 20 + [6] * 18, [2, 3], [12, 12, 12, 13, 11, 12, 14]]
    out = []
    for hap in (False, True):
        f.is_haploid = hap
        for c in cases:
            if not c:
                continue
            g, p = f.find_genotype_based_on_observed_repeats(list(c))
            out.append({"observed": c, "haploid": hap, "genotype": None if g is None else list(g), "max_prob": p})
    copies = [{"read_length": rl, "pattern_len": pl,
               "copies": make_finder("A", "A", "A" * pl, None).get_copies_for_hmm(rl)}
              for rl in (100, 148, 150, 151, 250) for pl in (6, 7, 12, 14, 30, 60, 100)]
    write("genotype_cases", {"kind": "genotype", "cases": out, "copies_for_hmm": copies})


def reference_fixture_answers():
    """Known answers of the reference's own fixture tests/data/hmm_utils.json: a real
    250-bp read and the 275 state names of its Viterbi path.  The fixture is data held by the reference's
    tests; the path summaries of advntr/hmm_utils.py:155-286 on it are recorded as expected outputs."""
    d = json.load(open("/root/reference/tests/data/hmm_utils.json"))
    names = d["visited_states"].split(",")
    vpath = [(0, types.SimpleNamespace(name="start"))] + \
            [(0, types.SimpleNamespace(name=n)) for n in names] + [(0, types.SimpleNamespace(name="end"))]
    repeats, states = hmm_utils.extract_repeating_segments_from_read(d["sequence"], names)
    ans = {"ru": hmm_utils.get_number_of_repeats_in_vpath(vpath),
           "matches": hmm_utils.get_number_of_matches_in_vpath(vpath),
           "repeat_bp": hmm_utils.get_number_of_repeat_bp_matches_in_vpath(vpath),
           "left_bp": hmm_utils.get_left_flanking_region_size_in_vpath(vpath),
           "right_bp": hmm_utils.get_right_flanking_region_size_in_vpath(vpath),
           "unit_lengths": hmm_utils.get_repeating_pattern_lengths(names),
           "repeats": repeats,
           "alignment": hmm_utils.get_multiple_alignment_of_viterbi_paths(repeats, states)}
    assert repeats == d["correct_repeats"] and ans["alignment"] == d["alignment"]
    write("reference_fixture_hmm_utils", {"kind": "reference_fixture", "visited_states": names,
                                          "sequence": d["sequence"], "correct_repeats": d["correct_repeats"],
                                          "alignment": d["alignment"], "answers": ans})


def main():
    fixture_alignment = json.load(open("/root/reference/tests/data/hmm_utils.json"))["alignment"]
    rng = np.random.default_rng(7)
    p12, p14, p5, p6, p30 = (rand_seq(rng, k) for k in (12, 14, 5, 6, 30))
    # toy: F=8, L=5, C=2 (the survey's wiring probe); forward values too
    adv_locus("toy_f8_l5_c2", 11, 8, [p5], [p5], 2, 0.05, 24, 20, -1.2, with_forward=True,
              extra=["", "A", "ACGT"])
    # S300: flank 30, 12-bp pattern, copies 3 -> 315 states / 197 emitting / 1004 edges
    adv_locus("s300_f30_l12_c3", 12, 30, [p12], [p12], 3, 0.05, 48, 60, -1.0, with_forward=True,
              extra=["", "G"])
    # REF150: flank 150, 14-bp pattern, copies 11 -> 1413 / 921 / 4626 (the model adVNTR builds for 150-bp reads)
    adv_locus("ref150_f150_l14_c11", 13, 150, [p14], [p14], 11, 0.05, 40, 150, -0.9, with_forward=True)
    # multi-repeat profile from the reference fixture's own 8-row alignment (one insert-prone column)
    units = [r.replace("-", "") for r in fixture_alignment]
    adv_locus("msa8_f50_c4", 14, 50, units, fixture_alignment, 4, 0.05, 40, 100, -1.0, with_forward=True)
    # hand-aligned rows with a majority-gap (insert) column and a deleted column
    rows = ["ACGT-TAGGCA", "ACGT-TAGGCA", "ACGTCTAGGCA", "ACG--TAGGCA", "ACGT-TA-GCA", "ACGT-TTGGCA"]
    adv_locus("msa_gaps_f40_c5", 15, 40, [r.replace("-", "") for r in rows], rows, 5, 0.05, 40, 80, None, with_forward=True)
    # short pattern, many copies (copies = round(100/6+0.5) = 17)
    adv_locus("p6_f100_c17", 16, 100, [p6], [p6], 17, 0.05, 30, 100, -1.0, with_forward=True)
    # PacBio settings: error 0.3, flank 100, noisy long reads
    adv_locus("pacbio_f100_l30_c6", 17, 100, [p30], [p30], 6, 0.3, 24, 340, None, sub=0.04, ins=0.05,
              dele=0.04, with_forward=True)
    generic_model("generic_finite", 21, 9, 6, True, 40)
    generic_model("generic_infinite", 22, 7, 4, False, 40)
    genotype_cases()
    illumina_aggregation_cases()
    pacbio_cases()
    reference_fixture_answers()


if __name__ == "__main__":
    main()
