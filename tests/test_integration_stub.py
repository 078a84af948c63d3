"""INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would add (advntr/hip_scoring.py).  These
tests take that fenced block OUT OF THE DOCUMENT and execute it as written against a duck-typed baked model (states,
silent_start, start_index, end_index, graph.edges_iter(data=True), State.distribution.log_probability) rebuilt from a golden
fixture of the reference: the documented binding is held to the reference's own scores, summaries and recruit verdicts."""
import ctypes
import os

import numpy as np
import pytest

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "advntr_amd", "libadvntr_hip.so")


def stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    at = text.index("```python\n# advntr/hip_scoring.py")
    body = text[at + len("```python\n"):]
    return body[:body.index("\n```")]


class _Dist(object):
    def __init__(self, logp):
        self._lp = dict(zip("ACGT", logp))

    def log_probability(self, c):
        return self._lp[c]


class _State(object):
    def __init__(self, name, dist):
        self.name, self.distribution = name, dist


class _Graph(object):
    def __init__(self, edges):
        self._e = edges

    def edges_iter(self, data=False):
        for a, b, lp in self._e:
            yield (a, b, {"probability": lp}) if data else (a, b)


class DuckModel(object):
    """What the stub reads off a baked pomegranate model, filled from a golden fixture (tests/golden/make_golden.py)."""

    def __init__(self, g):
        gm = g["model"]
        self.silent_start, self.start_index, self.end_index = gm["silent_start"], gm["start_index"], gm["end_index"]
        self.states = [_State(n, _Dist(gm["emissions"][i]["logp"]) if i < self.silent_start else None)
                       for i, n in enumerate(gm["state_names"])]
        self.graph = _Graph([(self.states[a], self.states[b], lp) for a, b, lp in gm["edges"]])


def run_stub(cdll=None):
    src = stub_source().replace('ctypes.CDLL("libadvntr_hip.so")', "ctypes.CDLL(%r)" % LIB)
    ns = {}
    exec(compile(src, "INTEGRATION.md:hip_scoring", "exec"), ns)
    return ns


def test_stub_state_classes_carry_the_flank_bases():
    """CPU: the block parses and loads the library; its state classes equal the host mirror's INCLUDING the flank-base bits
    (without them SUM_LEFT_MATCH / SUM_RIGHT_MATCH stay 0 and recruit() rejects every read that touches a flank), and the arrays
    it hands to advntr_hmm_create are the oracle's CSR."""
    from advntr_amd.pomegranate import state_class_from_name
    from oracle.oracle import OracleModel
    import __graft_entry__ as entry
    entry.build()
    ns = run_stub()
    g = load_golden("s300_f30_l12_c3")
    model = DuckModel(g)
    F = len(g["left"])
    want = []
    for s in model.states:
        base = None
        if s.name.startswith("M") and s.name.endswith("_suffix"):
            base = "ACGT".index(g["left"][int(s.name[1:].split("_")[0]) - 1])
        if s.name.startswith("M") and s.name.endswith("_prefix"):
            base = "ACGT".index(g["right"][int(s.name[1:].split("_")[0]) - 1])
        want.append(state_class_from_name(s.name, base))
    assert sum(1 for c in want if c & 0x0400) == 2 * F
    seen = {}

    class Recorder(object):
        def advntr_hmm_create(self, m, ss, si, ei, E, in_ptr, in_src, in_logp, emis, cls):
            as_arr = lambda p, t, n: np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(t)), (n,)).copy()
            seen.update(m=m, ss=ss, si=si, ei=ei, E=E, in_ptr=as_arr(in_ptr, ctypes.c_int32, m + 1),
                        in_src=as_arr(in_src, ctypes.c_int32, E), in_logp=as_arr(in_logp, ctypes.c_double, E),
                        emis=as_arr(emis, ctypes.c_double, ss * 4), cls=as_arr(cls, ctypes.c_uint16, m))
            return 1
    ns["_L"] = Recorder()
    assert ns["upload"](model, g["left"], g["right"]) == 1
    assert list(seen["cls"]) == want
    O = OracleModel.from_golden(g)
    in_ptr, in_src, in_logp, _ = O.csr()
    assert np.array_equal(seen["in_ptr"], in_ptr) and np.array_equal(seen["in_src"], in_src)
    assert np.array_equal(seen["in_logp"], in_logp) and np.array_equal(seen["emis"], np.asarray(O.emis).reshape(-1))
    assert (seen["m"], seen["ss"], seen["si"], seen["ei"]) == (len(model.states), model.silent_start, model.start_index, model.end_index)
    # recruit() is VNTRFinder.recruit_read on a summary record
    rec = ns["recruit"]
    assert rec(-10.0, [3, 140, 36, 50, 50, 50, 44, 160], 150, None) is False         # right flank rate 0.88
    assert rec(-10.0, [3, 140, 36, 50, 50, 50, 45, 160], 150, None) is True
    assert rec(-151.0, [3, 140, 36, 50, 50, 50, 50, 160], 150, None) is False        # logp <= -read_length
    assert rec(-151.0, [3, 10, 36, 0, 0, 0, 0, 160], 150, -200.0) is True            # trained score: logp > min_score
    assert rec(-10.0, [0, 0, 0, 0, 0, 0, 0, 0], 150, None) is False                  # impossible read


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["s300_f30_l12_c3", "ref150_f150_l14_c11", "msa_gaps_f40_c5"])
def test_stub_scores_and_recruits_like_the_reference(name):
    """GPU: upload() + score() + recruit() of the documented stub on the golden reads: log-probabilities ==, the six path
    summaries exact, and recruit verdicts (with the locus's trained score and without one) equal to what the reference's own
    VNTRFinder.recruit_read returned for the same reads."""
    ns = run_stub()
    g = load_golden(name)
    model = DuckModel(g)
    h = ns["upload"](model, g["left"], g["right"])
    reads = [r["seq"] for r in g["reads"] if r["path"] is not None]
    logp, summ = ns["score"](h, reads)
    touched = 0
    for lp, s, r in zip(logp, summ, [r for r in g["reads"] if r["path"] is not None]):
        assert lp == r["logp"]
        assert (s[0], s[1], s[2], s[3], s[4], s[7]) == (r["ru"], r["matches"], r["repeat_bp"], r["left_bp"], r["right_bp"], len(r["path"]))
        n = len(r["seq"])
        ms = g["scaled_score"] * n if g.get("scaled_score") else None
        if "recruit" in r:
            assert ns["recruit"](lp, s, n, ms) == r["recruit"], r["seq"]
            assert ns["recruit"](lp, s, n, None) == r["recruit_noscore"], r["seq"]
        else:
            assert ns["recruit"](lp, s, n, None) is False
        touched += int(s[3] + s[4] > 0)
    assert touched > 0 and any(r.get("recruit") or r.get("recruit_noscore") for r in g["reads"])
    from advntr_amd import _lib
    _lib.load().advntr_hmm_destroy(ctypes.c_void_p(h))


@pytest.mark.gpu
def test_short_read_flow_stub_prints_the_stage_by_stage_rows(tmp_path, capsys):
    """INTEGRATION.md section 9 -- the body a maintainer gives GenomeAnalyzer.find_repeat_counts_from_short_reads -- taken out of
    the document and executed as written on a duck-typed analyzer (40 loci, their candidate reads planted in a FASTA file of 30 000
    reads): the rows it prints are the rows of the stage-by-stage route (filter stdout text parsed as the reference parses it,
    reads as str, models built up front, vntr_finder.genotype_loci)."""
    from advntr_amd import filtering, genome_analyzer, hmm_utils, models, vntr_finder, workloads
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    at = text.index("```python\n# advntr/genome_analyzer.py: the short-read flow on the engine")
    body = text[at + len("```python\n"):]
    ns = {}
    exec(compile(body[:body.index("\n```")], "INTEGRATION.md:short_read_flow", "exec"), ns)
    loci, cands = [], []
    for k in range(40):
        params, calls, (nm, nu) = workloads._c2_locus((k, 20240602, 150, 80, 40))
        loci.append(workloads.Locus(*params))
        cands.append(calls[:nm + nu])
    lines, fasta, rec_len, _ = workloads.make_illumina_pipeline_workload(loci, cands, 30000)
    fa = tmp_path / "unmapped.fa"
    fa.write_bytes(fasta)
    vntrs = {}
    for k, l in enumerate(loci):
        v = models.ReferenceVNTR(k + 1, l.units[0], 10000 * k, "chr1", None, None, len(l.units))
        v.init_from_xml(list(l.units), l.left, l.right)
        vntrs[v.id] = v

    class Analyzer(object):
        reference_vntrs, target_vntr_ids, outfmt, is_haploid = vntrs, sorted(vntrs), "text", False
    capsys.readouterr()
    ns["find_repeat_counts_from_short_reads"](Analyzer(), str(fa), False)
    got = capsys.readouterr().out
    # stage by stage
    kw_text = "".join("%d %s\n" % (v, " ".join(sorted(k))) for v, k in lines)
    names, lists = {}, {}
    for line in filtering.run(fasta, kw_text, 5).split("\n"):
        parts = line.split()
        if len(parts) >= 2 and parts[0].isdigit() and parts[1].isdigit():
            lists[int(parts[0])] = parts[2:]
        elif len(parts) >= 2:
            names[parts[0]] = parts[1]
    read_lists = [[names[nm] for nm in sorted(lists.get(vid, ()))] for vid in sorted(vntrs)]
    desc = [(l.left, l.right, l.units, l.copies) for l in loci]
    plain = vntr_finder.genotype_loci(hmm_utils.build_read_matcher_models(desc), read_lists)
    want = "".join(genome_analyzer.genotype_row("text", vntrs[vid], vid, g) for vid, g in zip(sorted(vntrs), plain))
    assert got == want and got.count("\n") == 80 and sum(g.copy_numbers is not None for g in plain) > 20
