"""The host-side model builder (advntr_amd.hmm_utils / pomegranate mirror / profile_hmm) against the
baked models captured from the reference: state order, CSR edge order, log-probs, emissions.

Structure (names, indices, edge order) must match exactly.  Log-probs go through numpy.exp / libm log
(dense_transition_matrix -> from_matrix, hmm.pyx:514,433); on the machine that produced the goldens they
are bit-identical, elsewhere numpy's SIMD exp may differ in the last bit, hence the 4-ulp allowance.
"""
import numpy as np
import pytest

from conftest import READ_MATCHER_GOLDENS, load_golden
from advntr_amd import hmm_utils, settings
from oracle import stepwise_builder


def build_from_golden(g, native=True):
    settings.MAX_ERROR_RATE = g["error_rate"]
    try:
        build = hmm_utils.get_read_matcher_model if native else stepwise_builder.get_read_matcher_model
        return build(g["left"], g["right"], g["aligned_repeats"], g["copies"])
    finally:
        settings.MAX_ERROR_RATE = 0.05


def _close(got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    fin = np.isfinite(want)
    assert np.all(np.abs(got[fin] - want[fin]) <= 4 * np.spacing(np.abs(want[fin])))
    exact = float(np.mean(got == want))
    assert exact > 0.999, exact


@pytest.mark.parametrize("native", [True, False], ids=["native", "stepwise"])
@pytest.mark.parametrize("name", READ_MATCHER_GOLDENS)
def test_read_matcher_matches_reference_bake(name, native):
    """native: the library's C++ builder (csrc/model_builder.h); stepwise: the call-by-call assembly through the
    pomegranate mirror (oracle/stepwise_builder.py, the checker of the native builder).  Both must reproduce the reference's baked model."""
    g = load_golden(name)
    gm = g["model"]
    m = build_from_golden(g, native)
    assert [s.name for s in m.states] == gm["state_names"]
    assert (m.silent_start, m.start_index, m.end_index) == (gm["silent_start"], gm["start_index"], gm["end_index"])
    if not native:                       # the construction graph itself: edges in graph.edges_iter() order
        idx = {s: i for i, s in enumerate(m.states)}
        edges = [(idx[a], idx[b], lp) for a, b, lp in m.graph.edges()]
        assert [(a, b) for a, b, _ in edges] == [(a, b) for a, b, _ in gm["edges"]]
        _close([e[2] for e in edges], [e[2] for e in gm["edges"]])
    emis_want = np.array([e["logp"] for e in gm["emissions"]])
    a = m.baked_arrays()
    assert np.array_equal(a["emis_logp"], emis_want)
    # CSR as the C ABI takes it == the oracle's CSR of the golden edge list (in-edge order = the tie-break)
    from oracle.oracle import OracleModel
    in_ptr, in_src, in_logp, finite = OracleModel.from_golden(g).csr()
    assert np.array_equal(a["in_ptr"], in_ptr) and np.array_equal(a["in_src"], in_src)
    _close(a["in_logp"], in_logp)
    assert bool(m.finite) == finite


def test_profile_parameters_multi_row():
    """profile_hmm.py:13-161 on the reference fixture's 8-row alignment and on rows with insert columns,
    compared through the emitted model (emission probabilities are part of the golden)."""
    for name in ("msa8_f50_c4", "msa_gaps_f40_c5"):
        g = load_golden(name)
        m = build_from_golden(g, native=False)      # the native builder keeps log-probabilities only (compared above)
        for s, e in zip(m.states[:m.silent_start], g["model"]["emissions"]):
            assert [s.distribution.parameters[0][c] for c in "ACGT"] == e["prob"], s.name


@pytest.mark.parametrize("native", [True, False])
def test_unaligned_repeats_are_refused(native):
    with pytest.raises(NotImplementedError):
        build = hmm_utils.get_read_matcher_model if native else stepwise_builder.get_read_matcher_model
        build("ACGTACGT", "TTGACCAA", ["ACGTT", "ACGT"], 2)


def _baked_equals(m, gm):
    assert [s.name for s in m.states] == gm["state_names"]
    assert (m.silent_start, m.start_index, m.end_index) == (gm["silent_start"], gm["start_index"], gm["end_index"])
    idx = {s: i for i, s in enumerate(m.states)}
    edges = [(idx[a], idx[b], lp) for a, b, lp in m.graph.edges()]
    assert [(a, b) for a, b, _ in edges] == [(a, b) for a, b, _ in gm["edges"]]
    got, want = np.array([e[2] for e in edges]), np.array([e[2] for e in gm["edges"]])
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    fin = np.isfinite(want)
    assert np.all(np.abs(got[fin] - want[fin]) <= 4 * np.spacing(np.abs(want[fin]))) and np.mean(got == want) > 0.99
    assert np.array_equal(m.baked_arrays()["emis_logp"], np.array([e["logp"] for e in gm["emissions"]]).reshape(-1, 4))


def test_bake_with_merging_equals_the_reference():
    """bake(merge='All' / 'Partial') (hmm.pyx:725-823: orphan removal, normalisation, folding of probability-1 silent
    states) on the 61 random models the reference's pomegranate baked (tests/golden/make_merge_golden.py)."""
    from advntr_amd import HiddenMarkovModel, State, DiscreteDistribution
    g = load_golden("bake_merge")
    seen = set()
    for case in g["cases"]:
        spec = case["spec"]
        m = HiddenMarkovModel("rnd")
        states = [State(DiscreteDistribution(dict(zip("ACGT", spec["dists"][i]))) if i < spec["n_emit"] else None, name=nm)
                  for i, nm in enumerate(spec["names"])]
        m.add_states(states)
        n = len(states)
        for a, b, p in spec["edges"]:
            m.add_transition(m.start if a == -1 else states[a], m.end if b == n else states[b], p)
        before = len(m.graph.nodes())
        m.bake(merge=case["merge"])
        _baked_equals(m, case["model"])
        seen.add((case["merge"], len(m.states) < before))
    assert seen == {("All", True), ("All", False), ("Partial", True), ("Partial", False)} or len(seen) >= 3


def test_repeat_finder_model_equals_the_reference():
    g = load_golden("bake_merge")
    for f in g["repeat_finder"]:
        m = hmm_utils.build_reference_repeat_finder_hmm([f["pattern"]], copies=f["copies"])
        _baked_equals(m, f["model"])


def test_model_json_round_trip_equals_the_reference():
    """HiddenMarkovModel.from_json / to_json (hmm.pyx:3023-3143): the JSON text the reference wrote loads into the model
    the reference loads from it (default bake: merge='All' folds 11 states of the read matcher), and the mirror's own
    to_json goes the same way."""
    from advntr_amd import HiddenMarkovModel
    g = load_golden("model_json")
    for case in g["cases"]:
        m = HiddenMarkovModel.from_json(case["json"])
        _baked_equals(m, case["loaded"])
        again = HiddenMarkovModel.from_json(m.to_json())
        assert [s.name for s in again.states] == case["loaded"]["state_names"]
    spec = g["cases"][0]["spec"]
    for native in (True, False):
        build = hmm_utils.get_read_matcher_model if native else stepwise_builder.get_read_matcher_model
        fresh = build(spec["left"], spec["right"], [spec["pattern"]], spec["copies"])
        _baked_equals(HiddenMarkovModel.from_json(fresh.to_json()), g["cases"][0]["loaded"])


def test_host_path_summaries_on_goldens():
    from advntr_amd.pomegranate import State
    for name in READ_MATCHER_GOLDENS:
        g = load_golden(name)
        names = g["model"]["state_names"]
        for r in g["reads"]:
            if r["path"] is None:
                continue
            vpath = [(i, State(None, names[i])) for i in r["path"]]
            assert hmm_utils.get_number_of_repeats_in_vpath(vpath) == r["ru"]
            assert hmm_utils.get_number_of_matches_in_vpath(vpath) == r["matches"]
            assert hmm_utils.get_number_of_repeat_bp_matches_in_vpath(vpath) == r["repeat_bp"]
            assert hmm_utils.get_left_flanking_region_size_in_vpath(vpath) == r["left_bp"]
            assert hmm_utils.get_right_flanking_region_size_in_vpath(vpath) == r["right_bp"]
            if "flank_rate" in r:
                assert hmm_utils.get_flanking_regions_matching_rate(vpath, r["seq"], g["left"], g["right"]) == r["flank_rate"]


def test_reference_fixture_known_answers_host():
    from advntr_amd.pomegranate import State
    g = load_golden("reference_fixture_hmm_utils")
    vpath = [(0, State(None, "s"))] + [(0, State(None, n)) for n in g["visited_states"]] + [(0, State(None, "e"))]
    a = g["answers"]
    assert hmm_utils.get_number_of_repeats_in_vpath(vpath) == a["ru"]
    assert hmm_utils.get_number_of_matches_in_vpath(vpath) == a["matches"]
    assert hmm_utils.get_number_of_repeat_bp_matches_in_vpath(vpath) == a["repeat_bp"]
    assert hmm_utils.get_left_flanking_region_size_in_vpath(vpath) == a["left_bp"]
    assert hmm_utils.get_right_flanking_region_size_in_vpath(vpath) == a["right_bp"]


@pytest.mark.parametrize("native", [True, False], ids=["native", "stepwise"])
def test_model_reestimated_from_viterbi_paths(native):
    """get_read_matcher_model(left, right, None, copies, vpaths) (hmm_utils.py:424-431, the model update of
    vntr_finder.py:667-697): the golden holds the (sequence, path) pairs the reference scored and the model it rebuilt
    from them (tests/golden/make_update_golden.py)."""
    from advntr_amd.pomegranate import State
    from oracle.oracle import OracleModel
    g = load_golden("model_update")
    vpaths = [(seq, [(0, State(None, n)) for n in names]) for seq, names in g["vpaths"]]
    assert hmm_utils.get_multiple_alignment_of_repeats_from_reads(vpaths) == g["alignment"]
    build = hmm_utils.get_read_matcher_model if native else stepwise_builder.get_read_matcher_model
    m = build(g["left"], g["right"], None, g["copies"], vpaths)
    gm = g["model"]
    assert [s.name for s in m.states] == gm["state_names"]
    assert (m.silent_start, m.start_index, m.end_index) == (gm["silent_start"], gm["start_index"], gm["end_index"])
    a = m.baked_arrays()
    assert np.array_equal(a["emis_logp"], np.array([e["logp"] for e in gm["emissions"]]))
    in_ptr, in_src, in_logp, _ = OracleModel.from_golden(g).csr()
    assert np.array_equal(a["in_ptr"], in_ptr) and np.array_equal(a["in_src"], in_src)
    _close(a["in_logp"], in_logp)
