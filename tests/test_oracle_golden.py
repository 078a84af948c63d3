"""Pins the CPU oracle (oracle/) against the golden vectors captured from the reference itself.

Bit-exact: log-probabilities are compared with ==, paths and the integer summaries exactly.
"""
import math

import pytest

from conftest import ALL_MODEL_GOLDENS, READ_MATCHER_GOLDENS, load_golden
from oracle import oracle as O


@pytest.mark.parametrize("name", ALL_MODEL_GOLDENS)
def test_viterbi_logp_and_path_bit_exact(name):
    g = load_golden(name)
    M = O.OracleModel.from_golden(g)
    assert len(g["reads"]) > 0
    for r in g["reads"]:
        logp, path = M.viterbi(r["seq"])
        assert logp == r["logp"] or (math.isinf(logp) and math.isinf(r["logp"])), (name, r["seq"])
        assert path == r["path"], (name, r["seq"])


@pytest.mark.parametrize("name", ALL_MODEL_GOLDENS)
def test_forward_bit_exact(name):
    g = load_golden(name)
    M = O.OracleModel.from_golden(g)
    seen = 0
    for r in g["reads"]:
        if "forward_logp" in r:
            seen += 1
            assert M.forward(r["seq"]) == r["forward_logp"], (name, r["seq"])
    if name in ("toy_f8_l5_c2", "s300_f30_l12_c3", "msa8_f50_c4", "generic_finite", "generic_infinite"):
        assert seen > 0


@pytest.mark.parametrize("name", READ_MATCHER_GOLDENS)
def test_path_summaries_and_recruit(name):
    g = load_golden(name)
    names_of = g["model"]["state_names"]
    left, right = g["left"], g["right"]
    for r in g["reads"]:
        if r["path"] is None:
            continue
        names = [names_of[i] for i in r["path"]][1:-1]
        assert O.number_of_repeats(names) == r["ru"]
        assert O.number_of_matches(names) == r["matches"]
        assert O.repeat_bp_matches(names) == r["repeat_bp"]
        assert O.left_flank_size(names) == r["left_bp"]
        assert O.right_flank_size(names) == r["right_bp"]
        if "flank_rate" in r:
            assert O.flanking_matching_rate(names, r["seq"], left, right) == r["flank_rate"]
            assert O.flanking_matching_rate(names, r["seq"], left, right, True) == r["flank_rate_acc"]
            ms = None if not g["scaled_score"] else g["scaled_score"] * len(r["seq"])
            assert O.recruit_read(r["logp"], names, ms, r["seq"], left, right) == r["recruit"]
            assert O.recruit_read(r["logp"], names, None, r["seq"], left, right) == r["recruit_noscore"]


def test_reference_fixture_known_answers():
    """tests/data/hmm_utils.json of the reference (a real 250-bp read and its Viterbi path)."""
    g = load_golden("reference_fixture_hmm_utils")
    names, a = g["visited_states"], g["answers"]
    assert O.number_of_repeats(names) == a["ru"] == 9
    assert O.number_of_matches(names) == a["matches"] == 246
    assert O.repeat_bp_matches(names) == a["repeat_bp"] == 119
    assert O.left_flank_size(names) == a["left_bp"] == 0
    assert O.right_flank_size(names) == a["right_bp"] == 131


def test_survey_shapes():
    """REF150 = 1413 states / 921 emitting / 4626 edges; S300 = 315 / 197 / 1004."""
    g = load_golden("ref150_f150_l14_c11")["model"]
    assert (len(g["state_names"]), g["silent_start"], len(g["edges"])) == (1413, 921, 4626)
    g = load_golden("s300_f30_l12_c3")["model"]
    assert (len(g["state_names"]), g["silent_start"], len(g["edges"])) == (315, 197, 1004)
