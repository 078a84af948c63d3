"""The reference's own unit tests of this module, /root/reference/tests/test_hmm_utils.py:13-30, replayed against the
mirror with the reference's fixture data (tests/data/hmm_utils.json, committed as a golden together with the answers
the reference functions give on it)."""
from conftest import load_golden
from advntr_amd import hmm_utils


def _data():
    return load_golden("reference_fixture_hmm_utils")


def test_extract_repeating_segments_from_read():
    d = _data()
    repeats, states = hmm_utils.extract_repeating_segments_from_read(d["sequence"], d["visited_states"])
    assert d["correct_repeats"] == repeats
    assert d["answers"]["repeats"] == repeats


def test_multiple_alignment_for_real_data():
    d = _data()
    repeats, states = hmm_utils.extract_repeating_segments_from_read(d["sequence"], d["visited_states"])
    alignment = hmm_utils.get_multiple_alignment_of_viterbi_paths(repeats, states)
    assert d["alignment"] == alignment


def test_multiple_alignment_for_two_sequences():
    repeats = ['ACTTA', 'ATTGA']
    states = [['M1', 'M2', 'M3', 'M4', 'M5'],
              ['M1', 'D2', 'M3', 'M4', 'I4', 'M5']]
    assert hmm_utils.get_multiple_alignment_of_viterbi_paths(repeats, states) == ['ACTT-A', 'A-TTGA']


def test_unit_lengths_and_segments():
    d = _data()
    assert hmm_utils.get_repeating_pattern_lengths(d["visited_states"]) == d["answers"]["unit_lengths"]
    region = "".join(d["correct_repeats"])
    assert hmm_utils.get_repeat_segments_from_visited_states_and_region(d["visited_states"], region) == d["correct_repeats"]
