"""Frameshift identification (vntr_finder.py:256-309): the reference's own unit tests
(/root/reference/tests/test_frameshift_identification.py) replayed, and find_frameshift_from_selected_reads against
what the reference's method returned on reads it selected itself (tests/golden/frameshift.json.gz)."""
import pytest

from conftest import load_golden
from advntr_amd import vntr_finder


@pytest.mark.parametrize("observed,expected", [(14, True), (18, True), (7, True), (3, True), (2, False), (1, False), (0, False)])
def test_identify_frameshift_reference_unit_tests(observed, expected):
    avg_bp_coverage = 14.0
    assert vntr_finder.identify_frameshift(avg_bp_coverage, observed, 1 / avg_bp_coverage) is expected


def test_frameshift_from_the_reads_the_reference_selected():
    g = load_golden("frameshift")
    seen = set()
    for case in g["cases"]:
        selected = [(seq, names) for seq, names in case["selected"]]
        got = vntr_finder.find_frameshift_from_selected_reads(len(case["pattern"]), case["vntr_length"], selected)
        assert got == case["frameshift"], case["name"]
        seen.add(got)
    assert seen == {None, "I5T", "D8"}


@pytest.mark.gpu
def test_frameshift_end_to_end_on_the_gpu():
    """Reads -> both strands with PATH output on the GPU -> selection -> test: the same call as the reference made, and the
    selected reads (sequence and state names along the path) are the ones the reference selected."""
    from advntr_amd import hmm_utils
    g = load_golden("frameshift")
    for case in g["cases"]:
        model = hmm_utils.get_read_matcher_model(case["left"][-150:], case["right"][:150], case["repeat_segments"], case["copies"])
        got = vntr_finder.find_frameshift(model, len(case["pattern"]), case["vntr_length"], case["reads"])
        assert got == case["frameshift"], case["name"]


@pytest.mark.gpu
def test_cli_frameshift(tmp_path):
    import json
    import subprocess
    import sys
    from conftest import ROOT
    g = load_golden("frameshift")
    case = [c for c in g["cases"] if c["name"] == "insertion_both"][0]
    loci = [{"id": 5, "left": case["left"], "right": case["right"], "pattern": case["pattern"],
             "repeat_segments": case["repeat_segments"], "scaled_score": None}]
    (tmp_path / "loci.json").write_text(json.dumps(loci))
    (tmp_path / "reads.fa").write_text("".join(">r%d\n%s\n" % (i, s) for i, s in enumerate(case["reads"])))
    out = subprocess.run([sys.executable, "-m", "advntr_amd", "genotype", "--loci", str(tmp_path / "loci.json"),
                          "--reads", str(tmp_path / "reads.fa"), "--frameshift"], cwd=ROOT, stdout=subprocess.PIPE,
                         check=True).stdout.decode()
    assert out == "5\nI5T\n"
