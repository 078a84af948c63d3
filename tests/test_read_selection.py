"""Read selection from an alignment (vntr_finder.py:701-767) against what the reference's own method selected
(tests/golden/read_selection.json.gz, written by tests/golden/make_selection_golden.py), plus the SAM-text reader and the
read filters on CPU."""
import gzip
import json
import os
import types

import pytest

from advntr_amd import sam_utils, settings

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "read_selection.json.gz")

SAM = "\n".join([
    "@HD\tVN:1.6\tSO:coordinate",
    "@SQ\tSN:chr1\tLN:100000",
    "@SQ\tSN:chr2\tLN:5000",
    "r1\t0\tchr1\t101\t60\t4M2D2M1I2M3S\t*\t0\t0\tACGTACGTACGT\tIIIIIIIIIIII",
    "r2\t1024\tchr1\t201\t0\t12M\t*\t0\t0\tACGTACGTACGT\t" + "".join(chr(33 + q) for q in (2, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40, 40)),
    "r3\t4\t*\t0\t0\t*\t*\t0\t0\tACGTNACGT\t*",
    "r4\t16\tchr2\t11\t30\t5M100N5M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII",
]) + "\n"


def test_parse_sam_fields():
    f = sam_utils.parse_sam(SAM)
    assert f.references == ["chr1", "chr2"]
    r1, r2, r3, r4 = f.reads
    assert (r1.query_name, r1.reference_name, r1.reference_start, r1.reference_end, r1.mapq) == ("r1", "chr1", 100, 110, 60)
    assert r1.query_qualities == [40] * 12 and not r1.is_unmapped and not r1.is_duplicate
    assert r2.is_duplicate and r2.query_qualities[0] == 2
    assert r3.is_unmapped and r3.reference_end is None and r3.query_qualities is None
    assert (r4.reference_start, r4.reference_end) == (10, 120)
    assert [r.query_name for r in f.head(2)] == ["r1", "r2"]


def test_fetch_is_half_open_overlap_in_file_order():
    f = sam_utils.parse_sam(SAM)
    names = lambda *a: [r.query_name for r in f.fetch(*a)]
    assert names("chr1", 0, 100) == []
    assert names("chr1", 0, 101) == ["r1"]
    assert names("chr1", 109, 201) == ["r1", "r2"]
    assert names("chr1", 110, 200) == []
    assert names("chr2", 50, 60) == ["r4"]              # the skipped region counts as spanned
    assert names("chr3", 0, 10 ** 9) == []


def test_reference_genome_of_alignment_file():
    ns = types.SimpleNamespace
    assert sam_utils.get_reference_genome_of_alignment_file(ns(references=["chr1", "chrM"])) == "HG19"
    assert sam_utils.get_reference_genome_of_alignment_file(ns(references=["1", "2", "MT"])) == "GRCh37"
    assert sam_utils.get_reference_genome_of_alignment_file(ns(references=["1", "chrUn"])) == "HG19"
    assert sam_utils.get_reference_genome_of_alignment_file(ns(references=["scaffold_1"])) is None


@pytest.mark.parametrize("mapq,quals,expected", [
    (60, [40] * 150, False),
    (0, [40] * 150, True),                                   # mapq <= MAPQ_CUTOFF
    (60, [40] * 135 + [5] * 15, True),                       # 10 % low-quality bases
    (60, [5, 40] * 7 + [40] * 136, False),                   # 7 isolated low bases: under budget, each followed by a good one
    (60, [40] * 100 + [5, 5, 5, 5] + [40] * 46, True),       # base 100 sees only low bases at 101, 102
    (60, [40] * 100 + [5, 5] + [40] * 48, False),            # base 100 sees 102, base 101 sees 102
    (60, [40] * 149 + [5], False),                           # positions past the end are "not low"
])
def test_is_low_quality_read_cases(mapq, quals, expected):
    # maximum run = int(0.10 * 150 / 4) = 3: base i passes if one of i+1, i+2 is not in the low-quality list
    r = types.SimpleNamespace(mapq=mapq, query_qualities=quals)
    assert sam_utils.is_low_quality_read(r) is expected


def test_settings_defaults_match_reference():
    # advntr/settings.py:16-19
    assert (settings.QUALITY_SCORE_CUTOFF, settings.LOW_QUALITY_BP_TO_DISCARD_READ, settings.MAPQ_CUTOFF) == (20, 0.10, 0)
    assert settings.MIN_READ_LENGTH is None


def _golden_cases():
    with gzip.open(GOLDEN, "rt") as fh:
        return json.load(fh)["cases"]


def test_golden_sam_text_round_trips():
    for case in _golden_cases():
        f = sam_utils.parse_sam(case["sam"])
        assert f.references == ["chr5", "chrX"]
        assert len(f.reads) == 70
        assert all(r.reference_end == r.reference_start + len(r.seq) for r in f.reads)
        picked = {s["query_name"] for s in case["selected"] if s["query_name"]}
        by_name = {r.query_name: r for r in f.reads}
        # nothing the reference selected is a duplicate, low quality, short or N-carrying
        for name in picked:
            r = by_name[name]
            assert not r.is_duplicate and "N" not in r.seq and len(r.seq) >= 135 and not sam_utils.is_low_quality_read(r)


@pytest.mark.gpu
@pytest.mark.parametrize("index", [0, 1])
def test_select_illumina_reads_matches_reference(index):
    from advntr_amd import vntr_finder
    case = _golden_cases()[index]
    saved = (settings.MAX_ERROR_RATE, settings.USE_TRAINED_HMMS)
    settings.MAX_ERROR_RATE, settings.USE_TRAINED_HMMS = 0.05, False
    try:
        segs = case["repeat_segments"]
        vntr = types.SimpleNamespace(id=3, pattern=case["pattern"], chromosome="chr5", start_point=case["start_point"],
                                     scaled_score=case["scaled_score"], left_flanking_region=case["left"],
                                     right_flanking_region=case["right"], get_repeat_segments=lambda: segs,
                                     get_length=lambda: sum(len(s) for s in segs))
        selected, model = vntr_finder.select_illumina_reads(vntr, sam_utils.parse_sam(case["sam"]), case["unmapped"])
    finally:
        settings.MAX_ERROR_RATE, settings.USE_TRAINED_HMMS = saved
    want = case["selected"]
    assert [s.sequence for s in selected] == [w["sequence"] for w in want]
    assert [s.logp for s in selected] == [w["logp"] for w in want]
    assert [s.query_name for s in selected] == [w["query_name"] for w in want]
    assert [s.mapq for s in selected] == [w["mapq"] for w in want]
    assert [s.is_mapped for s in selected] == [w["is_mapped"] for w in want]
    by_name = {r.query_name: r for r in sam_utils.parse_sam(case["sam"]).reads}
    assert all(s.reference_start == by_name[s.query_name].reference_start for s in selected if s.is_mapped)
    assert any(s.is_mapped for s in selected) and not all(s.is_mapped for s in selected)


@pytest.mark.gpu
def test_cli_genotype_from_alignment_text(tmp_path):
    """python -m advntr_amd genotype --alignment sample.sam: mapped reads over the locus plus the unmapped records; the
    donor of the golden case carries 6 copies of the 17-base unit on both haplotypes."""
    import subprocess
    import sys
    from conftest import ROOT
    case = _golden_cases()[0]
    loci = [{"id": 3, "left": case["left"], "right": case["right"], "pattern": case["pattern"],
             "repeat_segments": case["repeat_segments"], "scaled_score": None, "chromosome": "chr5",
             "start_point": case["start_point"]}]
    (tmp_path / "loci.json").write_text(json.dumps(loci))
    sam = case["sam"] + "".join("u%d\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t*\n" % (i, s) for i, s in enumerate(case["unmapped"]))
    (tmp_path / "sample.sam").write_text(sam)
    out = subprocess.run([sys.executable, "-m", "advntr_amd", "genotype", "--loci", str(tmp_path / "loci.json"),
                          "--alignment", str(tmp_path / "sample.sam")], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out == "3\n6/6\n"
    # without the mapped reads (same unmapped reads as a FASTA) fewer reads support the call, the genotype stays
    (tmp_path / "reads.fa").write_text("".join(">u%d\n%s\n" % (i, s) for i, s in enumerate(case["unmapped"])))
    out = subprocess.run([sys.executable, "-m", "advntr_amd", "genotype", "--loci", str(tmp_path / "loci.json"),
                          "--reads", str(tmp_path / "reads.fa")], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out.startswith("3\n")
