"""The text / BED / VCF result rows against what the reference's own writers print
(tests/golden/output_rows.json.gz, made by tests/golden/make_output_golden.py)."""
import types

from conftest import load_golden
from advntr_amd import genome_analyzer as ga


def _vntrs(g):
    out = {}
    for v in g["vntrs"]:
        segs = v["repeat_segments"]
        out[v["id"]] = types.SimpleNamespace(id=v["id"], chromosome=v["chromosome"], start_point=v["start_point"],
                                             gene_name=v["gene_name"], pattern=v["pattern"],
                                             estimated_repeats=v["estimated_repeats"],
                                             get_repeat_segments=lambda s=segs: s,
                                             get_length=lambda s=segs: sum(len(x) for x in s))
    return out


def test_rows_and_headers_equal_the_reference_output():
    g = load_golden("output_rows")
    assert ga.VERSION == g["version"]
    vntrs = _vntrs(g)
    n = 0
    for case in g["cases"]:
        hap, fmt = case["haploid"], case["outfmt"]
        header = {"text": "", "bed": ga.bed_header(hap), "vcf": ga.vcf_header(list(vntrs.values()), g["input_file"])}[fmt]
        assert header == case["header"], (fmt, hap)
        for row in case["rows"]:
            cn = None if row["copy_numbers"] is None else tuple(row["copy_numbers"])
            res = types.SimpleNamespace(copy_numbers=cn, recruited_reads_count=row["dp"], spanning_reads_count=row["sr"],
                                        flanking_reads_count=row["fr"], maximum_likelihood=row["ml"])
            got = ga.genotype_row(fmt, vntrs[row["vntr_id"]], row["vntr_id"], res, row["error"], hap)
            assert got == row["text"], (fmt, hap, row["vntr_id"], cn, row["error"])
            n += 1
    assert n == 288
