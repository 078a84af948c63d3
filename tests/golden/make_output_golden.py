#!/usr/bin/env python3
"""Generate tests/golden/output_rows.json.gz by RUNNING THE REFERENCE's result writers
(GenomeAnalyzer.print_bed_header / print_genotype_in_bed_format / print_vcf_header / print_genotype_in_vcf /
print_genotype_in_text_format, advntr/genome_analyzer.py:28-170) and capturing what they print.  TEST INFRASTRUCTURE;
only data is written.  genome_analyzer.py imports pysam/biopython at module level, so the method bodies are compiled
at run time from the reference file with `ast` and executed unchanged; nothing of them is stored.

    python tests/golden/make_output_golden.py
"""
import ast
import contextlib
import gzip
import io
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def load_writer_class():
    path = "/root/reference/advntr/genome_analyzer.py"
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GenomeAnalyzer"][0]
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name.startswith("print_")]
    mod = ast.Module(body=[ast.ClassDef(name="GenomeAnalyzer", bases=[], keywords=[], body=keep, decorator_list=[])],
                     type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {}
    exec(compile(mod, path, "exec"), ns)
    return ns["GenomeAnalyzer"]


def main():
    G = load_writer_class()
    vntrs = {}
    specs = [(301645, "chr21", 45196324, "CSTB", "CGCGGGGCGGGG", ["CGCGGGGCGGGG", "CGCGGGGCGGGG", "CGCGGGGCGGGC"], 3),
             (25561, "chr1", 1000, "None", "ACGTT", ["ACGTT", "ACGT", "ACGTT", "ACGTT"], 4),
             (7, "chrX", 5, None, "GA", ["GA", "GA"], 2)]
    for vid, chrom, start, gene, motif, segs, est in specs:
        vntrs[vid] = types.SimpleNamespace(id=vid, chromosome=chrom, start_point=start, gene_name=gene, pattern=motif,
                                           estimated_repeats=est, get_repeat_segments=lambda s=segs: s,
                                           get_length=lambda s=segs: sum(len(x) for x in s))
    results = [(2, 5), (3, 3), (5, 5), (3, 4), (4, 3), None, (2, 2), (4, 4)]
    cases = []
    for haploid in (False, True):
        g = G.__new__(G)
        g.is_haploid = haploid
        g.target_vntr_ids = list(vntrs)
        g.vntr_finder = {vid: types.SimpleNamespace(reference_vntr=v) for vid, v in vntrs.items()}
        g.input_file = "/data/run 1/sample_7.sorted.bam "
        for fmt in ("text", "bed", "vcf"):
            g.outfmt = fmt
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                if fmt == "bed":
                    g.print_bed_header()
                if fmt == "vcf":
                    g.print_vcf_header()
            header = buf.getvalue()
            rows = []
            for vid in vntrs:
                for k, cn in enumerate(results):
                    for err in (False, True):
                        res = types.SimpleNamespace(copy_numbers=cn, recruited_reads_count=30 + k, spanning_reads_count=10 + k,
                                                    flanking_reads_count=k, maximum_likelihood=0.123456789 * (k + 1) % 1.0)
                        buf = io.StringIO()
                        with contextlib.redirect_stdout(buf):
                            g.print_genotype(vid, res, err)
                        rows.append({"vntr_id": vid, "copy_numbers": cn, "error": err, "dp": 30 + k, "sr": 10 + k, "fr": k,
                                     "ml": res.maximum_likelihood, "text": buf.getvalue()})
            cases.append({"haploid": haploid, "outfmt": fmt, "header": header, "rows": rows})
    import advntr
    out = {"version": advntr.__version__, "input_file": "/data/run 1/sample_7.sorted.bam ",
           "vntrs": [{"id": v, "chromosome": c, "start_point": s, "gene_name": g_, "pattern": m, "repeat_segments": segs,
                      "estimated_repeats": e} for v, c, s, g_, m, segs, e in specs],
           "cases": cases}
    with gzip.open(os.path.join(HERE, "output_rows.json.gz"), "wt") as fh:
        json.dump(out, fh)
    print("cases", len(cases), "rows", sum(len(c["rows"]) for c in cases), "version", advntr.__version__)


if __name__ == "__main__":
    main()
