"""Thin command line over the engine: `python -m advntr_amd genotype --loci loci.json --reads reads.fa`.

Not a re-implementation of the reference's CLI (/root/reference/advntr/__main__.py, advntr_commands.py: BAM/CRAM
input is out of scope).  It strings the GPU stages together
for reads that are already extracted: keyword prefilter -> per-locus Viterbi scoring of both strands -> recruit ->
Illumina aggregation (or PacBio dominant copy numbers) -> the reference's text output (genome_analyzer.py:158-170:
the VNTR id on one line, the genotype `a/b` on the next).

Loci come from --loci loci.json or from --models FILE.db, the reference's sqlite model database (table `vntrs`,
advntr/models.py:120-161), optionally narrowed with --vntr-id.  Repeat segments of unequal length need --align-repeats
(the library's own aligner stands in for the reference's `muscle` call).

loci.json: [{"id": 301645, "left": "...", "right": "...", "pattern": "...", "repeat_segments": ["...", ...],
             "scaled_score": -1.1}, ...]   (repeat_segments pre-aligned when more than one, equal length)
reads.fa : two-line FASTA (name line, sequence line), the format adVNTR-Filtering reads (filtering/main.cc:247-252).
"""
import argparse
import json
import sys

import numpy as np


def _read_fasta(path):
    lines = open(path).read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    return [lines[k][1:] for k in range(0, len(lines) - 1, 2)], [lines[k + 1] for k in range(0, len(lines) - 1, 2)]


def genotype(args):
    from . import filtering, hmm_utils, settings, vntr_finder
    settings.ALIGN_REPEATS = bool(args.align_repeats)
    from . import models
    if args.models:
        wanted = set(args.vntr_id or ())
        loci = [{"id": v.id, "left": v.left_flanking_region, "right": v.right_flanking_region, "pattern": v.pattern,
                 "repeat_segments": v.get_repeat_segments() or [v.pattern],
                 "scaled_score": v.scaled_score if v.scaled_score else None, "vntr": v}
                for v in models.load_unique_vntrs_data(args.models)
                if (not wanted or v.id in wanted) and v.left_flanking_region and v.right_flanking_region]
    else:
        loci = json.load(open(args.loci))
        for loc in loci:        # optional fields of the BED / VCF rows
            v = models.ReferenceVNTR(loc["id"], loc["pattern"], loc.get("start_point", 0), loc.get("chromosome", "chrUn"),
                                     loc.get("gene_name"), loc.get("annotation"), len(loc["repeat_segments"]))
            v.init_from_xml(loc["repeat_segments"], loc["left"], loc["right"])
            loc["vntr"] = v
    from . import genome_analyzer
    if args.outfmt == "bed":
        sys.stdout.write(genome_analyzer.bed_header(args.haploid))
    elif args.outfmt == "vcf":
        sys.stdout.write(genome_analyzer.vcf_header([loc["vntr"] for loc in loci], args.reads))

    def emit(loc, result):
        sys.stdout.write(genome_analyzer.genotype_row(args.outfmt, loc["vntr"], loc["id"], result, False, args.haploid))
    names, seqs = _read_fasta(args.reads)
    settings.MAX_ERROR_RATE = 0.3 if args.pacbio else 0.05                      # advntr_commands.py:66-71
    if args.pacbio:
        for loc in loci:
            geno, prob = vntr_finder.get_dominant_copy_numbers_from_spanning_reads(
                loc["left"], loc["right"], loc["repeat_segments"], loc["pattern"], [s.upper() for s in seqs],
                accuracy_filter=args.accuracy_filter, is_haploid=args.haploid)
            emit(loc, vntr_finder.GenotypeResult(geno, len(seqs), len(seqs), 0, prob))
        return 0
    # Illumina: prefilter every read (both strands) against all loci at once
    fasta = "".join(">%d\n%s\n" % (i, s.upper()) for i, s in enumerate(seqs))
    keywords = {int(loc["id"]): filtering.get_keywords_for_filtering(loc["left"], loc["repeat_segments"], loc["right"],
                                                                       loc["pattern"], True, 15) for loc in loci}
    _, ids_fwd = filtering.get_filtered_read_ids(fasta, keywords, min_matches=args.min_matches)
    rc = "".join(">%d\n%s\n" % (i, vntr_finder.reverse_complement(s.upper()) if "N" not in s.upper() else s.upper())
                 for i, s in enumerate(seqs))
    _, ids_rev = filtering.get_filtered_read_ids(rc, keywords, min_matches=args.min_matches)
    read_length = int(np.median([len(s) for s in seqs[:5]])) if seqs else 150     # vntr_finder.py:714-718
    if args.frameshift:        # genome_analyzer.py:260-271: the id, then the frameshift state label or None
        for loc in loci:
            vid = int(loc["id"])
            picked = sorted(set(int(n) for n in ids_fwd.get(vid, ())) | set(int(n) for n in ids_rev.get(vid, ())))
            copies = vntr_finder.get_copies_for_hmm(read_length, len(loc["pattern"]))
            model = hmm_utils.get_read_matcher_model(loc["left"][-read_length:], loc["right"][:read_length],
                                                     loc["repeat_segments"], copies)
            result = vntr_finder.find_frameshift(model, len(loc["pattern"]), sum(len(x) for x in loc["repeat_segments"]),
                                                 [seqs[i] for i in picked], loc.get("scaled_score"))
            sys.stdout.write("%s\n%s\n" % (loc["id"], result))
        return 0
    # all models in one native build, all (read, strand, locus) calls in one engine batch
    specs, cands = [], []
    for loc in loci:
        vid = int(loc["id"])
        picked = sorted(set(int(n) for n in ids_fwd.get(vid, ())) | set(int(n) for n in ids_rev.get(vid, ())))
        cands.append([seqs[i] for i in picked])
        specs.append((loc["left"][-read_length:], loc["right"][:read_length], loc["repeat_segments"],
                      vntr_finder.get_copies_for_hmm(read_length, len(loc["pattern"]))))
    models = hmm_utils.build_read_matcher_models(specs)
    scored_all = vntr_finder.score_reads_multi(models, cands, [loc.get("scaled_score") for loc in loci], True)
    for loc, scored in zip(loci, scored_all):
        scored = [s for s in scored if s is not None]
        selected = [s.summary for s in scored if s.recruited and s.repeat_bp > 2]          # vntr_finder.py:251
        res = vntr_finder.find_repeat_count_from_selected_reads(selected, accuracy_filter=args.accuracy_filter,
                                                                is_haploid=args.haploid)
        emit(loc, res)
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m advntr_amd")
    sub = ap.add_subparsers(dest="cmd")
    g = sub.add_parser("genotype", help="RU-count genotypes of the given loci from extracted reads (GPU)")
    src = g.add_mutually_exclusive_group(required=True)
    src.add_argument("--loci", help="JSON list of loci")
    src.add_argument("--models", help="sqlite model database in the reference's format (table vntrs)")
    g.add_argument("--vntr-id", type=int, action="append", help="with --models: genotype only these ids")
    g.add_argument("--align-repeats", action="store_true",
                   help="align repeat segments of unequal length with the built-in aligner (instead of refusing them)")
    g.add_argument("--reads", required=True)
    g.add_argument("--pacbio", action="store_true", help="reads are trimmed spanning long reads (error rate 0.3)")
    g.add_argument("--haploid", action="store_true")
    g.add_argument("--accuracy-filter", action="store_true")
    g.add_argument("--min-matches", type=int, default=5)
    g.add_argument("-fs", "--frameshift", action="store_true",
                   help="search for a frameshift in the VNTR instead of a copy number (vntr_finder.py:256-309)")
    g.add_argument("--outfmt", choices=["text", "bed", "vcf"], default="text",
                   help="result rows as the reference writes them (genome_analyzer.py:28-170)")
    args = ap.parse_args(argv)
    if args.cmd == "genotype":
        return genotype(args)
    ap.print_help()
    return 2


if __name__ == "__main__":
    sys.exit(main())
