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



def _read_fasta(path):
    """Names and sequences of a FASTA file (sequence lines may be wrapped) or a four-line FASTQ file."""
    names, seqs = [], []
    with open(path) as fh:
        first = fh.readline()
        if first.startswith("@"):                                   # FASTQ: name, sequence, '+', qualities
            lines = [first] + fh.readlines()
            lines = [l.rstrip("\r\n") for l in lines]
            while lines and not lines[-1].strip():                  # trailing blank lines are not records
                lines.pop()
            if len(lines) % 4:
                raise ValueError("%s: FASTQ needs four lines per record (multi-line records are not supported)" % path)
            for i in range(0, len(lines), 4):
                if not lines[i].startswith("@") or not lines[i + 2].startswith("+"):
                    raise ValueError("%s: malformed FASTQ record at line %d" % (path, i + 1))
                names.append(lines[i][1:])
                seqs.append(lines[i + 1])
            return names, seqs
        chunks = None
        for line in [first] + fh.readlines():
            line = line.rstrip("\n")
            if line.startswith(">"):
                if chunks is not None:
                    seqs.append("".join(chunks))
                names.append(line[1:])
                chunks = []
            elif chunks is not None and line:
                chunks.append(line)
        if chunks is not None:
            seqs.append("".join(chunks))
    return names, seqs


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
    from . import comm as comm_mod, genome_analyzer, sharding
    comm = args.comm = comm_mod.init_from_env()       # one process per GPU under a launcher; None for a plain run
    rank = comm.rank if comm is not None else 0
    if rank == 0 and not args.frameshift:
        if args.outfmt == "bed":
            sys.stdout.write(genome_analyzer.bed_header(args.haploid))
        elif args.outfmt == "vcf":
            sys.stdout.write(genome_analyzer.vcf_header([loc["vntr"] for loc in loci], args.reads or args.alignment))

    def row(loc, result):
        return genome_analyzer.genotype_row(args.outfmt, loc["vntr"], loc["id"], result, False, args.haploid)

    def finish(rows):          # rows of all loci are back on rank 0, in locus order
        if rows is not None:
            sys.stdout.write("".join(rows))
        return 0
    names, seqs = _read_fasta(args.reads) if args.reads else ([], [])
    samfile = None
    if args.alignment:     # SAM text (`samtools view -h`): mapped reads are selected per locus, unmapped ones join --reads
        from . import sam_utils
        with open(args.alignment) as fh:
            samfile = sam_utils.parse_sam(fh.read())
        seqs = seqs + [r.seq for r in samfile.reads if r.is_unmapped]
    settings.MAX_ERROR_RATE = 0.3 if args.pacbio else 0.05                      # advntr_commands.py:66-71
    if args.pacbio:
        def pacbio_job(indices):
            if args.extract_spanning:
                # whole long reads: extraction, per-locus models, scoring and the call for all of this rank's loci at once
                # (find_repeat_count_from_pacbio_reads, vntr_finder.py:652-665; every read is a candidate of every locus)
                desc = [(loci[i]["left"], loci[i]["right"], loci[i]["repeat_segments"], loci[i]["pattern"]) for i in indices]
                res = vntr_finder.genotype_pacbio_loci(desc, [seqs] * len(desc), accuracy_filter=args.accuracy_filter,
                                                       is_haploid=args.haploid, chunks=max(1, min(8, len(desc) // 4)))
                return [row(loci[i], r) for i, r in zip(indices, res)]
            out = []
            for i in indices:
                loc = loci[i]
                spanning = [s.upper() for s in seqs]
                if args.extract_spanning:       # untrimmed long reads: keep the ones that span the VNTR, trimmed to its flanks
                    spanning = [t[0] for t in vntr_finder.extract_spanning_reads(loc["left"], loc["right"], seqs)[0]]
                geno, prob = vntr_finder.get_dominant_copy_numbers_from_spanning_reads(
                    loc["left"], loc["right"], loc["repeat_segments"], loc["pattern"], spanning,
                    accuracy_filter=args.accuracy_filter, is_haploid=args.haploid)
                out.append(row(loc, vntr_finder.GenotypeResult(geno, len(spanning), len(spanning), 0, prob)))
            return out
        return finish(sharding.run_sharded([len(loc["pattern"]) for loc in loci], pacbio_job, comm))
    # Illumina: prefilter every read against all loci at once.  The reference runs adVNTR-Filtering on the reads as
    # they are (genome_analyzer.py:173-199), so a read sequenced from the opposite strand only passes through keywords
    # that happen to be reverse-palindromic; --prefilter-both-strands (off by default = the reference's candidate sets)
    # also scans the reverse complements and unions the hits.
    fasta = "".join(">%d\n%s\n" % (i, s.upper()) for i, s in enumerate(seqs))
    keywords = {int(loc["id"]): filtering.get_keywords_for_filtering(loc["left"], loc["repeat_segments"], loc["right"],
                                                                       loc["pattern"], True, 15) for loc in loci}
    _, ids_fwd = filtering.get_filtered_read_ids(fasta, keywords, min_matches=args.min_matches)
    ids_rev = {}
    if args.prefilter_both_strands:
        rc = "".join(">%d\n%s\n" % (i, vntr_finder.reverse_complement(s.upper()) if "N" not in s.upper() else s.upper())
                     for i, s in enumerate(seqs))
        _, ids_rev = filtering.get_filtered_read_ids(rc, keywords, min_matches=args.min_matches)
    head = [len(r.seq) for r in samfile.head(5)] if samfile is not None else [len(s) for s in seqs[:5]]
    read_length = sorted(head)[len(head) // 2] if head else 150                   # vntr_finder.py:714-718
    def candidates(loc):
        vid = int(loc["id"])
        return [seqs[i] for i in sorted(set(int(n) for n in ids_fwd.get(vid, ())) | set(int(n) for n in ids_rev.get(vid, ())))]

    cands = [candidates(loc) for loc in loci]
    # whole loci go to ranks by estimated work (calls x states); one process per GPU under a launcher
    # (`python -m torch.distributed.run --nproc-per-node N -m advntr_amd genotype ...` or any other that sets RANK /
    # LOCAL_RANK / WORLD_SIZE / MASTER_PORT), rows gathered to rank 0 over RCCL; a plain loop over everything otherwise
    work = [max(1, len(c)) * (6 * read_length + 3 * vntr_finder.get_copies_for_hmm(read_length, len(loc["pattern"])) *
                              (len(loc["repeat_segments"][0]) + 1) + 18) for loc, c in zip(loci, cands)]
    if args.frameshift:        # genome_analyzer.py:260-271: the id, then the frameshift state label or None
        def frameshift_job(indices):
            out = []
            for i in indices:
                out.append(_frameshift_row(loci[i], cands[i], read_length, hmm_utils, vntr_finder))
            return out
        return finish(sharding.run_sharded(work, frameshift_job, comm))
    def genotype_job(indices):
        # all models of this rank's share in one native build, all (read, strand, locus) calls in one engine batch
        specs = [(loci[i]["left"][-read_length:], loci[i]["right"][:read_length], loci[i]["repeat_segments"],
                  vntr_finder.get_copies_for_hmm(read_length, len(loci[i]["pattern"]))) for i in indices]
        built = hmm_utils.build_read_matcher_models(specs)
        if samfile is not None:     # vntr_finder.py:701-767: mapped reads over the locus + the filtered unmapped ones
            for i in indices:
                loci[i]["vntr"].scaled_score = loci[i].get("scaled_score")
            picked, _ = vntr_finder.select_illumina_reads_multi([loci[i]["vntr"] for i in indices], samfile,
                                                                [cands[i] for i in indices], models=built)
            return [row(loci[i], vntr_finder.find_repeat_count_from_selected_reads(
                [s.summary for s in sel], accuracy_filter=args.accuracy_filter, is_haploid=args.haploid))
                for i, sel in zip(indices, picked)]
        # scoring (both strands), recruit rule, selection and the per-locus aggregation + genotype without a Python object
        # per read: one engine batch, then advntr_genotype_illumina on the summary records
        results = vntr_finder.genotype_loci(built, [cands[i] for i in indices], [loci[i].get("scaled_score") for i in indices],
                                            accuracy_filter=args.accuracy_filter, is_haploid=args.haploid)
        return [row(loci[i], res) for i, res in zip(indices, results)]
    return finish(sharding.run_sharded(work, genotype_job, comm))


def _frameshift_row(loc, cand, read_length, hmm_utils, vntr_finder):
    copies = vntr_finder.get_copies_for_hmm(read_length, len(loc["pattern"]))
    model = hmm_utils.get_read_matcher_model(loc["left"][-read_length:], loc["right"][:read_length],
                                             loc["repeat_segments"], copies)
    result = vntr_finder.find_frameshift(model, len(loc["pattern"]), sum(len(x) for x in loc["repeat_segments"]), cand,
                                         loc.get("scaled_score"))
    return "%s\n%s\n" % (loc["id"], result)


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
    g.add_argument("--reads", help="FASTA / FASTQ of extracted reads")
    g.add_argument("--alignment", help="SAM text of the sample (e.g. `samtools view -h sample.bam`): mapped reads over each "
                                       "VNTR are selected as vntr_finder.py:701-750 does, unmapped records are added to --reads")
    g.add_argument("--pacbio", action="store_true", help="reads are trimmed spanning long reads (error rate 0.3)")
    g.add_argument("--extract-spanning", action="store_true",
                   help="with --pacbio: the reads are whole long reads; find the ones that span each VNTR by aligning its "
                        "flanks (vntr_finder.py:324-371; GPU Smith-Waterman, parity with biopython unpinned) and trim them")
    g.add_argument("--haploid", action="store_true")
    g.add_argument("--accuracy-filter", action="store_true")
    g.add_argument("--min-matches", type=int, default=5)
    g.add_argument("--prefilter-both-strands", action="store_true",
                   help="also run the keyword prefilter on the reverse complement of every read (a superset of the "
                        "reference's candidate reads; the reference scans the reads as given, genome_analyzer.py:173-199)")
    g.add_argument("-fs", "--frameshift", action="store_true",
                   help="search for a frameshift in the VNTR instead of a copy number (vntr_finder.py:256-309)")
    g.add_argument("--outfmt", choices=["text", "bed", "vcf"], default="text",
                   help="result rows as the reference writes them (genome_analyzer.py:28-170)")
    args = ap.parse_args(argv)
    if args.cmd == "genotype":
        if not args.reads and not args.alignment:
            ap.error("genotype needs --reads and/or --alignment")
        if args.alignment and (args.pacbio or args.frameshift):
            ap.error("--alignment is the Illumina copy-number path")
        args.comm = None
        try:
            return genotype(args)
        finally:
            if args.comm is not None:
                args.comm.close()
    ap.print_help()
    return 2


if __name__ == "__main__":
    sys.exit(main())
