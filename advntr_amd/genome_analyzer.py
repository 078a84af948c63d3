"""Result rows downstream of the scoring path: the text / BED / VCF lines of
/root/reference/advntr/genome_analyzer.py:28-170 (GenomeAnalyzer.print_genotype and its three writers), as functions
that return the text instead of printing it.  `vntr` is an advntr_amd.models.ReferenceVNTR (or anything with its
fields), `result` a vntr_finder.GenotypeResult.  The analysis drivers of that class (BAM/CRAM input, pysam) are out of
scope; `python -m advntr_amd genotype --outfmt ...` uses these writers."""

VERSION = "1.5.0"       # what `from advntr import __version__` gives in the reference tree (advntr/__init__.py)


def bed_header(is_haploid=False):
    return '#CHROM\tStart\tEnd\tVNTR_ID\tGene\tMotif\tRefCopy\t%s\n' % ('R' if is_haploid else 'R1\tR2')


def genotype_in_bed_format(vntr, vntr_id, copy_numbers, encountered_error=False, is_haploid=False):
    start = vntr.start_point
    end = start + vntr.get_length()
    if encountered_error:
        repeats = "Error"
    elif copy_numbers is None:
        repeats = 'None' if is_haploid else 'None\tNone'
    elif is_haploid:
        repeats = str(copy_numbers[0])
    else:
        repeats = '\t'.join(str(cn) for cn in sorted(copy_numbers))
    return '%s\t%s\t%s\t%s\t%s\t%s\t%s\t%s\n' % (vntr.chromosome, start, end, vntr_id, vntr.gene_name, vntr.pattern,
                                                 len(vntr.get_repeat_segments()), repeats)


def vcf_header(vntrs, input_file, version=VERSION):
    lines = ["##fileformat=VCFv4.2",
             "##source=adVNTR ver. {}".format(version),
             '##INFO=<ID=END,Number=1,Type=Integer,Description="End position of variant">',
             '##INFO=<ID=VID,Number=1,Type=Integer,Description="VNTR ID">',
             '##INFO=<ID=RU,Number=1,Type=String,Description="Repeat motif">',
             '##INFO=<ID=RC,Number=1,Type=Integer,Description="Reference repeat unit count">',
             '##FILTER=<ID=ERR,Description="Error occurred while genotyping">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Read depth">',
             '##FORMAT=<ID=SR,Number=1,Type=Integer,Description="Spanning read count">',
             '##FORMAT=<ID=FR,Number=1,Type=Integer,Description="Flanking read count">',
             '##FORMAT=<ID=ML,Number=1,Type=Float,Description="Maximum likelihood">']
    for contig in sorted(set(v.chromosome[3:] for v in vntrs)):
        lines.append('##contig=<ID={}>'.format(contig))
    sample = input_file.strip().split("/")[-1].split(".")[0]
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + sample)
    return "\n".join(lines) + "\n"


def genotype_in_vcf(vntr, vntr_id, result, encountered_error=False):
    end = vntr.start_point + vntr.get_length()
    ref = ''.join(vntr.get_repeat_segments())
    motif = vntr.pattern
    gt, diff_count, diff_index = [], 0, -1
    if result.copy_numbers is None:
        gt = ['.', '.']
    else:
        for index, copy_number in enumerate(result.copy_numbers):
            if copy_number != vntr.estimated_repeats:
                diff_index = index
                diff_count += 1
                gt.append(diff_count)
                if len(set(result.copy_numbers)) == 1:          # homozygous for one non-reference allele
                    gt.append(diff_count)
                    break
            else:
                gt.append(0)
    if diff_count == 2:
        alt = motif * result.copy_numbers[0] + "," + motif * result.copy_numbers[1]
    elif diff_count == 1:
        alt = motif * result.copy_numbers[diff_index]
    else:
        alt = '.'
    info = "END=" + str(end) + ";VID=" + str(vntr_id) + ";RU=" + motif + ";RC=" + str(vntr.estimated_repeats)
    sample = "%s/%s:%s:%s:%s:%s" % (gt[0], gt[1], result.recruited_reads_count, result.spanning_reads_count,
                                    result.flanking_reads_count, "{0:.4f}".format(result.maximum_likelihood))
    return "{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\t{}\n".format(vntr.chromosome, vntr.start_point, '.', ref, alt, '.',
                                                             "ERR" if encountered_error else '.', info,
                                                             "GT:DP:SR:FR:ML", sample)


def genotype_in_text_format(vntr_id, copy_numbers, encountered_error=False, is_haploid=False):
    if encountered_error:
        return "%s\nError\n" % vntr_id
    if copy_numbers is None:
        return "%s\nNone\n" % vntr_id
    if is_haploid:
        return "%s\n%s\n" % (vntr_id, copy_numbers[0])
    return "%s\n%s\n" % (vntr_id, '/'.join(str(cn) for cn in sorted(copy_numbers)))


def genotype_row(outfmt, vntr, vntr_id, result, encountered_error=False, is_haploid=False):
    """GenomeAnalyzer.print_genotype (genome_analyzer.py:28-34)."""
    if outfmt == 'bed':
        return genotype_in_bed_format(vntr, vntr_id, result.copy_numbers, encountered_error, is_haploid)
    if outfmt == 'vcf':
        return genotype_in_vcf(vntr, vntr_id, result, encountered_error)
    return genotype_in_text_format(vntr_id, result.copy_numbers, encountered_error, is_haploid)
