"""Mapped reads for the scoring path without pysam: a plain SAM-text reader and the read filters of the reference.

The reference reads BAM/CRAM through pysam (advntr/sam_utils.py, vntr_finder.py:701-750), which is not part of this
framework (BAM/CRAM decoding is out of scope); what the scoring path needs of an alignment file is a list of
records with a handful of fields, which `samtools view -h file.bam region` prints as SAM text.  This module parses that
text into objects with pysam's attribute names (query_name, reference_start, reference_end, mapq, seq,
query_qualities, is_unmapped, is_duplicate) and mirrors
  is_low_quality_read                      advntr/utils.py:20-38
  get_reference_genome_of_alignment_file   advntr/sam_utils.py:32-39
"""
import re

from . import settings

_CIGAR = re.compile(r"(\d+)([MIDNSHP=X])")


class SamRead(object):
    __slots__ = ("query_name", "flag", "reference_name", "reference_start", "reference_end", "mapq", "cigar", "seq",
                 "query_qualities")

    @property
    def is_unmapped(self):
        return bool(self.flag & 0x4)

    @property
    def is_duplicate(self):
        return bool(self.flag & 0x400)


class SamFile(object):
    """references (the @SQ names, in header order) and reads (in file order)."""

    def __init__(self, references, reads):
        self.references, self.reads = references, reads

    def head(self, n):
        return self.reads[:n]

    def fetch(self, reference, start, end):
        """Mapped reads of `reference` overlapping the half-open interval [start, end), in file order (what
        pysam.AlignmentFile.fetch yields on a coordinate-sorted, indexed file)."""
        for r in self.reads:
            if r.reference_name != reference or r.is_unmapped:
                continue
            r_end = r.reference_end if r.reference_end is not None else r.reference_start + 1
            if r.reference_start < end and r_end > start:
                yield r


def parse_sam(text):
    """SAM text (header optional) -> SamFile.  POS is 1-based in the text, reference_start 0-based as in pysam;
    reference_end = one past the last aligned reference base (None for unmapped reads or '*' CIGARs)."""
    references, reads = [], []
    for line in text.split("\n"):
        if not line:
            continue
        if line.startswith("@"):
            if line.startswith("@SQ"):
                for field in line.split("\t")[1:]:
                    if field.startswith("SN:"):
                        references.append(field[3:])
            continue
        f = line.split("\t")
        r = SamRead()
        r.query_name, r.flag, r.reference_name = f[0], int(f[1]), f[2]
        r.reference_start, r.mapq, r.cigar = int(f[3]) - 1, int(f[4]), f[5]
        r.seq = f[9]
        r.query_qualities = None if f[10] == "*" else [ord(c) - 33 for c in f[10]]
        r.reference_end = None
        if not r.is_unmapped and r.cigar != "*":
            span = sum(int(n) for n, op in _CIGAR.findall(r.cigar) if op in "MDN=X")
            r.reference_end = r.reference_start + span
        reads.append(r)
    return SamFile(references, reads)


def get_reference_genome_of_alignment_file(samfile):
    result = None
    if '1' in samfile.references:
        result = 'GRCh37'
    for reference in samfile.references:
        if reference.startswith('chr'):
            result = 'HG19'
    return result


def is_low_quality_read(read):
    """utils.py:20-38: poor mapping quality, >= 10 % bases under Q20, or a low-quality base that is not followed by a
    better one within a quarter of that budget."""
    if read.mapq <= settings.MAPQ_CUTOFF:
        return True
    quals = read.query_qualities
    low = [i for i, q in enumerate(quals) if q < settings.QUALITY_SCORE_CUTOFF]
    if len(low) >= settings.LOW_QUALITY_BP_TO_DISCARD_READ * len(quals):
        return True
    run = int(settings.LOW_QUALITY_BP_TO_DISCARD_READ * len(quals) / 4)
    low_set = set(low)
    for i in low:
        if not any(j not in low_set for j in range(i + 1, i + run)):
            return True
    return False
