"""The sqlite model database either side of the scoring path: the `vntrs` table that holds, per locus, the inputs of
get_read_matcher_model (pattern, flanking regions, repeat segments) and the trained recruitment score.

Mirror of the row format and the accessors of /root/reference/advntr/models.py:
  create_vntrs_database             models.py:120-131   (schema)
  load_unique_vntrs_data            models.py:134-161   (row -> ReferenceVNTR, 'None' strings, comma-joined segments)
  save_reference_vntr_to_database   models.py:207-220
  update_trained_score_in_database  models.py:189-194
  get_largest_id_in_database        models.py:223-231,  delete_vntr_from_database  models.py:234-239
and of the data fields of ReferenceVNTR (/root/reference/advntr/reference_vntr.py:7-67).  Building a database from
VNTRseek output / a reference genome (models.py:20-118, reference_vntr.py:69-140) is out of scope.
"""
import os
import sqlite3

from . import settings

_COLUMNS = ("id, nonoverlapping, chromosome, ref_start, gene_name, annotation, pattern, left_flanking, "
            "right_flanking, repeats, scaled_score")


class ReferenceVNTR(object):
    def __init__(self, vntr_id, pattern, start_point, chromosome, gene_name, annotation, estimated_repeats=None,
                 chromosome_sequence=None, scaled_score=0):
        self.non_overlapping = True
        self.has_homologous = False
        self.id = vntr_id
        self.pattern = pattern
        self.start_point = start_point
        self.chromosome = chromosome
        self.gene_name = gene_name
        self.annotation = annotation
        self.estimated_repeats = estimated_repeats
        self.repeat_segments = []
        self.left_flanking_region = None
        self.right_flanking_region = None
        self.chromosome_sequence = chromosome_sequence
        self.scaled_score = scaled_score

    def _key(self):
        return (self.non_overlapping, self.id, self.pattern, self.start_point, self.chromosome, self.gene_name,
                self.annotation, self.estimated_repeats, sorted(self.repeat_segments), self.left_flanking_region,
                self.right_flanking_region, self.chromosome_sequence, self.scaled_score)

    def __eq__(self, other):
        return isinstance(other, ReferenceVNTR) and self._key() == other._key()

    def __ne__(self, other):
        return not self == other

    def init_from_xml(self, repeat_segments, left_flanking_region, right_flanking_region):
        self.repeat_segments = repeat_segments
        self.left_flanking_region = None if left_flanking_region == 'None' else left_flanking_region
        self.right_flanking_region = None if right_flanking_region == 'None' else right_flanking_region

    def is_non_overlapping(self):
        return self.non_overlapping

    def has_homologous_vntr(self):
        return self.has_homologous

    def get_length(self):
        return sum(len(e) for e in self.repeat_segments)

    def get_repeat_segments(self):
        return self.repeat_segments


def _db_path(db_file):
    path = db_file if db_file is not None else getattr(settings, "TRAINED_MODELS_DB", None)
    if path is None:
        raise ValueError("no database given and settings.TRAINED_MODELS_DB is not set")
    return path


def create_vntrs_database(db_file):
    folder = os.path.dirname(db_file)
    if folder and not os.path.exists(folder):
        os.makedirs(folder)
    db = sqlite3.connect(db_file)
    db.execute("CREATE TABLE vntrs(id INTEGER PRIMARY KEY, nonoverlapping TEXT, chromosome TEXT, ref_start INTEGER, "
               "gene_name TEXT, annotation TEXT, pattern TEXT, left_flanking TEXT, right_flanking TEXT, repeats TEXT, "
               "scaled_score REAL default 0)")
    db.commit()
    db.close()


def load_unique_vntrs_data(db_file=None):
    db = sqlite3.connect(_db_path(db_file))
    vntrs = []
    for row in db.execute("SELECT %s FROM vntrs" % _COLUMNS):
        # every non-numeric cell goes through str(): NULL becomes the string 'None' (models.py:145-150)
        cells = [c if isinstance(c, (int, float)) else str(c) for c in row]
        vntr_id, overlap, chrom, start, gene, annotation, pattern, left_flank, right_flank, segments, score = cells
        repeat_segments = segments.split(',') if "," in segments else []
        vntr = ReferenceVNTR(int(vntr_id), pattern, int(start), chrom, gene, annotation, len(repeat_segments),
                             scaled_score=score)
        vntr.init_from_xml(repeat_segments, left_flank, right_flank)
        vntr.non_overlapping = overlap == 'True'
        vntrs.append(vntr)
    db.close()
    return vntrs


def save_reference_vntr_to_database(ref_vntr, db_file=None):
    db = sqlite3.connect(_db_path(db_file))
    db.execute("INSERT INTO vntrs(%s) VALUES(?,?,?,?,?,?,?,?,?,?,?)" % _COLUMNS,
               (ref_vntr.id, "True" if ref_vntr.non_overlapping else "False", ref_vntr.chromosome,
                ref_vntr.start_point, ref_vntr.gene_name, ref_vntr.annotation, ref_vntr.pattern,
                ref_vntr.left_flanking_region, ref_vntr.right_flanking_region,
                ','.join(ref_vntr.get_repeat_segments()), ref_vntr.scaled_score))
    db.commit()
    db.close()


def update_trained_score_in_database(vntr_id, scaled_recruitment_score, db_file=None):
    db = sqlite3.connect(_db_path(db_file))
    db.execute("UPDATE vntrs SET scaled_score=? WHERE id=?", (scaled_recruitment_score, vntr_id))
    db.commit()
    db.close()


def get_largest_id_in_database(db_file=None):
    db = sqlite3.connect(_db_path(db_file))
    result = 0
    for row in db.execute("SELECT MAX(id) FROM vntrs"):
        if row[0] is not None:
            result = row[0]
    db.close()
    return result


def delete_vntr_from_database(vntr_id, db_file=None):
    db = sqlite3.connect(_db_path(db_file))
    db.execute("DELETE FROM vntrs WHERE id=?", (int(vntr_id),))
    db.commit()
    db.close()


def init_from_vntrseek_data(vntr, chromosome_sequence, flanking_region_size=500):
    """ReferenceVNTR.init_from_vntrseek_data (reference_vntr.py:42-48) with the chromosome passed in: the region the
    VNTRseek estimate covers (cut at the first N) is segmented into repeat units by the repeat-finder HMM on the GPU
    (hmm_utils.find_repeat_segments), then the flanking regions are taken around the segmented VNTR."""
    from .hmm_utils import find_repeat_segments
    estimated_length = int(len(vntr.pattern) * vntr.estimated_repeats)
    region = chromosome_sequence[vntr.start_point:vntr.start_point + estimated_length].upper()
    if region.find('N') != -1:
        region = region[:region.find('N')]
    vntr.repeat_segments = find_repeat_segments(vntr.pattern, vntr.estimated_repeats, region)
    end_of_repeats = vntr.start_point + vntr.get_length()
    vntr.left_flanking_region = chromosome_sequence[vntr.start_point - flanking_region_size:vntr.start_point].upper()
    vntr.right_flanking_region = chromosome_sequence[end_of_repeats:end_of_repeats + flanking_region_size].upper()
    vntr.chromosome_sequence = None
    return vntr
