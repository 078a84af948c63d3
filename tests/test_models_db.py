"""The sqlite `vntrs` row format: advntr_amd.models against a database written by the reference's own models.py and
the objects its loader returns from it (tests/golden/vntrs_small.db + .json.gz, made by tests/golden/make_db_golden.py);
mirrors /root/reference/tests/test_models.py (save -> load round trip)."""
import os
import shutil
import sqlite3

from conftest import GOLDEN_DIR, load_golden
from advntr_amd import models

DB = os.path.join(GOLDEN_DIR, "vntrs_small.db")
FIELDS = ("id", "pattern", "start_point", "chromosome", "gene_name", "annotation", "estimated_repeats", "repeat_segments",
          "left_flanking_region", "right_flanking_region", "scaled_score", "non_overlapping")


def test_load_matches_the_reference_loader():
    want = load_golden("vntrs_small")
    got = models.load_unique_vntrs_data(DB)
    assert len(got) == len(want["vntrs"]) == 8
    for v, w in zip(got, want["vntrs"]):
        for f in FIELDS:
            assert getattr(v, f) == w[f], (v.id, f)
            assert type(getattr(v, f)) is type(w[f]), (v.id, f)
        assert v.get_length() == w["length"]
    assert models.get_largest_id_in_database(DB) == want["largest_id"]
    # quirks of the format the reference's loader has: NULL gene -> 'None', one segment -> [], NULL flank -> None
    assert any(v.gene_name == "None" for v in got)
    assert any(v.repeat_segments == [] for v in got)
    assert any(v.left_flanking_region is None for v in got)


def test_written_rows_equal_the_reference_writer(tmp_path):
    """Re-save every loaded locus with this writer: the rows must equal the ones the reference wrote (cell by cell,
    same sqlite types), except where the loader is lossy by design (NULL text comes back as 'None', a lone segment
    as [])."""
    mine = str(tmp_path / "out" / "copy.db")
    models.create_vntrs_database(mine)
    loaded = models.load_unique_vntrs_data(DB)
    for v in loaded:
        models.save_reference_vntr_to_database(v, mine)
    q = "SELECT id, nonoverlapping, chromosome, ref_start, gene_name, annotation, pattern, left_flanking, " \
        "right_flanking, repeats, scaled_score, typeof(scaled_score), typeof(ref_start) FROM vntrs ORDER BY id"
    ref_rows = sqlite3.connect(DB).execute(q).fetchall()
    my_rows = sqlite3.connect(mine).execute(q).fetchall()
    assert len(ref_rows) == len(my_rows)
    for r, m in zip(ref_rows, my_rows):
        for k, (a, b) in enumerate(zip(r, m)):
            if a is None:
                assert b in (None, "None")
            elif k == 9 and "," not in a:
                assert b == ""
            else:
                assert a == b, (r[0], k)
    schema = lambda p: sqlite3.connect(p).execute("SELECT sql FROM sqlite_master WHERE name='vntrs'").fetchone()[0]
    assert " ".join(schema(mine).split()) == " ".join(schema(DB).split())


def test_round_trip_update_delete(tmp_path):
    db = str(tmp_path / "m.db")
    models.create_vntrs_database(db)
    v = models.ReferenceVNTR(1, "CACA", 1000, "chr1", "GENE", "Coding", 2, scaled_score=0)
    v.init_from_xml(["CACA", "CACA"], "ACGT" * 10, "TTGA" * 10)
    models.save_reference_vntr_to_database(v, db)
    assert models.load_unique_vntrs_data(db) == [v]              # tests/test_models.py of the reference
    models.update_trained_score_in_database(1, -1.25, db)
    assert models.load_unique_vntrs_data(db)[0].scaled_score == -1.25
    assert models.get_largest_id_in_database(db) == 1
    models.delete_vntr_from_database(1, db)
    assert models.load_unique_vntrs_data(db) == [] and models.get_largest_id_in_database(db) == 0


def test_database_loci_build_models():
    """The loaded rows feed the native builder directly (equal-length segments only; see DESIGN.md on muscle)."""
    from advntr_amd import hmm_utils, vntr_finder
    loci = []
    for v in models.load_unique_vntrs_data(DB):
        segs = v.get_repeat_segments()
        if v.left_flanking_region and segs and len(set(len(s) for s in segs)) == 1:
            loci.append((v.left_flanking_region[-150:], v.right_flanking_region[:150], segs,
                         vntr_finder.get_copies_for_hmm(150, len(v.pattern))))
    assert len(loci) >= 3
    built = hmm_utils.build_read_matcher_models(loci)
    for (l, r, segs, c), m in zip(loci, built):
        L = len(segs[0])
        assert m.n_states == 6 * 150 + 3 * c * (L + 1) + 18
