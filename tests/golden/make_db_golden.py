#!/usr/bin/env python3
"""Generate tests/golden/vntrs_small.db (+ .json.gz) by RUNNING THE REFERENCE's model-database code in this container
(advntr/models.py: create_vntrs_database, save_reference_vntr_to_database, update_trained_score_in_database,
load_unique_vntrs_data).  TEST INFRASTRUCTURE; only data is written:
  vntrs_small.db       the sqlite file the reference wrote,
  vntrs_small.json.gz  the fields of the ReferenceVNTR objects the reference loads back from it.

    python oracle/tools/build_reference.py && python tests/golden/make_db_golden.py
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_BUILD = os.environ.get("ADVNTR_REF_BUILD", "/tmp/advntr_ref_build")
sys.path[:0] = [os.path.join(REPO, "oracle", "tools", "nx111"), os.path.join(REPO, "oracle", "tools", "stubs"), REF_BUILD]

import types                                          # noqa: E402
import numpy as np                                    # noqa: E402
# import-only placeholders for the biopython names advntr/models.py:6 and reference_vntr.py:1 import at module level
# (biopython is absent here); the database functions exercised below never touch them
import Bio                                            # noqa: E402
for _name in ("Seq", "SeqRecord", "SeqIO", "pairwise2", "SearchIO"):
    _m = types.ModuleType("Bio." + _name)
    sys.modules["Bio." + _name] = _m
    setattr(Bio, _name, _m)
from advntr import models, settings                   # noqa: E402  (the reference)
from advntr.reference_vntr import ReferenceVNTR       # noqa: E402

db = os.path.join(HERE, "vntrs_small.db")
if os.path.exists(db):
    os.remove(db)
models.create_vntrs_database(db)
rng = np.random.default_rng(77)
dna = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, n))
rows = []
for k in range(8):
    plen = int(rng.integers(6, 40))
    pattern = dna(plen)
    segs = []
    for _ in range(int(rng.integers(2, 7))):
        s = list(pattern)
        if rng.random() < 0.5:
            s[int(rng.integers(0, plen))] = "ACGT"[int(rng.integers(0, 4))]
        if k % 3 == 2 and rng.random() < 0.5:
            s.pop(int(rng.integers(0, len(s))))          # unequal lengths, as real loci have
        segs.append("".join(s))
    v = ReferenceVNTR(100 + 7 * k, pattern, int(rng.integers(1000, 10 ** 8)), "chr%d" % (1 + k), None if k % 2 else "GENE%d" % k,
                      None if k % 4 == 0 else "Coding", len(segs), scaled_score=0)
    v.init_from_xml(segs, dna(500), dna(500))
    if k == 3:
        v.non_overlapping = False
    if k == 5:
        v.init_from_xml([pattern], dna(40), dna(40))      # a single segment holds no comma: it loads back as []
    if k == 6:
        v.init_from_xml(segs, None, None)                 # NULL flanks
    models.save_reference_vntr_to_database(v, db)
settings.TRAINED_MODELS_DB = db
models.update_trained_score_in_database(100 + 7 * 2, -1.1375)
models.update_trained_score_in_database(100 + 7 * 4, -0.93)
loaded = models.load_unique_vntrs_data(db)
out = []
for v in loaded:
    out.append({"id": v.id, "pattern": v.pattern, "start_point": v.start_point, "chromosome": v.chromosome,
                "gene_name": v.gene_name, "annotation": v.annotation, "estimated_repeats": v.estimated_repeats,
                "repeat_segments": v.repeat_segments, "left_flanking_region": v.left_flanking_region,
                "right_flanking_region": v.right_flanking_region, "scaled_score": v.scaled_score,
                "non_overlapping": v.non_overlapping, "length": v.get_length()})
with gzip.open(os.path.join(HERE, "vntrs_small.json.gz"), "wt") as f:
    json.dump({"vntrs": out, "largest_id": models.get_largest_id_in_database()}, f)
print("wrote %d loci, db %d bytes" % (len(out), os.path.getsize(db)))
