"""Run-time constants of the scoring path (mirror of the fields of /root/reference/advntr/settings.py that
the path reads).  MAX_ERROR_RATE: 0.05 for Illumina, 0.3 for PacBio/nanopore (advntr_commands.py:66-71)."""
MAX_ERROR_RATE = 0.05
# The reference aligns the repeat units of a locus with the external program `muscle` (profile_hmm.py:166-171).  That
# program is not part of this framework: by default repeat units must come aligned (equal-length rows) and unequal
# lengths are refused; True lets the library's own progressive aligner (csrc/repeat_msa.h) stand in -- models built
# that way are NOT claimed to equal a muscle-based build.
ALIGN_REPEATS = False
TRAINED_MODELS_DB = None        # path of the sqlite `vntrs` database (advntr/settings.py:10); see advntr_amd/models.py
USE_TRAINED_HMMS = False        # advntr/settings.py:9: load / store per-locus HMMs as JSON (vntr_finder.py:116-138)
TRAINED_HMMS_DIR = 'vntr_data/'
# read filters of the mapped-read loop (advntr/settings.py:24-27,34)
QUALITY_SCORE_CUTOFF = 20
LOW_QUALITY_BP_TO_DISCARD_READ = 0.10
MAPQ_CUTOFF = 0
MIN_READ_LENGTH = None
