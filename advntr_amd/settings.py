"""Run-time constants of the scoring path (mirror of the fields of /root/reference/advntr/settings.py that
the path reads).  MAX_ERROR_RATE: 0.05 for Illumina, 0.3 for PacBio/nanopore (advntr_commands.py:66-71)."""
MAX_ERROR_RATE = 0.05
