"""bench.py in pieces: cli (arguments, launcher), main (the line), passes, rehearsal, records / illumina / upstream (sub-records)."""
