#!/bin/bash
# SQ counter passes over one bench launch set (run on the GPU box): scripts/pmc_sq.sh <out_dir> [lib.so]
# Each pass is its own rocprofv3 run with --pmc only (no tracing), as the pool requires.
out=$1; lib=$2
root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"; do
  i=$((i+1))
  ADVNTR_HIP_LIB=${lib:+$root/$lib} rocprofv3 --pmc $set --output-format csv -d $root/$out/pass$i -- python3 $root/bench.py --no-cpu --steps 1 --warmup 0 > $root/$out/pass$i.log 2>&1
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "viterbi_columns" not in row.get("Kernel_Name", "") and "viterbi_rows" not in row.get("Kernel_Name", ""): continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
with open(out + "/sq_summary.txt", "w") as w:
    for k in sorted(tot):
        line = "%-28s %18.0f  (dispatches %d)" % (k, tot[k], n[k])
        print(line); w.write(line + "\n")
PY
