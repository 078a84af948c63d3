#!/bin/bash
# one PMC pass (instruction counts) + kernel time for a given engine build: scripts/pmc_quick.sh <out_dir> <lib.so>
out=$1; lib=$2; root=$(pwd)
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
ADVNTR_HIP_LIB=${lib:+$root/$lib} rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_ADD_F64 --output-format csv -d $root/$out/p -- python3 $root/bench.py --no-cpu --steps 1 --warmup 0 > $root/$out/p.log 2>&1
cd $root
ADVNTR_HIP_LIB=${lib:+$root/$lib} python3 bench.py --no-cpu --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib kernel_ms', d['roofline']['kernel_ms'])"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "viterbi_columns" not in row.get("Kernel_Name", "") and "viterbi_rows" not in row.get("Kernel_Name", ""): continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot): print("%-24s %16.0f per launch" % (k, tot[k] / n[k]))
PY
