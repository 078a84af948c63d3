#!/bin/bash
# SQ counters of the long-read kernel on config C4 for the shipped engine and the builds under exp/: scripts/pmc_c4.sh <out_dir>
out=$1; root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for f in shipped $(cd $root && ls exp/*.so 2>/dev/null); do
  tag=$(basename $f .so)
  if [ "$f" = shipped ]; then unset ADVNTR_HIP_LIB; else export ADVNTR_HIP_LIB=$root/$f; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $root/$out/$tag -- python3 $root/bench.py --workload c4 --no-cpu --steps 1 --warmup 0 > $root/$out/$tag.log 2>&1 < /dev/null
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for tag in sorted(os.path.basename(p)[:-4] for p in glob.glob(out + "/*.log")):
    tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (out, tag), recursive=True):
        for row in csv.DictReader(open(f)):
            if "viterbi_rows_long_kernel" not in row.get("Kernel_Name", ""): continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for f in glob.glob("%s/%s/**/*kernel_trace.csv" % (out, tag), recursive=True):
        for row in csv.DictReader(open(f)):
            if "viterbi_rows_long_kernel" in row.get("Kernel_Name", ""):
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print(tag, "ms under profiler: %.2f" % (sum(dur) / max(len(dur), 1)), {k: round(tot[k] / n[k] / 1e9, 3) for k in sorted(tot)})
PY
