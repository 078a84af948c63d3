#!/bin/bash
# Instruction counters of the S300 (and C1) bench launch for several engine builds: scripts/pmc_ab.sh <out_dir> lib1.so lib2.so ... ("-" = shipped)
out=$1; shift; root=$(pwd)
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  path=$root/$lib; [ "$lib" = "-" ] && path=
  tag=$(basename $lib .so); [ "$lib" = "-" ] && tag=shipped
  for wl in s300 c1; do
    ADVNTR_HIP_LIB=$path rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv \
      -d $root/$out/${tag}_$wl -- python3 $root/bench.py --workload $wl --no-cpu --no-s300 --steps 2 --warmup 1 > $root/$out/${tag}_$wl.log 2>&1
  done
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_s300") + glob.glob(out + "/*_c1")):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "viterbi_rows_kernel" not in row.get("Kernel_Name", ""): continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    print(os.path.basename(d), {k: round(tot[k] / n[k] / 1e6, 1) for k in sorted(tot)})
PY
