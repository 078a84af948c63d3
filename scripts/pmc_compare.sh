#!/bin/bash
# Run on the GPU box: SQ counters of the bench kernel for the shipped engine and for experimental builds under exp/
# scripts/pmc_compare.sh gpurun_out/pmc_cmp [bench args]
out=$1; shift; root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for f in shipped $(cd $root && ls exp/*.so 2>/dev/null); do
  tag=$(basename $f .so)
  if [ "$f" = shipped ]; then unset ADVNTR_HIP_LIB; else export ADVNTR_HIP_LIB=$root/$f; fi
  i=0
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $root/$out/${tag}_$i -- python3 $root/bench.py --no-cpu --no-s300 --steps 1 --warmup 0 "$@" > $root/$out/${tag}_$i.log 2>&1 < /dev/null
  done
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for tag in sorted(set(os.path.basename(p).rsplit("_", 1)[0] for p in glob.glob(out + "/*_1"))):
    tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (out, tag), recursive=True):
        for row in csv.DictReader(open(f)):
            if "viterbi_rows_kernel<5, 2>" not in row.get("Kernel_Name", ""): continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    for f in glob.glob("%s/%s_*/**/*kernel_trace.csv" % (out, tag), recursive=True):
        for row in csv.DictReader(open(f)):
            if "viterbi_rows_kernel<5, 2>" in row.get("Kernel_Name", ""):
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print(tag, "ms under profiler: %.3f" % (sum(dur) / max(len(dur), 1)), {k: round(tot[k] / n[k] / 1e6, 2) for k in sorted(tot)})
PY
