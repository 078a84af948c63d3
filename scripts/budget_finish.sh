#!/bin/bash
# Instruction budget of the finish phase (tail states / traceback / summaries) of viterbi_rows_kernel: SQ_INSTS_VALU per launch of
# the S300 and C1 bench launches for the shipped build and for builds that leave one piece out (exp/budget_*.so, made by
# scripts/build_variant.sh with -DADVNTR_BUDGET_*; their results are wrong by construction, only the counters are read).
#   scripts/budget_finish.sh <out_dir>        (on the GPU box; the variants must have been built before: they travel with the tree)
out=$1; root=$(pwd)
mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
for v in shipped no_tail no_tb no_summary no_finish; do
  lib=$root/exp/budget_$v.so; [ $v = shipped ] && lib=
  for wl in s300 c1; do
    ADVNTR_HIP_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv \
      -d $root/$out/${v}_$wl -- python3 $root/bench.py --workload $wl --no-cpu --no-s300 --steps 2 --warmup 1 > $root/$out/${v}_$wl.log 2>&1
    ADVNTR_HIP_LIB=$lib python3 $root/bench.py --workload $wl --no-cpu --no-s300 --steps 10 --warmup 2 2>/dev/null > $root/$out/${v}_$wl.json
  done
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for v in ("shipped", "no_tail", "no_tb", "no_summary", "no_finish"):
    for wl in ("s300", "c1"):
        tot = collections.defaultdict(float); n = collections.defaultdict(int)
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, v, wl), recursive=True):
            for row in csv.DictReader(open(f)):
                if "viterbi_rows_kernel" not in row.get("Kernel_Name", ""): continue
                tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        try:
            ms = json.load(open("%s/%s_%s.json" % (out, v, wl)))["roofline"]["kernel_ms"]
        except Exception:
            ms = None
        res["%s/%s" % (wl, v)] = dict({k: tot[k] / n[k] for k in tot}, kernel_ms=ms)
json.dump(res, open(out + "/budget.json", "w"), indent=1)
for k in sorted(res): print(k, {a: (round(b / 1e6, 1) if a != "kernel_ms" else b) for a, b in res[k].items()})
PY
