#!/bin/bash
# Run on the GPU box: SQ counters of the prefilter kernel (scripts/filter_bench.py).  scripts/pmc_filter.sh gpurun_out/<tag>
out=$1; root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
NO_CPU=1 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $root/$out/pmc_kwf -- python3 $root/scripts/filter_bench.py > $root/$out/pmc_kwf.log 2>&1 < /dev/null
NO_CPU=1 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $root/$out/pmc_kwf2 -- python3 $root/scripts/filter_bench.py > $root/$out/pmc_kwf2.log 2>&1 < /dev/null
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_kwf*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "keyword_filter" not in row["Kernel_Name"]: continue
        agg[(f[-40:-30], row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
# the largest dispatch of each pass = the 2 M-read launch
best = {}
for (f, d), c in agg.items():
    for k, v in c.items():
        if v > best.get(k, 0): best[k] = v
json.dump(best, open(out + "/filter_pmc.json", "w"), indent=1)
print(json.dumps(best, indent=1))
PY
