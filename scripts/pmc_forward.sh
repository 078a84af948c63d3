#!/bin/bash
# Run on the GPU box: SQ counters of the sum-product kernel (scripts/forward_bench.py).  scripts/pmc_forward.sh gpurun_out/<tag>
out=$1; root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $root/$out/pmc_fwd -- python3 $root/scripts/forward_bench.py > $root/$out/pmc_fwd.log 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_WR --output-format csv -d $root/$out/pmc_fwd2 -- python3 $root/scripts/forward_bench.py > $root/$out/pmc_fwd2.log 2>&1 < /dev/null
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_fwd*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "forward_rows" not in row["Kernel_Name"]: continue
        tot[row["Dispatch_Id"] + f[-40:-30]][row["Counter_Name"]].append(float(row["Counter_Value"]))
agg = collections.defaultdict(list)
for d, c in tot.items():
    for k, v in c.items(): agg[k].append(sum(v))
for k, v in sorted(agg.items()):
    print(k, len(v), ["%.4g" % x for x in v[-4:]])
PY
