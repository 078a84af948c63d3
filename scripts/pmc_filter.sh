#!/bin/bash
# PMC pass of the prefilter bench: scripts/pmc_filter.sh gpurun_out/<tag>
out=$1; root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
export NO_CPU=1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $root/$out/p1 -- python3 $root/scripts/filter_bench.py > $root/$out/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $root/$out/p2 -- python3 $root/scripts/filter_bench.py > $root/$out/p2.log 2>&1
# (a third pass with FETCH_SIZE + TCC_HIT_sum + TCC_MISS_sum aborted inside rocprofv3 on this pool and then sat until the
# call's limit: every pass now runs under its own timeout, and that combination is not requested)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/$out/p3 -- python3 $root/scripts/filter_bench.py > $root/$out/p3.log 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
best = {}
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "keyword_filter" in r.get("Kernel_Name", "")]
    # the largest dispatch (2 M reads) of each counter
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in by.items(): best[k] = max(v)
for k in sorted(best): print("%-24s %16.0f" % (k, best[k]))
PY
