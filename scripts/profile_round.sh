#!/bin/bash
# Run on the GPU box: kernel-trace stats + PMC passes of the default bench command -> <out_dir>
#   scripts/profile_round.sh gpurun_out/r01_v4
out=$1; root=$(pwd); mkdir -p $root/$out
python3 bench.py > $out/bench.json 2> $out/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/bench.py --no-cpu > $root/$out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $root/$out/pmc_$c -- python3 $root/bench.py --no-cpu --steps 1 --warmup 0 > $root/$out/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_ADD_F64 --output-format csv -d $root/$out/pmc_sq1 -- python3 $root/bench.py --no-cpu --steps 1 --warmup 0 > $root/$out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $root/$out/pmc_sq2 -- python3 $root/bench.py --no-cpu --steps 1 --warmup 0 > $root/$out/pmc_sq2.log 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "viterbi_columns" not in row.get("Kernel_Name", "") and "viterbi_rows" not in row.get("Kernel_Name", ""): continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
res = {k: tot[k] / n[k] for k in sorted(tot)}
json.dump(res, open(out + "/pmc_per_launch.json", "w"), indent=1)
print(json.dumps(res, indent=1))
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
