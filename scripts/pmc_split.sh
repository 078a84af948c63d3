#!/bin/bash
# PMC counters of the split launch (measured, not kept: HISTORY.md, round 6) per kernel and shape, separate rocprofv3 --pmc passes.
# The code lives in commit a878944 only.  Before the gpurun call, in the build container:
#   git worktree add /tmp/split_tree a878944 && scripts/build_variant.sh split /tmp/split_tree
#   sed 's/_lib.FLAG_SPLIT_FINISH/256/' /tmp/split_tree/scripts/ab_split.py > exp/ab_split.py    (+ a SHAPE filter on its shape list)
# then on the GPU box: bash scripts/pmc_split.sh  ->  gpurun_out/r6_splitpmc/split_pmc.json (profiles/r06_split_finish_ab.json)
root=$(pwd); out=$root/gpurun_out/r6_splitpmc; mkdir -p $out
export ADVNTR_HIP_LIB=$root/exp/split.so
cd /tmp && export TMPDIR=/tmp
for shape in ref150 s300; do
  export SHAPE=$shape
  for c in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS" FETCH_SIZE WRITE_SIZE; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/${shape}_$tag -- python3 $root/exp/ab_split.py --rounds 1 --iters 1 > $out/${shape}_$tag.log 2>&1 < /dev/null
  done
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = {}
for shape in ("ref150", "s300"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (out, shape), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if "viterbi_rows" not in name: continue
            key = "finish" if "finish" in name else ("sweep_only" if "true" in name else "fused")
            tot[key][row["Counter_Name"]] += float(row["Counter_Value"]); n[key][row["Counter_Name"]] += 1
    res[shape] = {k: {c: tot[k][c] / n[k][c] for c in tot[k]} for k in tot}
    for k in res[shape]:
        c = res[shape][k]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            c["hbm_bytes_fetch_x2"] = (c["WRITE_SIZE"] + 2 * c["FETCH_SIZE"]) * 1024
        c["launches_averaged"] = dict(n[k])
json.dump(res, open(out + "/split_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
