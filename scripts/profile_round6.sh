#!/bin/bash
# Run on the GPU box, in ONE call on the final tree: everything profiles/r06_* is made of.  scripts/profile_round6.sh gpurun_out/r06_prof
# (every profiler pass under its own timeout: a pass that hangs must not eat the call)
# scripts/profile_round6.sh OUT lines: only the bench lines / records at the end (no profiler passes)
out=$1; root=$(pwd); mkdir -p $root/$out
B="--no-cpu --no-s300 --no-c2 --in-flight 1"       # (the profiler passes trace one launch at a time)
if [ "$2" != "lines" ]; then
for w in c1 s300 c2 c4; do timeout 200 python3 bench.py --workload $w $B --steps 2 > $out/${w}_quick.json 2> $out/${w}_quick.err; done
cd /tmp && export TMPDIR=/tmp
declare -A passes=([c1]="--steps 20 --warmup 5" [s300]="--steps 20 --warmup 5" [c2]="--steps 10 --warmup 3" [c4]="--steps 5 --warmup 2")
for w in c1 s300 c2 c4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace_$w -- python3 $root/bench.py --workload $w $B ${passes[$w]} > $root/$out/trace_$w.log 2>&1 < /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $root/$out/pmc_${w}_$c -- python3 $root/bench.py --workload $w $B --steps 1 --warmup 0 > $root/$out/pmc_${w}_$c.log 2>&1 < /dev/null
  done
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $root/$out/pmc_${w}_sq -- python3 $root/bench.py --workload $w $B --steps 1 --warmup 0 > $root/$out/pmc_${w}_sq.log 2>&1 < /dev/null
done
# the end-to-end run (builder, recruit kernels, genotype caller) under the kernel trace: what the device does in it
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace_e2e -- python3 $root/bench.py --no-upstream --no-cpu --steps 2 > $root/$out/trace_e2e.log 2>&1 < /dev/null
fi
cd $root
# the bench lines kept under profiles/: the default line, C2 / C4 alone, both strong-scaling lines over RCCL on one rank,
# config 5 at full size, the workgroup clocks of an 8-rank share (measurement build)
timeout 600 python3 bench.py > $out/c1_bench.json 2> $out/c1_bench.err
timeout 300 python3 bench.py --workload c2 --no-cpu --steps 5 --warmup 2 > $out/c2_bench.json 2> $out/c2_bench.err
timeout 300 python3 bench.py --workload c4 --no-cpu --steps 5 --warmup 2 > $out/c4_bench.json 2> $out/c4_bench.err
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29651 timeout 300 python3 bench.py --workload c3 --no-cpu --steps 5 --warmup 1 > $out/c3_1gpu_rccl_bench.json 2> $out/c3_rccl.err
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29652 timeout 300 python3 bench.py --workload c4 --no-cpu --steps 3 --warmup 1 > $out/c4_1gpu_rccl_bench.json 2> $out/c4_rccl.err
timeout 600 python3 scripts/pacbio_full_size.py 8960 $out/c5_full_size.json > /dev/null 2> $out/c5_full_size.err
# the host side of an 8-rank job on this one box: 8 rank processes share the GPU through the host communicator; per-rank throttle
# counters of the control group around the timed region
timeout 600 python3 bench.py --workload c3 --emulate-ranks 8 --processes --no-cpu --steps 10 --warmup 3 > $out/c3_emulate_8_processes.json 2> $out/c3_emulate_8.err
(cat /proc/self/cgroup; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null; nproc) > $out/host_cgroup.txt 2>&1
# (measurement build of the SAME tree: scripts/build_variant.sh wgclocks -DADVNTR_WG_CLOCKS before the call)
if [ -f exp/wgclocks.so ]; then
  for w in "c4 1120" "c2 840" "ref150" "s300"; do timeout 200 python3 scripts/wg_clocks.py $w 2>/dev/null | tail -1; done > $out/wg_clocks.jsonl
fi
[ "$2" = "lines" ] && exit 0
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
sections = []
for w in ("c1", "s300", "c2", "c4"):
    try:
        bench = json.load(open("%s/%s_quick.json" % (out, w)))
    except Exception as e:
        print("no quick line for", w, e); continue
    kernel = bench["config"]["kernel"]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("%s/pmc_%s_*/**/*counter_collection.csv" % (out, w), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if not name.startswith("void " + kernel.split("<")[0]) or kernel not in name: continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    c = {k: tot[k] / n[k] for k in tot}
    sec = {"workload": w, "calls": bench["config"]["calls_this_rank"], "kernel": kernel, "counters_per_launch": c,
           "launches_averaged": dict(n)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB here; FETCH_SIZE counts half of the bytes of wide reads on
        # gfx950 (MI355X_MICROARCH.md, HBM section): x2
        sec["hbm_bytes_per_launch_fetch_x2"] = (c["WRITE_SIZE"] + 2 * c["FETCH_SIZE"]) * 1024
    if "SQ_INSTS_VALU" in c:
        sec["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
    sections.append(sec)
    for f in glob.glob("%s/trace_%s/**/*kernel_stats.csv" % (out, w), recursive=True):
        open("%s/%s_kernel_stats.csv" % (out, w), "w").write(open(f).read())
for f in glob.glob("%s/trace_e2e/**/*kernel_stats.csv" % out, recursive=True):
    open("%s/e2e_kernel_stats.csv" % out, "w").write(open(f).read())
json.dump({"note": "per-launch counters of the dominant kernel of `python bench.py --workload W` (rocprofv3 --pmc, separate "
                   "passes); FETCH_SIZE/WRITE_SIZE in KiB as reported, hbm_bytes = (WRITE + 2 x FETCH) x 1024",
           "sections": sections}, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(sections, indent=1)[:1800])
PY
