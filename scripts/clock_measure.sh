#!/bin/bash
# Run on the GPU box: the shader clock under fp64 load -- (1) the micro-benchmark's own cycle and real-time counters,
# (2) GRBM_GUI_ACTIVE over the dispatch duration of the bench kernel (MI355X_MICROARCH.md, DVFS: effective clock =
# GRBM_GUI_ACTIVE / kernel wall time).   scripts/clock_measure.sh gpurun_out/<tag>
out=$1; root=$(pwd); mkdir -p $root/$out
for w in 1 3 8; do ./scripts/ubench/clock_probe $w 40000; done > $out/clock_probe.txt 2>&1
cat $out/clock_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $root/$out/pmc_clock -- python3 $root/bench.py --no-cpu --no-s300 --steps 10 --warmup 3 > $root/$out/pmc_clock.log 2>&1 < /dev/null
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
trace = {}
for f in glob.glob(out + "/pmc_clock/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        trace[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["Start_Timestamp"]), int(row["End_Timestamp"]))
rows = []
for f in glob.glob(out + "/pmc_clock/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        t = trace.get(row["Dispatch_Id"])
        if t is None:
            s, e = row.get("Start_Timestamp"), row.get("End_Timestamp")
            if not s: continue
            t = (row["Kernel_Name"], int(s), int(e))
        rows.append((t[0], float(row["Counter_Value"]), t[2] - t[1]))
by = collections.defaultdict(list)
for name, cyc, ns in rows:
    if ns > 0: by[name.split("(")[0][:60]].append((cyc, ns))
res = {}
for k, v in by.items():
    v = v[len(v) // 3:]            # the later launches: the device has warmed up
    res[k] = {"launches": len(v), "mean_ms": sum(ns for _, ns in v) / len(v) / 1e6,
              "grbm_gui_active_per_launch": sum(c for c, _ in v) / len(v),
              "effective_clock_mhz": sum(c for c, _ in v) / sum(ns for _, ns in v) * 1e3}
json.dump(res, open(out + "/clock_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
