#!/bin/bash
# The 8-rank rehearsal of the C3 set (bench.py --workload c3 --emulate-ranks 8) for several engine builds inside one gpurun call:
# whole-set time, slowest share, projected efficiency.   scripts/ab_rehearsal.sh exp/a.so exp/b.so -     ("-" = the shipped build)
root=$(pwd)
for lib in "$@"; do
  path=$root/$lib; [ "$lib" = "-" ] && path=
  ADVNTR_HIP_LIB=$path python3 bench.py --workload c3 --emulate-ranks 8 --no-cpu --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['scale_rehearsal']
print('%-24s whole %.2f ms  shares %s  eff %.4f  kernels-only %.4f' % ('$lib', r['whole_set']['loop_ms'], [round(x['loop_ms'], 2) for x in r['shares']], r['projected_efficiency'], r['projected_efficiency_kernels_only']))"
done
