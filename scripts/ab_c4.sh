#!/bin/bash
# A/B of engine builds on the C4 launch (8 960 PacBio loci, viterbi_rows_long_kernel) inside one gpurun call
root=$(pwd)
for round in 1 2; do
  for lib in "$@"; do
    path=$root/$lib; [ "$lib" = "-" ] && path=
    ADVNTR_HIP_LIB=$path python3 bench.py --workload c4 --no-cpu --steps 3 --warmup 1 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-20s c4 kernel_ms %.3f  frac %.4f' % ('$lib', d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
