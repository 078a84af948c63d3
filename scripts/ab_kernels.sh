#!/bin/bash
# A/B of engine builds inside ONE gpurun call (boxes of the pool differ by several percent): kernel ms of the S300 and C1 bench
# launches for each library given (paths relative to the repo root; "-" = the shipped build), three rounds, interleaved.
#   scripts/ab_kernels.sh exp/a.so exp/b.so -
root=$(pwd)
for round in 1 2 3; do
  for lib in "$@"; do
    path=$root/$lib; [ "$lib" = "-" ] && path=
    for wl in s300 c1; do
      ADVNTR_HIP_LIB=$path python3 bench.py --workload $wl --no-s300 --no-cpu --steps 20 --warmup 3 2>/dev/null | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s %-5s kernel_ms %.4f  frac %.4f' % ('$lib', '$wl', d['roofline']['kernel_ms'], d['roofline']['frac']))"
    done
  done
done
