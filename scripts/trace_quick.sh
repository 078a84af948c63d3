#!/bin/bash
# Run on the GPU box: kernel-trace stats of one bench command -> <out_dir>/trace, summary printed
#   scripts/trace_quick.sh gpurun_out/x [bench args]
out=$1; shift; root=$(pwd); mkdir -p $root/$out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/bench.py --no-cpu "$@" > $root/$out/trace.log 2>&1 < /dev/null
cd $root
tail -1 $out/trace.log | cut -c1-400
for f in $(find $out/trace -name "*kernel_stats.csv"); do head -6 $f | cut -c1-220; done
