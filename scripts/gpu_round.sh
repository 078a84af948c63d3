#!/bin/bash
# Run on the GPU box: the GPU test suite, the default bench line, and the kernel-trace summary of the bench command.
#   scripts/gpu_round.sh gpurun_out/<tag> [pytest args]
out=$1; shift; root=$(pwd); mkdir -p $root/$out
timeout 2400 python3 -m pytest tests -m gpu -x -q "$@" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 600 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
cut -c1-1500 $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/trace -- python3 $root/bench.py --no-cpu --no-s300 > $root/$out/trace.log 2>&1 < /dev/null
cd $root
for f in $(find $out/trace -name "*kernel_stats.csv"); do head -6 $f | cut -c1-220; done
