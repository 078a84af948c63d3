#!/bin/bash
# end_to_end record of the default bench line for several host-thread settings (ADVNTR_HOST_THREADS), one gpurun call
for t in "$@"; do
  ADVNTR_HOST_THREADS=$t python3 bench.py --no-upstream --no-cpu --steps 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())['end_to_end']
print('threads $t total', round(d['total_s'], 3), [round(x, 3) for x in d['total_s_of_each_pass']], {k: round(v, 3) for k, v in d['stage_s_overlapped'].items()})"
done
