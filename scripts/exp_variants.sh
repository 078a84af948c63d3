#!/bin/bash
# Run on the GPU box: bench every experimental engine build under exp/ (ADVNTR_HIP_LIB override), kernel ms per variant
for f in "" exp/*.so; do
  r=$(ADVNTR_HIP_LIB=${f:+$(pwd)/$f} timeout 300 python3 bench.py --no-cpu "$@" 2>/dev/null < /dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['value'], d.get('s300',{}).get('kernel_ms'))")
  echo "${f:-shipped} $r"
done
