#!/bin/bash
# Run on the GPU box: time the sum-product kernel of every experimental engine build under exp/ (ADVNTR_HIP_LIB override)
for f in "" exp/*.so; do
  r=$(ADVNTR_HIP_LIB=${f:+$(pwd)/$f} ADVNTR_EXP=1 timeout 200 python3 scripts/forward_bench.py 2>&1 < /dev/null | grep -E "forward_rows kernel|Error|error" | cut -c1-95)
  echo "${f:-shipped}"; echo "$r"
done
