#!/bin/bash
# A/B of SPACAP_LAB_* knobs on the pipelined cfg2 step: bash tools/lab/run_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...
export PYTHONUNBUFFERED=1
for v in "$@"; do
  echo "== $v"
  env $v MODE=step ARR=base timeout 400 python tools/lab/cumask_step.py 2>&1 | grep -E "wall|Error|error" 
done
