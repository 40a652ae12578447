#!/bin/bash
O=gpurun_out/stamps; mkdir -p $O
export ONLY=${ONLY:-relation_fused} CALLS=1
unset NO_MARKS
for v in "$@"; do
  echo "== $v"
  env $v timeout 300 python tools/lab/step_stamps.py 30 > $O/v.txt 2>&1
  grep -E "ms/step" $O/v.txt
  sed -n '/f:encoder/,/b:proposal/p' $O/v.txt | head -${HEADN:-60}
done
