#!/bin/bash
# Same-box A/B of the default bench line: the round's start (worktree _base at be7dfd2) against this tree, alternating.
for i in 1 2 3; do
  for d in _base .; do
    (cd $d && timeout 600 python bench.py --no-configs --no-cpu-baseline --no-drop-in --no-in-step 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d', round(d['value'],1), round(d['ms_per_step'],3))")
  done
done
