#!/bin/bash
# Same-box A/B of the default bench line: the round's start against this tree, alternating.  Needs a worktree of the older
# commit WITH its own built library under _base/ (not kept in the repository; it travels to the GPU box with the snapshot):
#   git worktree add _base <commit> && make -C _base/spacap3d_amd/csrc && echo _base/ >> .git/info/exclude
#   gpurun -- 'bash tools/lab/ab_base.sh';  git worktree remove --force _base
for i in 1 2 3; do
  for d in _base .; do
    (cd $d && timeout 600 python bench.py --no-configs --no-cpu-baseline --no-drop-in --no-in-step 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$d', round(d['value'],1), round(d['ms_per_step'],3))")
  done
done
