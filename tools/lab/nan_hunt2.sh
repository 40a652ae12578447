#!/bin/bash
run() { echo "== $*"; timeout 300 python bench.py --config cfg2 --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-drop-in --no-in-step "$@" 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print('  final_loss', d['final_loss'], 'ms', round(d['ms_per_step'],3))"; }
run --batch 2
run --batch 2 --no-graph
run --batch 2 --no-prefetch
run --batch 2 --no-graph --no-prefetch
run --batch 4
run --batch 3
run --batch 8
