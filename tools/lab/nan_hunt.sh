#!/bin/bash
run() { echo "== $*"; env "$@" timeout 300 python bench.py --config ${CFG:-cfg3} --batch 2 --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-drop-in --no-in-step 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print('  final_loss', d['final_loss'], 'ms', round(d['ms_per_step'],3))"; }
run A=1
run SPACAP_FORK_RELATION=0 SPACAP_FLUSH_MID=0
run SPACAP_FLUSH_MID=0
run SPACAP_FORK_RELATION=0
run SPACAP_SA_F32MFMA=1
run SPACAP_FPS_LEGACY=1
run SPACAP_WIDE_WGRAD_TR=0
