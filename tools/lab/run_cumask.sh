#!/bin/bash
# Lab driver: CU-mask experiments (round 6).  Output under gpurun_out/cumask/.
O=gpurun_out/cumask; mkdir -p $O
export PYTHONUNBUFFERED=1
MODE=probe timeout 300 python tools/lab/cumask_step.py > $O/probe.txt 2>&1
timeout 600 python bench.py > $O/bench_base.json 2> $O/bench_base.err
for a in base; do MODE=step ARR=$a timeout 400 python tools/lab/cumask_step.py > $O/step_$a.txt 2>&1; done
MODE=step ARR=base NORESERVE=1 timeout 400 python tools/lab/cumask_step.py > $O/step_base_noreserve.txt 2>&1
for a in mainonly all8 split splitgraph; do MODE=step ARR=$a SPACAP_LAB_CUS=248 timeout 400 python tools/lab/cumask_step.py > $O/step_$a.txt 2>&1; done
tail -n 30 $O/*.txt
