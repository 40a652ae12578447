#!/bin/bash
REPO=$(cd "$(dirname "$0")/../.." && pwd)
O=$REPO/gpurun_out/side_cost_spin; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/tools/lab/side_cost_spin.py > $O/run.log 2>&1
f=$(ls $O/trace/*/*kernel_trace.csv | head -1)
MARK=spin_kernel python3 $REPO/tools/lab/side_cost_diff.py $f 50 > $O/diff.txt 2>&1
rm -rf $O/trace
tail -3 $O/run.log; grep -A40 "first kernel start" $O/diff.txt
