#!/bin/bash
REPO=$(cd "$(dirname "$0")/../.." && pwd)
O=$REPO/gpurun_out/side_cost; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/tools/lab/side_cost.py > $O/run.log 2>&1
f=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 $REPO/tools/lab/side_cost_diff.py $f 70 > $O/diff.txt 2>&1
rm -rf $O/trace
cat $O/run.log | tail -3; head -120 $O/diff.txt
