#!/bin/bash
# rocprofv3 kernel trace of bench.py's graph-replayed steps on the GPU box: whole-process stats, the per-kernel table of the 10
# timed steps (tools/prof_window.py) and one step as a timeline (tools/prof_timeline.py) under gpurun_out/<tag>/.
#   bash tools/prof_step.sh <tag>
set -u
TAG=${1:-prof}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 5 --no-configs --no-in-step --no-cpu-baseline --no-drop-in > $OUT/bench.log 2>&1
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 $REPO/tools/prof_timeline.py $f 0 > $OUT/step_timeline.txt
python3 $REPO/tools/prof_window.py $f 10 400 > $OUT/timed_window_kernels.csv
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -2 $OUT/timed_window_kernels.csv
