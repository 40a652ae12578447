#!/bin/bash
# rocprofv3 kernel trace of the inference forward (greedy decoding) on the GPU box -> gpurun_out/<tag>/eval_window_kernels.csv
#   bash tools/prof_eval.sh <tag> [forwards]
set -u -o pipefail
TAG=${1:-eval}
N=${2:-5}
REPO=$(cd "$(dirname "$0")/.." && pwd)
case "$TAG" in */*|*..*|"") echo "prof_eval.sh: bad tag '$TAG'" >&2; exit 2;; esac
OUT="$REPO/gpurun_out/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
fail() { echo "prof_eval.sh: $1 FAILED" >&2; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/eval_profile_run.py $N > $OUT/eval.log 2> $OUT/eval.err \
  || { tail -20 $OUT/eval.err >&2; fail "rocprofv3 eval_profile_run.py"; }
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1) || fail "no kernel trace"
(cd $REPO/tools && python3 prof_between.py $f $N) > $OUT/eval_window_kernels.csv || fail "prof_between.py"
[ -s $OUT/eval_window_kernels.csv ] || fail "empty table"
rm -rf $OUT/trace
cat $OUT/eval.log
head -14 $OUT/eval_window_kernels.csv
