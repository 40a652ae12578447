#!/bin/bash
# rocprofv3 kernel trace of bench.py's graph-replayed steps on the GPU box: whole-process stats, the per-kernel table of the 10
# timed steps (tools/prof_window.py) and one step as a timeline (tools/prof_timeline.py) under gpurun_out/<tag>/.
#   bash tools/prof_step.sh <tag> [extra bench.py arguments, e.g. --config cfg5 --steps 5]
# Exits non-zero when any stage fails or writes an empty table (an empty table was committed as evidence once: round 3).
set -u -o pipefail
TAG=${1:-prof}
shift || true
STEPS=10
EXTRA=("$@")
for ((i = 0; i < ${#EXTRA[@]}; i++)); do
  if [ "${EXTRA[$i]}" == "--steps" ]; then STEPS=${EXTRA[$((i + 1))]}; fi
done
REPO=$(cd "$(dirname "$0")/.." && pwd)
case "$TAG" in */*|*..*|"") echo "prof_step.sh: bad tag '$TAG'" >&2; exit 2;; esac
OUT="$REPO/gpurun_out/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
fail() { echo "prof_step.sh: $1 FAILED" >&2; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps $STEPS --warmup 5 --no-configs --no-in-step --no-cpu-baseline --no-drop-in "${EXTRA[@]}" > $OUT/bench.log 2> $OUT/bench.err \
  || { tail -20 $OUT/bench.err >&2; fail "rocprofv3 bench.py"; }
grep -q '^{' $OUT/bench.log || fail "bench.py printed no JSON line"
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1) || fail "no kernel trace"
python3 $REPO/tools/prof_timeline.py $f 0 > $OUT/step_timeline.txt || fail "prof_timeline.py"
python3 $REPO/tools/prof_window.py $f $STEPS 400 > $OUT/timed_window_kernels.csv || fail "prof_window.py"
[ -s $OUT/timed_window_kernels.csv ] && [ -s $OUT/step_timeline.txt ] || fail "empty table"
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv || fail "no kernel stats"
rm -rf $OUT/trace
head -4 $OUT/timed_window_kernels.csv
