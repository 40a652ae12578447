# kernel-trace profile of the graph-replayed step: window table + one-step timeline under gpurun_out/prof/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python bench.py --steps 10 --warmup 5 --no-configs --no-in-step > gpurun_out/prof/bench.log 2>&1
f=$(ls gpurun_out/prof/*/*kernel_trace.csv | head -1)
python tools/prof_timeline.py $f 0 > gpurun_out/prof/timeline.txt
python tools/prof_window.py $f 10 400 > gpurun_out/prof/window.txt
rm -rf gpurun_out/prof/*/
