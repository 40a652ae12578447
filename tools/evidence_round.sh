#!/bin/bash
# One evidence set of a round, taken in ONE gpurun call at the final state (on the GPU box, from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/evidence_round.sh r06'
# GPU tests, the cfg2 window table / timeline / stats (tools/prof_step.sh), the PMC passes (tools/pmc_run.sh), the full default
# bench line, the window tables of cfg3 / cfg4 / cfg5 and of the inference forward -- all under gpurun_out/; copy what is to be
# judged into profiles/ afterwards (tools/pmc_collect.py for the counter passes).  The window table is copied into profiles/ before
# bench.py runs: the bench line's `roofline` names the function that table's first row names.
set -u
T=${1:-r06}
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${T}_gpu_tests.log 2>&1; tail -2 gpurun_out/${T}_gpu_tests.log
bash tools/prof_step.sh $T > gpurun_out/${T}_prof.log 2>&1; tail -2 gpurun_out/${T}_prof.log
cp gpurun_out/$T/timed_window_kernels.csv profiles/${T}_bench_timed_window_kernels.csv
bash tools/pmc_run.sh ${T}pmc > gpurun_out/${T}_pmc.log 2>&1; tail -2 gpurun_out/${T}_pmc.log
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; tail -c 300 gpurun_out/${T}_bench.json
for c in cfg3 cfg4 cfg5; do bash tools/prof_step.sh ${T}_$c --config $c --steps 5 > gpurun_out/${T}_$c.log 2>&1; tail -1 gpurun_out/${T}_$c.log; done
bash tools/prof_eval.sh ${T}_eval 5 > gpurun_out/${T}_eval.log 2>&1; tail -2 gpurun_out/${T}_eval.log
