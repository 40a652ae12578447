#!/bin/bash
# Collects the rocprofv3 evidence for the hot kernels on the GPU box (one counter group per pass, see tools/pmc_kernels.py).
#   bash tools/pmc_run.sh <tag>      -> gpurun_out/<tag>/{trace,fetch,write,mfma}/..., gpurun_out/<tag>/pmc.json
set -u
TAG=${1:-pmc}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PMC_ORDER_FILE=$OUT/order.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/pmc_kernels.py > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $REPO/tools/pmc_kernels.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $REPO/tools/pmc_kernels.py > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 $REPO/tools/pmc_kernels.py > $OUT/mfma.log 2>&1
python3 $REPO/tools/pmc_parse.py $OUT $OUT/pmc.json > $OUT/pmc_summary.txt 2>&1
# keep only what is small: the per-dispatch counter CSVs of the hot kernels are a few hundred rows
find $OUT -name "*agent_info.csv" -delete
cat $OUT/pmc_summary.txt
