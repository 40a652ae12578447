#!/bin/bash
# Lab: rocprofv3 counter passes over one lab script.   bash tools/lab/pmc_passes.sh <tag> <script.py> <kernel substring>
set -u
TAG=$1; SCRIPT=$2; KSUB=$3
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $REPO/$SCRIPT > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$KSUB" <<'PY'
import csv,sys,collections
csv.field_size_limit(1<<30)
rows=[r for r in csv.DictReader(open(sys.argv[1],newline='')) if sys.argv[2] in r["Kernel_Name"]]
agg=collections.OrderedDict()
for r in rows:
    k=(r["Kernel_Name"][:90], r["Counter_Name"])
    agg.setdefault(k,[]).append(float(r["Counter_Value"]))
for (k,c),v in agg.items():
    print(f"{k[-60:]:60s} {c:34s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_INST_CYCLES_VMEM
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
PASSES
