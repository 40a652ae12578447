#!/bin/bash
# L2 traffic of the SA1 furthest-point sampling kernel (lab)
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/pmc_fps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/fps_only.py <<PY
import sys
sys.path.insert(0, "$REPO"); sys.path.insert(0, "$REPO/tools")
import torch, kernel_cases as KC
c = KC.fps(8, 40000, 2048, torch.device("cuda:0"))
for _ in range(3): c["run"]()
torch.cuda.synchronize()
PY
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/a -- python3 /tmp/fps_only.py > $OUT/a.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/b -- python3 /tmp/fps_only.py > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob
csv.field_size_limit(1<<30)
for d in ("a", "b"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: print(d, "no csv"); continue
    last = {}
    for r in csv.DictReader(open(fs[0], newline="")):
        if "fps_bucket" in r["Kernel_Name"]:
            last.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in list(last.items())[-1:]:
        print(d, v)
PY
