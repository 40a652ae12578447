#!/bin/bash
# PMC counters of the split-bf16 streaming shared-MLP kernel (lab): clock, MFMA busy, LDS, waits, HBM traffic.
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/pmc_bf3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SPACAP_SA_BF16X3=2 BF3_SHAPES=2
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $REPO/tools/lab/bf3_variants.py run > $OUT/$name.log 2>&1
}
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM
run c FETCH_SIZE
run d WRITE_SIZE
run e TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1<<30)
for d in sorted(glob.glob("$OUT/[a-e]")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0], newline="")):
        if "bf3s" not in r["Kernel_Name"]: continue
        e = disp.setdefault(int(r["Dispatch_Id"]), {"g": int(r["Grid_Size"]), "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    seen = {}
    for k, e in disp.items():
        seen[e["g"]] = e
    for g, e in seen.items():
        print(d[-1], "grid", g, " ".join(f"{k}={v:.4g}" for k, v in e.items() if k not in ("g",)))
PY
