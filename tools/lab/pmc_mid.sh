#!/bin/bash
# PMC counters of sa_mid_fwd variants (lab): clock, MFMA busy, LDS conflicts / activity, wait breakdown.
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/pmc_mid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lab in 0 5; do
  for m16 in 0 1; do
    export SPACAP_SA_LAB=$lab
    if [ $m16 = 1 ]; then export SPACAP_SA_MFMA32=1; else unset SPACAP_SA_MFMA32; fi
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
      --output-format csv -d $OUT/l${lab}_m${m16} -- python3 $REPO/tools/lab/mid_variants.py run > $OUT/l${lab}_m${m16}.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1<<30)
for d in sorted(glob.glob("$OUT/l*_m*")):
    if not d[-1].isdigit(): continue
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0], newline="")):
        if "sa_mid_fwd" not in r["Kernel_Name"]: continue
        e = disp.setdefault(int(r["Dispatch_Id"]), {"g": int(r["Grid_Size"]), "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    seen = {}
    for k, e in disp.items():
        seen[e["g"]] = e       # last dispatch of each grid size (= each shape)
    for g, e in seen.items():
        us = e["t"] / 1e3
        clk = e["GRBM_GUI_ACTIVE"] / 8 / us * 1e-3
        print(f"{d[-5:]} grid {g:7d}: {us:7.1f} us  clock {clk:.2f} GHz  mfma_util {e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}  "
              f"lds_active/wave_cyc {e['SQ_LDS_IDX_ACTIVE'] / max(e['SQ_WAVE_CYCLES'], 1):.3f} lds_conflict/active {e['SQ_LDS_BANK_CONFLICT'] / max(e['SQ_LDS_IDX_ACTIVE'], 1):.3f}  "
              f"wait_inst {e['SQ_WAIT_INST_ANY'] / e['SQ_WAVE_CYCLES']:.3f} wait_any {e['SQ_WAIT_ANY'] / e['SQ_WAVE_CYCLES']:.3f} active {e['SQ_ACTIVE_INST_ANY'] / e['SQ_WAVE_CYCLES']:.3f}")
PY
