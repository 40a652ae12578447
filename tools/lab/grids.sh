#!/bin/bash
# Lab: grid / workgroup sizes of every kernel of one replayed step (rocprofv3 kernel trace of bench.py)
REPO=$(cd "$(dirname "$0")/../.." && pwd)
O=$REPO/gpurun_out/grids; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $REPO/bench.py --steps 4 --warmup 3 --no-configs --no-in-step --no-cpu-baseline --no-drop-in > $O/bench.log 2>&1
f=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 - "$f" > $O/grids.txt <<'PY'
import csv, sys, re
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
nk = "Kernel_Name" if "Kernel_Name" in rows[0] else "Name"
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(ev) if "adam_flat_kernel" in r[nk]]
step = ev[adam[-2] + 1: adam[-1] + 1]
t0 = int(step[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.split(r"\(", n, 1)[0][:60]
print("# start_us dur_us queue grid wg lds vgpr agpr kernel")
for r in step:
    g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0))); w = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)))
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} {r.get('Queue_Id','?'):>3} {g//max(w,1):6d} {w:5d} {r.get('LDS_Block_Size','?'):>7} {r.get('VGPR_Count','?'):>4} {r.get('Accum_VGPR_Count','?'):>4} {short(r[nk])}")
PY
rm -rf $O/trace
wc -l $O/grids.txt
