#!/bin/bash
# kernel trace (with queue ids) of tools/lab/quick_config_steps.py: which hardware queue each branch of the replayed step lands on
#   bash tools/lab/run_qtrace.sh <tag> cfgA cfgB ...
TAG=$1; shift
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT="$REPO/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-6} rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/lab/quick_config_steps.py "$@" > $OUT/log.txt 2>$OUT/err.txt
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY' > $OUT/queues.txt
import csv, sys, collections, os
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# split at fps_bucket_kernel launches (one per step on the side stream)
marks = [i for i, r in enumerate(rows) if "fps_bucket_kernel" in r["Kernel_Name"]]
print("fps launches", len(marks))
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    q = collections.Counter((r["Queue_Id"], r.get("Stream_Id", "")) for r in seg)
    dur = collections.defaultdict(float)
    for r in seg:
        dur[(r["Queue_Id"], r.get("Stream_Id", ""))] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fq = (rows[a]["Queue_Id"], rows[a].get("Stream_Id", ""))
    span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    rel = [r for r in seg if "rel_fused" in r["Kernel_Name"]]
    relq = sorted({(r["Queue_Id"], r.get("Stream_Id", "")) for r in rel})
    if os.environ.get("QNAMES") and len(rel) == 2:
        for k in sorted(q):
            names = collections.Counter()
            for r in seg:
                if (r["Queue_Id"], r.get("Stream_Id", "")) == k:
                    names[r["Kernel_Name"].split("(")[0].split("<")[0][-40:]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            print("      ", k, ", ".join(f"{n}:{v:.0f}" for n, v in names.most_common(6)))
    print(f"t={(int(rows[a]['Start_Timestamp'])-t0)/1e6:9.2f} ms span {span:8.1f} us  side={fq}  rel={relq}  " + "  ".join(f"{k}:{n}k/{dur[k]:.0f}us" for k, n in sorted(q.items())))
PY
rm -rf $OUT/trace
cat $OUT/log.txt
cat $OUT/queues.txt | tail -${TAILN:-40}
