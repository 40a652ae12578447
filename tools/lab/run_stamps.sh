#!/bin/bash
O=gpurun_out/stamps; mkdir -p $O
export ONLY=${ONLY:-l1in} NO_MARKS=1 CALLS=1
for v in "$@"; do
  echo "== $v"
  env $v timeout 300 python tools/lab/step_stamps.py 30 > $O/v.txt 2>&1
  grep -E "ms/step| END " $O/v.txt | grep -v "per entry" | head -${HEADN:-14}
  grep -B1 " END " $O/v.txt | grep -v "END\|--" | head -${HEADN:-14} > /dev/null
  python3 - <<'PY'
import re
rows=[]
for l in open("gpurun_out/stamps/v.txt"):
    m=re.match(r"\s+(.+?)\s+at\s+([\d.]+) us",l)
    if m: rows.append((m.group(1).strip(), float(m.group(2))))
for (n,t),(n2,t2) in zip(rows,rows[1:]):
    if n2==n+" END": print("   %-40s %8.1f us"%(n,t2-t))
PY
done
