#!/bin/bash
O=gpurun_out/stamps; mkdir -p $O
export ONLY=relation_fused,sa_mid_fwd_pool,adam NO_MARKS=1 CALLS=1
for v in "base" "GEOM=0" "SPACAP_PREFETCH_GRAPH=0" "SKEW=300" "NOSIDE=1"; do
  echo "== $v"
  if [ "$v" == "base" ]; then timeout 300 python tools/lab/step_stamps.py 30 > $O/v.txt 2>&1; else env $v timeout 300 python tools/lab/step_stamps.py 30 > $O/v.txt 2>&1; fi
  grep -E "ms/step|relation_fused|sa_mid_fwd_pool_f32 END|adam_flat_f32 END" $O/v.txt | grep -v "per entry" | head -14
done
