#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
for wl in $1; do for ma in $2; do
  timeout -k 10 120 python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-extra --no-cpu-baseline --bvh-min-active $ma > $OUT/sw_${wl}_$ma.log 2>&1 || exit 1
  python3 - $OUT/sw_${wl}_$ma.log $wl $ma <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-4s min_active %-3s Mrays/s %8.0f kernel_ms %9.3f lane_fill %.3f lanes/inner %.1f' % (sys.argv[2], sys.argv[3], d['value'], r['kernel_ms'], r['lane_fill'], r['bvh']['lanes_per_inner_round']))
PY
done; done
