#!/bin/bash
# tools/sweep_flag.sh "<workloads>" <bench flag> "<values>"
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
for wl in $1; do for v in $3; do
  timeout -k 10 120 python3 $R/bench.py --workload $wl --steps ${STEPS:-2} --warmup 1 --no-extra --no-cpu-baseline $2 $v > $OUT/sw_${wl}_$v.log 2>&1 || exit 1
  python3 - $OUT/sw_${wl}_$v.log $wl "$2 $v" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-4s %-22s Mrays/s %8.0f kernel_ms %9.3f waves %d lane_fill %.3f' % (sys.argv[2], sys.argv[3], d['value'], r['kernel_ms'], r['waves'], r['lane_fill']))
PY
done; done
