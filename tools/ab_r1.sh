#!/bin/bash
# same-box A/B of the round-1 tree (tools/ab/r1) against the working tree
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
for wl in ${WLS:-c4s c4}; do
  for which in r1 new; do
    if [ $which = r1 ]; then B=$R/tools/ab/r1/bench.py; else B=$R/bench.py; fi
    timeout -k 10 120 python3 $B --workload $wl --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/ab_${which}_$wl.log 2>&1 || exit 1
    python3 - $OUT/ab_${which}_$wl.log $which $wl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(sys.argv[2], sys.argv[3], 'Mrays/s %.0f kernel_ms %.2f iters %d waves %d lane_fill %.3f' % (d['value'], r['kernel_ms'], r.get('bvh',{}) and r['bvh']['iterations'] or 0, r['waves'], r['lane_fill']))
PY
  done
done
