#!/bin/bash
# same-box A/B of older trees (tools/ab/<name>, each with its own bench.py and library) against the working tree ("new")
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
WLS=$1; shift
for rep in 1 2; do
for wl in $WLS; do
  for which in "$@"; do
    if [ $which = new ]; then B=$R/bench.py; else B=$R/tools/ab/$which/bench.py; fi
    timeout -k 10 120 python3 $B --workload $wl --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/ab_${which}_$wl.log 2>&1 || { echo FAILED $which $wl; tail -3 $OUT/ab_${which}_$wl.log; exit 1; }
    python3 - $OUT/ab_${which}_$wl.log $which $wl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-8s %-4s Mrays/s %8.0f kernel_ms %9.3f waves %d lane_fill %.3f' % (sys.argv[2], sys.argv[3], d['value'], r['kernel_ms'], r['waves'], r['lane_fill']))
PY
  done
done
done
