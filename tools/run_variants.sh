#!/bin/bash
# A/B of library builds on one box: tools/run_variants.sh "<workloads>" <variant> ...   (variants = tools/ab/librt_hip_<v>.so,
# copied over the library of the box's scratch copy of the tree; nothing outside that copy is touched)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
WLS=$1; shift
cp $R/raytracer_amd/csrc/librt_hip.so /tmp/librt_hip_tree.so
for rep in 1 2; do
for v in "$@"; do
  if [ $v = tree ]; then cp /tmp/librt_hip_tree.so $R/raytracer_amd/csrc/librt_hip.so; else cp $R/tools/ab/librt_hip_$v.so $R/raytracer_amd/csrc/librt_hip.so; fi
  for wl in $WLS; do
    timeout -k 10 120 python3 $R/bench.py --workload $wl --steps ${STEPS:-3} --warmup 1 --no-extra --no-cpu-baseline ${EXTRA:-} > $OUT/var_${v}_$wl.log 2>&1 || { echo "FAILED $v $wl"; tail -3 $OUT/var_${v}_$wl.log; exit 1; }
    python3 - $OUT/var_${v}_$wl.log $v $wl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-10s %-4s Mrays/s %8.0f kernel_ms %9.3f waves %d lane_fill %.3f' % (sys.argv[2], sys.argv[3], d['value'], r['kernel_ms'], r['waves'], r['lane_fill']))
PY
  done
done
done
cp /tmp/librt_hip_tree.so $R/raytracer_amd/csrc/librt_hip.so
