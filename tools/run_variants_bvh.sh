#!/bin/bash
# like run_variants.sh, printing the tree statistics
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; WLS=$1; shift
cp $R/raytracer_amd/csrc/librt_hip.so /tmp/librt_hip_tree.so
for v in "$@"; do
  if [ $v = tree ]; then cp /tmp/librt_hip_tree.so $R/raytracer_amd/csrc/librt_hip.so; else cp $R/tools/ab/librt_hip_$v.so $R/raytracer_amd/csrc/librt_hip.so; fi
  for wl in $WLS; do
    timeout -k 10 120 python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-extra --no-cpu-baseline > $OUT/var_${v}_$wl.log 2>&1 || { echo FAILED $v $wl; exit 1; }
    python3 - $OUT/var_${v}_$wl.log $v $wl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; b=r['bvh']
        print('%-10s %-4s Mrays/s %8.0f kernel_ms %9.3f nodes/ray %.2f prims/ray %.2f' % (sys.argv[2], sys.argv[3], d['value'], r['kernel_ms'], b['node_visits_per_ray'], b['prim_tests_per_ray']))
PY
  done
done
cp /tmp/librt_hip_tree.so $R/raytracer_amd/csrc/librt_hip.so
