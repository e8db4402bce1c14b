#!/bin/bash
# same-box A/B of library builds: tools/ab_libs.sh "<workloads>" "<tags>" [steps]
# tag "main" = raytracer_amd/csrc/librt_hip.so, any other tag = librt_hip_<tag>.so (make variant TAG=<tag> DEFS=...)
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
for rep in 1 2; do
for wl in $1; do
  for tag in $2; do
    if [ $tag = main ]; then unset RT_HIP_LIBRARY; else export RT_HIP_LIBRARY=$R/raytracer_amd/csrc/librt_hip_$tag.so; fi
    timeout -k 10 150 python3 $R/bench.py --workload $wl --steps ${3:-3} --warmup 1 --no-extra --no-cpu-baseline ${EXTRA:-} > $OUT/abl_${tag}_$wl.log 2>&1 || { echo FAILED $tag $wl; tail -3 $OUT/abl_${tag}_$wl.log; exit 1; }
    python3 - $OUT/abl_${tag}_$wl.log $tag $wl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-10s %-4s Mrays/s %8.0f kernel_ms %9.3f waves %d lane_fill %.3f iters %d' % (sys.argv[2], sys.argv[3], d['value'], r['kernel_ms'], r['waves'], r['lane_fill'], r['rays_by_kind']['primary'] and (r.get('bvh') or {}).get('iterations', 0)))
PY
  done
done
done
