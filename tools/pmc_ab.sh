#!/bin/bash
# VALU counters of two library builds on one workload, same box: tools/pmc_ab.sh <workload> "<tags>" ["extra bench args"]
set -u
R=$GRAFT_REPO_ROOT; WL=$1
cd /tmp && export TMPDIR=/tmp
for tag in $2; do
  if [ $tag = main ]; then unset RT_HIP_LIBRARY; else export RT_HIP_LIBRARY=$R/raytracer_amd/csrc/librt_hip_$tag.so; fi
  OUT=$R/gpurun_out/prof_ab_${tag}_$WL; rm -rf $OUT; mkdir -p $OUT
  CMD="python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-extra --no-cpu-baseline ${3:-}"
  timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1 || exit 2
  echo "=== $tag $WL"; python3 $R/profiles/summarize.py $OUT $OUT/summary | grep -v "^$"
done
