#!/bin/bash
# all profiles of a round in one go: tools/prof_all.sh <tag> "<workloads>"
set -u
TAG=$1; WLS=$2
for wl in $WLS; do
  bash profiles/run_profile.sh ${TAG}_$wl $wl || { echo "profile $wl failed: $?"; exit 1; }
  python3 profiles/summarize.py gpurun_out/prof_${TAG}_$wl gpurun_out/${TAG}_${wl}_rocprof > /dev/null || exit 1
  rm -rf gpurun_out/prof_${TAG}_$wl
  echo "== $wl done"
done
