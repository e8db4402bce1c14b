#!/bin/bash
# extra PMC passes: profiles/run_pmc.sh <tag> <workload> "<counters>" ["extra bench args"]
set -u
TAG=$1; WL=$2; CTRS=$3; EXTRA=${4:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
CMD="python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extra --no-cpu-baseline $EXTRA"
timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $OUT/pmc_x -o p -- $CMD > $OUT/pmc_x.log 2>&1 || exit 2
