#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + separate PMC passes.
# usage: profiles/run_profile.sh <tag> [workload] ["extra bench.py args"]
set -u
TAG=${1:-r01}; WL=${2:-c2}; EXTRA=${3:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
CMD="python3 ${BENCH:-$R/bench.py} --workload $WL --steps 3 --warmup 1 --no-extra --no-cpu-baseline $EXTRA"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc1 -o p -- $CMD > $OUT/pmc1.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc2 -o p -- $CMD > $OUT/pmc2.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- $CMD > $OUT/pmc3.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- $CMD > $OUT/pmc4.log 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $OUT/pmc5 -o p -- $CMD > $OUT/pmc5.log 2>&1 || exit 6
