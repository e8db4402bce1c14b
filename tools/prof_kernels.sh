# usage: tools/prof_kernels.sh <workload> [extra bench args]: per-dispatch kernel durations (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; WL=$1; shift; OUT=$R/gpurun_out/prof_k_$WL; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $R/bench.py --workload $WL --steps 5 --warmup 2 --no-extra --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import sqlite3,sys,glob
c=sqlite3.connect(glob.glob(sys.argv[1]+'/trace/*.db')[0])
rows=list(c.execute("select name,start,end,grid_x from kernels order by start"))
t0=None
rows=[r for r in rows if 'rt::' in r[0]]
for name,s,e,g in rows[-8:]:
    short=name.split('(')[0][-40:]
    print('%-42s grid %8d  start +%9.1f us  dur %9.1f us' % (short,g,(s-(t0 or s))/1e3,(e-s)/1e3)); t0=t0 or s
PY
