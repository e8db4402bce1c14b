cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline $2 > gpurun_out/r2_exp_$1.log 2>&1 || echo FAILED $1; }
rm -f gpurun_out/r2_exp_*
run c5_r8 "--workload c5 --steps 3 --warmup 1 --bvh-ring 8"
run c5_r16 "--workload c5 --steps 3 --warmup 1 --bvh-ring 16"
run c3_r16 "--workload c3 --steps 10 --warmup 2 --bvh-ring 16"
run c3_r8 "--workload c3 --steps 10 --warmup 2 --bvh-ring 8"
for m in 20 24 32 36; do run c3_m$m "--workload c3 --steps 10 --warmup 2 --bvh-min-active $m"; done
for m in 20 36; do run c5_r8_m$m "--workload c5 --steps 3 --warmup 1 --bvh-ring 8 --bvh-min-active $m"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2_exp_*.log')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('r2_exp_')[1][:-4].ljust(14), round(d['value']), 'kernel_ms',round(r['kernel_ms'],3),'waves',r['waves'],'fill',round(r['lane_fill'],3))
    except Exception as e: print(f,'ERR',e)
PY
