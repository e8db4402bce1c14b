# usage: tools/run_bvh.sh <tag>  — BVH parity tests, then c3 / c4 / c5 bench lines
cd $GRAFT_REPO_ROOT
TAG=$1
timeout -k 10 600 python -u -m pytest tests/test_gpu_bvh.py tests/test_gpu_variants.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r2_bvh_${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/r2_bvh_${TAG}_tests.log; exit 1; }
tail -1 gpurun_out/r2_bvh_${TAG}_tests.log
timeout -k 10 200 python bench.py --workload c3 --steps 10 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r2_bvh_${TAG}_c3.log 2>&1
timeout -k 10 200 python bench.py --workload c4 --steps 2 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r2_bvh_${TAG}_c4.log 2>&1
timeout -k 10 200 python bench.py --workload c5 --steps 3 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/r2_bvh_${TAG}_c5.log 2>&1
python - $TAG <<'PY'
import json,sys
for f in ['c3','c4','c5']:
    try:
        d=json.loads(open('gpurun_out/r2_bvh_%s_%s.log'%(sys.argv[1],f)).read().strip().splitlines()[-1]); r=d['roofline']; b=r['bvh']; it=b['iterations']
        print(f, round(d['value']), 'kernel_ms',round(r['kernel_ms'],2),'waves',r['waves'],'fill',round(r['lane_fill'],3),'rounds/iter in',round(b['inner_rounds']/it,2),'lf',round(b['leaf_rounds']/it,2),'nodes/ray',round(b['node_visits_per_ray'],2),'prims/ray',round(b['prim_tests_per_ray'],2),'us/iter',round(r['kernel_ms']*1e3*r['waves']/it,2))
    except Exception as e: print(f,'ERR',e)
PY
