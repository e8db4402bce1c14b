export RT_HIP_LIBRARY=$GRAFT_REPO_ROOT/raytracer_amd/csrc/librt_hip_phase.so
for wl in c4 c3 c5; do timeout -k 10 150 python3 bench.py --workload $wl --steps 1 --warmup 1 --no-extra --no-cpu-baseline > gpurun_out/s3_phase_$wl.log 2>/dev/null; python3 - gpurun_out/s3_phase_$wl.log $wl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; b=r['bvh']; p=b['phase_clk_gen_scan_trav_shade']; t=sum(p)
        print(sys.argv[2], 'kernel_ms %.1f'%r['kernel_ms'], 'gen/scan/trav/shade', ['%.3f'%(x/t) for x in p], 'iters', b['iterations'], 'clk/iter/wave %.0f'%(t/b['iterations']))
PY
done
