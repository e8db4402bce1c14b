cd $GRAFT_REPO_ROOT
for wl in c1 c2 c4s; do
timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/r2_split_$wl.log 2>&1
done
python - <<'PY'
import json
for f in ['c1','c2','c4s']:
    d=json.loads(open('gpurun_out/r2_split_%s.log'%f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f, round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'kernel_ms',round(r['kernel_ms'],3),'waves',r['waves'],'mean',round(r['mean_wave_clk']),'max',r['max_wave_clk'],'fill',round(r['lane_fill'],3))
PY
