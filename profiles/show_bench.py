import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split('/')[-1], "Mrays/s=%.0f ms=%.2f kern_ms=%.2f fill=%.3f iters=%s prim_iters=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["lane_fill"], r["iters"], r["prim_iters"]), "waves=%s max_wave_iters=%s max_clk=%s mean_clk=%.0f" % (r.get("waves"), r.get("max_wave_iters"), r.get("max_wave_clk"), r.get("mean_wave_clk", 0)))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
