import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split('/')[-1], "Mrays/s=%.0f ms=%.2f kern_ms=%.2f fill=%.3f frac=%.4f flops/ray=%.0f waves=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["lane_fill"], r["frac"], r["flops_per_ray"], r["waves"]))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
