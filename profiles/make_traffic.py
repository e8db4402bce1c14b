#!/usr/bin/env python3
"""profiles/make_traffic.py <tag>: traffic.json from the <tag>_<workload>_rocprof.json summaries in this directory
(HBM bytes per launch of the trace kernels = FETCH_SIZE KB x 2 (gfx950 correction) + WRITE_SIZE KB), stamped with the
hash of the kernel sources they were taken with; bench.py reports `traffic` only while that hash still matches.
Two-pass frames (c1) have two trace kernels per frame: their bytes are added."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1]
out = {"kernel_sources_sha256": bench.kernel_sources_sha256(), "tag": tag, "bytes_per_launch": {}, "kernels": {}}
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", tag + "_*_rocprof.json"))):
    wl = os.path.basename(path)[len(tag) + 1:-len("_rocprof.json")]
    d = json.load(open(path))
    total, names = 0.0, []
    for k, c in d["counters"].items():
        # (bench.py renders one more frame with every counter on — rt_scene_counters — after its timed frames: that
        # launch is the instantiation whose last template argument is `true`, and is not what `traffic` describes)
        if "trace_kernel" in k and k.split("(")[0].rstrip("> ").endswith(", false") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            total += (c["FETCH_SIZE"] * 2 + c["WRITE_SIZE"]) * 1024
            names.append(k)
    if names:
        out["bytes_per_launch"][wl] = int(total)
        out["kernels"][wl] = names
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
