"""Which library renders the 1M-sphere frame like the brute-force oracle?  usage: c5_check.py render <out.npz> | compare a.npz b.npz"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bench
from raytracer_amd.scene import Scene

def scene():
    cfg = dict(bench.WORKLOADS["c5"]); surf, extra = bench.workload_scene("c5"); cfg.update(extra)
    return Scene(surfaces=surf, **cfg)

if sys.argv[1] == "render":
    s = scene(); s.renderScene()
    percol = np.array([s.renderColumns(x, x + 1, 1)[4] for x in range(s.width)], dtype=np.int64)
    np.savez(sys.argv[2], rgba=s.frameBuffer, ids=s.objectIds, rays=s.rayCount, percol=percol)
    print("rays", s.rayCount, int(percol.sum()))
else:
    import oracle_lib as O
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    diff = np.any(a["rgba"] != b["rgba"], axis=2)
    cols = np.nonzero(a["percol"] != b["percol"])[0]
    print("rays", int(a["rays"]), int(b["rays"]), "differing pixels", int(diff.sum()), "columns with different ray counts", len(cols), cols[:10])
    s = scene()
    import threading
    def beat():
        while True:
            time.sleep(45); print('  ... oracle running', flush=True)
    threading.Thread(target=beat, daemon=True).start()
    for x in cols[:1]:
        t0 = time.time()
        ref = O.render(s.surfaces, s.camera, s.params(int(x), int(x) + 1, 1), nthreads=16)
        ea = np.max(np.abs(a["rgba"][x] - ref["rgba"][0])); eb = np.max(np.abs(b["rgba"][x] - ref["rgba"][0]))
        print("column %d: oracle %.0f s, %d rays; A %d rays, B %d rays; max |A - oracle| = %.3g, max |B - oracle| = %.3g"
              % (x, time.time() - t0, ref["rays"], a["percol"][x], b["percol"][x], ea, eb), flush=True)
