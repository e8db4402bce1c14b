"""Small frames: two passes against one pass (rt_params.two_pass), kernel ms per frame, for the sphere scene and the mesh."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, scenes
from raytracer_amd.scene import Scene
dev = torch.device("cuda", 0)
mesh = scenes.heightfield_mesh(224)
for name, surf, kw in (("c1 640x480x1 spheres", None, dict(width=640, height=480, samplesPerPixel=1, maxTraceDepth=5, jitter=0)),
                       ("c3_1 800x640x1 mesh", mesh, dict(width=800, height=640, samplesPerPixel=1, maxTraceDepth=5, jitter=1)),
                       ("400x320x4 mesh", mesh, dict(width=400, height=320, samplesPerPixel=4, maxTraceDepth=5, jitter=1)),
                       ("320x240x1 spheres", None, dict(width=320, height=240, samplesPerPixel=1, maxTraceDepth=5, jitter=0))):
    for tp in (0, 1):
        s = Scene(surfaces=surf, two_pass=tp, seed=1, **kw)
        rgba = torch.zeros((s.width, s.height, 4), dtype=torch.float32, device=dev)
        depth = torch.zeros((s.width, s.height), dtype=torch.float32, device=dev)
        stats = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, stats.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, stats.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        rays = int(stats[0].item())
        print("%-24s %s: %.3f ms per frame, %.2f Grays/s" % (name, "one pass " if tp else "two passes", ms, rays / ms / 1e6))
        s.close()
