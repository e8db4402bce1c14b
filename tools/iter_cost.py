"""Time per 64-ray loop iteration of the linear-scan kernel (kernel ms / iterations), for A/B builds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi
from raytracer_amd.scene import Scene
dev = torch.device("cuda", 0)
for kw in (dict(width=800, height=640, samplesPerPixel=16, maxTraceDepth=4), dict(width=3840, height=2160, samplesPerPixel=16, maxTraceDepth=5)):
    s = Scene(jitter=1, seed=1, **kw)
    rgba = torch.zeros((s.width, s.height, 4), dtype=torch.float32, device=dev)
    depth = torch.zeros((s.width, s.height), dtype=torch.float32, device=dev)
    stats = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        e0.record()
        s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
    st = stats.cpu().tolist()
    ms = e0.elapsed_time(e1)
    print(kw, "ms %.3f rays %d iterations %d -> %.3f ns per iteration (chip-wide), %.1f Grays/s, lane fill %.3f" % (ms, st[0], st[2], ms * 1e6 / st[2], st[0] / ms / 1e6, st[0] / (64.0 * st[2])))
