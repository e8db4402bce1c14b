"""Debug (library built with -DRT_KIND_STATS): share of loop iterations whose child rays all leave one kind of surface."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi
from raytracer_amd.scene import Scene
dev = torch.device("cuda", 0)
for kw in (dict(width=800, height=640, samplesPerPixel=16, maxTraceDepth=4), dict(width=1920, height=1080, samplesPerPixel=64, maxTraceDepth=5)):
    s = Scene(jitter=1, seed=1, **kw)
    rgba = torch.zeros((s.width, s.height, 4), dtype=torch.float32, device=dev)
    depth = torch.zeros((s.width, s.height), dtype=torch.float32, device=dev)
    stats = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
    s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = stats.cpu().tolist()
    print(kw, "iterations", st[2], "with children", st[17], "one kind only", st[16], "(%.3f)" % (st[16] / max(1, st[17])), "with dielectric children", st[19], "(%.3f)" % (st[19] / max(1, st[17])))
