"""Debug (library built with -DRT_WAVE_TIMES=<pass>: 0 = one-pass frames, 1 / 2 = the passes of a small frame): when the
waves of that pass start and end, on the device's 100 MHz clock (stats words 12..17, 19 of the debug build).
usage: python tools/wave_times.py [workload of bench.py, default c1]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracer_amd.scene import Scene
from raytracer_amd import abi, host

wl = sys.argv[1] if len(sys.argv) > 1 else "c1"
cfg = dict(bench.WORKLOADS[wl])
surfaces, extra_kw = bench.workload_scene(wl)
cfg.update(extra_kw)
s = Scene(surfaces=surfaces, **cfg)
dev = torch.device("cuda:0")
npix = s.width * s.height
rgba = torch.zeros(npix * 4, dtype=torch.float32, device=dev)
depth = torch.zeros(npix, dtype=torch.float32, device=dev)
stats = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
for it in range(3):
    s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    w = [int(x) & (2**64 - 1) for x in stats.cpu().tolist()]
    n = w[19]
    inv = lambda x: (~x) & (2**64 - 1)
    s0, s1, e0, e1 = inv(w[14]), w[15], inv(w[16]), w[17]
    ms, me = w[12] / n, w[13] / n
    us = lambda t: (t - s0) / 100.0
    print("%s waves %d  start: first 0, mean %.1f, last %.1f us | end: first %.1f, mean %.1f, last %.1f us" %
          (wl, n, us(ms), us(s1), us(e0), us(me), us(e1)))
