"""Debug (library built with -DRT_WAVE_TIMES=<pass>): when the waves of one pass of a small frame start and end, on the
device's 100 MHz clock (stats words 12..17, 19 of the debug build).  usage: python tools/wave_times.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd.scene import Scene
from raytracer_amd import abi

tp = int(sys.argv[1]) if len(sys.argv) > 1 else 0
s = Scene(width=640, height=480, samplesPerPixel=1, maxTraceDepth=5, jitter=0, two_pass=tp)
dev = torch.device("cuda:0")
rgba = torch.zeros(640 * 480 * 4, dtype=torch.float32, device=dev)
depth = torch.zeros(640 * 480, dtype=torch.float32, device=dev)
stats = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
for it in range(4):
    s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    w = [int(x) & (2**64 - 1) for x in stats.cpu().tolist()]
    n = w[19]
    inv = lambda x: (~x) & (2**64 - 1)
    s0, s1, e0, e1 = inv(w[14]), w[15], inv(w[16]), w[17]
    ms, me = w[12] / n, w[13] / n
    us = lambda t: (t - s0) / 100.0
    print("waves %d  start: first 0, mean %.1f, last %.1f us | end: first %.1f, mean %.1f, last %.1f us" %
          (n, us(ms), us(s1), us(e0), us(me), us(e1)))
