"""Where the time of one c1 frame through rt_render goes: rt_render with registered (pinned) and
unregistered host buffers, rt_render_device + wait, and bare device-to-host copies of the same sizes."""
import ctypes as C
import sys
import time
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raytracer_amd import abi
from raytracer_amd.scene import Scene

W, H = 640, 480
N = 50


def t_loop(f, n=N):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


s = Scene(width=W, height=H, samplesPerPixel=1, maxTraceDepth=5, jitter=0)
s.upload()
print("frameBuffer base:", type(s.frameBuffer.base).__name__, type(getattr(s.frameBuffer.base, "base", None)).__name__)
print("rt_render, registered buffers, with ids   : %.3f ms" % t_loop(s.renderScene))
lib = s._lib
p = s.params()
md, rays = C.c_float(0), C.c_uint64(0)
def no_ids():
    abi.check(lib, lib.rt_render(s._scene, C.byref(s.camera), C.byref(p), s.frameBuffer.ctypes.data, s.depthBuffer.ctypes.data, None, C.byref(md), C.byref(rays)))
print("rt_render, registered buffers, no ids     : %.3f ms" % t_loop(no_ids))
fb, db, ids = np.zeros_like(s.frameBuffer), np.zeros_like(s.depthBuffer), np.zeros_like(s.objectIds)
def unreg():
    abi.check(lib, lib.rt_render(s._scene, C.byref(s.camera), C.byref(p), fb.ctypes.data, db.ctypes.data, ids.ctypes.data, C.byref(md), C.byref(rays)))
print("rt_render, pageable buffers, with ids     : %.3f ms" % t_loop(unreg))
def unreg_noids():
    abi.check(lib, lib.rt_render(s._scene, C.byref(s.camera), C.byref(p), fb.ctypes.data, db.ctypes.data, None, C.byref(md), C.byref(rays)))
print("rt_render, pageable buffers, no ids       : %.3f ms" % t_loop(unreg_noids))

dev = torch.device("cuda", 0)
rgba = torch.zeros((W, H, 4), dtype=torch.float32, device=dev)
depth = torch.zeros((W, H), dtype=torch.float32, device=dev)
stats = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def devr():
    s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, stats.data_ptr(), st)
    torch.cuda.synchronize()
print("rt_render_device + synchronize            : %.3f ms" % t_loop(devr))
def devr_nostats():
    s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, None, st)
    torch.cuda.synchronize()
print("rt_render_device (no stats) + synchronize : %.3f ms" % t_loop(devr_nostats))
hp = torch.zeros((W, H, 4), dtype=torch.float32).pin_memory()
hd = torch.zeros((W, H), dtype=torch.float32).pin_memory()
def copies():
    hp.copy_(rgba, non_blocking=True); hd.copy_(depth, non_blocking=True)
    torch.cuda.synchronize()
print("two pinned D2H copies (4.9 + 1.2 MB)      : %.3f ms" % t_loop(copies))
hpp = torch.zeros((W, H, 4), dtype=torch.float32)
def copies_pageable():
    hpp.copy_(rgba)
print("one pageable D2H copy (4.9 MB)            : %.3f ms" % t_loop(copies_pageable))
def sync_only():
    torch.cuda.synchronize()
print("empty synchronize                         : %.3f ms" % t_loop(sync_only))
