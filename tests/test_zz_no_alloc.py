"""rt_render_device_noalloc (include/rt_abi.h): the entry point for callers that cannot allocate.  It must give what
rt_render_device gives, bit for bit, on two calls in a row (the second with a smaller input: the static per-GPU
scratch keeps the first call's contents), must not change the amount of free device memory, and must reject
frames beyond its stated maximum with a clear error."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, host, scenes
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu


def _device_buffers(torch, ncols, height):
    dev = torch.device("cuda", 0)
    return dict(rgba=torch.full((ncols, height, 4), -7.0, dtype=torch.float32, device=dev),
                depth=torch.full((ncols, height), -7.0, dtype=torch.float32, device=dev),
                ids=torch.full((ncols, height), -7, dtype=torch.int64, device=dev),
                stats=torch.full((abi.RT_STATS_WORDS,), -7, dtype=torch.int64, device=dev))


def _render(torch, scene, noalloc, cols=(0, None, 1)):
    xb, xe, xs = cols
    p = scene.params(xb, xe, xs)
    b = _device_buffers(torch, host.num_columns(p), scene.height)
    fn = scene.renderColumnsDeviceNoAlloc if noalloc else scene.renderColumnsDevice
    fn(b["rgba"].data_ptr(), b["depth"].data_ptr(), b["ids"].data_ptr(), b["stats"].data_ptr(),
       torch.cuda.current_stream().cuda_stream, xb, xe, xs)
    torch.cuda.synchronize()
    scene.status()
    return {k: v.cpu().numpy() for k, v in b.items()}


def _same(a, b):
    assert np.array_equal(a["rgba"], b["rgba"]) and np.array_equal(a["depth"], b["depth"]) and np.array_equal(a["ids"], b["ids"])
    # ray count, max depth, child rays by kind (the loop-iteration, clock and wave counts depend on the number of waves)
    for w in (0, 1, 3, 4, 5):
        assert a["stats"][w] == b["stats"][w], w
    assert a["stats"][0] > 0


def _free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_matches_the_allocating_entry_point_twice_in_a_row():
    import torch
    big = Scene(width=320, height=240, samplesPerPixel=2, maxTraceDepth=5, jitter=1, seed=3)
    small = Scene(width=96, height=64, samplesPerPixel=3, maxTraceDepth=3, jitter=1, seed=4, backgroundColor=(0.1, 0.2, 0.3))
    want_big, want_small = _render(torch, big, False), _render(torch, small, False)
    got_big = _render(torch, big, True)          # first call: the larger frame
    got_small = _render(torch, small, True)      # second call: smaller, over the first call's leftovers
    _same(got_big, want_big)
    _same(got_small, want_small)
    # a column partition through the same entry point (what a multi-GPU caller does)
    _same(_render(torch, big, True, (1, big.width, 3)), _render(torch, big, False, (1, big.width, 3)))


def test_noalloc_bvh_scene_and_no_device_memory_is_taken():
    import torch
    m = np.concatenate([np.frombuffer(bytes(host.flatten_surfaces(host.SceneDefinitions.scene_spheres)), dtype=abi.surface_dtype, count=12).copy(),
                        scenes.heightfield_mesh(24, environment=False, base=-5.9, amplitude=1.0)])
    big = Scene(surfaces=m, width=200, height=150, samplesPerPixel=2, maxTraceDepth=4, jitter=1, seed=21)
    small = Scene(surfaces=m, width=64, height=48, samplesPerPixel=1, maxTraceDepth=6, jitter=0, seed=2)
    want_big, want_small = _render(torch, big, False), _render(torch, small, False)
    _render(torch, big, True)                    # warm: code objects, occupancy query
    bufs = _device_buffers(torch, big.width, big.height)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    before = _free_bytes()
    big.renderColumnsDeviceNoAlloc(bufs["rgba"].data_ptr(), bufs["depth"].data_ptr(), bufs["ids"].data_ptr(), bufs["stats"].data_ptr(), stream)
    torch.cuda.synchronize()
    assert _free_bytes() == before, "the allocation-free entry point changed the free device memory"
    big.status()
    _same({k: v.cpu().numpy() for k, v in bufs.items()}, want_big)
    _same(_render(torch, small, True), want_small)


def test_noalloc_rejects_frames_beyond_its_maximum():
    import torch
    s = Scene(width=1280, height=1024, samplesPerPixel=1, maxTraceDepth=1, jitter=0)      # 1 310 720 > 2^20 pixels
    b = _device_buffers(torch, 8, 8)
    with pytest.raises(abi.RtError) as e:
        s.renderColumnsDeviceNoAlloc(b["rgba"].data_ptr(), b["depth"].data_ptr(), None, None, 0)
    assert e.value.code == abi.RT_E_UNSUPPORTED and "at most" in str(e.value)
    # the same frame in two column halves fits
    half = _render(torch, s, True, (0, 640, 1))
    assert np.isfinite(half["rgba"]).all() and half["stats"][0] >= 640 * 1024
