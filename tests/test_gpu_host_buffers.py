"""The frame an F# host gets (rt_render, host buffers; Scene.fs:117-121 fills frameBuffer / depthBuffer, :28-29):
pinned buffers from rt_host_alloc and ordinary host arrays receive the same frame bit for bit; without an object-ID buffer
nothing is written for the IDs; depth and IDs (copied on a second stream while a small frame's second pass still runs)
belong to the same frame as the colours, also when frames of different scenes alternate on one handle's buffers."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu


def _render_into(s, fb, db, ids):
    p, md, rays = s.params(), C.c_float(0.0), C.c_uint64(0)
    abi.check(s._lib, s._lib.rt_render(s._scene, C.byref(s.camera), C.byref(p), fb.ctypes.data, db.ctypes.data,
                                       ids.ctypes.data if ids is not None else None, C.byref(md), C.byref(rays)))
    return md.value, rays.value


@pytest.mark.parametrize("kw", [dict(width=640, height=480, samplesPerPixel=1, maxTraceDepth=5, jitter=0),       # two passes
                                dict(width=201, height=97, samplesPerPixel=3, maxTraceDepth=3, jitter=1, two_pass=1)])
def test_pinned_and_pageable_buffers_receive_the_same_frame(kw):
    s = Scene(**kw)
    s.renderScene()                                         # into the Scene's own pinned buffers
    assert type(s.frameBuffer.base.base).__name__ == "PinnedBlock"
    fb, db, ids = np.zeros_like(np.asarray(s.frameBuffer)), np.zeros_like(np.asarray(s.depthBuffer)), np.zeros_like(np.asarray(s.objectIds))
    md, rays = _render_into(s, fb, db, ids)                 # into ordinary numpy arrays
    assert np.array_equal(fb, s.frameBuffer) and np.array_equal(db, s.depthBuffer) and np.array_equal(ids, s.objectIds)
    assert md == s.maxFrameBufferDepth and rays == s.rayCount
    assert (ids != 0).any() and (db < 1e30).any()


def test_no_object_id_buffer_no_object_id_traffic():
    s = Scene(width=160, height=120, samplesPerPixel=2, maxTraceDepth=3, jitter=1)
    s.renderScene()
    fb, db = np.zeros_like(np.asarray(s.frameBuffer)), np.zeros_like(np.asarray(s.depthBuffer))
    md, rays = _render_into(s, fb, db, None)
    assert np.array_equal(fb, s.frameBuffer) and np.array_equal(db, s.depthBuffer) and rays == s.rayCount


def test_depth_and_ids_belong_to_the_frame_they_arrive_with():
    """Two different frames alternate on one handle: the early copies of depth / IDs must never pair with the colours of
    the other frame (the staging block is reused by every frame of a handle)."""
    s = Scene(width=320, height=240, samplesPerPixel=1, maxTraceDepth=4, jitter=0)
    s.renderScene()
    a = (s.frameBuffer.copy(), s.depthBuffer.copy(), s.objectIds.copy())
    cam_b = Scene(width=320, height=240, samplesPerPixel=1, maxTraceDepth=4, jitter=0, cameraOriginWS=(6.0, 3.0, 12.0)).camera
    cam_a = s.camera
    s.camera = cam_b
    s.renderScene()
    b = (s.frameBuffer.copy(), s.depthBuffer.copy(), s.objectIds.copy())
    assert not np.array_equal(a[1], b[1])
    for i in range(6):
        s.camera = cam_a if i % 2 == 0 else cam_b
        s.renderScene()
        want = a if i % 2 == 0 else b
        assert np.array_equal(s.frameBuffer, want[0]) and np.array_equal(s.depthBuffer, want[1]) and np.array_equal(s.objectIds, want[2])


def test_host_alloc_and_free():
    lib = abi.load_hip_library()
    ptr = C.c_void_p()
    assert lib.rt_host_alloc(1 << 20, C.byref(ptr)) == abi.RT_OK and ptr.value
    buf = (C.c_ubyte * (1 << 20)).from_address(ptr.value)
    buf[0], buf[(1 << 20) - 1] = 7, 9                      # host-accessible
    assert (buf[0], buf[(1 << 20) - 1]) == (7, 9)
    lib.rt_host_free(ptr)
