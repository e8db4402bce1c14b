"""rt_render_multi: renderScene over several scene handles from one host thread, interleaved columns
per share, shares placed at stride n in the root device's frame.  One GPU is available to the tests, so the shares
are separate handles on device 0 (the partition, the placement kernel and the reduction do not care where a handle
lives; across devices the only difference is that the placement stores travel over xGMI).  Also: two gloo ranks
that render their columns with the HIP kernel (the multi-process path of bench.py, with gloo standing in for RCCL
because both ranks share the one GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

from raytracer_amd import abi, host, scenes
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(width=101, height=64, samplesPerPixel=3, maxTraceDepth=4, jitter=1, seed=12)


def _frame(s):
    return s.frameBuffer.copy(), s.depthBuffer.copy(), s.objectIds.copy(), s.maxFrameBufferDepth, s.rayCount


def _equal(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3] == b[3] and a[4] == b[4]


def test_one_share_equals_rt_render_bit_for_bit():
    s = Scene(**KW)
    s.renderScene()
    want = _frame(s)
    s.renderSceneMulti()
    _equal(_frame(s), want)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_n_shares_give_the_same_frame(n):
    s = Scene(**KW)
    s.renderScene()
    want = _frame(s)
    others = [Scene(**KW) for _ in range(n - 1)]
    s.frameBuffer[:] = -1
    s.renderSceneMulti(others)
    _equal(_frame(s), want)


def test_more_shares_than_columns_and_bvh_scene():
    m = np.concatenate([np.frombuffer(bytes(host.flatten_surfaces(host.SceneDefinitions.scene_spheres)), dtype=abi.surface_dtype, count=12).copy(),
                        scenes.heightfield_mesh(12, environment=False, base=-5.9, amplitude=1.0)])
    kw = dict(surfaces=m, width=5, height=40, samplesPerPixel=2, maxTraceDepth=3, jitter=1, seed=2)
    s = Scene(**kw)
    s.renderScene()
    want = _frame(s)
    s.renderSceneMulti([Scene(**kw) for _ in range(6)])      # 7 shares, 5 columns
    _equal(_frame(s), want)


def test_multi_argument_errors():
    s = Scene(**KW)
    s.upload()
    import ctypes as C
    lib = abi.load_hip_library()
    p = s.params(1, s.width, 2)
    arr = (C.c_void_p * 2)(s._scene, s._scene)
    md, rays = C.c_float(0), C.c_uint64(0)
    rc = lib.rt_render_multi(arr, 1, C.byref(s.camera), C.byref(p), s.frameBuffer.ctypes.data, s.depthBuffer.ctypes.data,
                             None, C.byref(md), C.byref(rays))
    assert rc == abi.RT_E_INVALID and b"whole frame" in lib.rt_last_error()
    p = s.params()
    rc = lib.rt_render_multi(arr, 2, C.byref(s.camera), C.byref(p), s.frameBuffer.ctypes.data, s.depthBuffer.ctypes.data,
                             None, C.byref(md), C.byref(rays))
    assert rc == abi.RT_E_INVALID and b"twice" in lib.rt_last_error()


def test_two_handles_render_concurrently_on_their_own_streams():
    """rt_render works on the scene's own stream and no longer synchronises the device: a render on one handle while
    another handle's asynchronous work is still queued must not wait for it or disturb it."""
    import torch
    a, b = Scene(**KW), Scene(width=64, height=48, samplesPerPixel=2, maxTraceDepth=3, jitter=1, seed=1)
    a.renderScene()
    want = _frame(a)
    dev = torch.device("cuda", 0)
    rgba = torch.zeros((b.width, b.height, 4), dtype=torch.float32, device=dev)
    depth = torch.zeros((b.width, b.height), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream()
    b.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), None, None, side.cuda_stream)
    a.renderScene()
    _equal(_frame(a), want)
    side.synchronize()
    b.renderScene()
    assert np.array_equal(rgba.cpu().numpy(), b.frameBuffer)


# ---- two processes, HIP kernel in each, frame gathered with torch.distributed (gloo over CPU tensors) -------------
def _free_port():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from raytracer_amd import multi
    from raytracer_amd.scene import Scene as S

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = S(**KW)
        xb, xe, xs, ncols = multi.columns_of_rank(s.width, rank, world)
        rgba, depth, ids, md, rays = s.renderColumns(xb, xe, xs)          # the HIP kernel, through the C ABI
        assert rgba.shape[0] == ncols
        g = multi.FrameGatherer(s.width, s.height, torch.device("cpu"), rank, world, with_ids=True)
        total, maxd = g.gather(torch.from_numpy(rgba), torch.from_numpy(depth), torch.from_numpy(ids.astype(np.int64)), rays, md)
        if rank == 0:
            np.savez(out_path, rgba=g.rgba.numpy(), depth=g.depth.numpy(), ids=g.ids.numpy(), rays=total, maxd=maxd)
    finally:
        dist.destroy_process_group()


def test_two_processes_render_with_the_hip_kernel_and_gather(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    s = Scene(**KW)
    s.renderScene()
    assert np.array_equal(got["rgba"], s.frameBuffer) and np.array_equal(got["depth"], s.depthBuffer)
    assert np.array_equal(got["ids"].astype(np.uint64), s.objectIds)
    assert int(got["rays"]) == s.rayCount and np.float32(got["maxd"]) == np.float32(s.maxFrameBufferDepth)
