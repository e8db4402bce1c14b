"""`Scene` — host-side mirror of the reference's renderer type (raytracer/Scene.fs:16-144).

Same members as the F# class (`renderScene`, `saveFrameBuffer`, `saveDepthBuffer`) and the same
state (`frameBuffer`, `depthBuffer`, `maxFrameBufferDepth`); the constants the reference
hard-codes as `let` bindings (Scene.fs:18-43) are constructor arguments with the reference's
values as defaults.  `renderScene` is one call through the C ABI (include/rt_abi.h) into
librt_hip.so; there is no CPU rendering path here.
"""
import ctypes as C
import math

import numpy as np

from . import abi, host


class Scene:
    def __init__(self, surfaces=None, width=800, height=640, samplesPerPixel=1, maxTraceDepth=5,
                 backgroundColor=(0.0, 0.0, 0.0), cameraOriginWS=(-3.0, 6.0, 15.0), lookAt=(-1.0, 1.0, -10.0),
                 fov=None, seed=1, jitter=1, lambertian_samples=0, device=0, waves_per_cu=0, pool_blocks=0,
                 estimator=0, bvh_ring=0, bvh_min_active=0, two_pass=0):
        if isinstance(surfaces, np.ndarray):
            self.surfaces = np.ascontiguousarray(surfaces)          # flat rt_surface records (raytracer_amd.scenes)
        else:
            self.surfaces = list(host.SceneDefinitions.scene_spheres if surfaces is None else surfaces)   # Scene.fs:33
        self.width, self.height = int(width), int(height)                                             # Scene.fs:18-19
        self.samplesPerPixel, self.maxTraceDepth = int(samplesPerPixel), int(maxTraceDepth)           # Scene.fs:20,31
        self.backgroundColor = tuple(backgroundColor)                                                 # Scene.fs:32
        self.camera = host.make_camera(cameraOriginWS, lookAt, (0.0, 1.0, 0.0), fov)                  # Scene.fs:36-43
        self.seed, self.jitter, self.lambertian_samples = int(seed), int(jitter), int(lambertian_samples)
        self.device, self.waves_per_cu, self.pool_blocks = int(device), int(waves_per_cu), int(pool_blocks)
        self.estimator, self.bvh_ring, self.bvh_min_active = int(estimator), int(bvh_ring), int(bvh_min_active)
        self.two_pass = int(two_pass)
        self.frameBuffer = np.zeros((self.width, self.height, 4), dtype=np.float32)                  # Scene.fs:28
        self.depthBuffer = np.full((self.width, self.height), np.finfo(np.float32).max, dtype=np.float32)  # Scene.fs:29
        self.objectIds = np.zeros((self.width, self.height), dtype=np.uint64)
        self.maxFrameBufferDepth = 0.0                                                                # Scene.fs:30
        self.rayCount = 0
        self._lib = None
        self._scene = C.c_void_p()
        self._pinned = False

    # -- plumbing ---------------------------------------------------------------------------
    def params(self, x_begin=0, x_end=None, x_step=1):
        return host.make_params(self.width, self.height, self.samplesPerPixel, self.maxTraceDepth,
                                self.backgroundColor, self.jitter, self.seed, self.lambertian_samples,
                                x_begin, x_end, x_step, self.waves_per_cu, self.pool_blocks,
                                self.estimator, self.bvh_ring, self.bvh_min_active, self.two_pass)

    def upload(self):
        """Flatten the surfaces and upload them once to HBM (rt_scene_create)."""
        if self._scene:
            return self._scene
        self._lib = abi.load_hip_library()
        arr = host.flatten_surfaces(self.surfaces)
        abi.check(self._lib, self._lib.rt_scene_create(arr, len(self.surfaces), self.device, C.byref(self._scene)))
        return self._scene

    def _pin_buffers(self):
        """The frame buffers move to pinned host memory (rt_host_alloc) before the first render into them, so that
        rt_render copies by DMA; the arrays keep their block alive for as long as anyone holds them (abi.PinnedBlock)."""
        if self._pinned:
            return
        self._pinned = True
        try:
            for name in ("frameBuffer", "depthBuffer", "objectIds"):
                old = getattr(self, name)
                new = abi.PinnedBlock(self._lib, old.nbytes).array(old.dtype, old.shape)
                new[...] = old
                setattr(self, name, new)
        except abi.RtError:
            pass                  # pageable buffers work too, only slower

    def close(self):
        if self._scene:
            self._lib.rt_scene_destroy(self._scene)
            self._scene = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- Scene.fs:117-121 -------------------------------------------------------------------
    def renderScene(self):
        self.upload()
        self._pin_buffers()
        p = self.params()
        md, rays = C.c_float(0.0), C.c_uint64(0)
        abi.check(self._lib, self._lib.rt_render(self._scene, C.byref(self.camera), C.byref(p),
                                                 self.frameBuffer.ctypes.data, self.depthBuffer.ctypes.data,
                                                 self.objectIds.ctypes.data, C.byref(md), C.byref(rays)))
        self.maxFrameBufferDepth, self.rayCount = md.value, rays.value

    def renderColumns(self, x_begin, x_end, x_step=1):
        """Render the columns x_begin, x_begin+x_step, ... < x_end into compact host arrays
        (the multi-GPU partition of include/rt_abi.h). Returns (rgba, depth, ids, max_depth, rays)."""
        self.upload()
        p = self.params(x_begin, x_end, x_step)
        ncols = host.num_columns(p)
        rgba = np.zeros((ncols, self.height, 4), dtype=np.float32)
        depth = np.zeros((ncols, self.height), dtype=np.float32)
        ids = np.zeros((ncols, self.height), dtype=np.uint64)
        md, rays = C.c_float(0.0), C.c_uint64(0)
        abi.check(self._lib, self._lib.rt_render(self._scene, C.byref(self.camera), C.byref(p), rgba.ctypes.data,
                                                 depth.ctypes.data, ids.ctypes.data, C.byref(md), C.byref(rays)))
        return rgba, depth, ids, md.value, rays.value

    def renderColumnsDevice(self, d_rgba, d_depth, d_ids=None, d_stats=None, stream=None,
                            x_begin=0, x_end=None, x_step=1):
        """Asynchronous render into DEVICE buffers (raw pointers, e.g. torch `data_ptr()`), on `stream`
        (a hipStream_t value, e.g. `torch.cuda.current_stream().cuda_stream`).  d_stats: abi.RT_STATS_WORDS x u64."""
        self.upload()
        p = self.params(x_begin, x_end, x_step)
        abi.check(self._lib, self._lib.rt_render_device(self._scene, C.byref(self.camera), C.byref(p),
                                                        C.c_void_p(stream or 0), C.c_void_p(d_rgba), C.c_void_p(d_depth),
                                                        C.c_void_p(d_ids or 0), C.c_void_p(d_stats or 0)))

    def renderColumnsDeviceNoAlloc(self, d_rgba, d_depth, d_ids=None, d_stats=None, stream=None,
                                   x_begin=0, x_end=None, x_step=1):
        """`renderColumnsDevice` through rt_render_device_noalloc: same results, no hipMalloc inside the call
        (static per-GPU scratch; at most abi.RT_NOALLOC_MAX_PIXELS pixels)."""
        self.upload()
        p = self.params(x_begin, x_end, x_step)
        abi.check(self._lib, self._lib.rt_render_device_noalloc(self._scene, C.byref(self.camera), C.byref(p),
                                                                C.c_void_p(stream or 0), C.c_void_p(d_rgba), C.c_void_p(d_depth),
                                                                C.c_void_p(d_ids or 0), C.c_void_p(d_stats or 0)))

    def renderSceneMulti(self, others=()):
        """Scene.renderScene on several GPUs from one host thread (rt_render_multi): `self` is the root (its device
        holds the gathered frame, its buffers receive it), `others` are Scenes with the same surfaces on other
        devices (or on the same one: the partition does not care).  Share i renders the columns i, i+n, ..."""
        scenes = [self] + list(others)
        for s in scenes:
            s.upload()
        self._pin_buffers()
        arr = (C.c_void_p * len(scenes))(*[s._scene for s in scenes])
        p = self.params()
        md, rays = C.c_float(0.0), C.c_uint64(0)
        abi.check(self._lib, self._lib.rt_render_multi(arr, len(scenes), C.byref(self.camera), C.byref(p),
                                                       self.frameBuffer.ctypes.data, self.depthBuffer.ctypes.data,
                                                       self.objectIds.ctypes.data, C.byref(md), C.byref(rays)))
        self.maxFrameBufferDepth, self.rayCount = md.value, rays.value

    def status(self):
        self.upload()
        abi.check(self._lib, self._lib.rt_scene_status(self._scene))

    def counters(self, level):
        """rt_scene_counters: 1 = the frames of this scene fill every word of d_stats (rays by kind, BVH visits ...),
        0 (default) = ray count, largest depth and wave clocks only."""
        self.upload()
        abi.check(self._lib, self._lib.rt_scene_counters(self._scene, int(level)))

    def findClosestIntersection(self, origins, directions):
        """Batch form of Surface.fs:49-58 on the device: returns (t, index) arrays."""
        self.upload()
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        t = np.empty(len(o), dtype=np.float32)
        idx = np.empty(len(o), dtype=np.int32)
        abi.check(self._lib, self._lib.rt_find_closest(self._scene, len(o), o.ctypes.data, d.ctypes.data,
                                                       t.ctypes.data, idx.ctypes.data))
        return t, idx

    # -- Scene.fs:123-144 (PNG instead of JPEG: lossless, no ImageSharp) -------------------------
    @staticmethod
    def _to_rgba32(v):
        """ImageSharp `Rgba32(Vector4)` (Scene.fs:128,140): v*255 + 0.5, clamped to [0,255], truncated to a byte."""
        with np.errstate(over="ignore", invalid="ignore"):
            s = np.asarray(v, dtype=np.float32) * np.float32(255.0) + np.float32(0.5)
        s = np.where(np.isnan(s), np.float32(0.0), s)
        return np.clip(s, 0.0, 255.0).astype(np.uint8)

    def frameBufferImage(self):
        """[H, W, 4] u8: the image saveFrameBuffer hands to the encoder (image.Item(px,py) <- Rgba32 frameBuffer[px,py])."""
        return self._to_rgba32(self.frameBuffer).transpose(1, 0, 2)

    def depthBufferImage(self):
        """[H, W, 4] u8: Vector4(Vector3(depth / maxFrameBufferDepth), 1) per pixel (Scene.fs:139); a miss holds
        Single.MaxValue and saturates to white."""
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            g = self.depthBuffer / np.float32(self.maxFrameBufferDepth)
        rgba = np.stack([g, g, g, np.ones_like(g)], axis=-1)
        return self._to_rgba32(rgba).transpose(1, 0, 2)

    def saveFrameBuffer(self, path="sphere.png"):
        from PIL import Image
        Image.fromarray(self.frameBufferImage(), "RGBA").convert("RGB").save(path)

    def saveDepthBuffer(self, path="depth.png"):
        from PIL import Image
        Image.fromarray(self.depthBufferImage(), "RGBA").convert("RGB").save(path)
