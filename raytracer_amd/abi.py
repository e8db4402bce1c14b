"""ctypes view of include/rt_abi.h (struct layouts, constants, librt_hip.so loader).

Only plumbing lives here: the structs an F# host would declare for P/Invoke
restated for Python, and the loader for the in-tree HIP library.  There is no CPU fallback:
if librt_hip.so is missing or no HIP device is present, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

RT_OK = 0
RT_E_INVALID = -1
RT_E_NO_DEVICE = -2
RT_E_HIP = -3
RT_E_OVERFLOW = -4
RT_E_UNSUPPORTED = -5
RT_E_RANGE = -6

ESTIMATOR_BRANCHING, ESTIMATOR_SINGLE_PATH = 0, 1
KIND_NOSURFACE, KIND_LAMBERTIAN, KIND_METAL, KIND_DIELECTRIC = 0, 1, 2, 3
GEOM_SPHERE, GEOM_PLANE, GEOM_TRIANGLE = 0, 1, 2

RT_ABI_VERSION = 8
RT_STATS_WORDS = 20
RT_NOALLOC_SCRATCH_BYTES = 64 << 20
RT_NOALLOC_MAX_PIXELS = 1 << 20


class rt_surface(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("geom", C.c_int32),
        ("id", C.c_uint64),
        ("albedo", C.c_float * 3),
        ("emission", C.c_float * 3),
        ("fuzz", C.c_float),
        ("ior", C.c_float),
        ("sample_count", C.c_int32),
        ("g", C.c_float * 12),
    ]


class rt_camera(C.Structure):
    _fields_ = [
        ("cam_to_world", C.c_float * 16),
        ("look_at", C.c_float * 3),
        ("fov", C.c_float),
    ]


class rt_params(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("spp", C.c_int32),
        ("max_trace_depth", C.c_int32),
        ("background", C.c_float * 3),
        ("jitter", C.c_int32),
        ("seed", C.c_uint64),
        ("lambertian_samples", C.c_int32),
        ("x_begin", C.c_int32),
        ("x_end", C.c_int32),
        ("x_step", C.c_int32),
        ("waves_per_cu", C.c_int32),
        ("pool_blocks", C.c_int32),
        ("estimator", C.c_int32),
        ("bvh_ring", C.c_int32),
        ("bvh_min_active", C.c_int32),
        ("two_pass", C.c_int32),
    ]


# numpy view of rt_surface with the same layout (for bulk scene generation)
surface_dtype = np.dtype(
    [
        ("kind", "<i4"),
        ("geom", "<i4"),
        ("id", "<u8"),
        ("albedo", "<f4", (3,)),
        ("emission", "<f4", (3,)),
        ("fuzz", "<f4"),
        ("ior", "<f4"),
        ("sample_count", "<i4"),
        ("g", "<f4", (12,)),
    ],
    align=True,
)
assert surface_dtype.itemsize == C.sizeof(rt_surface), (surface_dtype.itemsize, C.sizeof(rt_surface))

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIBRARY selects another build of the same library (tools/ab_libs.sh: same-box A/B of kernel variants built with
# `make -C raytracer_amd/csrc variant TAG=... DEFS=...`); it is still this HIP library or nothing.
LIB_HIP_PATH = os.environ.get("RT_HIP_LIBRARY") or os.path.join(_PKG_DIR, "csrc", "librt_hip.so")

# every symbol include/rt_abi.h declares
ABI_SYMBOLS = (
    "rt_scene_create",
    "rt_render",
    "rt_render_device",
    "rt_render_device_noalloc",
    "rt_render_multi",
    "rt_find_closest",
    "rt_scene_status",
    "rt_scene_counters",
    "rt_host_alloc",
    "rt_host_free",
    "rt_scene_destroy",
    "rt_last_error",
    "rt_abi_version",
)

_lib = None


def load_hip_library():
    """dlopen the in-tree librt_hip.so and declare its signatures. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_HIP_PATH):
        raise RuntimeError(
            f"{LIB_HIP_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the render path)"
        )
    lib = C.CDLL(LIB_HIP_PATH, mode=C.RTLD_GLOBAL)
    P = C.POINTER
    lib.rt_scene_create.argtypes = [P(rt_surface), C.c_int64, C.c_int32, P(C.c_void_p)]
    lib.rt_scene_create.restype = C.c_int
    lib.rt_render.argtypes = [C.c_void_p, P(rt_camera), P(rt_params), C.c_void_p, C.c_void_p,
                              C.c_void_p, P(C.c_float), P(C.c_uint64)]
    lib.rt_render.restype = C.c_int
    lib.rt_render_device.argtypes = [C.c_void_p, P(rt_camera), P(rt_params), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rt_render_device.restype = C.c_int
    lib.rt_render_device_noalloc.argtypes = lib.rt_render_device.argtypes
    lib.rt_render_device_noalloc.restype = C.c_int
    lib.rt_render_multi.argtypes = [P(C.c_void_p), C.c_int32, P(rt_camera), P(rt_params), C.c_void_p, C.c_void_p,
                                    C.c_void_p, P(C.c_float), P(C.c_uint64)]
    lib.rt_render_multi.restype = C.c_int
    lib.rt_find_closest.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rt_find_closest.restype = C.c_int
    lib.rt_scene_status.argtypes = [C.c_void_p]
    lib.rt_scene_status.restype = C.c_int
    lib.rt_scene_counters.argtypes = [C.c_void_p, C.c_int]
    lib.rt_scene_counters.restype = C.c_int
    lib.rt_host_alloc.argtypes = [C.c_uint64, P(C.c_void_p)]
    lib.rt_host_alloc.restype = C.c_int
    lib.rt_host_free.argtypes = [C.c_void_p]
    lib.rt_host_free.restype = None
    lib.rt_scene_destroy.argtypes = [C.c_void_p]
    lib.rt_scene_destroy.restype = None
    lib.rt_last_error.argtypes = []
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_abi_version.argtypes = []
    lib.rt_abi_version.restype = C.c_int
    if lib.rt_abi_version() != RT_ABI_VERSION:
        raise RuntimeError("librt_hip.so ABI version mismatch: rebuild")
    _lib = lib
    return lib


class PinnedBlock:
    """`nbytes` of pinned host memory from rt_host_alloc, exposed to numpy (`np.asarray(block)` is a u8 view whose base
    keeps this object, and with it the memory, alive; the memory is returned when the last view is gone)."""

    def __init__(self, lib, nbytes):
        ptr = C.c_void_p()
        check(lib, lib.rt_host_alloc(nbytes, C.byref(ptr)))
        self._lib, self.ptr, self.nbytes = lib, ptr.value, nbytes
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def array(self, dtype, shape):
        return np.asarray(self).view(dtype).reshape(shape)

    def __del__(self):
        try:
            if self.ptr:
                self._lib.rt_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class RtError(RuntimeError):
    """A C-ABI call returned a negative code (the F# wrapper would `failwith`, cf. Camera.fs:45)."""

    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


def check(lib, rc):
    if rc != RT_OK:
        raise RtError(rc, lib.rt_last_error().decode("utf-8", "replace"))
