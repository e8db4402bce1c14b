"""ctypes binding of oracle/librt_oracle.so — the CPU checker.  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi, host

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(_ROOT, "oracle")
ORACLE_PATH = os.path.join(ORACLE_DIR, "librt_oracle.so")

FLAG_LITERAL_PLANE = 1
FLAG_LITERAL_LEAVES = 2

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_PATH):
        build()
    lib = C.CDLL(ORACLE_PATH)
    P = C.POINTER
    lib.rto_render.argtypes = [P(abi.rt_surface), C.c_int64, P(abi.rt_camera), P(abi.rt_params), C.c_uint, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_float), P(C.c_uint64)]
    lib.rto_render.restype = C.c_int
    lib.rto_round.argtypes = [C.c_float, C.c_int]
    lib.rto_round.restype = C.c_float
    lib.rto_pow5.argtypes = [C.c_float]
    lib.rto_pow5.restype = C.c_float
    lib.rto_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
    lib.rto_sincos_2pi.argtypes = [C.c_float, P(C.c_float), P(C.c_float)]
    lib.rto_intersect.argtypes = [P(abi.rt_surface), P(C.c_float), P(C.c_float), C.c_float,
                                  P(C.c_int), P(C.c_float), P(C.c_int)]
    lib.rto_generate_samples.argtypes = [P(abi.rt_surface), P(C.c_float), P(C.c_float), C.c_float,
                                         C.c_uint64, C.c_uint64, C.c_uint64, P(C.c_float), P(C.c_float)]
    lib.rto_generate_samples.restype = C.c_int
    lib.rto_generate_samples_single_path.argtypes = [P(abi.rt_surface), P(C.c_float), P(C.c_float), C.c_float,
                                                     C.c_uint64, C.c_uint64, C.c_uint64, P(C.c_float), P(C.c_float)]
    lib.rto_generate_samples_single_path.restype = C.c_int
    lib.rto_random_float.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    lib.rto_random_float.restype = C.c_float
    lib.rto_find_closest.argtypes = [P(abi.rt_surface), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rto_find_closest.restype = None
    lib.rto_set_raymap.argtypes = [C.c_void_p]
    lib.rto_set_raymap.restype = None
    lib.rto_hemisphere_sample.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, P(C.c_float)]
    lib.rto_unit_sphere_sample.argtypes = [C.c_uint64, C.c_uint64, P(C.c_float)]
    lib.rto_frame_around.argtypes = [P(C.c_float), P(C.c_float), P(C.c_float)]
    lib.rto_plane_rotation.argtypes = [P(C.c_float), P(C.c_float)]
    lib.rto_create_look_at.argtypes = [P(C.c_float)] * 4
    lib.rto_invert.argtypes = [P(C.c_float), P(C.c_float)]
    lib.rto_invert.restype = C.c_int
    _lib = lib
    return lib


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def render(surfaces, cam, params, flags=0, nthreads=0):
    """Oracle render. Returns dict(rgba[ncols,H,4], depth[ncols,H], ids[ncols,H], max_depth, rays)."""
    lib = load()
    arr = host.flatten_surfaces(surfaces) if isinstance(surfaces, (list, tuple, np.ndarray)) else surfaces
    n = len(surfaces)
    ncols, H = host.num_columns(params), params.height
    rgba = np.empty((ncols, H, 4), dtype=np.float32)
    depth = np.empty((ncols, H), dtype=np.float32)
    ids = np.empty((ncols, H), dtype=np.uint64)
    md, rc = C.c_float(0), C.c_uint64(0)
    r = lib.rto_render(arr, n, C.byref(cam), C.byref(params), flags, nthreads,
                       rgba.ctypes.data, depth.ctypes.data, ids.ctypes.data, C.byref(md), C.byref(rc))
    if r != 0:
        raise RuntimeError(f"rto_render failed: {r}")
    return dict(rgba=rgba, depth=depth, ids=ids, max_depth=md.value, rays=rc.value)


def find_closest(surfaces, origins, directions):
    """Oracle findClosestIntersection for a batch of rays. Returns (t, index)."""
    lib = load()
    arr = host.flatten_surfaces(surfaces) if isinstance(surfaces, (list, tuple, np.ndarray)) else surfaces
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    t = np.empty(len(o), dtype=np.float32)
    idx = np.empty(len(o), dtype=np.int32)
    lib.rto_find_closest(arr, len(surfaces), len(o), o.ctypes.data, d.ctypes.data, t.ctypes.data, idx.ctypes.data)
    return t, idx


def intersect(surface, o, d, dot_view=1.0):
    """Hitable.Intersect + IntersectionAcceptable for one surface. Returns (has, t, acceptable)."""
    lib = load()
    s = surface.flatten() if isinstance(surface, host.Surface) else surface
    has, acc, t = C.c_int(0), C.c_int(0), C.c_float(0)
    lib.rto_intersect(C.byref(s), f3(o), f3(d), dot_view, C.byref(has), C.byref(t), C.byref(acc))
    return bool(has.value), t.value, bool(acc.value)


def generate_samples(surface, o, d, t, seed=1, ps=0, node=0):
    """Surface.GenerateSamples. Returns (n, dirs[n,3], shading[n,3])."""
    lib = load()
    s = surface.flatten() if isinstance(surface, host.Surface) else surface
    dirs, sh = (C.c_float * 12)(), (C.c_float * 12)()
    n = lib.rto_generate_samples(C.byref(s), f3(o), f3(d), t, seed, ps, node, dirs, sh)
    return n, np.array(dirs[:3 * n], dtype=np.float32).reshape(n, 3), np.array(sh[:3 * n], dtype=np.float32).reshape(n, 3)


def generate_samples_single_path(surface, o, d, t, seed=1, ps=0, node=0):
    """Surface.GenerateSamples under RT_ESTIMATOR_SINGLE_PATH. Returns (n, dir[3], shading[3])."""
    lib = load()
    s = surface.flatten() if isinstance(surface, host.Surface) else surface
    dr, sh = (C.c_float * 3)(), (C.c_float * 3)()
    n = lib.rto_generate_samples_single_path(C.byref(s), f3(o), f3(d), t, seed, ps, node, dr, sh)
    return n, np.array(dr[:], dtype=np.float32), np.array(sh[:], dtype=np.float32)
