"""Host-side scene description, mirroring the reference's F# API for the render path.

Names follow the reference: `Material` (Material.fs:9-18), `Sphere` (Geometry/Sphere.fs:8),
`Plane` (Geometry/Plane.fs:12), `NoSurface`/`Lambertian`/`Metal`/`Dielectric`
(Surface.fs:44,60,85,112), `SurfaceNormal` (Numerics.fs:28), `WorldToCamera`/`CameraToWorld`
(Camera.fs:41-45), the scene lists of SceneDefinitions.fs and `Scene` (Scene.fs:16).
Nothing here computes pixels: objects only flatten themselves into `rt_surface` records for
the C ABI (include/rt_abi.h).  All arithmetic is float32, one rounding per operation, in the
order System.Numerics uses (see DESIGN.md "Numerics contract").
"""
import ctypes as C
import math

import numpy as np

from . import abi

f32 = np.float32


# ---------------------------------------------------------------------------------------------
# System.Numerics subset (float32, scalar op order)
# ---------------------------------------------------------------------------------------------
def vec3(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross(a, b):
    return vec3(
        f32(f32(a[1] * b[2]) - f32(a[2] * b[1])),
        f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
        f32(f32(a[0] * b[1]) - f32(a[1] * b[0])),
    )


def length(a):
    return f32(np.sqrt(dot(a, a)))


def div_scalar(a, s):
    """Vector3 operator/(Vector3, float): multiply by the reciprocal."""
    inv = f32(f32(1.0) / f32(s))
    return vec3(f32(a[0] * inv), f32(a[1] * inv), f32(a[2] * inv))


def normalize(a):
    return div_scalar(a, length(a))


def SurfaceNormal(a, b, c):
    """Numerics.fs:28"""
    return normalize(vec3(a, b, c))


def create_look_at(position, target, up):
    """Matrix4x4.CreateLookAt, row-major M11..M44."""
    position, target, up = (np.asarray(v, dtype=np.float32) for v in (position, target, up))
    z = normalize(position - target)
    x = normalize(cross(up, z))
    y = cross(z, x)
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, :3] = (x[0], y[0], z[0])
    m[1, :3] = (x[1], y[1], z[1])
    m[2, :3] = (x[2], y[2], z[2])
    m[3, :3] = (-dot(x, position), -dot(y, position), -dot(z, position))
    m[3, 3] = 1.0
    return m


def invert(M):
    """Matrix4x4.Invert: cofactor expansion in float32. Returns (success, inverse)."""
    a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p = (f32(v) for v in np.asarray(M, dtype=np.float32).reshape(16))
    kp_lo, jp_ln, jo_kn = k * p - l * o, j * p - l * n, j * o - k * n
    ip_lm, io_km, in_jm = i * p - l * m, i * o - k * m, i * n - j * m
    a11 = +(f * kp_lo - g * jp_ln + h * jo_kn)
    a12 = -(e * kp_lo - g * ip_lm + h * io_km)
    a13 = +(e * jp_ln - f * ip_lm + h * in_jm)
    a14 = -(e * jo_kn - f * io_km + g * in_jm)
    det = a * a11 + b * a12 + c * a13 + d * a14
    if abs(det) < np.finfo(np.float32).smallest_subnormal:
        return False, np.full((4, 4), np.nan, dtype=np.float32)
    inv = f32(1.0) / det
    r = np.zeros(16, dtype=np.float32)
    r[0], r[4], r[8], r[12] = a11 * inv, a12 * inv, a13 * inv, a14 * inv
    r[1] = -(b * kp_lo - c * jp_ln + d * jo_kn) * inv
    r[5] = +(a * kp_lo - c * ip_lm + d * io_km) * inv
    r[9] = -(a * jp_ln - b * ip_lm + d * in_jm) * inv
    r[13] = +(a * jo_kn - b * io_km + c * in_jm) * inv
    gp_ho, fp_hn, fo_gn = g * p - h * o, f * p - h * n, f * o - g * n
    ep_hm, eo_gm, en_fm = e * p - h * m, e * o - g * m, e * n - f * m
    r[2] = +(b * gp_ho - c * fp_hn + d * fo_gn) * inv
    r[6] = -(a * gp_ho - c * ep_hm + d * eo_gm) * inv
    r[10] = +(a * fp_hn - b * ep_hm + d * en_fm) * inv
    r[14] = -(a * fo_gn - b * eo_gm + c * en_fm) * inv
    gl_hk, fl_hj, fk_gj = g * l - h * k, f * l - h * j, f * k - g * j
    el_hi, ek_gi, ej_fi = e * l - h * i, e * k - g * i, e * j - f * i
    r[3] = -(b * gl_hk - c * fl_hj + d * fk_gj) * inv
    r[7] = +(a * gl_hk - c * el_hi + d * ek_gi) * inv
    r[11] = -(a * fl_hj - b * el_hi + d * ej_fi) * inv
    r[15] = +(a * fk_gj - b * ek_gi + c * ej_fi) * inv
    return True, r.reshape(4, 4)


def WorldToCamera(position, target, up):
    """Camera.fs:41"""
    return create_look_at(position, target, up)


def CameraToWorld(world_to_camera):
    """Camera.fs:43-45"""
    ok, inv = invert(world_to_camera)
    if not ok:
        raise RuntimeError("Matrix4x4 Invert Failed")
    return inv


# ---------------------------------------------------------------------------------------------
# Material.fs / ImageSharp named colours (bytes / 255, Material.fs:14)
# ---------------------------------------------------------------------------------------------
class Rgba32:
    White = (255, 255, 255)
    Red = (255, 0, 0)
    Green = (0, 128, 0)
    DarkGreen = (0, 100, 0)
    RoyalBlue = (65, 105, 225)
    AntiqueWhite = (250, 235, 215)
    IndianRed = (205, 92, 92)

    @staticmethod
    def to_vector3(rgb):
        return vec3(*(f32(c) / f32(255.0) for c in rgb))


class Material:
    """Material.fs:9-18: (color) | (rgba32) | (rgba32, albedoFactor, rgba32, emittingFactor)."""

    def __init__(self, color, albedo_factor=None, emittance=None, emitting_factor=None):
        col = Rgba32.to_vector3(color) if isinstance(color, tuple) and all(isinstance(c, int) for c in color) \
            else np.asarray(color, dtype=np.float32)
        if albedo_factor is None:
            self.Albedo, self.Emmitance = col, vec3(0, 0, 0)
        else:
            self.Albedo = (f32(albedo_factor) * col).astype(np.float32)
            self.Emmitance = (f32(emitting_factor) * Rgba32.to_vector3(emittance)).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# Geometry
# ---------------------------------------------------------------------------------------------
class Sphere:
    """Geometry/Sphere.fs:8"""

    def __init__(self, center, radius):
        self.Center = np.asarray(center, dtype=np.float32)
        self.Radius = f32(radius)

    def flatten(self, s):
        s.geom = abi.GEOM_SPHERE
        s.g[0:4] = [*self.Center, self.Radius]


class NumericsPlane:
    """System.Numerics.Plane(normal, d) and Plane.CreateFromVertices."""

    def __init__(self, normal, d):
        self.Normal = np.asarray(normal, dtype=np.float32)
        self.D = f32(d)

    @staticmethod
    def CreateFromVertices(p1, p2, p3):
        p1, p2, p3 = (np.asarray(p, dtype=np.float32) for p in (p1, p2, p3))
        n = normalize(cross(p2 - p1, p3 - p1))
        return NumericsPlane(n, -dot(n, p1))


class Plane:
    """Geometry/Plane.fs:12 — plane, optional rectangle centre, width, height."""

    def __init__(self, plane, center=None, width=None, height=None):
        self.Plane = plane
        self.Center = None if center is None else np.asarray(center, dtype=np.float32)
        self.Width, self.Height = width, height

    def flatten(self, s):
        s.geom = abi.GEOM_PLANE
        s.g[0:4] = [*self.Plane.Normal, self.Plane.D]
        if self.Center is not None:
            s.g[4] = 1.0
            s.g[5:8] = list(self.Center)
            s.g[8], s.g[9] = f32(self.Width), f32(self.Height)


class Triangle:
    """A triangle Hitable.  NOT in the reference (Geometry/AABB.fs is empty, there is no triangle type): the
    build's own primitive obeying the Hitable contract of Geometry/Core.fs:26-43 (include/rt_abi.h)."""

    def __init__(self, v0, v1, v2):
        self.V0, self.V1, self.V2 = (np.asarray(v, dtype=np.float32) for v in (v0, v1, v2))

    def flatten(self, s):
        s.geom = abi.GEOM_TRIANGLE
        s.g[0:9] = [*self.V0, *self.V1, *self.V2]


# ---------------------------------------------------------------------------------------------
# Surface.fs
# ---------------------------------------------------------------------------------------------
class Surface:
    kind = abi.KIND_NOSURFACE

    def __init__(self, id, geometry, material, fuzz=0.0, ior=0.0, sample_count=0):
        self.ID, self.Geometry, self.Material = int(id), geometry, material
        self.fuzz, self.ior, self.sample_count = fuzz, ior, sample_count

    def flatten(self):
        s = abi.rt_surface()
        s.kind, s.id = self.kind, self.ID
        s.albedo[:] = list(self.Material.Albedo)
        s.emission[:] = list(self.Material.Emmitance)
        s.fuzz, s.ior, s.sample_count = self.fuzz, self.ior, self.sample_count
        self.Geometry.flatten(s)
        return s


class NoSurface(Surface):
    kind = abi.KIND_NOSURFACE


class Lambertian(Surface):
    kind = abi.KIND_LAMBERTIAN


class Metal(Surface):
    kind = abi.KIND_METAL

    def __init__(self, id, geometry, material, fuzz):
        super().__init__(id, geometry, material, fuzz=fuzz)


class Dielectric(Surface):
    kind = abi.KIND_DIELECTRIC

    def __init__(self, id, geometry, material, refractive_index):
        super().__init__(id, geometry, material, ior=refractive_index)


def flatten_surfaces(surfaces):
    """`Surface list -> rt_surface[]` (what `scene_spheres |> Array.ofList` becomes, Scene.fs:33).
    A numpy array of dtype `abi.surface_dtype` (bulk-generated scenes) is passed through as a pointer."""
    if isinstance(surfaces, np.ndarray):
        assert surfaces.dtype == abi.surface_dtype and surfaces.flags["C_CONTIGUOUS"]
        return surfaces.ctypes.data_as(C.POINTER(abi.rt_surface))
    arr = (abi.rt_surface * max(len(surfaces), 1))()
    for i, s in enumerate(surfaces):
        arr[i] = s.flatten() if isinstance(s, Surface) else s
    return arr


# ---------------------------------------------------------------------------------------------
# SceneDefinitions.fs — every list, in module-initialisation order so the IDs match
# (assignIDAndIncrement, SceneDefinitions.fs:16-22).
# ---------------------------------------------------------------------------------------------
class _Ids:
    def __init__(self):
        self.id = 1

    def __call__(self):
        v = self.id
        self.id += 1
        return v


def _build_scene_definitions():
    nid = _Ids()
    W = Rgba32
    n111 = SurfaceNormal(-1.0, -1.0, 1.0)
    floor = lambda: Plane(NumericsPlane.CreateFromVertices((-1.0, -6.0, 0.0), (1.0, -6.0, 0.0), (0.0, -6.0, -1.0)))
    d = {}
    d["lights"] = [
        NoSurface(nid(), Plane(NumericsPlane(n111, 20.0), (n111 * f32(-20.0)).astype(np.float32), 40.0, 13.0),
                  Material(W.White, 0.0, W.White, 1.0)),
        NoSurface(nid(), Plane(NumericsPlane(SurfaceNormal(1.0, 0.0, 0.0), 15.0), (-15.0, 5.0, -10.0), 20.0, 10.0),
                  Material(W.White, 0.0, W.White, 1.0)),
    ]
    d["lights_simple"] = [
        NoSurface(nid(), Sphere((-15.0, 5.0, -10.0), 5.0), Material(W.White, 0.0, W.White, 1.0)),
        NoSurface(nid(), Sphere((-5.0, 11.0, -13.0), 5.0), Material(W.White, 0.0, W.White, 1.0)),
    ]
    d["spheres"] = [
        Dielectric(nid(), Sphere((4.0, -2.0, -12.0), 2.0), Material(W.White), 1.3),
        Metal(nid(), Sphere((3.0, -1.0, -19.0), 3.5), Material(W.RoyalBlue), 0.7),
        Metal(nid(), Sphere((-1.5, -4.5, -7.0), 1.5), Material(W.White), 0.0),
        Lambertian(nid(), Sphere((2.0, -4.5, -7.0), 1.5), Material(W.Red)),
        Lambertian(nid(), Sphere((6.0, -4.5, -7.0), 1.5), Material(W.DarkGreen)),
        Dielectric(nid(), Sphere((2.5, -3.0, -3.0), 1.5), Material(W.White), 1.5),
        Dielectric(nid(), Sphere((-5.1, 2.0, -11.0), 5.0), Material(W.White), 1.5),
    ]
    d["spheres_submission"] = [
        Lambertian(nid(), Sphere((4.0, -1.0, -15.0), 3.5), Material(W.RoyalBlue)),
        Metal(nid(), Sphere((-1.5, -4.5, -7.0), 1.5), Material(W.White), 0.0),
        Lambertian(nid(), Sphere((2.0, -4.5, -7.0), 1.5), Material(W.Red)),
        Lambertian(nid(), Sphere((6.0, -4.5, -7.0), 1.5), Material(W.DarkGreen)),
        Dielectric(nid(), Sphere((2.5, -3.0, -3.0), 1.5), Material(W.White), 1.5),
        Dielectric(nid(), Sphere((-5.1, 2.0, -11.0), 5.0), Material(W.White), 1.5),
    ]
    d["spheres_simple"] = [Lambertian(nid(), Sphere((4.0, -1.0, -15.0), 3.5), Material(W.RoyalBlue))]
    d["spheres_caustics"] = [Dielectric(nid(), Sphere((-5.1, -2.0, -11.0), 3.0), Material(W.White), 1.5)]
    d["spheres_scene_2"] = [
        Lambertian(nid(), Sphere((4.0, -1.0, -15.0), 3.5), Material(W.RoyalBlue)),
        Lambertian(nid(), Sphere((2.0, -4.5, -7.0), 1.5), Material(W.Red)),
        Lambertian(nid(), Sphere((6.0, -4.5, -7.0), 1.5), Material(W.DarkGreen)),
        Metal(nid(), Sphere((-1.5, -4.5, -7.0), 1.5), Material(W.White), 0.0),
        Dielectric(nid(), Sphere((-5.1, -2.0, -11.0), 3.0), Material(W.White), 1.5),
        Dielectric(nid(), Sphere((2.5, -3.0, -3.0), 1.5), Material(W.White), 1.5),
    ]

    def plane_list(order, fuzz):
        back = lambda: Lambertian(nid(), Plane(NumericsPlane(SurfaceNormal(0.0, 0.0, -1.0), 17.0), (0.0, 0.0, 17.0), 30.0, 10.0), Material(W.White))
        red = lambda: Lambertian(nid(), Plane(NumericsPlane(SurfaceNormal(0.0, 0.0, 1.0), 30.0), (-5.0, 10.0, -30.0), 25.0, 20.0), Material(W.IndianRed))
        mirror = lambda: Metal(nid(), Plane(NumericsPlane(SurfaceNormal(-1.0, 0.0, 1.0), 15.0), (15.0, 3.0, -15.0), 10.0, 6.0), Material(W.AntiqueWhite), fuzz)
        fl = lambda: Lambertian(nid(), floor(), Material(vec3(1.0, 1.0, 1.0)))
        return [{"floor": fl, "back": back, "red": red, "mirror": mirror}[k]() for k in order]

    d["planes"] = plane_list(["floor", "back", "red", "mirror"], 0.01)
    d["planes_submission"] = plane_list(["floor", "back", "red", "mirror"], 0.0)
    d["planes_simple"] = plane_list(["floor"], 0.0)
    d["planes_scene_2"] = plane_list(["floor", "back", "mirror", "red"], 0.0)
    d["scene_spheres"] = d["spheres_scene_2"] + d["planes_scene_2"] + d["lights"]   # SceneDefinitions.fs:119
    return d


class SceneDefinitions:
    """Namespace holding the lists of SceneDefinitions.fs (built once, like F# module init)."""

    _lists = _build_scene_definitions()
    lights = _lists["lights"]
    lights_simple = _lists["lights_simple"]
    spheres = _lists["spheres"]
    spheres_submission = _lists["spheres_submission"]
    spheres_simple = _lists["spheres_simple"]
    spheres_caustics = _lists["spheres_caustics"]
    spheres_scene_2 = _lists["spheres_scene_2"]
    planes = _lists["planes"]
    planes_submission = _lists["planes_submission"]
    planes_simple = _lists["planes_simple"]
    planes_scene_2 = _lists["planes_scene_2"]
    scene_spheres = _lists["scene_spheres"]


# ---------------------------------------------------------------------------------------------
# Scene.fs:16-43 — the constants renderScene closes over, as ABI structs
# ---------------------------------------------------------------------------------------------
def make_camera(origin=(-3.0, 6.0, 15.0), look_at=(-1.0, 1.0, -10.0), up=(0.0, 1.0, 0.0), fov=None):
    """Scene.fs:36-43: cameraWS = CameraToWorld (WorldToCamera origin lookAt UnitY); fov = PI/4."""
    cam = abi.rt_camera()
    cw = CameraToWorld(WorldToCamera(origin, look_at, up))
    cam.cam_to_world[:] = [float(v) for v in cw.reshape(16)]
    cam.look_at[:] = [float(f32(v)) for v in look_at]
    cam.fov = float(f32(math.pi) / f32(4.0)) if fov is None else float(f32(fov))
    return cam


def make_params(width=800, height=640, spp=1, max_trace_depth=5, background=(0.0, 0.0, 0.0), jitter=1,
                seed=1, lambertian_samples=0, x_begin=0, x_end=None, x_step=1, waves_per_cu=0, pool_blocks=0,
                estimator=0, bvh_ring=0, bvh_min_active=0, two_pass=0):
    """Scene.fs:18-20,31-32 defaults."""
    p = abi.rt_params()
    p.width, p.height, p.spp, p.max_trace_depth = width, height, spp, max_trace_depth
    p.background[:] = list(background)
    p.jitter, p.seed, p.lambertian_samples = jitter, seed, lambertian_samples
    p.x_begin, p.x_end, p.x_step = x_begin, width if x_end is None else x_end, x_step
    p.waves_per_cu, p.pool_blocks = waves_per_cu, pool_blocks
    p.estimator, p.bvh_ring, p.bvh_min_active, p.two_pass = estimator, bvh_ring, bvh_min_active, two_pass
    return p


def num_columns(p):
    return (p.x_end - p.x_begin + p.x_step - 1) // p.x_step
