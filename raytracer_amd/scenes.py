"""Synthetic large scenes of BASELINE.json configs 3-5: a ~100k-triangle mesh and
the many-sphere stress scene.  The reference has neither (its only geometry is the dozen surfaces of
SceneDefinitions.fs); these are generated from fixed formulas and seeds as flat `rt_surface` records
(numpy view `abi.surface_dtype`), together with the reference's floor plane and lights so that the
shading path is the reference's.
"""
import numpy as np

from . import abi, host


def _records(n):
    return np.zeros(n, dtype=abi.surface_dtype)


def reference_environment(first_id):
    """Floor (scene_spheres idx 6) and the two emissive planes (idx 10, 11) of SceneDefinitions.fs, as records."""
    sd = host.SceneDefinitions
    env = [sd.planes_scene_2[0]] + sd.lights
    arr = host.flatten_surfaces(env)
    out = _records(len(env))
    for i in range(len(env)):
        out[i] = np.frombuffer(bytes(arr[i]), dtype=abi.surface_dtype)[0]
        out[i]["id"] = first_id + i
    return out


def heightfield_mesh(n=224, seed=7, extent=(-12.0, 12.0, -26.0, -2.0), base=-5.5, amplitude=2.5, environment=True):
    """n x n grid of quads over the floor, two Lambertian triangles each: 2 n^2 triangles (n = 224 gives
    100352).  Height = base + amplitude * smooth value noise from a seeded lattice; colours vary with height."""
    rng = np.random.default_rng(seed)
    lat = rng.random((9, 9))
    u = np.linspace(0, 8, n + 1)
    i0 = np.minimum(u.astype(int), 7)
    f = u - i0
    w = f * f * (3 - 2 * f)
    rows = lat[i0][:, i0] * np.outer(1 - w, 1 - w) + lat[i0 + 1][:, i0] * np.outer(w, 1 - w) \
        + lat[i0][:, i0 + 1] * np.outer(1 - w, w) + lat[i0 + 1][:, i0 + 1] * np.outer(w, w)
    x0, x1, z0, z1 = extent
    xs = np.linspace(x0, x1, n + 1, dtype=np.float32)
    zs = np.linspace(z0, z1, n + 1, dtype=np.float32)
    ys = (base + amplitude * rows).astype(np.float32)            # [ix, iz]
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    P = np.stack([X, ys, Z], axis=-1).astype(np.float32)         # [n+1, n+1, 3]
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    # counter-clockwise seen from above => normals point up
    t1 = np.concatenate([a, c, b], axis=-1).reshape(-1, 9)
    t2 = np.concatenate([a, d, c], axis=-1).reshape(-1, 9)
    tris = np.empty((2 * n * n, 9), dtype=np.float32)
    tris[0::2], tris[1::2] = t1, t2
    out = _records(len(tris))
    out["kind"], out["geom"] = abi.KIND_LAMBERTIAN, abi.GEOM_TRIANGLE
    out["id"] = 1000 + np.arange(len(tris), dtype=np.uint64)
    out["g"][:, :9] = tris
    h = ((tris[:, 1] - base) / amplitude).clip(0, 1)
    out["albedo"] = np.stack([0.35 + 0.5 * h, 0.55 + 0.25 * h, 0.3 + 0.2 * (1 - h)], axis=-1).astype(np.float32)
    if environment:
        out = np.concatenate([out, reference_environment(1)])
    return out


def many_spheres(n=1_000_000, seed=42, environment=True):
    """config 5: centres uniform in [-500,500] x [0,50] x [-1000,0], radius U[0.2,1], 60 % Lambertian, 25 % Metal
    (fuzz U[0,0.5]), 15 % Dielectric (ior 1.5), albedo U[0.2,0.9]^3; plus the reference's floor and lights."""
    rng = np.random.default_rng(seed)
    out = _records(n)
    out["geom"] = abi.GEOM_SPHERE
    out["id"] = 1000 + np.arange(n, dtype=np.uint64)
    c = rng.random((n, 3))
    out["g"][:, 0] = (-500 + 1000 * c[:, 0]).astype(np.float32)
    out["g"][:, 1] = (50 * c[:, 1]).astype(np.float32)
    out["g"][:, 2] = (-1000 * c[:, 2]).astype(np.float32)
    out["g"][:, 3] = (0.2 + 0.8 * rng.random(n)).astype(np.float32)
    k = rng.random(n)
    out["kind"] = np.where(k < 0.6, abi.KIND_LAMBERTIAN, np.where(k < 0.85, abi.KIND_METAL, abi.KIND_DIELECTRIC))
    out["fuzz"] = np.where(out["kind"] == abi.KIND_METAL, 0.5 * rng.random(n), 0).astype(np.float32)
    out["ior"] = np.where(out["kind"] == abi.KIND_DIELECTRIC, 1.5, 0).astype(np.float32)
    out["albedo"] = (0.2 + 0.7 * rng.random((n, 3))).astype(np.float32)
    if environment:
        out = np.concatenate([out, reference_environment(1)])
    return out


# camera of config 5: above the sphere field looking along -z
MANY_SPHERES_CAMERA = dict(cameraOriginWS=(0.0, 80.0, 60.0), lookAt=(0.0, 10.0, -400.0))
