"""Host-side mirror of the reference's scene API: IDs, constants and camera of SceneDefinitions.fs /
Scene.fs / Camera.fs, and the struct layouts of include/rt_abi.h."""
import ctypes as C

import numpy as np

import oracle_lib as O
from raytracer_amd import abi, host

SD = host.SceneDefinitions


def test_scene_spheres_ids_and_order():
    """module-initialisation order of assignIDAndIncrement (SceneDefinitions.fs:16-22)."""
    s = SD.scene_spheres
    assert [x.ID for x in s] == [20, 21, 22, 23, 24, 25, 35, 36, 37, 38, 1, 2]
    kinds = [x.kind for x in s]
    assert kinds == [1, 1, 1, 2, 3, 3, 1, 1, 2, 1, 0, 0]
    assert [type(x.Geometry).__name__ for x in s] == ["Sphere"] * 6 + ["Plane"] * 6
    assert s[6].Geometry.Center is None and all(x.Geometry.Center is not None for x in s[7:])
    assert [x.ID for x in SD.lights_simple] == [3, 4] and [x.ID for x in SD.spheres] == list(range(5, 12))
    assert [x.ID for x in SD.planes] == [26, 27, 28, 29] and [x.ID for x in SD.planes_simple] == [34]


def test_scene_constants():
    s = SD.scene_spheres
    assert np.allclose(s[0].Geometry.Center, (4, -1, -15)) and s[0].Geometry.Radius == 3.5
    assert np.allclose(s[0].Material.Albedo, np.array([65, 105, 225]) / 255, atol=1e-7)
    assert s[4].ior == 1.5 and s[3].fuzz == 0.0
    floor = s[6].Geometry.Plane
    assert np.allclose(floor.Normal, (0, 1, 0)) and floor.D == 6.0       # y = -6 from the three vertices
    light = s[10]
    assert np.allclose(light.Material.Emmitance, 1.0) and np.allclose(light.Material.Albedo, 0.0)
    n = host.SurfaceNormal(-1.0, -1.0, 1.0)
    assert np.allclose(light.Geometry.Center, n * np.float32(-20.0))
    assert light.Geometry.Width == 40.0 and light.Geometry.Height == 13.0


def test_camera_matches_oracle_lookat_and_invert():
    """Camera.fs:41-45 = Matrix4x4.CreateLookAt + Matrix4x4.Invert, bit for bit against the oracle's copy."""
    lib = O.load()
    pos, tgt, up = (-3.0, 6.0, 15.0), (-1.0, 1.0, -10.0), (0.0, 1.0, 0.0)
    view = (C.c_float * 16)()
    lib.rto_create_look_at(O.f3(pos), O.f3(tgt), O.f3(up), view)
    inv = (C.c_float * 16)()
    assert lib.rto_invert(view, inv) == 1
    assert np.array_equal(host.WorldToCamera(pos, tgt, up).reshape(16), np.array(view[:], dtype=np.float32))
    cam = host.make_camera()
    assert np.array_equal(np.array(cam.cam_to_world[:], dtype=np.float32), np.array(inv[:], dtype=np.float32))
    assert np.allclose(np.array(cam.cam_to_world[12:15]), pos, atol=1e-5)       # translation = eye
    assert abs(cam.fov - np.pi / 4) < 1e-7


def test_flatten_round_trip():
    arr = host.flatten_surfaces(SD.scene_spheres)
    assert arr[3].kind == abi.KIND_METAL and arr[3].geom == abi.GEOM_SPHERE and arr[3].id == 23
    assert arr[7].geom == abi.GEOM_PLANE and arr[7].g[4] == 1.0 and list(arr[7].g[5:10]) == [0.0, 0.0, 17.0, 30.0, 10.0]
    assert arr[6].g[4] == 0.0
