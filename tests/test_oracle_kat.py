"""Known-answer tests that pin the CPU oracle (oracle/rt_oracle.c) function by function.

The reference ships no tests, fixtures or golden vectors for this path, so the
oracle's parity with the F# is "unpinned" by reference data; these tests pin it against hand-derived
values of the reference's formulas and against the published known-answer
vectors of the Philox generator that replaces the reference's (missing) Henzai.Core RNG.
"""
import ctypes as C
import math

import numpy as np

import oracle_lib as O
from raytracer_amd import host
from raytracer_amd.host import Dielectric, Lambertian, Material, Metal, NoSurface, NumericsPlane, Plane, Rgba32, Sphere

f32 = np.float32


def test_philox4x32_10_random123_kat():
    """kat_vectors of Random123 1.09 for philox4x32-10 (Salmon et al., SC'11)."""
    lib = O.load()
    kats = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kats:
        out = (C.c_uint32 * 4)()
        lib.rto_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert list(out) == want


def test_round_is_midpoint_to_even():
    """Numerics.fs:9 -> MathF.Round(x, digits)."""
    lib = O.load()
    assert lib.rto_round(0.0005, 3) == 0.0          # 0.5 -> 0 (even)
    assert lib.rto_round(0.0015, 3) == f32(0.002)   # 1.5 -> 2
    assert lib.rto_round(0.0025, 3) == f32(0.002)   # 2.5 -> 2 (0.0025f is just below the midpoint anyway)
    assert lib.rto_round(1.000004, 5) == 1.0
    assert lib.rto_round(-0.0004, 3) == 0.0
    assert lib.rto_round(3e9, 3) == f32(3e9)        # |x| >= 1e8 is returned unchanged


def test_sphere_hit_miss_tangent_inside():
    """Sphere.fs:18-43: ray (0,0,0)->(0,0,-1) vs sphere (0,0,-5) r=1 => t = 4; tangent => miss by
    Round(disc,3) = 0; from inside => the far root."""
    s = Lambertian(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.White))
    has, t, ok = O.intersect(s, (0, 0, 0), (0, 0, -1))
    assert has and ok and t == 4.0
    has, t, ok = O.intersect(s, (1.0, 0, 0), (0, 0, -1))       # grazing: disc = 0
    assert not has and not ok
    has, t, ok = O.intersect(s, (0, 0, -5.0), (0, 0, -1))      # origin at the centre: roots -1, +1
    assert has and ok and t == 1.0
    has, t, ok = O.intersect(s, (0, 0, -7.0), (0, 0, -1))      # sphere behind the ray
    assert has and not ok                                      # t = -1 fails t > TMin
    has, t, ok = O.intersect(s, (3.0, 0, 0), (0, 0, -1))       # clean miss: disc < 0
    assert not has and not ok


def test_plane_tmin_tmax_and_dotview():
    """Plane.fs:36-49: t = (-D - N.O)/(N.d); |N.d| < TMin => no hit; accept t in (TMin, TMax/dotView]."""
    floor = Lambertian(1, Plane(NumericsPlane((0.0, 1.0, 0.0), 6.0)), Material(Rgba32.White))   # y = -6
    has, t, ok = O.intersect(floor, (0, 0, 0), (0, -1, 0))
    assert has and ok and t == 6.0
    has, t, ok = O.intersect(floor, (0, 0, 0), (1, 0, 0))      # parallel
    assert not has and not ok
    has, t, ok = O.intersect(floor, (0, 0, 0), (0, 1, 0))      # pointing away: t = -6
    assert has and not ok
    d = np.array([math.sqrt(1 - 0.01 ** 2), -0.01, 0.0], dtype=np.float32)   # t = 600 > TMax = 500
    has, t, ok = O.intersect(floor, (0, 0, 0), d)
    assert has and not ok and abs(t - 600.0) < 1e-2
    has, t, ok = O.intersect(floor, (0, 0, 0), d, dot_view=0.5)                # TMax/dotView = 1000
    assert has and ok


def test_bounded_plane_rectangle_edges():
    """Plane.fs:19-34 on scene plane idx7: N=(0,0,-1), D=17 (z = 17), centre (0,0,17), 30 x 10."""
    back = host.SceneDefinitions.planes_scene_2[1]
    o = np.array([0.0, 0.0, 0.0], dtype=np.float32)

    def hit(x, y):
        d = np.array([x, y, 17.0], dtype=np.float64)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        return O.intersect(back, o, d)[2]

    assert hit(0.0, 0.0)
    assert hit(14.9, 4.9) and hit(-14.9, -4.9)
    assert not hit(15.1, 0.0) and not hit(-15.1, 0.0)
    assert not hit(0.0, 5.1) and not hit(0.0, -5.1)


def test_plane_rotation_is_the_reference_formula():
    """Numerics.fs:63-76 uses cos(|a x b|) (= cos(sin theta)), not cos(theta): R is not orthonormal."""
    lib = O.load()
    n = host.SurfaceNormal(-1.0, 0.0, 1.0)
    R = (C.c_float * 16)()
    lib.rto_plane_rotation(O.f3(n), R)
    R = np.array(R[:], dtype=np.float64).reshape(4, 4)
    b = np.array([0.0, 0.0, -1.0])
    w = np.cross(np.array(n, dtype=np.float64), b)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    want = np.eye(3) + W + W @ W / (1.0 + math.cos(np.linalg.norm(w)))
    assert np.allclose(R[:3, :3], want, atol=1e-6)
    assert not np.allclose(R[:3, :3] @ R[:3, :3].T, np.eye(3), atol=1e-3)


def test_dielectric_normal_incidence():
    """Surface.fs:119-181 at normal incidence, ior 1.5: R0 = 0.04 = F, two samples with shading
    2*albedo*F = 0.08 and 2*albedo*(1-F) = 1.92, reflection straight back, refraction straight on."""
    s = Dielectric(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.White), 1.5)
    n, dirs, sh = O.generate_samples(s, (0, 0, 0), (0, 0, -1), 4.0)
    assert n == 2
    assert np.allclose(dirs[0], (0, 0, 1), atol=1e-6) and np.allclose(dirs[1], (0, 0, -1), atol=1e-6)
    assert np.allclose(sh[0], 0.08, atol=1e-6) and np.allclose(sh[1], 1.92, atol=1e-6)


def test_dielectric_total_internal_reflection_is_one_sample():
    """Inside glass at a grazing angle: Refract fails (Surface.fs:124-130), F = 1, Round(F,3) = 1 => 1 sample."""
    s = Dielectric(1, Sphere((0.0, 0.0, 0.0), 1.0), Material(Rgba32.White), 1.5)
    o = np.array([0.0, 0.9, 0.0], dtype=np.float32)
    d = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    has, t, ok = O.intersect(s, o, d)
    assert ok
    n, dirs, sh = O.generate_samples(s, o, d, t)
    assert n == 1 and np.allclose(sh[0], 1.0)


def test_lambertian_four_samples_weight_2_albedo_cos():
    """Surface.fs:60-83: 4 samples, shading = (albedo/pi) * cos / (1/(2 pi)) = 2 albedo cos, directions in
    the hemisphere of the normal, unit length."""
    alb = host.Rgba32.to_vector3(Rgba32.RoyalBlue)
    s = Lambertian(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.RoyalBlue))
    n, dirs, sh = O.generate_samples(s, (0, 0, 0), (0, 0, -1), 4.0, seed=3, ps=17, node=0)
    assert n == 4
    normal = np.array([0, 0, 1.0])
    for i in range(4):
        cos = float(dirs[i] @ normal)
        assert cos >= -1e-6 and abs(np.linalg.norm(dirs[i]) - 1) < 1e-6
        assert np.allclose(sh[i], 2 * alb * cos, atol=2e-6)
    assert len({tuple(d) for d in dirs.round(5).tolist()}) == 4     # distinct draws per child node


def test_metal_fuzz_zero_is_a_mirror_and_clamps_fuzz():
    """Surface.fs:85-109: fuzz 0 => out = reflect(d, n); shading = albedo; fuzz is clamped to [0,1]."""
    s = Metal(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.AntiqueWhite), 0.0)
    d = np.array([0.6, 0.0, -0.8], dtype=np.float32)
    o = np.array([-3.0, 0.0, 0.0], dtype=np.float32)          # hits the sphere front at (0,0,-4)
    has, t, ok = O.intersect(s, o, d)
    assert ok and abs(t - 5.0) < 1e-5
    n, dirs, sh = O.generate_samples(s, o, d, t)
    assert n == 1 and np.allclose(dirs[0], (0.6, 0.0, 0.8), atol=1e-5)
    assert np.allclose(sh[0], host.Rgba32.to_vector3(Rgba32.AntiqueWhite))
    s5 = Metal(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.White), 5.0)
    s1 = Metal(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.White), 1.0)
    assert np.array_equal(O.generate_samples(s5, o, d, t, seed=9)[1], O.generate_samples(s1, o, d, t, seed=9)[1])


def test_nosurface_has_no_samples():
    s = NoSurface(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.White, 0.0, Rgba32.White, 1.0))
    assert O.generate_samples(s, (0, 0, 0), (0, 0, -1), 4.0)[0] == 0


def test_hemisphere_sampler_is_uniform_y_up():
    """RandomInUnitHemisphere: unit vectors, y >= 0, E[y] = 1/2 and E[x] = E[z] = 0 for the uniform pdf 1/(2 pi)."""
    lib = O.load()
    out = (C.c_float * 3)()
    v = np.empty((20000, 3))
    for i in range(len(v)):
        lib.rto_hemisphere_sample(5, i, 1 + (i % 4), out)
        v[i] = out[:]
    assert np.all(v[:, 1] >= 0) and np.allclose(np.linalg.norm(v, axis=1), 1, atol=1e-6)
    assert abs(v[:, 1].mean() - 0.5) < 0.01 and abs(v[:, 0].mean()) < 0.01 and abs(v[:, 2].mean()) < 0.01
    assert abs((v[:, 1] ** 2).mean() - 1 / 3) < 0.01


def test_sincos_2pi_accuracy():
    lib = O.load()
    s, c = C.c_float(), C.c_float()
    for u in np.linspace(0, 1, 4097)[:-1]:
        lib.rto_sincos_2pi(float(f32(u)), C.byref(s), C.byref(c))
        assert abs(s.value - math.sin(2 * math.pi * float(f32(u)))) < 3e-7
        assert abs(c.value - math.cos(2 * math.pi * float(f32(u)))) < 3e-7


def test_frame_is_orthonormal():
    lib = O.load()
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        nt, nb = (C.c_float * 3)(), (C.c_float * 3)()
        lib.rto_frame_around(O.f3(n), nt, nb)
        nt, nb = np.array(nt[:]), np.array(nb[:])
        assert abs(nt @ n) < 1e-6 and abs(nb @ n) < 1e-6 and abs(nt @ nb) < 1e-6
        assert abs(np.linalg.norm(nt) - 1) < 1e-6 and abs(np.linalg.norm(nb) - 1) < 1e-6


def test_pow5_rounds_once():
    lib = O.load()
    for x in (0.0, 0.25, 0.3333333, 0.96, 1.0):
        assert lib.rto_pow5(x) == f32(float(f32(x)) ** 5)


def test_depth_accounting_matches_the_recursion():
    """Scene.fs:51-72: with maxTraceDepth = d rays are intersected at levels 0..d+1 and never deeper.
    Camera inside a Lambertian sphere: every primary hits (1 ray) and scatters 4 children outward, which
    all miss (sphere normals point outward, Sphere.fs:37): 5 rays per pixel whatever d is.  On the floor of
    scene_spheres the count is bounded by the full 4-ary tree and grows with d."""
    W = Rgba32.White
    ball = [Lambertian(1, Sphere((0.0, 0.0, 0.0), 50.0), Material(W))]
    cam = host.make_camera(origin=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0))
    for d in (0, 1, 3):
        r = O.render(ball, cam, host.make_params(4, 4, 1, d, jitter=0), nthreads=1)
        assert r["rays"] == 16 * 5
    prev = 0
    for d in (0, 1, 2, 3):
        r = O.render(host.SceneDefinitions.scene_spheres, host.make_camera(), host.make_params(16, 12, 1, d, jitter=0), nthreads=1)
        assert prev <= r["rays"] <= 16 * 12 * sum(4 ** k for k in range(d + 2))
        prev = r["rays"]
    assert prev > 16 * 12 * 5


def test_triangle_known_answers():
    """The build's own triangle Hitable (no reference counterpart; include/rt_abi.h RT_GEOM_TRIANGLE): Moeller-Trumbore
    with fused multiply-adds.  Hand-derived answers on the unit right triangle in the plane z = -2."""
    tri = host.Lambertian(7, host.Triangle((0.0, 0.0, -2.0), (1.0, 0.0, -2.0), (0.0, 1.0, -2.0)), host.Material(host.Rgba32.Red))
    down = (0.0, 0.0, -1.0)
    has, t, acc = O.intersect(tri, (0.25, 0.25, 0.0), down)          # interior: t = 2 exactly
    assert has and acc and t == 2.0
    has, t, acc = O.intersect(tri, (0.25, 0.25, -4.0), (0.0, 0.0, 1.0))   # from behind: no back-face culling
    assert has and acc and t == 2.0
    assert O.intersect(tri, (0.5, 0.5, 0.0), down)[0]               # on the hypotenuse, u + v = 1: inside (<=)
    assert O.intersect(tri, (0.0, 0.0, 0.0), down)[0]               # on the vertex v0, u = v = 0: inside (>=)
    assert not O.intersect(tri, (0.51, 0.5, 0.0), down)[0]          # just outside the hypotenuse
    assert not O.intersect(tri, (-0.01, 0.5, 0.0), down)[0]         # u < 0
    assert not O.intersect(tri, (0.25, 0.25, 0.0), (1.0, 0.0, 0.0))[0]   # parallel to the plane: |det| < 1e-12
    has, t, acc = O.intersect(tri, (0.25, 0.25, -2.0005), (0.0, 0.0, 1.0))   # hit closer than TMin = 0.001
    assert has and not acc and abs(t - 0.0005) < 1e-6
    has, t, acc = O.intersect(tri, (0.25, 0.25, 0.0), (0.0, 0.0, 1.0))   # pointing away: t = -2, not acceptable
    assert has and not acc and t == -2.0


# ---- single-path estimator (Dielectric.Scatter, Surface.fs:155-166; notes.txt:21-54) ------------
def test_single_path_dielectric_scatter_picks_reflection_with_probability_F():
    """Normal incidence, ior 1.5: reflectProb = 0.04.  Scatter reflects when RandomFloat <= reflectProb, else
    refracts; either way shading = BRDF * (1/PDF) = Albedo (Surface.fs:28,35-36,162,166)."""
    lib = O.load()
    s = Dielectric(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.AntiqueWhite), 1.5)
    alb = host.Rgba32.to_vector3(Rgba32.AntiqueWhite)
    nrefl = 0
    trials = 20000
    for ps in range(trials):
        n, d, sh = O.generate_samples_single_path(s, (0, 0, 0), (0, 0, -1), 4.0, seed=11, ps=ps, node=0)
        assert n == 1 and np.array_equal(sh, alb)
        u = lib.rto_random_float(11, ps, 1)                   # the draw is keyed by the child node 4*0+1
        assert 0.0 <= u < 1.0
        if u <= np.float32(0.04):
            nrefl += 1
            assert np.allclose(d, (0, 0, 1), atol=1e-6)
        else:
            assert np.allclose(d, (0, 0, -1), atol=1e-6)
    assert abs(nrefl / trials - 0.04) < 0.005


def test_single_path_total_internal_reflection_never_refracts():
    s = Dielectric(1, Sphere((0.0, 0.0, 0.0), 1.0), Material(Rgba32.White), 1.5)
    o = np.array([0.0, 0.9, 0.0], dtype=np.float32)
    d = np.array([1.0, 0.0, 0.0], dtype=np.float32)
    _, t, ok = O.intersect(s, o, d)
    assert ok
    _, dirs, _ = O.generate_samples(s, o, d, t)               # branching: the single reflected sample
    for ps in range(200):
        n, dr, sh = O.generate_samples_single_path(s, o, d, t, seed=2, ps=ps)
        assert n == 1 and np.array_equal(dr, dirs[0]) and np.allclose(sh, 1.0)


def test_single_path_lambertian_and_metal_are_the_first_branching_sample():
    """One child per hit through the default GenerateSamples with SampleCount 1: child node 4k+1, i.e. exactly sample 0
    of the branching estimator, with the same shading (the division by the sample count happens in rayTrace)."""
    for s in (Lambertian(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.RoyalBlue)),
              Metal(2, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.White), 0.4)):
        n4, dirs, sh4 = O.generate_samples(s, (0, 0, 0), (0, 0, -1), 4.0, seed=3, ps=17, node=5)
        n1, d1, sh1 = O.generate_samples_single_path(s, (0, 0, 0), (0, 0, -1), 4.0, seed=3, ps=17, node=5)
        assert n1 == 1 and n4 >= 1 and np.array_equal(d1, dirs[0]) and np.array_equal(sh1, sh4[0])
    light = NoSurface(1, Sphere((0.0, 0.0, -5.0), 1.0), Material(Rgba32.White, 0.0, Rgba32.White, 1.0))
    assert O.generate_samples_single_path(light, (0, 0, 0), (0, 0, -1), 4.0)[0] == 0


def test_single_path_render_has_one_ray_per_level_and_the_same_mean():
    """A path intersects at most maxTraceDepth + 2 rays; its expectation is the branching estimator's (checked
    loosely on the frame mean here, to 45 dB on the GPU in tests/test_gpu_estimator.py)."""
    cam = host.make_camera()
    sc = host.SceneDefinitions.scene_spheres
    a = O.render(sc, cam, host.make_params(40, 30, 256, 4, jitter=0, seed=5, estimator=1))
    b = O.render(sc, cam, host.make_params(40, 30, 16, 4, jitter=0, seed=6))
    assert a["rays"] <= 40 * 30 * 256 * 6 and np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["depth"], b["depth"])
    la, lb = a["rgba"][:, :, :3].astype(np.float64) ** 2, b["rgba"][:, :, :3].astype(np.float64) ** 2
    assert abs(la.mean() - lb.mean()) < 0.01 * lb.mean() + 1e-3
