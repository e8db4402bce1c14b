"""Seeded random scenes and parameter combinations against the oracle: sphere counts on both sides of the BVH threshold
(with and without a small mesh; a cluster in front of the reference camera, or a sparse field a kilometre across), every estimator / pass / ring / sample-count / depth / background combination the draw
produces, odd frame sizes, and a random column partition.  Bit-exact object IDs, depth, hit mask and ray count, RGB
within 1e-4 (tests/test_gpu_parity.py::_compare)."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, scenes
from raytracer_amd.scene import Scene
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _random_scene(rng):
    if rng.random() < 0.25:
        # config 5's kilometre-wide sparse field seen from config 5's camera: rays travel hundreds of units, where the
        # scan's sphere test is fuzzy and the tree walks the ray as a cone (rt_bvh.h)
        return scenes.many_spheres(int(rng.choice([70, 300, 2000, 9000])), seed=int(rng.integers(1 << 30))), dict(scenes.MANY_SPHERES_CAMERA)
    n = int(rng.choice([3, 12, 40, 70, 300]))
    s = scenes.many_spheres(n, seed=int(rng.integers(1 << 30)), environment=False)
    # a cluster in front of the reference camera instead of the kilometre-wide field
    c = rng.random((n, 3))
    s["g"][:, 0] = (-9 + 16 * c[:, 0]).astype(np.float32)
    s["g"][:, 1] = (-5.5 + 9 * c[:, 1]).astype(np.float32)
    s["g"][:, 2] = (-24 + 20 * c[:, 2]).astype(np.float32)
    s["g"][:, 3] = (0.3 + 1.2 * rng.random(n)).astype(np.float32)
    if rng.random() < 0.3:
        s["emission"][: max(1, n // 8)] = (2.0 * rng.random((max(1, n // 8), 3))).astype(np.float32)
    parts = [s, scenes.reference_environment(1)]
    if rng.random() < 0.4:
        parts.insert(0, scenes.heightfield_mesh(int(rng.choice([6, 14])), seed=int(rng.integers(1000)), environment=False,
                                                base=-5.9, amplitude=float(rng.uniform(0.3, 1.5))))
    camera = {}
    if rng.random() < 0.2:
        # the same cluster through a long lens from 20 or 60 times as far: primaries start hundreds of units from every
        # primitive (the widened ray for spheres AND triangles), their bounces right next to them, in the same waves
        k = float(rng.choice([20.0, 60.0]))
        o, la = np.array([-3.0, 6.0, 15.0]), np.array([-1.0, 1.0, -10.0])
        camera = dict(cameraOriginWS=tuple(la + (o - la) * k), lookAt=tuple(la), fov=2.0 * np.arctan(np.tan(np.pi / 8.0) / k))
    return np.concatenate(parts), camera


# RT_FUZZ_CASES / RT_FUZZ_BASE in the environment widen the draw for a one-off soak (the committed default stays small)
_CASES = int(os.environ.get("RT_FUZZ_CASES", "40"))
_BASE = int(os.environ.get("RT_FUZZ_BASE", "1000"))


@pytest.mark.parametrize("case", range(_CASES))
def test_random_scene_and_parameters(case):
    rng = np.random.default_rng(_BASE + case)
    surfaces, camera = _random_scene(rng)
    estimator = int(rng.random() < 0.35)
    depth = int(rng.choice([0, 1, 2, 3, 4, 5, 14])) if estimator or rng.random() < 0.5 else int(rng.choice([0, 1, 2, 3, 4]))
    spp = int(rng.choice([1, 2, 3, 9, 16]))
    lamb = int(rng.choice([0, 1, 2, 3]))
    if depth >= 5 and not estimator and lamb != 1:
        spp = min(spp, 2)                      # keep the oracle's branching recursion affordable
    if depth == 14 and not estimator:
        lamb = 1                               # 4^14 rays per sample otherwise
    w, h = int(rng.integers(17, 72)), int(rng.integers(11, 56))
    bg = (0.0, 0.0, 0.0) if rng.random() < 0.6 else tuple(float(x) for x in rng.uniform(0.0, 0.4, 3))
    sc = Scene(surfaces=surfaces, width=w, height=h, samplesPerPixel=spp, maxTraceDepth=depth, jitter=int(rng.random() < 0.7),
               seed=int(rng.integers(1, 1 << 40)), lambertian_samples=lamb, backgroundColor=bg, estimator=estimator,
               two_pass=int(rng.random() < 0.3), bvh_ring=int(rng.choice([0, 0, 16])),
               bvh_min_active=int(rng.choice([0, 0, 7, 40])), pool_blocks=int(rng.choice([0, 0, 3])), **camera)
    if (_BASE + case) % 3 == 0:
        sc.counters(1)                         # the instantiations with every statistic on (rt_scene_counters)
    _compare(sc)
    # a random interleaved column share renders exactly its columns of that frame
    n = int(rng.integers(2, 5)); r = int(rng.integers(0, n))
    full = sc.frameBuffer.copy(); fd = sc.depthBuffer.copy()
    rgba, depth_s, ids, _, _ = sc.renderColumns(r, w, n)
    assert np.array_equal(rgba, full[r::n]) and np.array_equal(depth_s, fd[r::n])
    sc.close()
