"""GPU parity: the HIP render loop (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): hit mask and object IDs bit-exact; RGB and depth within 1e-4 per
pixel.  Because both sides use the same counter-based RNG keyed by (pixel-sample, tree node), the
stochastic (jitter / multi-sample) renders are held to the same bar, and the number of
closest-hit queries must agree exactly (identical ray trees).
"""
import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import host
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu

TOL = 1e-4   # north_star: RGB / depth within 1e-4 per pixel
SD = host.SceneDefinitions


def _compare(scene, nthreads=0):
    scene.renderScene()
    ref = O.render(scene.surfaces, scene.camera, scene.params(), nthreads=nthreads)
    assert np.array_equal(scene.objectIds, ref["ids"]), "object IDs differ"
    assert np.array_equal(scene.depthBuffer < 1e30, ref["depth"] < 1e30), "hit mask differs"
    hit = ref["depth"] < 1e30
    assert np.max(np.abs(scene.depthBuffer[hit] - ref["depth"][hit]), initial=0.0) <= TOL
    assert np.array_equal(scene.depthBuffer[~hit], ref["depth"][~hit])
    assert scene.rayCount == ref["rays"], (scene.rayCount, ref["rays"])
    assert scene.maxFrameBufferDepth == ref["max_depth"]
    err = np.max(np.abs(scene.frameBuffer - ref["rgba"]))
    assert err <= TOL, f"max |rgba - oracle| = {err}"
    return err


def test_c1_deterministic_640x480():
    """configs[0]: 640x480, 1 spp, pinhole, maxTraceDepth 5."""
    _compare(Scene(width=640, height=480, samplesPerPixel=1, maxTraceDepth=5, jitter=0))


def test_source_resolution_800x640():
    """The checked-in constants of Scene.fs:18-31."""
    _compare(Scene(width=800, height=640, samplesPerPixel=1, maxTraceDepth=5, jitter=0))


def test_c2_stochastic_16spp_depth4():
    """configs[1] at reduced resolution: 16 spp, 4 bounces, jitter on."""
    _compare(Scene(width=200, height=160, samplesPerPixel=16, maxTraceDepth=4, jitter=1, seed=1))


@pytest.mark.parametrize("spp", [1, 2, 3, 5, 64, 100])
def test_sample_counts(spp):
    _compare(Scene(width=40, height=30, samplesPerPixel=spp, maxTraceDepth=3, jitter=1, seed=7))


@pytest.mark.parametrize("depth", [0, 1, 2, 7])
def test_trace_depths(depth):
    _compare(Scene(width=96, height=64, samplesPerPixel=1, maxTraceDepth=depth, jitter=1, seed=3))


@pytest.mark.parametrize("blocks", [2, 4, 6])
def test_pool_blocks_spill(blocks):
    """Small LDS pools force the spill-to-HBM path of the record pool; results must not change."""
    _compare(Scene(width=160, height=120, samplesPerPixel=2, maxTraceDepth=5, jitter=1, seed=5, pool_blocks=blocks))


@pytest.mark.parametrize("waves", [1, 3])
def test_waves_per_cu(waves):
    _compare(Scene(width=100, height=75, samplesPerPixel=2, maxTraceDepth=4, jitter=1, seed=5, waves_per_cu=waves))


@pytest.mark.parametrize("lamb", [1, 2, 3])
def test_lambertian_sample_count(lamb):
    _compare(Scene(width=96, height=64, samplesPerPixel=4, maxTraceDepth=5, jitter=1, seed=11, lambertian_samples=lamb))


def test_nonzero_background():
    _compare(Scene(width=96, height=64, samplesPerPixel=2, maxTraceDepth=2, jitter=1, seed=2,
                   backgroundColor=(0.25, 0.5, 0.75)))


def test_deep_trace_node64():
    """max_trace_depth > 13 switches the tree-node index to 64 bits; fan-out 1 keeps it tractable."""
    _compare(Scene(width=64, height=48, samplesPerPixel=2, maxTraceDepth=16, jitter=1, seed=4, lambertian_samples=1))


@pytest.mark.parametrize("name", ["spheres", "spheres_submission", "spheres_simple", "spheres_caustics"])
def test_alternate_scene_lists(name):
    """SceneDefinitions.fs:32-110: Metal fuzz 0.7 / 0.01, ior 1.3, sphere lights."""
    surfaces = getattr(SD, name) + SD.planes + SD.lights_simple
    _compare(Scene(surfaces=surfaces, width=120, height=96, samplesPerPixel=2, maxTraceDepth=4, jitter=1, seed=9))


def test_empty_scene():
    s = Scene(surfaces=[], width=16, height=8, backgroundColor=(0.5, 0.25, 1.0))
    _compare(s)
    assert s.rayCount == 16 * 8 and np.all(s.objectIds == 0)


def test_column_partition_matches_full_frame():
    """The multi-GPU partition: interleaved columns rendered separately equal the full frame bit for bit
    (radiance is summed in fixed point, so the result does not depend on which wave traced what)."""
    _partition_check(4, 0.0)
    _partition_check(64, 0.0)


def _partition_check(spp, tol):
    s = Scene(width=96 if spp > 4 else 160, height=64 if spp > 4 else 120, samplesPerPixel=spp, maxTraceDepth=4,
              jitter=1, seed=6)
    s.renderScene()
    total = 0
    for r in range(3):
        rgba, depth, ids, md, rays = s.renderColumns(r, s.width, 3)
        assert np.max(np.abs(rgba - s.frameBuffer[r::3])) <= tol
        assert np.array_equal(depth, s.depthBuffer[r::3])
        assert np.array_equal(ids, s.objectIds[r::3])
        total += rays
    assert total == s.rayCount


def test_run_to_run_determinism():
    s = Scene(width=320, height=240, samplesPerPixel=4, maxTraceDepth=5, jitter=1, seed=8)
    s.renderScene()
    a = s.frameBuffer.copy()
    s.renderScene()
    assert np.array_equal(a, s.frameBuffer)


def test_find_closest_matches_oracle():
    """findClosestIntersection (Surface.fs:49-58) on random rays, incl. rays from inside spheres."""
    rng = np.random.default_rng(0)
    n = 200000
    o = rng.uniform(-12, 12, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    s = Scene()
    t, idx = s.findClosestIntersection(o, d)
    rt, ridx = O.find_closest(s.surfaces, o, d)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(t, rt)     # bit-exact ray parameters
