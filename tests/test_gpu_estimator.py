"""the single-path estimator (rt_params.estimator = RT_ESTIMATOR_SINGLE_PATH) — one child per hit,
Dielectric through its stochastic Scatter (Surface.fs:155-166), the accumulator form of notes.txt:21-54 — on the GPU
against the oracle's restatement (same bar as the branching estimator: identical ray trees, IDs, depth, ray counts;
RGB within 1e-4), and statistically against the branching estimator it must agree with in expectation."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, host, scenes
from raytracer_amd.scene import Scene
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu
SD = host.SceneDefinitions
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SP = abi.ESTIMATOR_SINGLE_PATH


@pytest.mark.parametrize("depth,spp", [(5, 4), (16, 2), (0, 3)])
def test_single_path_matches_oracle(depth, spp):
    s = Scene(width=120, height=90, samplesPerPixel=spp, maxTraceDepth=depth, jitter=1, seed=5, estimator=SP)
    _compare(s)
    # a path has one ray per level: primary + at most maxTraceDepth + 1 bounces
    assert s.rayCount <= 120 * 90 * spp * (depth + 2)


def test_single_path_alternate_scenes_and_background():
    surfaces = SD.spheres + SD.planes + SD.lights_simple          # ior 1.3, fuzz 0.7 / 0.01, sphere lights
    _compare(Scene(surfaces=surfaces, width=96, height=72, samplesPerPixel=4, maxTraceDepth=6, jitter=1, seed=9,
                   estimator=SP, backgroundColor=(0.2, 0.3, 0.4)))


def test_single_path_ignores_sample_count_overrides():
    surfaces = list(SD.scene_spheres)
    s = Scene(surfaces=surfaces, width=64, height=48, samplesPerPixel=2, maxTraceDepth=4, jitter=1, seed=2, estimator=SP,
              lambertian_samples=3)
    t = Scene(surfaces=surfaces, width=64, height=48, samplesPerPixel=2, maxTraceDepth=4, jitter=1, seed=2, estimator=SP)
    _compare(s)
    t.renderScene()
    assert np.array_equal(s.frameBuffer, t.frameBuffer) and s.rayCount == t.rayCount


def test_single_path_through_the_bvh():
    m = np.concatenate([np.frombuffer(bytes(host.flatten_surfaces(SD.scene_spheres)), dtype=abi.surface_dtype, count=12).copy(),
                        scenes.heightfield_mesh(12, environment=False, base=-5.9, amplitude=1.0)])
    _compare(Scene(surfaces=m, width=120, height=90, samplesPerPixel=3, maxTraceDepth=8, jitter=1, seed=5, estimator=SP))


def test_single_path_converges_to_the_branching_estimator():
    """Both estimators have the same expectation.  Reference: the oracle's BRANCHING render at 32768 spp (committed
    fixture, pinhole, maxTraceDepth 4); GPU: single path, 16384 spp, another seed (measured with the oracle's
    single-path mode: 42.8 dB at 4096 spp, 48.5 dB at 16384)."""
    want = np.load(os.path.join(G, "oracle_ref_40x30_hi.npz"))["rgba"]
    s = Scene(width=40, height=30, samplesPerPixel=16384, maxTraceDepth=4, jitter=0, seed=99, estimator=SP)
    s.renderScene()
    mse = float(np.mean((s.frameBuffer[:, :, :3].astype(np.float64) - want[:, :, :3]) ** 2))
    psnr = 10 * np.log10(1.0 / mse)
    assert psnr >= 45.0, psnr


def test_bad_estimator_is_rejected():
    with pytest.raises(abi.RtError) as e:
        Scene(width=8, height=8, estimator=2).renderScene()
    assert e.value.code == abi.RT_E_INVALID and "estimator" in str(e.value)
