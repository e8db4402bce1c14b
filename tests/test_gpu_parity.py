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


def test_c2_full_size_800x640_16spp_depth4():
    """configs[1] at its full size: 274 M rays; the oracle on the box's host cores needs a few seconds."""
    import os
    _compare(Scene(width=800, height=640, samplesPerPixel=16, maxTraceDepth=4, jitter=1, seed=1),
             nthreads=min(16, os.cpu_count() or 1))


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


# ---- committed fixtures (tests/golden, generated by the oracle; see make_golden.py) --------------
import os

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,kw", [
    ("oracle_c1_64x48.npz", dict(width=64, height=48, samplesPerPixel=1, maxTraceDepth=5, jitter=0, seed=1)),
    ("oracle_c2_48x36.npz", dict(width=48, height=36, samplesPerPixel=4, maxTraceDepth=4, jitter=1, seed=7)),
])
def test_gpu_matches_committed_golden(name, kw):
    want = np.load(os.path.join(G, name))
    s = Scene(**kw)
    s.renderScene()
    assert np.array_equal(s.objectIds, want["ids"])
    assert np.array_equal(s.depthBuffer, want["depth"])                 # bit-exact ray parameters
    assert np.max(np.abs(s.frameBuffer - want["rgba"])) <= TOL
    assert s.rayCount == int(want["rays"]) and np.float32(s.maxFrameBufferDepth) == want["max_depth"]


def test_psnr_against_high_spp_reference():
    """north_star: stochastic multi-sample renders reach PSNR >= 40 dB against a high-spp reference.
    Reference = the oracle at 32768 spp (seed 1001, committed); GPU = 4096 spp with another seed, so the
    two share no random numbers and the check is statistical, not a replay (measured: 1024 spp 44 dB,
    16384 spp 54.5 dB).  Both use the pinhole primary ray: the reference draws ONE sub-pixel offset per
    pixel and seed (Scene.fs:98-104, Camera.fs:8-17), not one per sample, so renders with different seeds
    look through slightly different primary rays and differ by up to 1.0 at silhouette pixels whatever the
    sample count (measured 26-30 dB between seeds at 65536 spp with jitter on) -- that is the reference's
    estimator, not noise that more samples remove."""
    want = np.load(os.path.join(G, "oracle_ref_40x30_hi.npz"))["rgba"]
    s = Scene(width=40, height=30, samplesPerPixel=4096, maxTraceDepth=4, jitter=0, seed=77)
    s.renderScene()
    mse = float(np.mean((s.frameBuffer[:, :, :3].astype(np.float64) - want[:, :, :3]) ** 2))
    psnr = 10 * np.log10(1.0 / mse)
    assert psnr >= 40.0, psnr


# ---- BASELINE.json sizes: size-independent properties ------------------------------------------
def test_c2_full_size_partition_and_determinism():
    """configs[1] at full size (800x640, 16 spp, 4 bounces): two interleaved half-frames equal the full
    frame bit for bit, their ray counts add up, and a second render is identical."""
    s = Scene(width=800, height=640, samplesPerPixel=16, maxTraceDepth=4, jitter=1, seed=1)
    s.renderScene()
    full, rays, md = s.frameBuffer.copy(), s.rayCount, s.maxFrameBufferDepth
    tot, mds = 0, []
    for r in range(2):
        rgba, depth, ids, m, n = s.renderColumns(r, s.width, 2)
        assert np.array_equal(rgba, full[r::2]) and np.array_equal(ids, s.objectIds[r::2])
        tot += n
        mds.append(m)
    assert tot == rays and max(mds) == md
    s.renderScene()
    assert np.array_equal(full, s.frameBuffer) and s.rayCount == rays
    assert np.all(s.frameBuffer[:, :, 3] == 1.0) and np.isfinite(s.frameBuffer).all()
    # depth output of configs[1]: hits carry t in (TMin, maxFrameBufferDepth], misses FLT_MAX
    hit = s.objectIds != 0
    assert np.all(s.depthBuffer[hit] > 0.001) and s.depthBuffer[hit].max() == md
    assert np.all(s.depthBuffer[~hit] == np.finfo(np.float32).max)


def test_c4s_4k_64spp_sphere_scene_frame_properties():
    """The 4K 64-spp frame of the sphere scene (bench workload c4s; the mixed scene of configs[3] is in test_gpu_bvh.py): primary visibility equals that of the 1-spp pinhole-free frame
    (jitter is per pixel, not per sample: Scene.fs:98-104), every pixel is finite with alpha 1, and the
    ray count is within the 4-ary bound."""
    s = Scene(width=3840, height=2160, samplesPerPixel=64, maxTraceDepth=5, jitter=1, seed=1)
    s.renderScene()
    one = Scene(width=3840, height=2160, samplesPerPixel=1, maxTraceDepth=0, jitter=1, seed=1)
    one.renderScene()
    assert np.array_equal(s.objectIds, one.objectIds) and np.array_equal(s.depthBuffer, one.depthBuffer)
    assert np.isfinite(s.frameBuffer).all() and np.all(s.frameBuffer[:, :, 3] == 1.0)
    npx = 3840 * 2160 * 64
    assert npx <= s.rayCount <= npx * 5461
    # emitters seen directly are exactly white after gamma
    lights = np.isin(s.objectIds, [1, 2])
    assert lights.any() and np.all(s.frameBuffer[lights][:, :3] == 1.0)


def test_rccl_collectives_of_the_frame_gather():
    """The collectives FrameGatherer issues for N > 1 (one gather per output into views of a [world, mc, H, ...] tensor,
    one f64 all-gather of the scalars), run on RCCL with a one-rank group; two real ranks: test_gpu_bench_launch.py."""
    import torch
    import torch.distributed as dist
    from raytracer_amd import multi
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dev = torch.device("cuda", 0)
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
    try:
        s = Scene(width=64, height=48, samplesPerPixel=1, maxTraceDepth=3, jitter=0)
        s.renderScene()
        g = multi.FrameGatherer(s.width, s.height, dev, rank=0, world=1, with_ids=True)
        rgba, depth, ids = g.render_targets()
        rgba.copy_(torch.from_numpy(s.frameBuffer))
        depth.copy_(torch.from_numpy(s.depthBuffer))
        ids.copy_(torch.from_numpy(s.objectIds.astype(np.int64)))
        recv = torch.zeros((1,) + tuple(g.part_rgba.shape), dtype=torch.float32, device=dev)
        dist.gather(g.part_rgba, list(recv.unbind(0)), dst=0)
        both = torch.zeros((1, 2), dtype=torch.float64, device=dev)
        dist.all_gather(list(both.unbind(0)), torch.tensor([float(s.rayCount), float(s.maxFrameBufferDepth)], dtype=torch.float64, device=dev))
        torch.cuda.synchronize()
        assert torch.equal(recv[0], rgba) and int(both[0, 0].item()) == s.rayCount
        assert float(both[0, 1].item()) == float(np.float32(s.maxFrameBufferDepth))
        total, maxd = g.gather(rgba, depth, ids, s.rayCount, s.maxFrameBufferDepth)
        assert total == s.rayCount and np.float32(maxd) == np.float32(s.maxFrameBufferDepth)
        assert np.array_equal(g.rgba.cpu().numpy(), s.frameBuffer) and np.array_equal(g.depth.cpu().numpy(), s.depthBuffer)
        assert np.array_equal(g.ids.cpu().numpy().astype(np.uint64), s.objectIds)
    finally:
        dist.destroy_process_group()
