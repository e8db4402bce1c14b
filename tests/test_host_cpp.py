"""The C++ host mirror (include/rt_host.hpp) above the C ABI: the reference is compiled F#, so the compiled-language
host of this build is C++.  CPU part: its SceneDefinitions / camera flatten to exactly the bytes the Python mirror
produces (both restate SceneDefinitions.fs, Material.fs, Camera.fs:41-45).  GPU part: a C++ program rendering through
raytracer::Scene::renderScene gets exactly the frame the Python Scene gets."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import abi, host

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_cpp")
SD = host.SceneDefinitions
LISTS = ["lights", "lights_simple", "spheres", "spheres_submission", "spheres_simple", "spheres_caustics",
         "spheres_scene_2", "planes", "planes_submission", "planes_simple", "planes_scene_2", "scene_spheres"]


def _build(target):
    subprocess.run(["make", "-C", HERE, target], check=True, capture_output=True)
    return os.path.join(HERE, target)


def test_cpp_scene_definitions_and_camera_flatten_like_the_python_mirror():
    out = subprocess.run([_build("dump_scene")], check=True, capture_output=True).stdout
    expect = b""
    for name in LISTS:
        lst = getattr(SD, name)
        expect += bytes(host.flatten_surfaces(lst))[: C.sizeof(abi.rt_surface) * len(lst)]
    expect += bytes(host.make_camera())
    tri = host.Lambertian(77, host.Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)), host.Material(host.Rgba32.Red)).flatten()
    expect += bytes(tri)
    assert len(out) == len(expect)
    assert out == expect


def _read_ppm(path):
    raw = open(path, "rb").read()
    magic, dims, maxval, data = raw.split(b"\n", 3)
    w, h = (int(x) for x in dims.split())
    assert magic == b"P6" and maxval == b"255" and len(data) == w * h * 3
    return np.frombuffer(data, dtype=np.uint8).reshape(h, w, 3)


def test_cpp_save_frame_and_depth_buffer_pack_like_the_python_image_stage(tmp_path):
    """saveFrameBuffer / saveDepthBuffer of include/rt_host.hpp (Scene.fs:123-144: Rgba32(Vector4) pack, depth / max,
    white = miss) against Scene.frameBufferImage() / depthBufferImage(), byte for byte, on buffers that hold every
    special case: values below 0 and above 1, exact .5 boundaries, NaN, infinities, Single.MaxValue depths."""
    from raytracer_amd.scene import Scene
    w, h = 37, 23
    rng = np.random.default_rng(5)
    s = Scene(width=w, height=h)
    fb = rng.uniform(-0.2, 1.2, (w, h, 4)).astype(np.float32)
    fb[0, 0] = (np.nan, np.inf, -np.inf, 1.0)
    fb[1, :, 0] = (np.arange(h, dtype=np.float32) + np.float32(0.5)) / np.float32(255.0)     # pack boundaries
    fb[2, :, 1] = np.arange(h, dtype=np.float32) / np.float32(255.0)
    db = rng.uniform(0.5, 40.0, (w, h)).astype(np.float32)
    db[rng.uniform(size=(w, h)) < 0.2] = np.finfo(np.float32).max                       # misses (Scene.fs:29)
    s.frameBuffer, s.depthBuffer = fb, db
    s.maxFrameBufferDepth = float(db[db < 1e30].max())
    src = tmp_path / "in.bin"
    with open(src, "wb") as f:
        f.write(np.array([w, h], dtype=np.int32).tobytes())
        f.write(np.array([s.maxFrameBufferDepth], dtype=np.float32).tobytes())
        f.write(fb.tobytes())
        f.write(db.tobytes())
    frame, depth = tmp_path / "frame.ppm", tmp_path / "depth.ppm"
    subprocess.run([_build("save_images"), str(src), str(frame), str(depth)], check=True, timeout=60)
    assert np.array_equal(_read_ppm(frame), s.frameBufferImage()[:, :, :3])
    assert np.array_equal(_read_ppm(depth), s.depthBufferImage()[:, :, :3])


@pytest.mark.gpu
@pytest.mark.parametrize("shares", [1, 3])
def test_cpp_render_scene_matches_python_scene(tmp_path, shares):
    """raytracer::Scene::renderScene through rt_render (one handle) and rt_render_multi (three handles, interleaved
    columns), twice on handles uploaded once: the frame the Python Scene gets."""
    from raytracer_amd.scene import Scene
    exe = os.path.join(HERE, "render_scene")
    if not os.path.exists(exe):
        exe = _build("render_scene")
    w, h, spp, depth, jitter = 96, 72, 2, 4, 1
    out = tmp_path / "frame.bin"
    subprocess.run([exe, str(w), str(h), str(spp), str(depth), str(jitter), str(out)] + ([str(shares)] if shares > 1 else []),
                   check=True, timeout=120)
    raw = out.read_bytes()
    n = w * h
    fb = np.frombuffer(raw, dtype=np.float32, count=n * 4).reshape(w, h, 4)
    db = np.frombuffer(raw, dtype=np.float32, count=n, offset=n * 16).reshape(w, h)
    ids = np.frombuffer(raw, dtype=np.uint64, count=n, offset=n * 20).reshape(w, h)
    md = np.frombuffer(raw, dtype=np.float32, count=1, offset=n * 28)[0]
    rays = np.frombuffer(raw, dtype=np.uint64, count=1, offset=n * 28 + 4)[0]
    s = Scene(width=w, height=h, samplesPerPixel=spp, maxTraceDepth=depth, jitter=jitter)
    s.renderScene()
    assert np.array_equal(fb, s.frameBuffer) and np.array_equal(db, s.depthBuffer) and np.array_equal(ids, s.objectIds)
    assert md == np.float32(s.maxFrameBufferDepth) and int(rays) == s.rayCount


def test_multi_gpu_placement_arithmetic():
    """rt_partition.h (used by rt_render_multi and its placement kernel) compiled for the host: every share's compact
    elements cover the frame exactly once, and the Python partition of bench.py (multi.columns_of_rank) agrees."""
    out = subprocess.run([_build("partition_check")], check=True, capture_output=True, text=True).stdout
    assert out.startswith("ok ")
    from raytracer_amd import multi
    for width in (1, 5, 101, 3840):
        for world in (1, 3, 8):
            for r in range(world):
                xb, xe, xs, ncols = multi.columns_of_rank(width, r, world)
                assert (xb, xs) == (r, world) and ncols == len(range(r, width, world))
