"""bench.py for N > 1 the way the driver starts it (`python bench.py --gpus N`, no launcher), and the multi-GPU paths
on real devices.  The tests that need two physical GPUs run by themselves on a box that has them and skip on a one-GPU
box; there the same code is rehearsed with two ranks sharing device 0 (`--backend gloo`)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import abi, host, launch, scenes
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _ndev():
    import torch
    return torch.cuda.device_count()


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in launch.RANK_ENV}
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=900, env=env)


def _line(r):
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_self_launched_ranks_share_one_gpu_and_gather_the_frame():
    one = _line(_bench("--gpus", "1", "--workload", "c2", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"))
    two = _line(_bench("--gpus", "2", "--backend", "gloo", "--workload", "c2", "--steps", "2", "--warmup", "1"))
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "strong"
    # the same frame: the two shares trace exactly the rays of the whole frame
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]
    seen = two["ranks_seen"]
    assert seen["world_size"] == 2 and [d["rank"] for d in seen["devices"]] == [0, 1]
    assert len({d["pid"] for d in seen["devices"]}) == 2
    pr = two["per_rank"]
    assert len(pr["kernel_ms"]) == 2 and pr["kernel_ms_min"] > 0 and sum(pr["rays_last_frame"]) == one["config"]["rays_per_frame"]
    assert two["gather_ms"] > 0 and two["value"] > 0


def test_bench_line_carries_the_contract():
    """One JSON line on stdout with the keys the driver reads, the roofline and the CPU baseline (the oracle timed on
    the host cores, a bounded sample), and nothing else on stdout."""
    r = _bench("--workload", "c1", "--steps", "3", "--warmup", "1", "--no-extra")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f32"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["workload"].startswith("c1:") and d["config"]["rays_per_frame"] == 16886754
    roof = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert key in roof, key
    assert roof["peak"] == 157.3 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0.0 < roof["frac"] < 1.0
    # value = rays of all frames / wall time of the timed region; the kernel bracket is inside it
    assert abs(d["value"] - d["config"]["rays_per_frame"] / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-9
    assert roof["kernel_ms"] <= d["ms_per_step"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == "Mrays/s" and cpu["value"] > 0 and "640x480" in cpu["sample"]


def test_asking_for_more_gpus_than_the_box_has_is_a_message_not_a_traceback():
    n = _ndev() + 1
    r = _bench("--gpus", str(n), "--workload", "c1", "--steps", "1")
    assert r.returncode == 2 and f"needs {n} visible GPU" in r.stderr and "Traceback" not in r.stderr and r.stdout.strip() == ""


# ---- two or more physical GPUs ---------------------------------------------------------------------------------
needs2 = pytest.mark.skipif(_ndev() < 2, reason="needs two GPUs (runs by itself on a multi-GPU box)")


@needs2
def test_bench_two_gpus_over_rccl():
    one = _line(_bench("--gpus", "1", "--workload", "c2", "--steps", "2", "--no-extra", "--no-cpu-baseline"))
    two = _line(_bench("--gpus", "2", "--workload", "c2", "--steps", "2"))
    seen = two["ranks_seen"]
    assert seen["world_size"] == 2 and seen["backend"] == "nccl"
    assert sorted(d["device_index"] for d in seen["devices"]) == [0, 1]
    assert two["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]


def _frame(s):
    return s.frameBuffer.copy(), s.depthBuffer.copy(), s.objectIds.copy(), s.maxFrameBufferDepth, s.rayCount


def _equal(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3] == b[3] and a[4] == b[4]


def _mixed():
    base = np.frombuffer(bytes(host.flatten_surfaces(host.SceneDefinitions.scene_spheres)), dtype=abi.surface_dtype, count=12).copy()
    return np.concatenate([base, scenes.heightfield_mesh(24, environment=False, base=-5.9, amplitude=1.0)])


@needs2
@pytest.mark.parametrize("bvh", [False, True])
def test_render_multi_over_physical_devices_equals_rt_render(bvh):
    """rt_render_multi with one handle per device: peer stores over xGMI into device 0's frame (Scene.fs:119-121:
    independent pixels; :47-48: max depth)."""
    n = min(_ndev(), 8)
    kw = dict(width=203, height=96, samplesPerPixel=3, maxTraceDepth=4, jitter=1, seed=5)
    if bvh:
        kw["surfaces"] = _mixed()
    s = Scene(**kw)
    s.renderScene()
    want = _frame(s)
    others = [Scene(device=d, **kw) for d in range(1, n)]
    for _ in range(2):                                   # the second frame reuses the cached peer set-up
        s.frameBuffer[:] = -1
        s.renderSceneMulti(others)
        _equal(_frame(s), want)


def _rccl_worker(rank, world, port, out_path, kw):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from raytracer_amd import multi
    from raytracer_amd.scene import Scene as S

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        s = S(device=rank, **kw)
        xb, xe, xs, ncols = multi.columns_of_rank(s.width, rank, world)
        g = multi.FrameGatherer(s.width, s.height, dev, rank, world, with_ids=True)
        rgba, depth, ids = g.render_targets()
        stats = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
        s.renderColumnsDevice(rgba.data_ptr(), depth.data_ptr(), ids.data_ptr(), stats.data_ptr(),
                              torch.cuda.current_stream().cuda_stream, xb, xe, xs)
        st = stats.cpu()
        s.status()
        maxd = torch.tensor([int(st[1].item()) & 0xffffffff], dtype=torch.int64).view(torch.float32)[0].item()
        total, gmax = g.gather(rgba, depth, ids, int(st[0].item()), maxd)
        if rank == 0:
            np.savez(out_path, rgba=g.rgba.cpu().numpy(), depth=g.depth.cpu().numpy(), ids=g.ids.cpu().numpy(), rays=total, maxd=gmax)
    finally:
        dist.destroy_process_group()


@needs2
def test_two_rccl_ranks_gather_the_single_gpu_frame(tmp_path):
    import torch.multiprocessing as mp
    kw = dict(width=101, height=64, samplesPerPixel=3, maxTraceDepth=4, jitter=1, seed=12)
    out = str(tmp_path / "frame.npz")
    mp.spawn(_rccl_worker, args=(2, launch.free_port(), out, kw), nprocs=2, join=True)
    got = np.load(out)
    s = Scene(**kw)
    s.renderScene()
    assert np.array_equal(got["rgba"], s.frameBuffer) and np.array_equal(got["depth"], s.depthBuffer)
    assert np.array_equal(got["ids"].astype(np.uint64), s.objectIds)
    assert int(got["rays"]) == s.rayCount and np.float32(got["maxd"]) == np.float32(s.maxFrameBufferDepth)
