"""The N>1 path on CPU: two gloo ranks partition the columns as bench.py does, render their share
(with the ORACLE standing in for the HIP kernel: this container has no GPU), and gather the frame with
raytracer_amd.multi.FrameGatherer.  Rank 0's gathered frame must equal the single-process render bit for
bit, the ray counts must add up and the max depth must be the global maximum."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as O
    from raytracer_amd import host, multi

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cam = host.make_camera()
        sc = host.SceneDefinitions.scene_spheres
        xb, xe, xs, ncols = multi.columns_of_rank(width, rank, world)
        p = host.make_params(width, height, 2, 3, jitter=1, seed=4, x_begin=xb, x_end=xe, x_step=xs)
        assert host.num_columns(p) == ncols
        r = O.render(sc, cam, p, nthreads=2)
        g = multi.FrameGatherer(width, height, torch.device("cpu"), rank, world, with_ids=True)
        rays, maxd = g.gather(torch.from_numpy(r["rgba"]), torch.from_numpy(r["depth"]),
                              torch.from_numpy(r["ids"].astype(np.int64)), r["rays"], r["max_depth"])
        if rank == 0:
            np.savez(out_path, rgba=g.rgba.numpy(), depth=g.depth.numpy(), ids=g.ids.numpy(), rays=rays, maxd=maxd)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width", [(2, 33), (3, 32)])
def test_two_rank_partition_and_gather(tmp_path, world, width):
    import oracle_lib as O
    from raytracer_amd import host

    height = 20
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), width, height, out), nprocs=world, join=True)
    got = np.load(out)
    full = O.render(host.SceneDefinitions.scene_spheres, host.make_camera(),
                    host.make_params(width, height, 2, 3, jitter=1, seed=4))
    assert np.array_equal(got["rgba"], full["rgba"])
    assert np.array_equal(got["depth"], full["depth"])
    assert np.array_equal(got["ids"].astype(np.uint64), full["ids"])
    assert int(got["rays"]) == full["rays"]
    assert np.float32(got["maxd"]) == np.float32(full["max_depth"])


def test_columns_of_rank_cover_the_frame():
    from raytracer_amd import multi
    for width in (1, 7, 64, 3840):
        for world in (1, 2, 3, 8):
            cols = []
            for r in range(world):
                xb, xe, xs, n = multi.columns_of_rank(width, r, world)
                mine = list(range(xb, xe, xs)) if n else []
                assert len(mine) == n <= multi.max_columns(width, world)
                cols += mine
            assert sorted(cols) == list(range(width))
