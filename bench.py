#!/usr/bin/env python3
"""bench.py — Mrays/s of the render loop on N MI355X (one process per GPU).

`python bench.py --gpus N` works with or without a launcher: under torch.distributed.run (WORLD_SIZE set) this process
is one rank; started plainly with N > 1 it becomes the parent of N fresh rank processes (raytracer_amd/launch.py).

A "step" is one full frame of the workload: every rank renders its interleaved image columns with
the HIP kernel (device buffers, no PCIe in the timed region), the framebuffer and depth buffer are
gathered to rank 0 over RCCL/xGMI, and ray count / max depth are all-reduced.  The frame is fixed
as N grows (strong scaling): value = rays of the whole frame / max-over-ranks time.

A ray = one findClosestIntersection query (Surface.fs:49): primary + every bounce ray intersected.

Workloads (BASELINE.json configs; camera of Scene.fs:36-43 unless noted):
  c4   scene_spheres + 100k-triangle mesh, 3840x2160, 64 spp, maxTraceDepth 5  — configs[3], BVH (default)
  c4s  scene_spheres only (12 surfaces, linear scan), 3840x2160, 64 spp, maxTraceDepth 5
  c2   scene_spheres, 800x640, 16 spp, maxTraceDepth 4, jitter on    — configs[1]
  c1   scene_spheres, 640x480, 1 spp, maxTraceDepth 5, pinhole       — configs[0]
  c3   100k-triangle mesh + floor + lights, 800x640, 16 spp, maxTraceDepth 5     — configs[2], BVH
  c3_1 the same at 1 spp ("spp 1 & 16")
  c5   1M spheres + floor + lights, 1920x1080, 4 spp, maxTraceDepth 16, Lambertian SampleCount 1 — configs[4], BVH
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    "c4": dict(width=3840, height=2160, samplesPerPixel=64, maxTraceDepth=5, jitter=1, seed=1),
    "c3": dict(width=800, height=640, samplesPerPixel=16, maxTraceDepth=5, jitter=1, seed=1),
    "c3_1": dict(width=800, height=640, samplesPerPixel=1, maxTraceDepth=5, jitter=1, seed=1),      # "spp 1 & 16"
    "c5": dict(width=1920, height=1080, samplesPerPixel=4, maxTraceDepth=16, jitter=1, seed=1, lambertian_samples=1),
    "c4s": dict(width=3840, height=2160, samplesPerPixel=64, maxTraceDepth=5, jitter=1, seed=1),
    "c2": dict(width=800, height=640, samplesPerPixel=16, maxTraceDepth=4, jitter=1, seed=1),
    "c1": dict(width=640, height=480, samplesPerPixel=1, maxTraceDepth=5, jitter=0, seed=1),
}

SCENE_NAMES = {"c4": "scene_spheres + 100352-triangle mesh (BVH)", "c3": "100352-triangle mesh + floor + 2 lights (BVH)",
               "c3_1": "100352-triangle mesh + floor + 2 lights (BVH)",
               "c5": "1M spheres + floor + 2 lights (BVH)"}
_scene_cache = {}


def workload_scene(wl):
    """(surfaces or None for scene_spheres, extra Scene kwargs) of a workload."""
    import numpy as np
    from raytracer_amd import abi, host, scenes
    if wl not in _scene_cache:
        if wl in ("c3", "c3_1"):
            _scene_cache[wl] = (scenes.heightfield_mesh(224), {})
        elif wl == "c4":
            arr = host.flatten_surfaces(host.SceneDefinitions.scene_spheres)
            base = np.frombuffer(bytes(arr), dtype=abi.surface_dtype, count=12).copy()
            mesh = scenes.heightfield_mesh(224, environment=False, base=-5.9, amplitude=1.2)
            _scene_cache[wl] = (np.concatenate([base, mesh]), {})
        elif wl == "c5":
            _scene_cache[wl] = (scenes.many_spheres(1_000_000), dict(scenes.MANY_SPHERES_CAMERA))
        else:
            _scene_cache[wl] = (None, {})
    return _scene_cache[wl]


# Algorithmic FP32 work per ray, counted from the reference's F# (SURVEY.md section 8(d)); sqrt and
# divide count 1.  Closest hit on scene_spheres = 6 sphere tests (27) + 6 plane tests (13) + 5
# rectangle tests (25) + 12 PointForRay (6) = 437 flops per ray.  Generating the ray adds 90 flops per
# Lambertian child, 60 per Metal child, 110 per Dielectric child and 30 per primary; the kernel counts
# its rays by kind, so flops per launch = 437 rays + 90 nL + 60 nM + 110 nD + 30 nPrimary.
FLOPS_SCAN, FLOPS_LAMB, FLOPS_METAL, FLOPS_DIEL, FLOPS_PRIMARY = 437.0, 90.0, 60.0, 110.0, 30.0
# MI355X_MICROARCH.md: peak FP32 vector.  Measured with tools/valu_ops.hip and tools/clock_probe.hip (profiles/
# r02_valu_ops.txt, r03_clock_probe.txt; shader clock 2.34-2.39 GHz under load): that rate is reached by v_pk_fma_f32
# (two FMAs per lane every ~4.2 cycles: 147 TFLOP/s) and by v_fma_f32 with an inline-constant operand (2.2 cycles); a
# v_fma_f32 with three register operands issues every 3.5-4.1 cycles per SIMD, v_add/v_sub/v_mul_f32 every ~2.2 (one
# flop each: 0.45 of the peak), compares, selects, min/max, conversions and integer multiplies every ~4.1, v_sqrt/v_rcp
# every ~8.  The render loop is reference-order arithmetic without contraction, so its own ceiling is well below `peak`
# (DESIGN.md section 6).
VALU_PEAK_TFLOPS = 157.3
PEAK_NOTE = ("157.3 TFLOP/s is v_pk_fma_f32 (measured 147) or v_fma_f32 with two register operands and a constant; "
             "reference-order f32 arithmetic without contraction issues v_add/v_mul at 2.2, three-register fma at 3.5-4.1 and "
             "compares/selects/min/max at 4.1 cycles per wave64 instruction per SIMD, at a shader clock measured at "
             "2.34-2.39 GHz under load (profiles/r03_clock_probe.txt, r02_valu_ops.txt, DESIGN.md section 6)")


# BVH scenes: 20 flops per box test (four per visit of a 4-wide node), 40 per triangle test, 27
# per sphere test, plus the surfaces scanned per ray (planes; in c4 also the 6 spheres of scene_spheres, which a mesh
# scene keeps out of the tree); bytes: 128 per node visit, 48 per primitive record.
BVH_COST = {"c3": dict(scan=19.0 + 2 * 44.0, prim=40.0), "c3_1": dict(scan=19.0 + 2 * 44.0, prim=40.0), "c4": dict(scan=FLOPS_SCAN, prim=40.0),
            "c5": dict(scan=19.0 + 2 * 44.0, prim=27.0)}


def launch_flops(d, wl="c4s"):
    prim = d["rays"] - d["lamb"] - d["metal"] - d["diel"]
    gen = FLOPS_LAMB * d["lamb"] + FLOPS_METAL * d["metal"] + FLOPS_DIEL * d["diel"] + FLOPS_PRIMARY * prim
    if wl in BVH_COST:
        c = BVH_COST[wl]
        return c["scan"] * d["rays"] + 80.0 * d["nodes"] + c["prim"] * d["prims"] + gen
    return FLOPS_SCAN * d["rays"] + gen


KERNEL_SOURCES = ("rt_kernels.h", "rt_bvh.h", "rt_device.h", "rt_partition.h", "rt_bvh_build.h", "rt_hip.hip")


def kernel_sources_sha256():
    """Identifies the kernels a profile was taken with (there is no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "raytracer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def measured_traffic(workload):
    """HBM bytes per launch of rt::trace_kernel from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE x2 + WRITE_SIZE, see profiles/README.md).  None when that workload was not profiled or when the
    kernel sources have changed since (profiles/traffic.json records their hash): a stale figure is not reported."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if t.get("kernel_sources_sha256") != kernel_sources_sha256():
            return None
        return t.get("bytes_per_launch", {}).get(workload)
    except (OSError, ValueError):
        return None


def cpu_baseline():
    """The oracle ("port" of the reference's F# loop) on the host cores of this box, config c1."""
    import oracle_lib as O
    from raytracer_amd import host

    cam = host.make_camera()
    w = WORKLOADS["c1"]
    p = host.make_params(w["width"], w["height"], w["samplesPerPixel"], w["maxTraceDepth"], jitter=w["jitter"], seed=w["seed"])
    surfaces = host.SceneDefinitions.scene_spheres
    O.render(surfaces, cam, host.make_params(64, 48, 1, 5, jitter=0), nthreads=1)     # load + warm
    t0 = time.perf_counter()
    r = O.render(surfaces, cam, p, nthreads=1)
    t1 = time.perf_counter() - t0
    ncores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    t0 = time.perf_counter()
    r2 = O.render(surfaces, cam, p, nthreads=ncores)
    tn = time.perf_counter() - t0
    return {
        "value": r["rays"] / t1 / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
        "sample": "config c1 whole frame: 640x480, 1 spp, maxTraceDepth 5, %d rays, reference-faithful recursion" % r["rays"],
        "all_cores": {"value": r2["rays"] / tn / 1e6, "cores": ncores},
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--waves-per-cu", type=int, default=0)
    ap.add_argument("--pool-blocks", type=int, default=0)
    ap.add_argument("--bvh-ring", type=int, default=0, help="tuning: 8 or 16 entries of the traversal stack in LDS (0 = by tree size)")
    ap.add_argument("--bvh-min-active", type=int, default=0, help="tuning: refill threshold of the BVH kernels (0 = 24, or 20 when 8 or more loose surfaces are scanned)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend for --gpus N > 1 (nccl = RCCL; gloo only to rehearse several ranks on one GPU)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    from raytracer_amd import launch
    if args.gpus > 1 and not launch.launched_as_rank():
        # Started without a launcher (`python bench.py --gpus N`): this process becomes the parent of N fresh rank
        # processes and never touches the GPU itself (no torch import here, no exec).
        sys.exit(launch.relaunch_self(args.gpus, os.path.abspath(__file__)))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` "
              f"or under torch.distributed.run --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)

    # The contract is ONE JSON line on stdout.  Libraries chat there too (gloo prints its connection report to stdout):
    # from here on file descriptor 1 is stderr, and the JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from raytracer_amd import abi, multi
    from raytracer_amd.scene import Scene

    share_one_gpu = args.backend == "gloo"          # rehearsal: every rank on device 0, gloo instead of RCCL
    ndev = torch.cuda.device_count()                # (counting devices does not initialise the GPU)
    need = 1 if share_one_gpu else world
    if ndev < need:
        if rank == 0:
            print(f"bench.py: --gpus {world} needs {need} visible GPU(s), this box has {ndev}: nothing was run "
                  "(the render path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    if share_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if share_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev)
    gdev = torch.device("cpu") if share_one_gpu else dev     # where the gathered frame lives

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make(wl):
        cfg = dict(WORKLOADS[wl])
        surfaces, extra_kw = workload_scene(wl)
        cfg.update(extra_kw)
        sc = Scene(surfaces=surfaces, device=local_rank, waves_per_cu=args.waves_per_cu, pool_blocks=args.pool_blocks,
                   bvh_ring=args.bvh_ring, bvh_min_active=args.bvh_min_active, **cfg)
        t0 = time.perf_counter()
        sc.upload()                                   # rt_scene_create: flatten, host BVH build, upload (once per scene)
        create_ms = (time.perf_counter() - t0) * 1e3
        xb, xe, xs, ncols = multi.columns_of_rank(sc.width, rank, world)
        g = multi.FrameGatherer(sc.width, sc.height, gdev, rank, world)
        if share_one_gpu:
            bufs = dict(rgba=torch.zeros((ncols, sc.height, 4), dtype=torch.float32, device=dev),
                        depth=torch.zeros((ncols, sc.height), dtype=torch.float32, device=dev))
        else:
            # the kernel renders straight into the gather's send buffers
            r, d, _ = g.render_targets()
            bufs = dict(rgba=r, depth=d)
        bufs["stats"] = torch.zeros(abi.RT_STATS_WORDS, dtype=torch.int64, device=dev)
        return sc, (xb, xe, xs), bufs, g, create_ms

    diag = {}

    def step(sc, cols, bufs, g, ev=None):
        stream = torch.cuda.current_stream().cuda_stream
        if ev:
            ev[0].record()
        sc.renderColumnsDevice(bufs["rgba"].data_ptr(), bufs["depth"].data_ptr(), None, bufs["stats"].data_ptr(),
                               stream, cols[0], cols[1], cols[2])
        if ev:
            ev[1].record()
        st = bufs["stats"].cpu()        # syncs this stream: ray count + max depth bits of this rank
        rays = int(st[0].item())
        diag["rays"], diag["iters"] = rays, int(st[2].item())
        diag["lamb"], diag["metal"], diag["diel"], diag["sum_wave_clk"], diag["waves"] = (int(st[i].item()) for i in (3, 4, 5, 6, 7))
        diag["nodes"], diag["prims"] = int(st[8].item()), int(st[9].item())
        diag["rounds_inner"], diag["rounds_leaf"] = int(st[10].item()), int(st[11].item())
        diag["phase_clk"] = [int(st[i].item()) for i in (12, 13, 14, 15)]
        diag["max_wave_clk"] = int(st[18].item())
        maxd = torch.tensor([int(st[1].item()) & 0xffffffff], dtype=torch.int64).view(torch.float32)[0].item()
        if ev:
            ev[2].record()
        if share_one_gpu:
            out = g.gather(bufs["rgba"].cpu(), bufs["depth"].cpu(), None, rays, maxd)
        else:
            out = g.gather(bufs["rgba"], bufs["depth"], None, rays, maxd)
        if ev:
            ev[3].record()
        return out

    def run(wl, steps, warmup):
        sc, cols, bufs, g, create_ms = make(wl)
        for _ in range(warmup):
            step(sc, cols, bufs, g)
        sc.status()
        evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(steps)]
        barrier()
        t0 = time.perf_counter()
        rays = 0
        for i in range(steps):
            r, _ = step(sc, cols, bufs, g, evs[i])
            rays += r
        barrier()
        dt = time.perf_counter() - t0
        sc.status()
        # The timed frames count what renderScene itself produces (rays, largest depth).  The inputs of the flop model —
        # rays by kind, BVH node visits and primitive tests, loop iterations — come from ONE more frame of the same
        # workload rendered with every counter on (rt_scene_counters; 1-3 % slower, outside the timed region): the ray
        # trees, and with them those counts, are identical from frame to frame.
        lean_rays = diag["rays"]
        timed_clocks = {k: diag[k] for k in ("sum_wave_clk", "waves", "max_wave_clk")}     # wave clocks: of the last TIMED frame
        sc.counters(1)
        step(sc, cols, bufs, g)
        sc.status()
        assert diag["rays"] == lean_rays, (diag["rays"], lean_rays)
        diag.update(timed_clocks)
        kern_ms = sum(e[0].elapsed_time(e[1]) for e in evs) / steps
        gather_ms = sum(e[2].elapsed_time(e[3]) for e in evs) / steps
        tt = torch.tensor([dt], dtype=torch.float64, device=gdev)
        per_rank = None
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            mine = torch.tensor([kern_ms, gather_ms, float(diag["rays"]), dt * 1e3 / steps], dtype=torch.float64, device=gdev)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = torch.stack(allr).cpu().tolist()
        sc.close()
        return dict(rays=rays, dt=float(tt.item()), kernel_ms=kern_ms, gather_ms=gather_ms, create_ms=create_ms, per_rank=per_rank)

    res = run(args.workload, args.steps, args.warmup)
    rays, dt, kern_ms = res["rays"], res["dt"], res["kernel_ms"]
    rays_per_frame = rays / args.steps
    ranks_seen = None
    if world > 1:
        # proof that the process group saw `world` ranks, each on a device of its own
        props = torch.cuda.get_device_properties(local_rank)
        me = {"rank": rank, "device_index": local_rank, "name": props.name, "uuid": str(getattr(props, "uuid", "")),
              "pci_bus_id": getattr(props, "pci_bus_id", None), "pid": os.getpid()}
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks_seen = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "devices": seen}
    out = None
    if rank == 0:
        cfg = WORKLOADS[args.workload]
        flops = launch_flops(diag, args.workload)                     # rank 0's last launch (interleaved columns: equal shares to <1 %)
        achieved = flops / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/s (primary+bounce)", "value": rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s, %dx%d, %d spp, maxTraceDepth %d, jitter %d" % (
                           args.workload, SCENE_NAMES.get(args.workload, "scene_spheres (12 surfaces, linear scan)"),
                           cfg["width"], cfg["height"], cfg["samplesPerPixel"], cfg["maxTraceDepth"], cfg["jitter"]),
                       "rays_per_frame": rays_per_frame,
                       "partition": "interleaved columns, %d rank(s), %s gather to rank 0" % (world, "gloo (one shared GPU: rehearsal)" if share_one_gpu else "RCCL"),
                       "scene_create_ms": res["create_ms"]},
            "roofline": {"bound": "valu", "achieved": achieved, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / VALU_PEAK_TFLOPS, "traffic": measured_traffic(args.workload) if world == 1 else None,
                         "kernel": "rt::trace_kernel (+ memsets and rt::resolve_kernel, <1 % of the bracket)",
                         "counts_from": "one extra frame of the same workload with rt_scene_counters(1), after the timed region (identical ray trees; the timed frames count rays and largest depth only)",
                         "peak_note": PEAK_NOTE,
                         "kernel_ms": kern_ms, "flops_per_launch": flops, "flops_per_ray": flops / max(1, diag["rays"]),
                         "rays_by_kind": {"lambertian": diag["lamb"], "metal": diag["metal"], "dielectric": diag["diel"],
                                          "primary": diag["rays"] - diag["lamb"] - diag["metal"] - diag["diel"]},
                         "bvh": ({"node_visits_per_ray": diag["nodes"] / max(1, diag["rays"]), "prim_tests_per_ray": diag["prims"] / max(1, diag["rays"]),
                                  "l1_request_GBps": (128.0 * diag["nodes"] + 48.0 * diag["prims"]) / (kern_ms * 1e-3) / 1e9,   # node and record bytes the lanes ask of L1/L2, not HBM traffic
                                  "inner_rounds": diag["rounds_inner"], "leaf_rounds": diag["rounds_leaf"], "iterations": diag["iters"],
                                  "lanes_per_inner_round": diag["nodes"] / max(1, diag["rounds_inner"]),
                                  "phase_clk_gen_scan_trav_shade": diag["phase_clk"]}
                                 if args.workload in BVH_COST else None),
                         "lane_fill": diag["rays"] / max(1, diag["iters"] * 64), "waves": diag["waves"],
                         "mean_wave_clk": diag["sum_wave_clk"] / max(1, diag["waves"]), "max_wave_clk": diag["max_wave_clk"]},
            "gather_ms": res["gather_ms"],
        }
        if world > 1:
            pr = res["per_rank"]
            out["ranks_seen"] = ranks_seen
            out["per_rank"] = {"kernel_ms": [r[0] for r in pr], "kernel_ms_min": min(r[0] for r in pr), "kernel_ms_max": max(r[0] for r in pr),
                               "gather_ms": [r[1] for r in pr], "rays_last_frame": [int(r[2]) for r in pr],
                               "ms_per_step": [r[3] for r in pr]}
    if rank == 0 and world == 1 and not args.no_extra:
        extra = {}
        for wl, st in (("c1", 20), ("c2", 20), ("c3", 5), ("c3_1", 20), ("c5", 3), ("c4s", 2)):
            if wl == args.workload:
                continue
            r = run(wl, st, 1)
            extra[wl] = {"Mrays_per_s": r["rays"] / r["dt"] / 1e6, "ms_per_frame": r["dt"] / st * 1e3, "kernel_ms": r["kernel_ms"],
                         "rays_per_frame": r["rays"] / st, "scene_create_ms": r["create_ms"]}
        # the same frames through rt_render (HOST buffers: staging + PCIe copies of rgba/depth/ids inside the call);
        # reported beside the device-resident figures, never as `value`
        hostbuf = {}
        import ctypes as C
        import numpy as np
        for wl, st in (("c1", 20), ("c2", 5)):
            cfg = dict(WORKLOADS[wl])
            s = Scene(device=local_rank, **cfg)

            def timed(f, n):
                f(); f()
                t0 = time.perf_counter()
                for _ in range(n):
                    f()
                return (time.perf_counter() - t0) / n

            # Scene keeps its frame buffers in pinned memory of the library (rt_host_alloc): DMA straight into them
            t = timed(s.renderScene, st)
            hostbuf[wl] = {"Mrays_per_s": s.rayCount / t / 1e6, "ms_per_frame": t * 1e3, "buffers": "pinned (rt_host_alloc), rgba + depth + object ids"}
            # the same call into ordinary (pageable) host arrays: what round 2 measured
            fb, db, ids = np.zeros_like(s.frameBuffer), np.zeros_like(s.depthBuffer), np.zeros_like(s.objectIds)
            p, md, rays = s.params(), C.c_float(0.0), C.c_uint64(0)

            def pageable():
                abi.check(s._lib, s._lib.rt_render(s._scene, C.byref(s.camera), C.byref(p), fb.ctypes.data, db.ctypes.data,
                                                   ids.ctypes.data, C.byref(md), C.byref(rays)))
            t = timed(pageable, st)
            hostbuf[wl + "_pageable"] = {"Mrays_per_s": rays.value / t / 1e6, "ms_per_frame": t * 1e3, "buffers": "pageable numpy arrays"}
            s.close()
        extra["host_buffers_pcie_inclusive"] = hostbuf
        out["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
