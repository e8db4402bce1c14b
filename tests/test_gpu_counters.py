"""rt_scene_counters (include/rt_abi.h): which words of d_stats a scene's frames fill.  Level 0 (default) is what
`Scene.renderScene` itself produces — the ray count and maxFrameBufferDepth (Scene.fs:47-48) — level 1 adds the inputs of
bench.py's flop model.  The frame must not depend on the level, and the two levels are different kernel instantiations
(template parameter COUNT of trace_kernel), so each is held to the other and to the oracle here."""
import numpy as np
import pytest

import oracle_lib as O
from raytracer_amd import abi, host, scenes
from raytracer_amd.scene import Scene

pytestmark = pytest.mark.gpu


def _mixed():
    base = np.frombuffer(bytes(host.flatten_surfaces(host.SceneDefinitions.scene_spheres)), dtype=abi.surface_dtype, count=12).copy()
    return np.concatenate([base, scenes.heightfield_mesh(12, environment=False, base=-5.9, amplitude=1.0)])


def _frame(s, noalloc=False):
    import torch
    dev = torch.device("cuda", 0)
    rgba = torch.zeros((s.width, s.height, 4), dtype=torch.float32, device=dev)
    depth = torch.zeros((s.width, s.height), dtype=torch.float32, device=dev)
    ids = torch.zeros((s.width, s.height), dtype=torch.int64, device=dev)
    stats = torch.full((abi.RT_STATS_WORDS,), -1, dtype=torch.int64, device=dev)
    render = s.renderColumnsDeviceNoAlloc if noalloc else s.renderColumnsDevice
    render(rgba.data_ptr(), depth.data_ptr(), ids.data_ptr(), stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s.status()
    return rgba.cpu().numpy(), depth.cpu().numpy(), ids.cpu().numpy(), [int(x) for x in stats.cpu().tolist()]


@pytest.mark.parametrize("mesh", [False, True])
@pytest.mark.parametrize("kw", [dict(two_pass=1), dict(two_pass=0), dict(two_pass=1, estimator=abi.ESTIMATOR_SINGLE_PATH)])
def test_frames_do_not_depend_on_the_counters(mesh, kw):
    s = Scene(surfaces=_mixed() if mesh else None, width=120, height=90, samplesPerPixel=2, maxTraceDepth=5, jitter=1, seed=3, **kw)
    lean = _frame(s)
    s.counters(1)
    full = _frame(s)
    full_noalloc = _frame(s, noalloc=True)        # the entry point that allocates nothing runs the same instantiations
    assert np.array_equal(full[0], full_noalloc[0])
    for w in (0, 1, 3, 4, 5, 8, 9):            # (iterations and rounds depend on the number of waves, these do not)
        assert full[3][w] == full_noalloc[3][w], w
    s.counters(0)
    again = _frame(s)
    for a, b in ((lean, full), (lean, again)):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3][0] == b[3][0] and a[3][1] == b[3][1]
    ref = O.render(s.surfaces, s.camera, s.params())
    assert lean[3][0] == ref["rays"]
    # level 0: the counters of level 1 stay zero; level 1: the rays by kind add up, and a BVH scene reports its visits
    for w in (2, 3, 4, 5, 8, 9, 10, 11):
        assert lean[3][w] == 0 and again[3][w] == 0, w
    st = full[3]
    primaries = s.width * s.height * s.samplesPerPixel
    assert st[3] + st[4] + st[5] + primaries == st[0]
    assert st[2] * 64 >= st[0] and st[7] > 0
    assert (st[8] > 0 and st[9] > 0 and st[10] > 0 and st[11] > 0) if mesh else (st[8] == 0 and st[9] == 0)
    s.close()


def test_counters_reject_other_levels_and_null():
    lib = abi.load_hip_library()
    assert lib.rt_scene_counters(None, 1) == abi.RT_E_INVALID
    s = Scene(width=16, height=16, samplesPerPixel=1, maxTraceDepth=1, jitter=0)
    s.upload()
    assert lib.rt_scene_counters(s._scene, 2) == abi.RT_E_INVALID
    assert lib.rt_scene_counters(s._scene, 1) == 0
    s.close()
