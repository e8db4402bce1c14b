"""The C-ABI boundary: include/rt_abi.h, the ctypes mirror of its structs, and librt_hip.so's exports.
No compute call is made here (no GPU in this container): the library must load, export every symbol
the header declares, and fail loudly (RT_E_NO_DEVICE) instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from raytracer_amd import abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_abi.h")


def _declared_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_header_compiles_as_c_and_layouts_match_ctypes():
    src = '#include "%s"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%d %%d %%zu %%zu %%u %%d\\n",' \
          'sizeof(rt_surface),sizeof(rt_camera),sizeof(rt_params),offsetof(rt_params,seed),offsetof(rt_params,x_begin),' \
          'offsetof(rt_surface,g),offsetof(rt_params,pool_blocks),RT_ABI_VERSION,RT_STATS_WORDS,' \
          'offsetof(rt_params,estimator),offsetof(rt_params,two_pass),RT_NOALLOC_SCRATCH_BYTES,RT_NOALLOC_MAX_PIXELS);return 0;}\n' % HEADER
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", c, "-o", os.path.join(d, "t")], check=True)
        out = subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    want = [C.sizeof(abi.rt_surface), C.sizeof(abi.rt_camera), C.sizeof(abi.rt_params), abi.rt_params.seed.offset,
            abi.rt_params.x_begin.offset, abi.rt_surface.g.offset, abi.rt_params.pool_blocks.offset, abi.RT_ABI_VERSION, 20,
            abi.rt_params.estimator.offset, abi.rt_params.two_pass.offset, abi.RT_NOALLOC_SCRATCH_BYTES, abi.RT_NOALLOC_MAX_PIXELS]
    assert got == want
    assert (got[0], got[1], got[2]) == (104, 80, 80)
    assert abi.surface_dtype.itemsize == C.sizeof(abi.rt_surface)


def test_python_symbol_list_is_the_header():
    assert sorted(abi.ABI_SYMBOLS) == _declared_functions()


def test_library_exports_every_declared_symbol():
    lib = abi.load_hip_library()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.rt_abi_version() == abi.RT_ABI_VERSION
    assert lib.rt_last_error() is not None


def test_no_cpu_fallback_without_a_device():
    """On a box without a HIP device the product path fails loudly; nothing is rendered on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = abi.load_hip_library()
    arr = host.flatten_surfaces(host.SceneDefinitions.scene_spheres)
    scene = C.c_void_p()
    rc = lib.rt_scene_create(arr, 12, 0, C.byref(scene))
    assert rc == abi.RT_E_NO_DEVICE and not scene
    assert b"no HIP device" in lib.rt_last_error()
    ptr = C.c_void_p()
    assert lib.rt_host_alloc(4096, C.byref(ptr)) == abi.RT_E_NO_DEVICE and not ptr      # no pinned memory without a device either
    from raytracer_amd.scene import Scene
    with pytest.raises(abi.RtError):
        Scene(width=8, height=8).renderScene()


def test_argument_validation_messages():
    lib = abi.load_hip_library()
    assert lib.rt_scene_create(None, 3, 0, None) == abi.RT_E_INVALID
    assert lib.rt_scene_status(None) == abi.RT_E_INVALID
    assert lib.rt_find_closest(None, 0, None, None, None, None) == abi.RT_E_INVALID
    assert lib.rt_render_multi(None, 0, None, None, None, None, None, None, None) == abi.RT_E_INVALID
    assert lib.rt_render_device_noalloc(None, None, None, None, None, None, None, None) == abi.RT_E_INVALID
    assert lib.rt_host_alloc(4096, None) == abi.RT_E_INVALID
    ptr = C.c_void_p(1)
    assert lib.rt_host_alloc(0, C.byref(ptr)) == abi.RT_E_INVALID and not ptr
    lib.rt_host_free(None)                                                              # like free(NULL)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_amd/ or include/ may name it."""
    for base in ("raytracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".hpp", ".cpp")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "rt_oracle" not in text and "oracle_lib" not in text and "librt_oracle" not in text, f
