"""The F# shim (fsharp/Native.fs) cannot be compiled here (no .NET toolchain), so its struct declarations are parsed
and laid out by the C rules LayoutKind.Sequential follows (natural alignment of each primitive; System.Numerics
Vector3 / Vector4 / Matrix4x4 are sequential structs of float32, alignment 4) and compared field by field with the
ctypes mirror of include/rt_abi.h; every DllImport must name a function the header declares, with the same number of
parameters."""
import ctypes as C
import os
import re

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = open(os.path.join(ROOT, "fsharp", "Native.fs")).read()
HEADER = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rt_abi.h")).read(), flags=re.S)

# F# type -> (size, alignment) under LayoutKind.Sequential
FS_TYPES = {"int32": (4, 4), "int": (4, 4), "uint64": (8, 8), "float32": (4, 4),
            "Vector3": (12, 4), "Vector4": (16, 4), "Matrix4x4": (64, 4)}


def _fs_struct(name):
    m = re.search(r"type %s =\n((?:[ \t]+val mutable .*\n)+)" % name, NATIVE)
    assert m, name
    fields, off, align = [], 0, 1
    for line in m.group(1).splitlines():
        fm = re.match(r"\s+val mutable (\w+) : (\w+)", line)
        size, al = FS_TYPES[fm.group(2)]
        off = (off + al - 1) // al * al
        fields.append((fm.group(1), off, size))
        off += size
        align = max(align, al)
    return fields, (off + align - 1) // align * align


def _c_fields(struct):
    return [(n, getattr(struct, n).offset, getattr(struct, n).size) for n, _ in struct._fields_]


def _check(fs_name, struct, groups):
    """groups: C field name -> list of F# fields that cover it, in order."""
    fs, size = _fs_struct(fs_name)
    assert size == C.sizeof(struct), (fs_name, size)
    fs_iter = iter(fs)
    for cname, coff, csize in _c_fields(struct):
        covered = 0
        for want in groups.get(cname, [cname]):
            name, off, sz = next(fs_iter)
            assert name.lower() == want.lower(), (fs_name, name, want)
            assert off == coff + covered, (fs_name, name, off, coff + covered)
            covered += sz
        assert covered == csize, (fs_name, cname, covered, csize)
    assert next(fs_iter, None) is None


def test_struct_layouts_match_the_header():
    _check("RtSurface", abi.rt_surface, {"sample_count": ["sampleCount"], "g": ["g0", "g1", "g2"]})
    _check("RtCamera", abi.rt_camera, {"cam_to_world": ["camToWorld"], "look_at": ["lookAt"]})
    _check("RtParams", abi.rt_params, {"max_trace_depth": ["maxTraceDepth"], "lambertian_samples": ["lambertianSamples"],
                                       "x_begin": ["xBegin"], "x_end": ["xEnd"], "x_step": ["xStep"],
                                       "waves_per_cu": ["wavesPerCu"], "pool_blocks": ["poolBlocks"],
                                       "bvh_ring": ["bvhRing"], "bvh_min_active": ["bvhMinActive"], "two_pass": ["twoPass"]})
    assert (C.sizeof(abi.rt_surface), C.sizeof(abi.rt_camera), C.sizeof(abi.rt_params)) == (104, 80, 80)


def test_no_managed_arrays_inside_structs():
    """ByValArray fields would make the structs non-blittable (marshalled by copy): none may appear."""
    code = re.sub(r"//.*", "", NATIVE)
    assert "MarshalAs" not in code and "ByValArray" not in code
    for name in ("RtSurface", "RtCamera", "RtParams"):
        body = re.search(r"type %s =\n((?:[ \t]+val mutable .*\n)+)" % name, NATIVE).group(1)
        assert "[]" not in re.sub(r"//.*", "", body)


def test_every_dllimport_names_a_declared_function_with_the_same_arity():
    decl = {m.group(1): m.group(2) for m in re.finditer(r"\b(rt_[a-z_]+)\s*\(([^)]*)\)\s*;", HEADER)}
    imports = re.findall(r"extern \w+ (rt_[a-z_]+)\(([^)]*)\)", NATIVE.replace("[,]", "[*]"))
    assert {n for n, _ in imports} == set(decl), (sorted(decl), sorted(n for n, _ in imports))
    for name, params in imports:
        c_params = [p for p in decl[name].split(",") if p.strip() and p.strip() != "void"]
        fs_params = [p for p in params.split(",") if p.strip()]
        assert len(c_params) == len(fs_params), name
    assert "RtAbiVersion = %d" % abi.RT_ABI_VERSION in NATIVE


def test_fsproj_keeps_the_reference_compile_order():
    proj = open(os.path.join(ROOT, "fsharp", "raytracer.gpu.fsproj")).read()
    order = re.findall(r'<Compile Include="([^"]+)"', proj)
    ref = ["Numerics.fs", "Camera.fs", "Material.fs", "Geometry/Core.fs", "Geometry/Sphere.fs", "Geometry/Plane.fs",
           "Geometry/AABB.fs", "Surface.fs", "SceneDefinitions.fs", "Scene.fs", "Program.fs"]      # raytracer.fsproj:11-21
    assert [f for f in order if f in ref] == ref
    assert order.index("SceneDefinitions.fs") < order.index("Native.fs") < order.index("SceneGpu.fs") < order.index("Program.fs")


def test_frame_buffers_of_the_shim_are_pinned_arrays_with_the_reference_indexer():
    """SceneGpu keeps frameBuffer / depthBuffer (Scene.fs:28-29) in PinnedArray2D (rt_host_alloc memory): the type must
    offer the `.[px,py]` indexer over px*height + py that saveFrameBuffer / saveDepthBuffer use (Scene.fs:128,139), free
    its block, and the render calls must pass its pointer."""
    gpu = open(os.path.join(ROOT, "fsharp", "SceneGpu.fs")).read()
    assert "type PinnedArray2D<'T when 'T : unmanaged>" in NATIVE
    assert "rt_host_alloc(bytes, &ptr)" in NATIVE and "rt_host_free ptr" in NATIVE
    assert "member self.Item" in NATIVE and NATIVE.count("(px * height + py)") == 2
    for buf in ("frameBuffer", "depthBuffer", "objectIds"):
        assert re.search(r"let %s = new PinnedArray2D<" % buf, gpu), buf
        assert "%s.Pointer" % buf in gpu
        assert "(%s :> IDisposable).Dispose ()" % buf in gpu
