// Native.fs — P/Invoke surface of librt_hip.so (include/rt_abi.h, RT_ABI_VERSION 8) for the reference's F# host.
//
// Compile position: after Surface.fs / SceneDefinitions.fs and before SceneGpu.fs (see raytracer.gpu.fsproj).
// Every struct below is BLITTABLE: only primitive fields and System.Numerics Vector3 / Vector4 / Matrix4x4 (which are
// themselves sequential structs of float32), no managed arrays and no MarshalAs — the runtime pins and passes them
// without copying.  tests/test_fsharp_shim.py parses the `val mutable` lines of this file and checks field order,
// offsets and sizes (104 / 80 / 80 bytes) against the C header, because no .NET toolchain exists in the build image.
module Raytracer.Native

open System
open System.Numerics
open System.Runtime.InteropServices

// rt_surface (include/rt_abi.h): one `Surface` (Surface.fs:15) flattened
[<Struct; StructLayout(LayoutKind.Sequential)>]
type RtSurface =
    val mutable kind : int32            // 0 NoSurface, 1 Lambertian, 2 Metal, 3 Dielectric   (Surface.fs:44,60,85,112)
    val mutable geom : int32            // 0 Sphere, 1 Plane (Geometry/*.fs), 2 Triangle (the build's own Hitable)
    val mutable id : uint64             // Surface.ID (Surface.fs:25)
    val mutable albedo : Vector3        // Material.Albedo     (Material.fs:9-18)
    val mutable emission : Vector3      // Material.Emmitance
    val mutable fuzz : float32          // Metal.Fuzz          (Surface.fs:88)
    val mutable ior : float32           // Dielectric.RefractiveIndex (Surface.fs:117)
    val mutable sampleCount : int32     // 0 = the kind's default SampleCount
    val mutable g0 : Vector4            // sphere: center.xyz, radius | plane: normal.xyz, d      | triangle: v0.xyz, v1.x
    val mutable g1 : Vector4            //                            | hasCenter, center.xyz     | v1.yz, v2.xy
    val mutable g2 : Vector4            //                            | width, height, -, -       | v2.z, -, -, -

// rt_camera: what renderScene closes over (Scene.fs:36-43)
[<Struct; StructLayout(LayoutKind.Sequential)>]
type RtCamera =
    val mutable camToWorld : Matrix4x4  // cameraWS = CameraToWorld viewMatrix (Scene.fs:41), M11..M44
    val mutable lookAt : Vector3        // the target POINT (Scene.fs:37,76)
    val mutable fov : float32           // Scene.fs:43

// rt_params: Scene.fs:18-32 plus what the build adds
[<Struct; StructLayout(LayoutKind.Sequential)>]
type RtParams =
    val mutable width : int32
    val mutable height : int32
    val mutable spp : int32             // samplesPerPixel
    val mutable maxTraceDepth : int32
    val mutable background : Vector3    // backgroundColor
    val mutable jitter : int32          // 0 = pinhole, 1 = Camera.pertrube on
    val mutable seed : uint64
    val mutable lambertianSamples : int32
    val mutable xBegin : int32
    val mutable xEnd : int32
    val mutable xStep : int32
    val mutable wavesPerCu : int32
    val mutable poolBlocks : int32
    val mutable estimator : int32       // 0 = renderScene's branching estimator, 1 = single path (Dielectric.Scatter)
    val mutable bvhRing : int32
    val mutable bvhMinActive : int32
    val mutable twoPass : int32

[<Literal>]
let RtAbiVersion = 8

[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_abi_version()
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern nativeint rt_last_error()
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_scene_create(RtSurface[] surfaces, int64 n, int32 device, nativeint& scene)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern void rt_scene_destroy(nativeint scene)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_scene_status(nativeint scene)
// 0 (default): frames count their rays and largest depth only; 1: every word of the statistics array (1-3 % slower)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_scene_counters(nativeint scene, int32 level)
// optional: pinned host memory for the frame buffers (PinnedArray2D below)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_host_alloc(uint64 bytes, nativeint& out)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern void rt_host_free(nativeint ptr)
// rgba / depth / objectId: width*height elements laid out like `Array2D.create width height` (Scene.fs:28-29: element
// [px,py] at px*height + py).  Pass PinnedArray2D.Pointer (below), or the address of a GCHandle-pinned managed
// Vector4[,] / float32[,] (GCHandle.Alloc(a, GCHandleType.Pinned).AddrOfPinnedObject()) — the layout is the same.
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_render(nativeint scene, RtCamera& cam, RtParams& p, nativeint rgba, nativeint depth,
                     nativeint objectId, float32& maxDepth, uint64& rayCount)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_render_multi(nativeint[] scenes, int32 nscenes, RtCamera& cam, RtParams& p, nativeint rgba,
                           nativeint depth, nativeint objectId, float32& maxDepth, uint64& rayCount)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_render_device(nativeint scene, RtCamera& cam, RtParams& p, nativeint stream, nativeint dRgba,
                            nativeint dDepth, nativeint dObjectId, nativeint dStats)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_render_device_noalloc(nativeint scene, RtCamera& cam, RtParams& p, nativeint stream, nativeint dRgba,
                                    nativeint dDepth, nativeint dObjectId, nativeint dStats)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_find_closest(nativeint scene, int64 nrays, Vector3[] origins, Vector3[] directions, float32[] t, int32[] index)

/// failwith on a negative return code, with the library's message (the style of Camera.fs:45)
let check (rc : int) =
    if rc <> 0 then failwith (sprintf "rt_hip error %d: %s" rc (Marshal.PtrToStringAnsi(rt_last_error())))

/// A width x height array of a blittable element type in pinned host memory of the library (rt_host_alloc), indexed
/// like the reference's `Array2D.create width height` buffers: `a.[px, py]` is element px*height + py, so
/// `frameBuffer.[px,py]` in saveFrameBuffer / saveDepthBuffer (Scene.fs:128,139) compiles unchanged against it.
/// The library copies a frame into such a block by DMA; a managed array goes through a staged pageable copy.
type PinnedArray2D<'T when 'T : unmanaged> (width : int, height : int, init : 'T) =
    let bytes = uint64 width * uint64 height * uint64 sizeof<'T>
    let mutable ptr = 0n
    do
        check (rt_host_alloc(bytes, &ptr))
        let p = NativeInterop.NativePtr.ofNativeInt<'T> ptr
        for i in 0 .. width * height - 1 do NativeInterop.NativePtr.set p i init
    member self.Pointer = ptr
    member self.Length1 = width
    member self.Length2 = height
    member self.Item
        with get (px : int, py : int) : 'T =
            if px < 0 || px >= width || py < 0 || py >= height then raise (IndexOutOfRangeException())
            NativeInterop.NativePtr.get (NativeInterop.NativePtr.ofNativeInt<'T> ptr) (px * height + py)
        and set (px : int, py : int) (v : 'T) =
            if px < 0 || px >= width || py < 0 || py >= height then raise (IndexOutOfRangeException())
            NativeInterop.NativePtr.set (NativeInterop.NativePtr.ofNativeInt<'T> ptr) (px * height + py) v
    interface IDisposable with
        member self.Dispose () = if ptr <> 0n then (rt_host_free ptr; ptr <- 0n)
