// Native.fs — P/Invoke surface of librt_hip.so (include/rt_abi.h, RT_ABI_VERSION 7) for the reference's F# host.
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
let RtAbiVersion = 7

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
// optional: pin the frame buffers for DMA (the host keeps a GCHandle on them for as long as they are registered)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_host_register(nativeint ptr, uint64 bytes)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_host_unregister(nativeint ptr)
// frameBuffer : Vector4[,] and depthBuffer : float32[,] exactly as Scene.fs:28-29 declares them (Array2D.create width
// height: element [px,py] at px*height + py, the layout the library writes); blittable element types are pinned.
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_render(nativeint scene, RtCamera& cam, RtParams& p, Vector4[,] rgba, float32[,] depth,
                     uint64[,] objectId, float32& maxDepth, uint64& rayCount)
[<DllImport("rt_hip", CallingConvention = CallingConvention.Cdecl)>]
extern int rt_render_multi(nativeint[] scenes, int32 nscenes, RtCamera& cam, RtParams& p, Vector4[,] rgba,
                           float32[,] depth, uint64[,] objectId, float32& maxDepth, uint64& rayCount)
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
