// SceneGpu.fs — the drop-in for Scene.renderScene (Scene.fs:117-121) on MI355X.
//
// `SceneGpu` keeps the state of the reference's `Scene` type under the same names (frameBuffer, depthBuffer,
// maxFrameBufferDepth, Scene.fs:28-30) and the same constants (Scene.fs:18-43), so saveFrameBuffer / saveDepthBuffer
// (Scene.fs:123-144) can be pasted below unchanged.  A maintainer who prefers to keep one type adds the body of
// `renderScene` to Scene.fs as `member self.renderSceneGpu ()` instead; the fields it touches are the same.
// The reference's own renderScene remains the CPU path: the library has none.
module Raytracer.SceneGpu

open System
open System.Numerics
open Raytracer.Camera
open Raytracer.Surface
open Raytracer.Geometry.Sphere
open Raytracer.Geometry.Plane
open Raytracer.SceneDefinitions
open Raytracer.Native

/// Surface -> rt_surface (what `scene_spheres |> Array.ofList` feeds, Scene.fs:33)
let flatten (s : Surface) : RtSurface =
    let mutable r = RtSurface()
    r.id <- s.ID
    r.albedo <- s.Material.Albedo
    r.emission <- s.Material.Emmitance
    match s with
    | :? Lambertian -> r.kind <- 1
    | :? Metal as m -> r.kind <- 2; r.fuzz <- m.Fuzz                    // clamped again inside, Surface.fs:88
    | :? Dielectric as d -> r.kind <- 3; r.ior <- d.RefractiveIndex
    | _ -> r.kind <- 0                                                  // NoSurface
    match s.Geometry with
    | :? Sphere as sp ->
        r.geom <- 0
        r.g0 <- Vector4(sp.Center, sp.Radius)
    | :? Plane as pl ->
        r.geom <- 1
        r.g0 <- Vector4(pl.Plane.Normal, pl.Plane.D)
        match pl.Center, pl.Width, pl.Height with
        | Some c, Some w, Some h ->
            r.g1 <- Vector4(1.0f, c.X, c.Y, c.Z)
            r.g2 <- Vector4(w, h, 0.0f, 0.0f)
        | _ -> ()
    | _ -> failwith "unsupported geometry"
    r

type SceneGpu (devices : int list) =

    let width = 800
    let height = 640
    let samplesPerPixel = 1
    let maxTraceDepth = 5us
    let backgroundColor = Vector3.Zero
    // Scene.fs:28-29 `Array2D.create width height ...`, in pinned memory of the library: same `.[px,py]` indexing
    let frameBuffer = new PinnedArray2D<Vector4>(width, height, Vector4.Zero)
    let depthBuffer = new PinnedArray2D<float32>(width, height, System.Single.MaxValue)
    let objectIds = new PinnedArray2D<uint64>(width, height, 0UL)
    let mutable maxFrameBufferDepth = 0.0f
    let mutable rayCount = 0UL
    let surfaces : (Surface array) = scene_spheres |> Array.ofList

    let cameraOriginWS = Vector3(-3.0f,6.0f,15.0f)
    let lookAt = Vector3(-1.0f,1.0f,-10.0f)
    let viewMatrix = WorldToCamera cameraOriginWS lookAt Vector3.UnitY
    let cameraWS = CameraToWorld viewMatrix
    let fov = MathF.PI/4.0f

    // uploaded once: flattened SoA scene (+ BVH for large scenes) in the HBM of every device used
    let flat = surfaces |> Array.map flatten
    let handles =
        do if rt_abi_version() <> RtAbiVersion then failwith "librt_hip.so: ABI version mismatch"
        devices |> List.map (fun dev ->
            let mutable h = 0n
            check (rt_scene_create(flat, int64 flat.Length, dev, &h))
            h) |> Array.ofList

    new () = new SceneGpu([0])

    member self.FrameBuffer = frameBuffer
    member self.DepthBuffer = depthBuffer
    member self.MaxFrameBufferDepth = maxFrameBufferDepth
    member self.RayCount = rayCount

    /// Scene.renderScene (Scene.fs:117-121): one call; several devices split the frame by interleaved columns
    member self.renderScene () =
        let mutable cam = RtCamera()
        cam.camToWorld <- cameraWS
        cam.lookAt <- lookAt
        cam.fov <- fov
        let mutable p = RtParams()
        p.width <- width; p.height <- height; p.spp <- samplesPerPixel; p.maxTraceDepth <- int32 maxTraceDepth
        p.background <- backgroundColor
        p.jitter <- 1; p.seed <- 1UL
        p.xBegin <- 0; p.xEnd <- width; p.xStep <- 1
        if handles.Length = 1 then
            check (rt_render(handles.[0], &cam, &p, frameBuffer.Pointer, depthBuffer.Pointer, objectIds.Pointer, &maxFrameBufferDepth, &rayCount))
        else
            check (rt_render_multi(handles, handles.Length, &cam, &p, frameBuffer.Pointer, depthBuffer.Pointer, objectIds.Pointer, &maxFrameBufferDepth, &rayCount))

    interface IDisposable with
        member self.Dispose () =
            handles |> Array.iter rt_scene_destroy
            (frameBuffer :> IDisposable).Dispose ()
            (depthBuffer :> IDisposable).Dispose ()
            (objectIds :> IDisposable).Dispose ()
