/* rt_abi.h — C ABI of the MI355X render loop (librt_hip.so).
 *
 * This is the drop-in boundary for the reference's `Scene.renderScene : unit -> unit`
 * (raytracer/Scene.fs:117-121).  The reference has no FFI of its own; the seam is the
 * set of values renderScene closes over and the three buffers it fills:
 *
 *   inputs   surfaces           raytracer/Scene.fs:33   (Surface array, SceneDefinitions.fs:119)
 *            cameraWS, lookAt   raytracer/Scene.fs:36-41
 *            fov                raytracer/Scene.fs:43
 *            width,height,spp   raytracer/Scene.fs:18-20
 *            maxTraceDepth      raytracer/Scene.fs:31
 *            backgroundColor    raytracer/Scene.fs:32
 *   outputs  frameBuffer        raytracer/Scene.fs:28   Vector4[width,height]  -> rgba
 *            depthBuffer        raytracer/Scene.fs:29   float32[width,height]  -> depth
 *            maxFrameBufferDepth raytracer/Scene.fs:30                          -> max_depth
 *
 * All structs are blittable (an F# host passes them with [<DllImport>]).
 * Vector3 = 3 packed f32, Matrix4x4 = 16 f32 row-major M11..M44 (System.Numerics layout).
 * Buffers are indexed like `Array2D.create width height` (Scene.fs:28): linear = px*height + py.
 *
 * Conventions: every entry point returns 0 on success and a negative RT_E_* code on failure,
 * never throws across the boundary; rt_last_error() returns a thread-local message.
 * There is NO CPU fallback in this library: without a HIP device every call fails with RT_E_NO_DEVICE.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------ */
#define RT_OK              0
#define RT_E_INVALID      -1   /* bad argument (null pointer, size, enum value)              */
#define RT_E_NO_DEVICE    -2   /* no HIP device / device index out of range                  */
#define RT_E_HIP          -3   /* a HIP runtime call failed (message has the HIP error)      */
#define RT_E_OVERFLOW     -4   /* a wave's record pool overflowed: the proven bound of a linear-scan scene (a bug), or the
                                  heuristic capacity of a BVH scene — the frame is invalid; see rt_scene_status         */
#define RT_E_UNSUPPORTED  -5   /* valid request this build does not implement                */
#define RT_E_RANGE        -6   /* a radiance contribution was NaN, infinite or >= 2^31 in magnitude
                                  (the pixel sums are 32.32 fixed point); it was dropped, the image is invalid */

/* ---- surface kinds: the Surface subclasses of raytracer/Surface.fs ---------------------- */
#define RT_KIND_NOSURFACE  0   /* Surface.fs:44-47   0 samples, radiance = Emitted           */
#define RT_KIND_LAMBERTIAN 1   /* Surface.fs:60-83   SampleCount 4 (overridable)             */
#define RT_KIND_METAL      2   /* Surface.fs:85-109  SampleCount 1                           */
#define RT_KIND_DIELECTRIC 3   /* Surface.fs:112-181 1 or 2 deterministic samples            */

/* ---- geometry kinds: the Hitable subclasses of raytracer/Geometry ----------------------- */
#define RT_GEOM_SPHERE     0   /* Geometry/Sphere.fs:8                                       */
#define RT_GEOM_PLANE      1   /* Geometry/Plane.fs:12                                       */
#define RT_GEOM_TRIANGLE   2   /* not in the reference (Geometry/AABB.fs is empty); obeys the
                                  Hitable contract of Geometry/Core.fs:26-43: Moeller-Trumbore
                                  hit in f32 whose cross and dot products are fused
                                  multiply-adds (x*y - z*w = fma(x,y,-(z*w)); a.b =
                                  fma(az,bz,fma(ay,by,ax*bx))), accepted when t > TMin,
                                  constant normal
                                  Normalize((v1-v0) x (v2-v0)), never flipped.  Scenes with
                                  triangles or more than 64 surfaces are traced through a BVH
                                  built at rt_scene_create; results equal the linear scan's.  */

/* One `Surface` (Surface.fs:15) flattened: kind + material (Material.fs:9-18) + geometry.
 *   g[] for RT_GEOM_SPHERE  : center.xyz, radius                       (Sphere.fs:8)
 *   g[] for RT_GEOM_PLANE   : normal.xyz, d, has_center(0/1), center.xyz, width, height
 *                             (System.Numerics.Plane + the three options of Plane.fs:12)
 *   g[] for RT_GEOM_TRIANGLE: v0.xyz, v1.xyz, v2.xyz
 * sample_count: 0 = the kind's default (NoSurface 0, Lambertian rt_params.lambertian_samples,
 *   Metal 1); 1..4 overrides `SampleCount` (Surface.fs:20) for Lambertian and Metal. Ignored
 *   for Dielectric, whose GenerateSamples override fixes 1 or 2 (Surface.fs:168-181).        */
typedef struct rt_surface {
    int32_t  kind;
    int32_t  geom;
    uint64_t id;            /* Surface.ID (Surface.fs:10,25); 0 is reserved for "miss"        */
    float    albedo[3];     /* Material.Albedo                                               */
    float    emission[3];   /* Material.Emmitance                                            */
    float    fuzz;          /* Metal      (clamped to [0,1] as Surface.fs:88)                */
    float    ior;           /* Dielectric refractiveIndex                                    */
    int32_t  sample_count;
    float    g[12];
} rt_surface;

/* Camera state renderScene closes over (Scene.fs:36-43).  cam_to_world is what
 * `CameraToWorld (WorldToCamera origin lookAt UnitY)` returns (Camera.fs:41-45): the host
 * computes CreateLookAt + Invert.  look_at is the *target point* the reference normalises and
 * dots with the primary direction (Scene.fs:37,76).                                           */
typedef struct rt_camera {
    float cam_to_world[16];
    float look_at[3];
    float fov;              /* radians, Scene.fs:43                                          */
} rt_camera;

/* Render constants (Scene.fs:18-32) plus what the build adds: the counter-based RNG seed, the
 * deterministic-pinhole switch and the pixel-column partition used for multi-GPU.
 * Columns rendered: px = x_begin + k*x_step for k = 0,1,... while px < x_end
 * (x_begin=0,x_end=width,x_step=1 is the whole frame).  Output buffers are COMPACT over the
 * rendered columns: element (k, py) lives at k*height + py; with the whole-frame settings
 * that is exactly px*height + py.                                                            */
typedef struct rt_params {
    int32_t  width, height;       /* Scene.fs:18-19                                          */
    int32_t  spp;                 /* samplesPerPixel, Scene.fs:20                             */
    int32_t  max_trace_depth;     /* maxTraceDepth,  Scene.fs:31  (0..28)                     */
    float    background[3];       /* backgroundColor, Scene.fs:32                             */
    int32_t  jitter;              /* 0 = deterministic pinhole (pertrube offset 0), 1 = on    */
    uint64_t seed;                /* Philox4x32-10 key                                        */
    int32_t  lambertian_samples;  /* Lambertian SampleCount; 0 = 4 (Surface.fs:63); 1..4      */
    int32_t  x_begin, x_end, x_step;
    int32_t  waves_per_cu;        /* tuning: cap on resident persistent waves per CU; 0 = all that fit */
    int32_t  pool_blocks;         /* tuning: 64-record pool blocks kept in LDS per wave (rest spills to HBM); 0 = auto */
    int32_t  estimator;           /* RT_ESTIMATOR_BRANCHING (0, the reference's renderScene) or RT_ESTIMATOR_SINGLE_PATH */
    int32_t  bvh_ring;            /* tuning/testing: traversal-stack entries kept in LDS, 0 = by tree size, 8 or 16
                                     (selects the kernel variant a BVH scene runs; results do not depend on it)   */
    int32_t  bvh_min_active;      /* tuning/testing: lanes still walking below which a wave refills; 0 = 24 (20 for scenes that scan 8 or more loose surfaces), 1..64 */
    int32_t  two_pass;            /* tuning/testing: 0 = frames of at most 2^19 pixel-samples are rendered in two passes
                                     (primaries, then their subtrees shared among all waves), 1 = always one pass;
                                     the image does not depend on it.  rt_render_device_noalloc renders such a frame in
                                     one pass when the list of primary hit records (up to 4 x 64 B per pixel-sample)
                                     exceeds half of what is left of its scratch (640x480x1 already does)             */
} rt_params;

/* Estimators.  BRANCHING is what Scene.rayTrace computes (Scene.fs:51-72): every hit spawns SampleCount children
 * (Lambertian 4, Metal 1, Dielectric the deterministic 1-or-2 split of its GenerateSamples override,
 * Surface.fs:168-181).  SINGLE_PATH is SURVEY.md 8(f4): one child per hit for every kind — Lambertian and Metal
 * with SampleCount 1 through the default GenerateSamples (Surface.fs:37-41), Dielectric through its stochastic
 * `Scatter` (Surface.fs:155-166: reflect when RandomFloat <= reflectProb, else refract; shading = BRDF*(1/PDF) =
 * Albedo), i.e. the accumulator form of raytracer/notes.txt:21-54 with one ray per level.  Same expectation,
 * more variance per sample, cost linear in max_trace_depth (what makes 16 bounces tractable).                  */
#define RT_ESTIMATOR_BRANCHING   0
#define RT_ESTIMATOR_SINGLE_PATH 1

typedef struct rt_scene rt_scene;   /* opaque: device-resident SoA scene + workspace          */

/* Replaces `scene_spheres |> Array.ofList` (Scene.fs:33): flattens the surfaces to SoA,
 * precomputes the per-plane rectangle constants of Plane.fs:19-30 in the reference's f32
 * operation order, and uploads everything once to the HBM of `device`.                        */
int  rt_scene_create(const rt_surface* surfaces, int64_t n, int32_t device, rt_scene** out);

/* Replaces `Scene.renderScene ()` (Scene.fs:117-121).  Synchronous; host output buffers.
 * rgba  [ncols*height*4]  Vector4(sqrt(avg),1)            (Scene.fs:114)
 * depth [ncols*height]    t of the first sample's primary hit, FLT_MAX on a miss (Scene.fs:29,80)
 * object_id [ncols*height] Surface.ID of that hit, 0 on a miss; may be NULL
 * max_depth               maxFrameBufferDepth over the rendered columns (Scene.fs:47-48); may be NULL
 * ray_count               number of findClosestIntersection invocations (Surface.fs:49); may be NULL */
int  rt_render(rt_scene* scene, const rt_camera* cam, const rt_params* params,
               float* rgba, float* depth, uint64_t* object_id,
               float* max_depth, uint64_t* ray_count);

/* Same render, but asynchronous on `stream` (a hipStream_t; NULL = default stream) with DEVICE
 * output buffers, so a caller that owns device memory (torch, RCCL gather of the framebuffer)
 * never crosses PCIe.  d_stats points at RT_STATS_WORDS device u64: [0] ray count, [1] bit pattern
 * of max depth (f32) in the low word, [2] 64-ray loop iterations over all waves, [3] / [4] / [5] child
 * rays scattered by Lambertian / Metal / Dielectric surfaces (ray count minus these = primary rays),
 * [6] summed wave lifetimes in shader clocks, [7] waves, [8] BVH nodes visited, [9] BVH primitives
 * tested (0 for linear-scan scenes), [10] / [11] BVH traversal rounds with an inner-node / a leaf step,
 * [12..15] shader clocks summed over waves in the phases generate / plane scan / BVH traversal / shade+flush
 * (filled by builds with -DRT_PHASE_TIMING, else 0), [16], [17] unused, [18] the longest wave lifetime in shader clocks
 * ([6] / [7] is the mean), [19] unused; may be NULL.  Words [2]..[5] and [8]..[11] stay 0 unless rt_scene_counters(scene, 1)
 * was called (below).
 * No counterpart in the reference.                                                            */
#define RT_STATS_WORDS 20
int  rt_render_device(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                      void* stream, float* d_rgba, float* d_depth, uint64_t* d_object_id,
                      uint64_t* d_stats);

/* rt_render_device for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  NO counterpart in the reference.  Same arguments, same results bit for bit, but everything
 * the kernels need beyond the inputs and outputs — the per-pixel radiance sums, the record-pool spill area, the
 * traversal-stack backing store — lives in one static device array of RT_NOALLOC_SCRATCH_BYTES per GPU (held from
 * library load until the process ends) instead of buffers the scene grows with hipMalloc; the call performs no
 * hipMalloc/hipFree.  Limits: at most RT_NOALLOC_MAX_PIXELS rendered pixels per call (ncols*height), else
 * RT_E_UNSUPPORTED; the number of persistent waves shrinks until their spill areas fit (slower for deep traces, same
 * image); a small frame whose two-pass list does not fit is rendered in one pass (rt_params.two_pass).  All calls on
 * one GPU share that GPU's array: two calls that overlap in time on one GPU race — serialise them (one stream, or
 * events).                                                                                                          */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
#define RT_NOALLOC_MAX_PIXELS    (1 << 20)
int  rt_render_device_noalloc(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                              void* stream, float* d_rgba, float* d_depth, uint64_t* d_object_id,
                              uint64_t* d_stats);

/* `Scene.renderScene ()` (Scene.fs:117-121) on several GPUs of one node from ONE host thread:
 * scenes[i] (the same surfaces uploaded to any device, normally device i) renders the interleaved columns
 * i, i+n, ... of the whole frame on its own stream; each device then stores its block into the frame held in
 * the HBM of scenes[0]'s device (peer stores over xGMI), which is copied to the host buffers once.
 * max_depth is the maximum and ray_count the sum over the shares (Scene.fs:47-48).  Output layout and meaning as
 * rt_render; params->x_begin/x_end/x_step must describe the whole frame (0, width, 1).  With n = 1 the result is
 * rt_render's bit for bit; for any n the frame is identical (integer radiance sums).                             */
int  rt_render_multi(rt_scene* const* scenes, int32_t nscenes, const rt_camera* cam, const rt_params* params,
                     float* rgba, float* depth, uint64_t* object_id, float* max_depth, uint64_t* ray_count);

/* findClosestIntersection (Surface.fs:49-58) as a batch query over host arrays: for ray i with
 * origin origins[3i..] and (unit) direction directions[3i..], index[i] = array index of the
 * closest acceptable surface (-1 = none) and t[i] = its ray parameter (FLT_MAX if none).
 * Directions must be unit vectors as f32 arithmetic makes them (| |D|^2 - 1 | <= 2.4e-7), as every ray of the
 * reference is (Ray normalises: Core.fs:21, Numerics.fs:12-16): for a scene traced through the BVH, the guarantee that the tree offers every sphere
 * the scan's test would accept is derived for such directions.                                 */
int  rt_find_closest(rt_scene* scene, int64_t nrays, const float* origins, const float* directions,
                     float* t, int32_t* index);

/* After a render on `scene` has completed: 0, RT_E_OVERFLOW if any wave's ray pool overflowed or RT_E_RANGE if a
 * radiance contribution did not fit the fixed-point sums (results are then invalid).  rt_render and rt_render_multi
 * check this themselves and render an overflowed BVH frame again.
 * Recovery for callers of rt_render_device / rt_render_device_noalloc: a BVH scene's pool capacity is a heuristic
 * (rays finish out of order there, so the linear scan's proven bound of 64 records per level does not hold).  When
 * this call reports RT_E_OVERFLOW for a BVH scene it has DOUBLED the handle's capacity: render the frame again on the
 * same handle (repeat until 0; the capacity stays raised for the handle's lifetime).  For a linear-scan scene
 * RT_E_OVERFLOW is a bug.  (Test-only: with RT_TEST_TINY_SPILL=1 in the environment a BVH handle starts with no
 * spill capacity at all, so that this path runs: tests/test_gpu_overflow.py.)                                    */
int  rt_scene_status(rt_scene* scene);

/* Which words of d_stats the frames of `scene` fill.  level 0 (the default): the ray count [0], the largest depth [1] and
 * the wave clocks [6], [7], [18] — what `Scene.renderScene` itself produces (Scene.fs:47-48) plus what it costs; level 1:
 * every word (rays by kind, loop iterations, BVH visits and rounds: the inputs of bench.py's flop model).  The image, the
 * depth buffer, the IDs and words [0], [1] do not depend on the level; the extra counters cost a frame 1-3 % (they are
 * wave-uniform scalars carried through the whole render loop), which is why a frame does not pay for them unless asked.
 * Takes effect from the next render of the handle.  NO counterpart in the reference.                              */
int  rt_scene_counters(rt_scene* scene, int level);

/* Optional: pinned host memory for the frame buffers (hipHostMalloc / hipHostFree).  rt_render and rt_render_multi copy
 * a frame into buffers that live in such a block by DMA instead of through the runtime's pageable path (640x480x1:
 * rgba + depth in 0.13 instead of 0.45 ms).  NO counterpart in the reference, whose `frameBuffer` / `depthBuffer`
 * (Scene.fs:28-29) are managed arrays; fsharp/SceneGpu.fs keeps them in such a block behind the same `.[px,py]`
 * indexer.  Any other host memory keeps working as before.  (Registering memory the caller already owns —
 * hipHostRegister — is refused by the driver of the machines this was built on, so it is not offered.)          */
int  rt_host_alloc(uint64_t bytes, void** out);
void rt_host_free(void* ptr);

void rt_scene_destroy(rt_scene* scene);

/* Thread-local text of the last error on this thread ("" if none). Never NULL.               */
const char* rt_last_error(void);

/* ABI version: bumped whenever a struct layout or signature changes.                         */
int  rt_abi_version(void);
#define RT_ABI_VERSION 8

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
