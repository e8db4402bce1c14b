// rt_hip.hip — librt_hip.so: the C ABI of include/rt_abi.h over the HIP render kernel.
//
// Host side of the boundary that replaces Scene.renderScene (raytracer/Scene.fs:117-121): scene
// flattening (Scene.fs:33 `scene_spheres |> Array.ofList`), the per-plane constants of
// Plane.PointLiesInRectangle (Geometry/Plane.fs:19-30) hoisted out of the per-ray test, camera
// constants (Camera.fs:20-45, Scene.fs:100-102), launch geometry.  No CPU rendering path exists
// in this library: without a HIP device every entry point returns RT_E_NO_DEVICE.
#include "rt_kernels.h"
#include "rt_bvh_build.h"

#include "../../include/rt_abi.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

using namespace rt;
static_assert(kStatsWords == RT_STATS_WORDS, "statistics words");
// layout of a scene's control block (rt_scene::d_ctl), in bytes
constexpr size_t kCtlCounter = 256, kCtlCounterBytes = sizeof(unsigned long long) * kCounterParts * kCounterStride;
constexpr size_t kCtlCounter2 = kCtlCounter + kCtlCounterBytes, kCtlStats = kCtlCounter2 + kCtlCounterBytes;
constexpr size_t kCtlBytes = kCtlStats + sizeof(unsigned long long) * kStatsParts * kStatsWords;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(RT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

// ---- System.Numerics on the host, float32, one rounding per operation ----------------------
struct H3 { float x, y, z; };
inline float hdot(H3 a, H3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline H3 hdivs(H3 a, float s) { float inv = 1.0f / s; return { a.x * inv, a.y * inv, a.z * inv }; }
inline H3 hnormalize(H3 a) { return hdivs(a, sqrtf(hdot(a, a))); }

struct M4 { float m[4][4]; };
inline M4 mzero() { M4 r; memset(&r, 0, sizeof r); return r; }
inline M4 mmul(const M4& a, const M4& b)
{
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = ((a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j]) + a.m[i][3] * b.m[3][j];
    return r;
}
// Numerics.fs:51-55,59,63-76: R = I + [w]x + [w]x^2 * (1/(1+cos|w|)), w = a x b  (as written)
M4 rotation_between_unit_vectors(H3 a, H3 b)
{
    H3 w = hcross(a, b);
    M4 wx = mzero();
    wx.m[0][1] = -w.z; wx.m[0][2] = w.y;
    wx.m[1][0] = w.z;  wx.m[1][2] = -w.x;
    wx.m[2][0] = -w.y; wx.m[2][1] = w.x;
    M4 wx2 = mmul(wx, wx);
    float s = 1.0f / (1.0f + cosf(sqrtf(hdot(w, w))));
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = (((i == j) ? 1.0f : 0.0f) + wx.m[i][j]) + wx2.m[i][j] * s;
    return r;
}
// Vector4.Transform(Vector4(v, 0), M).xyz
inline H3 transform_w0(H3 v, const M4& M)
{
    return { ((v.x * M.m[0][0] + v.y * M.m[1][0]) + v.z * M.m[2][0]) + 0.0f * M.m[3][0],
             ((v.x * M.m[0][1] + v.y * M.m[1][1]) + v.z * M.m[2][1]) + 0.0f * M.m[3][1],
             ((v.x * M.m[0][2] + v.y * M.m[1][2]) + v.z * M.m[2][2]) + 0.0f * M.m[3][2] };
}

int default_samples(int kind, int sample_count, int lambertian_default)
{
    switch (kind) {
    case RT_KIND_LAMBERTIAN: return sample_count > 0 ? sample_count : lambertian_default;
    case RT_KIND_METAL:      return sample_count > 0 ? sample_count : 1;
    case RT_KIND_DIELECTRIC: return 2;
    default:                 return 0;
    }
}

}  // namespace

// ---- the allocation-free entry point's scratch (include/rt_abi.h rt_render_device_noalloc) ------------------
// One array per GPU the library's code object is loaded on, held until the process ends.  `scratch` is a
// host-side stand-in for the symbol, never a device address: the address on each device is looked up once
// (scratch_of) and only that pointer is handed to kernels, memsets and copies.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = { nullptr };

// Address of this device's copy of `scratch` (the device must be current).  Looked up once per device.
int scratch_of(int device, unsigned char** out)
{
    if (device < 0 || device >= kMaxDevices) return fail(RT_E_UNSUPPORTED, "device %d beyond the %d this library tracks", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) return fail(RT_E_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
        g_scratch_ptr[device] = static_cast<unsigned char*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return RT_OK;
}

}  // namespace

struct rt_scene {
    int device = 0;
    int64_t n = 0;
    hipStream_t stream = nullptr;              // rt_render / rt_find_closest / rt_render_multi work of this scene
    hipStream_t copy_stream = nullptr;         // rt_render: depth and object IDs leave for the host while the frame still renders
    hipEvent_t ev_primaries = nullptr;         // recorded behind the launch that writes depth and object IDs
    GeomRec* d_geom = nullptr;
    ShadeRec* d_shade = nullptr;
    // One control block, cleared by one memset per frame (a fill is a launch: a 640x480x1 frame of 0.7 ms paid for six):
    // status word | hand-out counters of the frame or its first pass | those of the second pass | partial statistics
    unsigned char* d_ctl = nullptr;
    int* d_status = nullptr;                   // -> d_ctl
    unsigned long long* d_counter = nullptr;   // -> d_ctl + kCtlCounter: work counters of the persistent waves
    unsigned long long* d_counter2 = nullptr;  // -> d_ctl + kCtlCounter2: those of the second pass of a small frame
    unsigned long long* d_statpart = nullptr;  // -> d_ctl + kCtlStats: kStatsParts partial statistics arrays (folded by resolve_kernel)
    unsigned long long* d_accum = nullptr;     // per-pixel 32.32 radiance sums (grown on demand)
    size_t accum_bytes = 0;
    uint32_t* d_spill = nullptr;               // pool blocks spilled by the persistent waves (grown on demand)
    size_t spill_bytes = 0;
    uint32_t* d_gstack = nullptr;              // backing store of the BVH traversal stacks (grown on demand)
    size_t gstack_bytes = 0;
    unsigned char* scratch = nullptr;          // this device's copy of the static scratch array (allocation-free entry)
    int num_cus = 0;
    int64_t nloose = 0;                        // entries of d_geom (all surfaces, or only the planes of a BVH scene)
    bool use_bvh = false;
    int spill_factor = 1;                      // BVH scenes: multiple of the in-order pool bound the spill area holds
    float4* d_nodes = nullptr;                 // the BVH: nodes, then the primitive records (one allocation, 32-bit offsets)
    uint32_t pdata_off = 0;                    // byte offset of the records
    int nnodes = 0, bvh_depth = 0;
    float lin_s = 0.0f, lin_far = 3.402823466e38f;
    float cone_k = 0.0f, tree_rmin = 0.0f, tree_rmax = 0.0f, tree_reach = 0.0f;    // spheres in the tree: how the traversal widens the ray (rt_bvh.h closest_hit_bvh)
    int64_t nprims = 0;                        // primitives in the BVH
    int counters = 0;                 // rt_scene_counters: 1 = the frames of this scene fill every word of d_stats
    int occ_smem[64] = { 0 };         // (0 = not queried yet) LDS size the cached occupancy below was queried for (per kernel variant)
    int occ_per_cu[64] = { 0 };
    unsigned char* d_list = nullptr;    // small frames: hit records of the primary pass (grown on demand)
    size_t list_bytes = 0;
    float* d_tmp = nullptr;             // rt_render / rt_find_closest staging (grown on demand)
    size_t tmp_bytes = 0;
    float* d_frame = nullptr;           // rt_render_multi: the gathered frame on the root scene's device
    size_t frame_bytes = 0;
    uint64_t* h_pin = nullptr;          // 64 bytes of pinned host memory: ray count, max depth bits, status word of a frame
                                        // (truly asynchronous copies: rt_render fetches them with the frame, one wait)
    bool peer_ok[kMaxDevices] = { false };   // rt_render_multi: this scene's device may store to device d (checked and enabled once)
};

namespace {

int ensure_buffer(void** ptr, size_t* have, size_t bytes)
{
    if (*have >= bytes) return RT_OK;
    if (*ptr) { HIP_TRY(hipFree(*ptr)); *ptr = nullptr; *have = 0; }
    HIP_TRY(hipMalloc(ptr, bytes));
    *have = bytes;
    return RT_OK;
}
int ensure_tmp(rt_scene* sc, size_t bytes) { return ensure_buffer(reinterpret_cast<void**>(&sc->d_tmp), &sc->tmp_bytes, bytes); }

int check_params(const rt_camera* cam, const rt_params* p)
{
    if (!cam || !p) return fail(RT_E_INVALID, "null camera or params");
    if (p->width <= 0 || p->height <= 0 || p->spp <= 0) return fail(RT_E_INVALID, "width, height and spp must be positive");
    if (p->max_trace_depth < 0 || p->max_trace_depth > 28) return fail(RT_E_INVALID, "max_trace_depth must be in 0..28");
    if (p->lambertian_samples < 0 || p->lambertian_samples > 4) return fail(RT_E_INVALID, "lambertian_samples must be in 0..4");
    if (p->x_step <= 0 || p->x_begin < 0 || p->x_end > p->width || p->x_begin > p->x_end)
        return fail(RT_E_INVALID, "bad column range x_begin/x_end/x_step");
    if (p->waves_per_cu < 0 || p->waves_per_cu > 32) return fail(RT_E_INVALID, "waves_per_cu must be in 0..32");
    if (p->pool_blocks < 0 || p->pool_blocks > 32) return fail(RT_E_INVALID, "pool_blocks must be in 0..32");
    if (p->estimator != RT_ESTIMATOR_BRANCHING && p->estimator != RT_ESTIMATOR_SINGLE_PATH) return fail(RT_E_INVALID, "estimator must be 0 or 1");
    if (p->bvh_ring != 0 && p->bvh_ring != 8 && p->bvh_ring != 16) return fail(RT_E_INVALID, "bvh_ring must be 0, 8 or 16");
    if (p->bvh_min_active < 0 || p->bvh_min_active > 64) return fail(RT_E_INVALID, "bvh_min_active must be in 0..64");
    if (p->two_pass != 0 && p->two_pass != 1) return fail(RT_E_INVALID, "two_pass must be 0 (automatic) or 1 (off)");
    return RT_OK;
}

// the sphere test's worst rounding error per squared distance (19 * 2^-24 = 1.13e-6: rt_bvh.h closest_hit_bvh), rounded up
constexpr float kSphereTestError = 1.2e-6f;
// the triangle test answers as if the ray were shifted sideways by up to 2^-24 (4 + 3.5 / sin(angle between the ray and an
// edge)) |O - v0| (rt_bvh.h): this slope covers every ray that makes more than 3 degrees with the edges
constexpr float kTriangleConeSlope = 4e-6f;
BvhRef bvh_ref(const rt_scene* sc)
{
    BvhRef b;
    b.nodes = sc->d_nodes; b.pdata_off = sc->pdata_off; b.nnodes = sc->nnodes;
    b.cone_k = sc->cone_k; b.rmin = sc->tree_rmin; b.rmax = sc->tree_rmax; b.reach = sc->tree_reach;
    b.lin_s = sc->lin_s; b.lin_far = sc->lin_far;
    return b;
}

// Where a launch keeps what is neither input nor output: the scene's own buffers, grown on demand (hipMalloc), or a
// carve-up of the static scratch array (no allocation; the grid shrinks until its spill and stack areas fit).
struct Workspace {
    bool fixed = false;
    unsigned char* base = nullptr;   // fixed: first free byte of the scratch array
    size_t avail = 0;                // fixed: bytes from there to its end
};

// refill threshold of the BVH kernels (lanes still walking); a wave needs a free sample slot to start a primary ray,
// so the default must not exceed the sample slots of any variant (the kernel makes progress either way, see
// trace_kernel "leave early", but a larger value only costs occupancy of the lanes)
constexpr int kMinActiveDefault = 24;
constexpr int kMinActiveManyLoose = 20;   // scenes that scan 8 or more loose surfaces per ray
static_assert(kMinActiveDefault <= slots_of(true, 8) && kMinActiveDefault <= slots_of(true, 16), "refill threshold above the sample slots");

#ifndef RT_SAMPLE_MAJOR_BELOW
#define RT_SAMPLE_MAJOR_BELOW 16384
#endif
#ifndef RT_BVH_GROUP_SHIFT
#define RT_BVH_GROUP_SHIFT 2
#endif
constexpr int kBvhGroupShift = RT_BVH_GROUP_SHIFT;
#ifndef RT_LIN_GROUP_SHIFT
#define RT_LIN_GROUP_SHIFT 2
#endif
constexpr int kLinGroupShift = RT_LIN_GROUP_SHIFT;
constexpr size_t kStripTreeBytes = 32u << 20;   // trees below this stay in one strip (launch of a sample-major frame)
constexpr int kDeepTrace = 8;           // maxTraceDepth above which the samples of a pixel are not kept together (launch_trace)
constexpr long long kSampleMajorBelow = RT_SAMPLE_MAJOR_BELOW;   // pixel-samples per wave below which a one-pass frame is handed out sample-major

// PHASE: 0 = one launch per frame; 1, 2 = the two passes of a small frame; 3 = one launch, handed out sample-major (trace_kernel)
// SINGLE: the single-path estimator (rt_params.estimator)
template <bool NODE64, bool BVH, int RING, int PHASE, bool SINGLE, bool COUNT>
int launch_trace(rt_scene* sc, ColdArgs& c, const RenderArgs& a, int waves_per_cu,
                 size_t smem, hipStream_t stream, const Workspace& ws)
{
    constexpr int phase = PHASE;
    const void* fn = reinterpret_cast<const void*>(&trace_kernel<NODE64, BVH, RING, PHASE, SINGLE, COUNT>);
    constexpr int NF = 11 + (NODE64 ? 1 : 0) + (BVH ? 1 : 0);
    // the attribute and the occupancy query cost host time: once per (kernel, LDS size) and scene
    constexpr int variant = (NODE64 ? 1 : 0) + ((BVH && RING == 16) ? 2 : 0) + 4 * PHASE + (SINGLE ? 16 : 0) + (COUNT ? 32 : 0);
    int& cached_smem = sc->occ_smem[variant];
    int& cached_per_cu = sc->occ_per_cu[variant];
    {   // hipFuncAttributeMaxDynamicSharedMemorySize belongs to the KERNEL (per device), not to a scene: two scenes that
        // run this instantiation with different pool sizes would otherwise lower each other's limit between a cached
        // query and a launch.  It only ever grows: one maximum per instantiation and device, under a mutex.
        static std::mutex attr_mutex;
        static int attr_max[kMaxDevices] = { 0 };
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (sc->device < 0 || sc->device >= kMaxDevices) return fail(RT_E_UNSUPPORTED, "device %d beyond the %d this library tracks", sc->device, kMaxDevices);
        if (attr_max[sc->device] < (int)smem) {
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_max[sc->device] = (int)smem;
        }
    }
    if (cached_smem != (int)smem) {
        int q = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, fn, 64, smem));
        // The query over-states what is resident: LDS is handed out in granules of 1 280 bytes (measured with
        // tools/launch_ramp.hip: 6 400 B per wave -> 25 waves per CU, 6 656 .. 7 680 B -> 21, 7 936 B -> 18), so the
        // 7 168 B of the linear-scan kernel put 21 waves on a CU, not the 22 the query reports.  A 22nd wave per CU
        // starts when another ends: harmless in a long frame, but the first pass of a small frame (waves of 45 us)
        // then runs as two generations.
        constexpr size_t kLdsPerCu = 160 * 1024, kLdsGranule = 1280;
        const int fit = (int)(kLdsPerCu / ((smem + kLdsGranule - 1) / kLdsGranule * kLdsGranule));
        if (q > fit) q = fit;
        cached_per_cu = q < 1 ? 1 : q;
        cached_smem = (int)smem;
    }
    int per_cu = cached_per_cu;
    if (waves_per_cu > 0 && waves_per_cu < per_cu) per_cu = waves_per_cu;
    // persistent waves: no more than fit on the chip, no more than there are 64-sample refills
    const unsigned long long chunks = (c.total + 63) / 64;
    unsigned long long grid = (unsigned long long)per_cu * (unsigned long long)sc->num_cus;
    if (phase == 0 || phase == 3) {
        // Frames that are small for the chip but too large for the two-pass scheme: with one refill per wave there is
        // no dynamic balancing left and the waves that drew expensive pixels finish alone.  Two thirds as many waves
        // as refills balances better (measured on 640x480x1 before it became a two-pass frame: 13.8 -> 15.0 Grays/s).
        if (grid * 3 > chunks * 2) grid = std::max<unsigned long long>(chunks * 2 / 3, 1ull);
    } else {
        // a pass hands out 32 pixel-samples (1) or 32 list entries (2) at a time: no more waves than hand-outs
        const unsigned long long handouts = phase == 1 ? (c.total + 31) / 32 : (4 * c.total + 31) / 32;
        if (grid > handouts) grid = std::max<unsigned long long>(handouts, 1ull);
    }
    const size_t wave_spill = (size_t)c.max_spill * NF * 64 * sizeof(uint32_t);
    const size_t wave_stack = BVH ? (size_t)kBvhStack * 64 * sizeof(uint32_t) : 0;
    if (ws.fixed) {
        // the image does not depend on the number of waves (integer sums): take as many as the scratch holds
        const unsigned long long fit = ws.avail / (wave_spill + wave_stack);
        const bool shrunk = fit < grid;
        if (shrunk) grid = fit;
        if (shrunk && grid < (unsigned long long)sc->num_cus)
            return fail(RT_E_UNSUPPORTED, "the %zu-byte scratch of rt_render_device_noalloc cannot hold one wave per CU for this frame and trace depth",
                        (size_t)RT_NOALLOC_SCRATCH_BYTES);
        c.spill = reinterpret_cast<uint32_t*>(ws.base);
        c.gstack = BVH ? reinterpret_cast<uint32_t*>(ws.base + (size_t)grid * wave_spill) : nullptr;
    } else {
        int rc = ensure_buffer(reinterpret_cast<void**>(&sc->d_spill), &sc->spill_bytes, (size_t)grid * wave_spill);
        if (rc) return rc;
        c.spill = sc->d_spill;
        c.gstack = nullptr;
        if (BVH) {
            rc = ensure_buffer(reinterpret_cast<void**>(&sc->d_gstack), &sc->gstack_bytes, (size_t)grid * wave_stack);
            if (rc) return rc;
            c.gstack = sc->d_gstack;
        }
    }
    // hand-out order of a sample-major one-pass frame (PHASE 3; rt_kernels.h, primary ray)
    // Samples of one pixel that stay together, G = 1 << group_shift.  Measured (A/B on one box, against the pixel-major
    // order): 800x640x16 linear scan +2.4 / +3.6 / +4.6 / +2.6 % with G = 1 / 2 / 4 / 8 (with G = 1 the tail shrinks
    // from 0.77 to 0.18 ms, but the bulk of the frame gets 6 % slower: the rays of an iteration no longer hit the same
    // kinds of surface, and the three scatter routines diverge); 100k-triangle mesh, 16 spp, depth 5: -1.5 / -0.7 /
    // +0.8 % with G = 1 / 2 / 4 (coherent primaries are worth what the balance gains); 1M spheres, 4 spp, depth 16:
    // +7 / +4.5 / 0 % (trees of thousands of rays under a few pixels).  Hence groups of 4, none for deep traces.
    c.group_shift = 0;
    if (a.max_depth <= kDeepTrace)
        while (c.group_shift < (BVH ? kBvhGroupShift : kLinGroupShift) && a.spp % (2 << c.group_shift) == 0) c.group_shift++;
    hipLaunchKernelGGL((trace_kernel<NODE64, BVH, RING, PHASE, SINGLE, COUNT>), dim3((unsigned)grid), dim3(64), smem, stream, c, a, sc->d_geom, sc->d_shade,
                       bvh_ref(sc));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// A scene goes through the BVH when it has triangles or more spheres than a wave-uniform scan should test.
constexpr int64_t kBvhThreshold = 64;
constexpr int64_t kLooseSpheres = 16;
// Frames of at most this many pixel-samples are rendered in two passes (render_on_stream); the list of primary hit
// records takes up to 4 x 64 bytes per pixel-sample.
constexpr unsigned long long kTwoPassMaxSamples = 1ull << 19;

int columns_of(const rt_params* p) { return (p->x_end - p->x_begin + p->x_step - 1) / p->x_step; }

// The render behind rt_render_device and rt_render_device_noalloc: memsets, trace_kernel, resolve_kernel on `stream`.
// `after_primaries` (optional) is recorded behind the launch that writes depth and object IDs: the first pass of a small
// frame, else the only one.
int render_on_stream(rt_scene* sc, const rt_camera* cam, const rt_params* p, hipStream_t stream,
                     float* d_rgba, float* d_depth, uint64_t* d_object_id, uint64_t* d_stats, bool noalloc,
                     hipEvent_t after_primaries = nullptr)
{
    if (!sc) return fail(RT_E_INVALID, "scene is null");
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (!d_rgba || !d_depth) return fail(RT_E_INVALID, "rgba and depth buffers are required");
    HIP_TRY(hipSetDevice(sc->device));

    const int ncols = columns_of(p);
    if (ncols <= 0) return RT_OK;
    const long long npix = (long long)ncols * p->height;
    if (npix > 0x7fffffffLL) return fail(RT_E_INVALID, "frame too large for one launch");
    const size_t accum_bytes = (size_t)npix * 4 * sizeof(unsigned long long);
    Workspace ws;
    unsigned long long* accum = nullptr;
    if (noalloc) {
        if (npix > (long long)RT_NOALLOC_MAX_PIXELS)
            return fail(RT_E_UNSUPPORTED, "rt_render_device_noalloc renders at most %d pixels per call (%lld asked): use rt_render_device",
                        RT_NOALLOC_MAX_PIXELS, npix);
        if (!sc->scratch) return fail(RT_E_HIP, "the scratch array was not located on device %d", sc->device);
        accum = reinterpret_cast<unsigned long long*>(sc->scratch);
        const size_t used = (accum_bytes + 255) & ~(size_t)255;
        ws.fixed = true; ws.base = sc->scratch + used; ws.avail = (size_t)RT_NOALLOC_SCRATCH_BYTES - used;
    } else {
        rc = ensure_buffer(reinterpret_cast<void**>(&sc->d_accum), &sc->accum_bytes, accum_bytes);
        if (rc) return rc;
        accum = sc->d_accum;
    }

    RenderArgs a; memset(&a, 0, sizeof a);
    ColdArgs c; memset(&c, 0, sizeof c);
    c.accum = accum; c.counter = sc->d_counter; c.depth = d_depth;
    c.ids = reinterpret_cast<unsigned long long*>(d_object_id);
    c.stats = d_stats ? sc->d_statpart : nullptr;
    c.status = sc->d_status;
    c.total = (unsigned long long)npix * (unsigned long long)p->spp;
    c.npix = (unsigned int)npix;
    // multiplier of the hand-out permutation of n pixels: near n/phi, coprime with n
    auto multiplier = [](unsigned long long n) -> unsigned int {
        if (n <= 1) return 0u;
        unsigned long long m = (unsigned long long)((double)n * 0.6180339887498949) | 1ull;
        auto gcd = [](unsigned long long x, unsigned long long y) { while (y) { unsigned long long t = x % y; x = y; y = t; } return x; };
        while (gcd(m, n) != 1) m += 2;
        return (unsigned int)(m % n);
    };
    c.scatter = multiplier((unsigned long long)npix);
    a.nsurf = (int)sc->nloose; c.H = p->height; a.spp = p->spp; a.max_depth = p->max_trace_depth;
    c.jitter = p->jitter ? 1 : 0;
    a.bg_nonzero = (p->background[0] != 0.0f || p->background[1] != 0.0f || p->background[2] != 0.0f) ? 1 : 0;
    c.x_begin = p->x_begin; c.x_step = p->x_step; c.ncols = ncols;
    a.seed_lo = (uint32_t)p->seed; a.seed_hi = (uint32_t)(p->seed >> 32);
    a.lamb = p->lambertian_samples ? p->lambertian_samples : 4;          // Surface.fs:63
    // measured on the 100k-triangle and 1M-sphere scenes (profiles/README.md): refill the wave when fewer than
    // 28 of its lanes still walk the tree (group thresholds of the traversal rounds: rt_bvh.h kBvhInnerMin / kBvhLeafMin)
    // (the longer the scan of loose surfaces, the dearer a half-empty generate / scan / shade pass: with the 12 scanned
    // surfaces of the mixed scene 20 is 1.6 % faster than 28, with the 3 of the mesh and sphere-field scenes 3-5 % slower)
    a.min_active = p->bvh_min_active > 0 ? p->bvh_min_active : (sc->nloose >= 8 ? kMinActiveManyLoose : kMinActiveDefault);
    for (int k = 0; k < 3; k++) c.bg[k] = p->background[k];
    // pool bound: 64 records per record level 0..max_trace_depth (+1 with a non-black background)
    // (DESIGN.md "Pool bound"); the top pool_blocks blocks live in LDS, the rest spills to HBM
    const int levels = p->max_trace_depth + 1 + a.bg_nonzero;
    a.nblk = p->pool_blocks > 0 ? p->pool_blocks : 2;   // measured: 2 blocks (19 waves/CU) beats 3, 4, 6
    if (a.nblk > levels) a.nblk = levels;
    if (a.nblk < 2 && levels >= 2) a.nblk = 2;
    // A BVH scene's rays finish out of order, so records of one level can be pushed while a few suspended rays of
    // the level above are still walking: the in-order bound does not hold and the capacity is a heuristic
    // (2x the bound, never seen exceeded); rt_render doubles spill_factor and renders again on RT_E_OVERFLOW.
    // (+1: the second pass of a small frame puts up to 32 list records on top of a pool that still holds up to 16)
    c.max_spill = (sc->use_bvh ? 2 * sc->spill_factor * levels : levels) + 1;
    // TEST KNOB (tests/test_gpu_overflow.py; include/rt_abi.h RT_E_OVERFLOW): with RT_TEST_TINY_SPILL set in the environment
    // a BVH scene's waves get spill_factor - 1 blocks (0, 1, 3, 7, ... as the factor doubles), so that the first frame
    // on a handle overflows and the recovery paths below (status_code, rt_render, rt_render_multi) actually run.
    if (sc->use_bvh) {
        const char* tiny = getenv("RT_TEST_TINY_SPILL");
        if (tiny && tiny[0] == '1') c.max_spill = sc->spill_factor - 1;
    }

    // Camera constants (Scene.fs:100-102, Camera.fs:27,32)
    const float* M = cam->cam_to_world;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.rot[3 * i + j] = M[4 * i + j];
    c.origin[0] = M[12]; c.origin[1] = M[13]; c.origin[2] = M[14];
    const H3 ln = hnormalize(H3{ cam->look_at[0], cam->look_at[1], cam->look_at[2] });
    c.look_n[0] = ln.x; c.look_n[1] = ln.y; c.look_n[2] = ln.z;
    c.fw = (float)p->width; c.fh = (float)p->height;
    c.aspect = c.fw / c.fh;
    c.tan_half = tanf(cam->fov / 2.0f);

    const bool node64 = p->max_trace_depth > 13;
    const int nf = 11 + (node64 ? 1 : 0) + (sc->use_bvh ? 1 : 0);
    // LDS entries of the traversal stacks: see trace_kernel's RING
    // 8 entries for every tree since the ring makes room for exactly the entries pushed (rt_bvh.h): the 1M-sphere tree,
    // which used to overflow the short ring into HBM (-11 %), now gains 10 % from the 15 instead of 12 waves per CU
    const int ring = p->bvh_ring ? p->bvh_ring : 8;
    const size_t smem = sizeof(uint32_t) * ((size_t)LdsLayout(slots_of(sc->use_bvh, ring), sc->use_bvh ? ring : 0).pool + (size_t)nf * 64 * a.nblk);
    if (smem > 160 * 1024) return fail(RT_E_UNSUPPORTED, "ray pool of %zu bytes exceeds the 160 KiB LDS", smem);

    // Small frames (few pixel-samples, each a ray tree of up to thousands of rays: 640x480x1 has 307 200 trees of up to
    // 3 928 rays, 55 on average, and a wave's fair share is ~50 iterations of 64 rays — less than one large tree) are
    // rendered in two passes: primaries first, their hit records to a list, then the subtrees of those records shared
    // out record by record among all the waves the chip holds.  (A pass is a launch: no wave ever waits for another.)
    const unsigned long long total = c.total;
    const size_t list_cap = 4 * (size_t)total;
    const size_t list_bytes = (size_t)(nf + 3) * list_cap * sizeof(uint32_t);
    bool two_pass = total <= kTwoPassMaxSamples && p->two_pass == 0;
    unsigned char* list_base = nullptr;
    if (two_pass) {
        if (noalloc) {
            if (list_bytes * 2 <= ws.avail) { list_base = ws.base; ws.base += (list_bytes + 255) & ~(size_t)255; ws.avail -= (list_bytes + 255) & ~(size_t)255; }
            else two_pass = false;
        } else {
            rc = ensure_buffer(reinterpret_cast<void**>(&sc->d_list), &sc->list_bytes, list_bytes);
            if (rc) return rc;
            list_base = sc->d_list;
        }
    }
    if (two_pass) {
        c.list = reinterpret_cast<uint32_t*>(list_base);
        c.list_cap = (unsigned int)list_cap;
        {   // inverse of the hand-out multiplier modulo npix (extended Euclid): position of a pixel in the hand-out order
            long long t0 = 0, t1 = 1, r0 = npix, r1 = c.scatter;
            while (r1 != 0) { const long long qq = r0 / r1, t2 = t0 - qq * t1, r2 = r0 - qq * r1; t0 = t1; t1 = t2; r0 = r1; r1 = r2; }
            c.scatter_inv = (unsigned int)((t0 % npix + npix) % npix);
        }
        // entries of children that never exist (misses, lights, surfaces with fewer than four samples) stay empty
        HIP_TRY(hipMemsetAsync(c.list + 9 * list_cap, 0, list_cap * sizeof(uint32_t), stream));
    }

    auto launch_cnt = [&](auto phase_tag, auto est_tag, auto cnt_tag) -> int {
        constexpr int PH = decltype(phase_tag)::value;
        constexpr bool SP = decltype(est_tag)::value;
        constexpr bool CN = decltype(cnt_tag)::value;
        if (sc->use_bvh)
            return ring == 8 ? (node64 ? launch_trace<true, true, 8, PH, SP, CN>(sc, c, a, p->waves_per_cu, smem, stream, ws)
                                       : launch_trace<false, true, 8, PH, SP, CN>(sc, c, a, p->waves_per_cu, smem, stream, ws))
                             : (node64 ? launch_trace<true, true, 16, PH, SP, CN>(sc, c, a, p->waves_per_cu, smem, stream, ws)
                                       : launch_trace<false, true, 16, PH, SP, CN>(sc, c, a, p->waves_per_cu, smem, stream, ws));
        return node64 ? launch_trace<true, false, 16, PH, SP, CN>(sc, c, a, p->waves_per_cu, smem, stream, ws)
                      : launch_trace<false, false, 16, PH, SP, CN>(sc, c, a, p->waves_per_cu, smem, stream, ws);
    };
    auto launch_est = [&](auto phase_tag, auto est_tag) -> int {
        // all the statistics only for a scene that asked for them and a caller that takes them (rt_scene_counters)
        return (sc->counters && d_stats) ? launch_cnt(phase_tag, est_tag, std::true_type()) : launch_cnt(phase_tag, est_tag, std::false_type());
    };
    auto launch = [&](auto phase_tag) -> int {
        return p->estimator == RT_ESTIMATOR_SINGLE_PATH ? launch_est(phase_tag, std::true_type()) : launch_est(phase_tag, std::false_type());
    };
    HIP_TRY(hipMemsetAsync(sc->d_ctl, 0, kCtlBytes, stream));        // status, both sets of counters, partial statistics
    HIP_TRY(hipMemsetAsync(accum, 0, accum_bytes, stream));
    if (two_pass) {
        rc = launch(std::integral_constant<int, 1>());
        if (rc) return rc;
        if (after_primaries) { HIP_TRY(hipEventRecord(after_primaries, stream)); after_primaries = nullptr; }
        c.counter = sc->d_counter2;
        rc = launch(std::integral_constant<int, 2>());
    } else if (c.total < (unsigned long long)kSampleMajorBelow * 16ull * (unsigned long long)sc->num_cus) {
        // a wave's share of the frame is small enough for one heavy refill to matter: sample-major hand-out (4K frames
        // keep the samples of a pixel together: they meet in the pixel cache, and their tail is under 1 % of the frame)
        // ... in one strip of columns per XCD (trace_kernel, take()): the permutation works inside a strip
        // (a tree that an XCD's L2 mostly holds gains nothing from the strips and loses 1 % to their imbalance — the
        // 100k-triangle mesh, 7.6 MB, 83 % L2 hits: one strip then, i.e. the other seven are empty and every wave
        // moves to the first at once; measured: profiles/r03_experiments.txt section 3)
        const size_t tree_bytes = sc->use_bvh ? (size_t)sc->pdata_off + (size_t)sc->nprims * 48 : 0;
        const int strips = (sc->use_bvh && tree_bytes < kStripTreeBytes) ? 1 : kXcdParts;
        c.part_pixels = (unsigned int)((npix + strips - 1) / strips);
        c.part_items = c.part_pixels * (unsigned int)p->spp;
        c.scatter = multiplier(c.part_pixels);
        c.scatter_tail = multiplier((unsigned long long)npix % c.part_pixels);      // (0 when the strips are all full)
        rc = launch(std::integral_constant<int, 3>());
    } else rc = launch(std::integral_constant<int, 0>());
    if (rc) return rc;
    if (after_primaries) HIP_TRY(hipEventRecord(after_primaries, stream));
    const unsigned pixel_blocks = (unsigned)((npix + 255) / 256);
    hipLaunchKernelGGL(resolve_kernel, dim3(pixel_blocks + (d_stats ? (unsigned)((kStatsWords + 3) / 4) : 0u)), dim3(256), 0, stream, accum, d_rgba,
                       npix, p->spp, pixel_blocks, sc->d_statpart, reinterpret_cast<unsigned long long*>(d_stats));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// What a frame's status word means.  A BVH scene's pool capacity is a heuristic (render_on_stream): when it was
// exceeded the handle's capacity is doubled HERE, whoever asked — rt_render and rt_render_multi then render the frame
// again, and a caller of rt_render_device / rt_render_device_noalloc that finds RT_E_OVERFLOW in rt_scene_status has
// room on its next render of the same handle (those callers could not raise the capacity themselves).
constexpr int kMaxSpillFactor = 1 << 10;
int status_code(rt_scene* sc, int st)
{
    if (st & kStatusOverflow) {
        if (sc->use_bvh && sc->spill_factor < kMaxSpillFactor) {
            sc->spill_factor *= 2;
            return fail(RT_E_OVERFLOW, "a wave's ray pool overflowed its capacity; this frame is invalid, the handle's capacity is now %dx: render again",
                        sc->spill_factor);
        }
        return fail(RT_E_OVERFLOW, "a wave's ray pool overflowed its bound; results are invalid");
    }
    if (st & kStatusRange) return fail(RT_E_RANGE, "a radiance contribution was not finite or reached 2^31; it was dropped and the image is invalid");
    return RT_OK;
}
constexpr int kOverflowRetries = 4;       // rt_render / rt_render_multi render an overflowed BVH frame again this often

// Compact staging of one scene's columns inside its d_tmp: stats | rgba (16 B each) | ids (8 B each) | depth (4 B each):
// every block starts on a multiple of its own element size whatever npix is (rgba is read and written as float4)
struct Staging { uint64_t* stats; uint64_t* ids; float* rgba; float* depth; };
static_assert((RT_STATS_WORDS * 8) % 16 == 0, "rgba block of the staging must be 16-byte aligned");
int staging_of(rt_scene* sc, size_t npix, Staging* s)
{
    const size_t off_rgba = RT_STATS_WORDS * 8, off_ids = off_rgba + npix * 16, off_depth = off_ids + npix * 8;
    const int rc = ensure_tmp(sc, off_depth + npix * 4);
    if (rc) return rc;
    char* base = reinterpret_cast<char*>(sc->d_tmp);
    s->stats = reinterpret_cast<uint64_t*>(base); s->ids = reinterpret_cast<uint64_t*>(base + off_ids);
    s->rgba = reinterpret_cast<float*>(base + off_rgba); s->depth = reinterpret_cast<float*>(base + off_depth);
    return RT_OK;
}

// Queues the copies of a frame's scalars (ray count, max depth bits, status word) into the scene's pinned block.
int fetch_scalars(rt_scene* sc, const Staging& st)
{
    HIP_TRY(hipMemcpyAsync(sc->h_pin, st.stats, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, sc->stream));
    HIP_TRY(hipMemcpyAsync(sc->h_pin + 2, sc->d_status, sizeof(int), hipMemcpyDeviceToHost, sc->stream));
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_last_error(void) { return g_err; }

int rt_scene_create(const rt_surface* surfaces, int64_t n, int32_t device, rt_scene** out)
{
    if (!out) return fail(RT_E_INVALID, "out is null");
    *out = nullptr;
    if (n < 0 || (n > 0 && !surfaces)) return fail(RT_E_INVALID, "bad surface array");
    if (n > (int64_t)1 << 30) return fail(RT_E_INVALID, "too many surfaces");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_E_NO_DEVICE, "no HIP device (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(RT_E_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);

    std::vector<GeomRec> geom;
    geom.reserve((size_t)std::min<int64_t>(n, 4096));
    int64_t nprim = 0, nsphere = 0; bool any_tri = false;
    for (int64_t i = 0; i < n; i++) {
        if (surfaces[i].geom == RT_GEOM_SPHERE || surfaces[i].geom == RT_GEOM_TRIANGLE) nprim++;
        if (surfaces[i].geom == RT_GEOM_SPHERE) nsphere++;
        if (surfaces[i].geom == RT_GEOM_TRIANGLE) any_tri = true;
    }
    // A handful of spheres in a mesh scene stay in the scanned list: in the tree each of them is a second kind of
    // leaf step that most rounds would have to run (measured on the 4K mixed scene).
    const bool loose_spheres = any_tri && nsphere <= kLooseSpheres;
    // (a scene of many planes and nothing else has no primitive to put in a tree: it is scanned)
    const bool use_bvh = (any_tri || n > kBvhThreshold) && nprim > 0;
    std::vector<BuildPrim> bprims;
    // A sphere or triangle with a NaN or an infinity in it is never hit by the scan (every comparison of its test fails, or
    // gives t = inf): it stays out of the tree, whose builder and boxes need numbers (its tree_rmax / rmin share included).
    auto finite_box = [](const BuildPrim& bp) {
        for (int k = 0; k < 3; k++) if (!std::isfinite(bp.lo[k]) || !std::isfinite(bp.hi[k])) return false;
        return true;
    };
    HostBvh hb;
    std::vector<float4> src_pdata;
    if (use_bvh) bprims.reserve((size_t)nprim);
    rt_scene* sc = new (std::nothrow) rt_scene;
    if (!sc) return fail(RT_E_INVALID, "out of host memory");
    sc->device = device; sc->n = n; sc->use_bvh = use_bvh;
    std::vector<ShadeRec> h_shade((size_t)n);
    float tree_rmax = 0.0f, tree_rmin = 0.0f, tree_tri_rmax = 0.0f; bool tree_spheres = false;
    for (int64_t i = 0; i < n; i++) {
        const rt_surface& s = surfaces[i];
        if (s.kind < 0 || s.kind > 3) { delete sc; return fail(RT_E_INVALID, "surface %lld: bad kind %d", (long long)i, s.kind); }
        if (s.sample_count < 0 || s.sample_count > 4) { delete sc; return fail(RT_E_INVALID, "surface %lld: sample_count must be 0..4", (long long)i); }
        GeomRec g; memset(&g, 0, sizeof g);
        ShadeRec h; memset(&h, 0, sizeof h);
        for (int c = 0; c < 3; c++) { h.albedo[c] = s.albedo[c]; h.emission[c] = s.emission[c]; }
        h.id = s.id;
        if (s.kind == RT_KIND_METAL) h.param = fmaxf(fminf(1.0f, s.fuzz), 0.0f);      // Surface.fs:88
        else if (s.kind == RT_KIND_DIELECTRIC) h.param = s.ior;
        int gtype;
        if (s.geom == RT_GEOM_SPHERE) {
            gtype = kGeomSphere;
            g.a[0] = s.g[0]; g.a[1] = s.g[1]; g.a[2] = s.g[2];
            g.a[3] = s.g[3] * s.g[3];                                                   // Square radius, Sphere.fs:21
            h.nc[0] = s.g[0]; h.nc[1] = s.g[1]; h.nc[2] = s.g[2];
            if (use_bvh && !loose_spheres) {
                BuildPrim bp;
                const float r = std::fabs(s.g[3]);
                for (int k = 0; k < 3; k++) { bp.lo[k] = s.g[k] - r; bp.hi[k] = s.g[k] + r; bp.c[k] = s.g[k]; }
                bp.ref = (uint32_t)src_pdata.size(); bp.surf = (uint32_t)i;
                src_pdata.push_back(make_float4(g.a[0], g.a[1], g.a[2], g.a[3]));
                if (finite_box(bp)) {
                    tree_rmax = std::max(tree_rmax, r); tree_rmin = tree_spheres ? std::min(tree_rmin, r) : r; tree_spheres = true;
                    bprims.push_back(bp);
                }
            }
        } else if (s.geom == RT_GEOM_TRIANGLE) {
            // the build's own Hitable (no reference counterpart): v0, e1 = v1 - v0, e2 = v2 - v0, constant
            // normal Normalize(e1 x e2)
            gtype = kGeomTriangle;
            const H3 v0 = { s.g[0], s.g[1], s.g[2] };
            const H3 e1 = { s.g[3] - v0.x, s.g[4] - v0.y, s.g[5] - v0.z }, e2 = { s.g[6] - v0.x, s.g[7] - v0.y, s.g[8] - v0.z };
            const H3 tn = hnormalize(hcross(e1, e2));
            h.nc[0] = tn.x; h.nc[1] = tn.y; h.nc[2] = tn.z;
            BuildPrim bp;
            for (int k = 0; k < 3; k++) {
                const float a0 = s.g[k], a1 = s.g[3 + k], a2 = s.g[6 + k];
                bp.lo[k] = std::min(a0, std::min(a1, a2)); bp.hi[k] = std::max(a0, std::max(a1, a2));
                bp.c[k] = (bp.lo[k] + bp.hi[k]) * 0.5f;
            }
            bp.ref = 0x80000000u | (uint32_t)src_pdata.size(); bp.surf = (uint32_t)i;
            src_pdata.push_back(make_float4(v0.x, v0.y, v0.z, e1.x));
            src_pdata.push_back(make_float4(e1.y, e1.z, e2.x, e2.y));
            src_pdata.push_back(make_float4(e2.z, 0.0f, 0.0f, 0.0f));
            if (finite_box(bp)) {
                tree_tri_rmax = std::max(tree_tri_rmax, 0.5f * std::sqrt((bp.hi[0] - bp.lo[0]) * (bp.hi[0] - bp.lo[0]) + (bp.hi[1] - bp.lo[1]) * (bp.hi[1] - bp.lo[1]) +
                                                                        (bp.hi[2] - bp.lo[2]) * (bp.hi[2] - bp.lo[2])));
                bprims.push_back(bp);
            }
        } else if (s.geom == RT_GEOM_PLANE) {
            const H3 nraw = { s.g[0], s.g[1], s.g[2] };
            const float d = s.g[3];
            const H3 pn = hnormalize(nraw);                                             // Plane.fs:15
            g.a[0] = nraw.x; g.a[1] = nraw.y; g.a[2] = nraw.z; g.a[3] = d;
            h.nc[0] = pn.x; h.nc[1] = pn.y; h.nc[2] = pn.z;
            if (s.g[4] != 0.0f) {
                if (!(s.g[8] >= 0.0f) || !(s.g[9] >= 0.0f)) { delete sc; return fail(RT_E_INVALID, "surface %lld: negative rectangle size", (long long)i); }
                gtype = kGeomPlaneRect;
                // Plane.fs:19-30: R, kern and the rotated centre are per-plane constants
                const M4 R = rotation_between_unit_vectors(pn, H3{ 0.0f, 0.0f, -1.0f });
                const H3 kern = (d > 0.0f) ? H3{ pn.x * (-1.0f * d), pn.y * (-1.0f * d), pn.z * (-1.0f * d) }
                                           : H3{ pn.x * d, pn.y * d, pn.z * d };
                const H3 center = { s.g[5], s.g[6], s.g[7] };
                const H3 b = { center.x - kern.x, center.y - kern.y, center.z - kern.z };
                const H3 tb = transform_w0(b, R);
                const H3 newB = { kern.x + tb.x, kern.y + tb.y, kern.z + tb.z };
                const float wOff = s.g[8] / 2.0f, hOff = s.g[9] / 2.0f;
                g.b[0] = R.m[0][0]; g.b[1] = R.m[1][0]; g.b[2] = R.m[2][0]; g.b[3] = newB.x + wOff;
                g.c[0] = R.m[0][1]; g.c[1] = R.m[1][1]; g.c[2] = R.m[2][1]; g.c[3] = newB.x - wOff;
                g.d[0] = kern.x; g.d[1] = kern.y; g.d[2] = kern.z; g.d[3] = newB.y + hOff;
                g.ymin = newB.y - hOff;
            } else {
                gtype = kGeomPlaneInf;
            }
        } else {
            delete sc;
            return fail(RT_E_INVALID, "surface %lld: bad geometry %d", (long long)i, s.geom);
        }
        g.type = gtype;
        g.index = (int32_t)i;
        const int ns = default_samples(s.kind, s.sample_count, 0);
        // a Lambertian without its own SampleCount takes rt_params.lambertian_samples at render time (bit 7)
        const uint32_t lamb_default = (s.kind == RT_KIND_LAMBERTIAN && s.sample_count == 0) ? 0x80u : 0u;
        h.info = (uint32_t)s.kind | ((uint32_t)ns << 2) | ((uint32_t)gtype << 5) | lamb_default;
        // the scalar-operand linear list: every surface of a small scene, only the planes of a BVH scene
        if (!use_bvh || s.geom == RT_GEOM_PLANE || (s.geom == RT_GEOM_SPHERE && loose_spheres)) geom.push_back(g);
        h_shade[(size_t)i] = h;
    }
    sc->nloose = (int64_t)geom.size();
    if (sc->nloose > kMaxPoolSurfaces) { delete sc; return fail(RT_E_UNSUPPORTED, "more than %d %s", kMaxPoolSurfaces, use_bvh ? "planes" : "surfaces without a sphere or triangle among them"); }
    if (use_bvh) {
        bvh_build(bprims, src_pdata, hb);
        if (3 * hb.depth + 1 > kBvhStack) { delete sc; return fail(RT_E_UNSUPPORTED, "BVH depth %d exceeds the traversal stack", hb.depth); }
        // the traversal addresses nodes and records by 32-bit byte offsets and leaf references hold 28-bit record indices
        if ((hb.nodes.size() * ((size_t)1 << (kBvhNodeShift - 7)) + hb.pdata.size()) * sizeof(float4) >= (1ull << 32) || bprims.size() >= (1ull << 28)) {
            delete sc; return fail(RT_E_UNSUPPORTED, "BVH of %zu primitives is too large", bprims.size());
        }
        sc->nnodes = (int)(hb.nodes.size() / 8); sc->bvh_depth = hb.depth; sc->nprims = (int64_t)bprims.size();
        // how the traversal widens the ray (rt_bvh.h closest_hit_bvh)
        if (tree_spheres) { sc->cone_k = kSphereTestError; sc->tree_rmin = tree_rmin; }
        if (any_tri) { sc->lin_s = kTriangleConeSlope; sc->lin_far = hb.pad / kTriangleConeSlope; }
        sc->tree_rmax = std::max(tree_rmax, tree_tri_rmax);
        for (int k = 0; k < 3; k++) {
            float m = 0.0f;
            for (const BuildPrim& bp : bprims) m = std::max(m, std::max(std::fabs(bp.lo[k]), std::fabs(bp.hi[k])));
            sc->tree_reach += m;
        }
    }

    hipError_t e = hipSetDevice(device);
    const size_t ng = geom.size();
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&sc->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->ev_primaries, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&sc->h_pin), 64, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&sc->d_geom), sizeof(GeomRec) * (ng > 0 ? ng : 1));
    if (e == hipSuccess && ng > 0) e = hipMemcpy(sc->d_geom, geom.data(), sizeof(GeomRec) * ng, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&sc->d_shade), sizeof(ShadeRec) * (n > 0 ? (size_t)n : 1));
    if (e == hipSuccess && n > 0) e = hipMemcpy(sc->d_shade, h_shade.data(), sizeof(ShadeRec) * (size_t)n, hipMemcpyHostToDevice);
    if (use_bvh && !hb.nodes.empty()) {
        if (kBvhNodeShift > 7) {     // experiment build: the same nodes, further apart (rt_bvh.h RT_BVH_NODE_SHIFT)
            const size_t stride = (size_t)1 << (kBvhNodeShift - 4), nn = hb.nodes.size() / 8;
            std::vector<float4> spread(nn * stride, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            for (size_t i = 0; i < nn; i++)
                for (int r = 0; r < 7; r++) spread[i * stride + node_row(r) / 16] = hb.nodes[i * 8 + r];
            hb.nodes.swap(spread);
        }
        const size_t nbytes = sizeof(float4) * hb.nodes.size(), pbytes = sizeof(float4) * hb.pdata.size();
        sc->pdata_off = (uint32_t)nbytes;
        if (RT_BVH_REF_OFFSETS) {
            // the references the traversal sees: half the byte offset of the node / of the leaf's first record from the
            // start of the allocation (rt_bvh.h: one mask and one shift give either address), leaf flag and count kept
            const size_t stride4 = (size_t)1 << (kBvhNodeShift - 4), nn = hb.nodes.size() / stride4;
            for (size_t i = 0; i < nn; i++) {
                uint32_t refs[4];
                memcpy(refs, &hb.nodes[i * stride4 + node_row(6) / 16], sizeof refs);
                for (uint32_t& r : refs) {
                    if (r & 0x80000000u) r = 0x80000000u | (uint32_t)((((size_t)nbytes + (size_t)((r & 0x7fffffffu) >> 3) * 48u) / 16u) << 3) | (r & 7u);
                    else r = r << (kBvhNodeShift - 1);
                }
                memcpy(&hb.nodes[i * stride4 + node_row(6) / 16], refs, sizeof refs);
            }
        }
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&sc->d_nodes), nbytes + pbytes);
        if (e == hipSuccess) e = hipMemcpy(sc->d_nodes, hb.nodes.data(), nbytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(reinterpret_cast<char*>(sc->d_nodes) + nbytes, hb.pdata.data(), pbytes, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&sc->d_ctl), kCtlBytes);
    if (e == hipSuccess) e = hipMemset(sc->d_ctl, 0, kCtlBytes);
    if (e == hipSuccess) {
        sc->d_status = reinterpret_cast<int*>(sc->d_ctl);
        sc->d_counter = reinterpret_cast<unsigned long long*>(sc->d_ctl + kCtlCounter);
        sc->d_counter2 = reinterpret_cast<unsigned long long*>(sc->d_ctl + kCtlCounter2);
        sc->d_statpart = reinterpret_cast<unsigned long long*>(sc->d_ctl + kCtlStats);
    }
    if (e == hipSuccess) e = hipDeviceGetAttribute(&sc->num_cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) {
        rt_scene_destroy(sc);
        return fail(RT_E_HIP, "scene upload: %s", hipGetErrorString(e));
    }
    // set-up of the allocation-free entry point: this device's address of the static scratch array, once per device
    const int rc = scratch_of(device, &sc->scratch);
    if (rc) { rt_scene_destroy(sc); return rc; }
    *out = sc;
    g_err[0] = 0;
    return RT_OK;
}

void rt_scene_destroy(rt_scene* sc)
{
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    if (sc->copy_stream) { (void)hipStreamSynchronize(sc->copy_stream); (void)hipStreamDestroy(sc->copy_stream); }
    if (sc->stream) { (void)hipStreamSynchronize(sc->stream); (void)hipStreamDestroy(sc->stream); }
    if (sc->ev_primaries) (void)hipEventDestroy(sc->ev_primaries);
    if (sc->d_geom) (void)hipFree(sc->d_geom);
    if (sc->d_shade) (void)hipFree(sc->d_shade);
    if (sc->d_ctl) (void)hipFree(sc->d_ctl);
    if (sc->d_accum) (void)hipFree(sc->d_accum);
    if (sc->d_spill) (void)hipFree(sc->d_spill);
    if (sc->d_gstack) (void)hipFree(sc->d_gstack);
    if (sc->d_nodes) (void)hipFree(sc->d_nodes);
    if (sc->d_list) (void)hipFree(sc->d_list);
    if (sc->d_tmp) (void)hipFree(sc->d_tmp);
    if (sc->d_frame) (void)hipFree(sc->d_frame);
    if (sc->h_pin) (void)hipHostFree(sc->h_pin);
    delete sc;
}

int rt_render_device(rt_scene* sc, const rt_camera* cam, const rt_params* p, void* stream_v,
                     float* d_rgba, float* d_depth, uint64_t* d_object_id, uint64_t* d_stats)
{
    return render_on_stream(sc, cam, p, static_cast<hipStream_t>(stream_v), d_rgba, d_depth, d_object_id, d_stats, false);
}

int rt_render_device_noalloc(rt_scene* sc, const rt_camera* cam, const rt_params* p, void* stream_v,
                             float* d_rgba, float* d_depth, uint64_t* d_object_id, uint64_t* d_stats)
{
    return render_on_stream(sc, cam, p, static_cast<hipStream_t>(stream_v), d_rgba, d_depth, d_object_id, d_stats, true);
}

int rt_scene_counters(rt_scene* sc, int level)
{
    if (!sc) return fail(RT_E_INVALID, "scene is null");
    if (level != 0 && level != 1) return fail(RT_E_INVALID, "counters level %d (0 or 1)", level);
    sc->counters = level;
    return RT_OK;
}

int rt_scene_status(rt_scene* sc)
{
    if (!sc) return fail(RT_E_INVALID, "scene is null");
    HIP_TRY(hipSetDevice(sc->device));
    int st = 0;
    HIP_TRY(hipMemcpy(&st, sc->d_status, sizeof st, hipMemcpyDeviceToHost));
    // (a frame's status word is cleared by the next render, so an overflow is seen, and the capacity doubled, once:
    // a second query without a render in between doubles it again, which is harmless)
    return status_code(sc, st);
}

// Pinned host memory for the caller's frame buffers (hipHostMalloc): rt_render / rt_render_multi copy a frame into it
// by DMA (measured on the 640x480x1 frame: 0.13 ms for rgba + depth) instead of through the runtime's pageable path
// (0.45 ms).  Pinning memory the caller already owns (hipHostRegister) is refused by this pool's driver ("invalid
// argument" for every flag and alignment), so the library hands the memory out instead.
int rt_host_alloc(uint64_t bytes, void** out)
{
    if (!out) return fail(RT_E_INVALID, "out is null");
    *out = nullptr;
    if (bytes == 0) return fail(RT_E_INVALID, "empty host block");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(RT_E_NO_DEVICE, "no HIP device (this library has no CPU path)");
    const hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return fail(RT_E_HIP, "hipHostMalloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e)); }
    return RT_OK;
}

void rt_host_free(void* ptr)
{
    if (ptr) (void)hipHostFree(ptr);
}

int rt_render(rt_scene* sc, const rt_camera* cam, const rt_params* p, float* rgba, float* depth, uint64_t* object_id,
              float* max_depth, uint64_t* ray_count)
{
    if (!sc) return fail(RT_E_INVALID, "scene is null");
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (!rgba || !depth) return fail(RT_E_INVALID, "rgba and depth buffers are required");
    HIP_TRY(hipSetDevice(sc->device));
    const size_t npix = (size_t)columns_of(p) * (size_t)p->height;
    if (npix == 0) { if (max_depth) *max_depth = 0.0f; if (ray_count) *ray_count = 0; return RT_OK; }
    Staging st;
    rc = staging_of(sc, npix, &st);
    if (rc) return rc;
    // The frame, its scalars and its status word are all queued behind the kernels on the scene's own stream and waited
    // for ONCE (other scenes on this device keep running: no device-wide synchronisation).  The status is looked at
    // afterwards: a frame that has to be rendered again (a BVH scene whose heuristic pool capacity was exceeded: doubled
    // by status_code) is simply copied again.  No object-ID traffic at all when the caller passes no buffer for them.
    // Depth and object IDs are final once the primaries are traced (Scene.fs:80: first sample's primary hit): they
    // leave on a second stream while the rest of the frame renders (the second pass of a small frame is 7/8 of it).
    for (int attempt = 0;; attempt++) {
        rc = render_on_stream(sc, cam, p, sc->stream, st.rgba, st.depth, object_id ? st.ids : nullptr, st.stats, false, sc->ev_primaries);
        if (rc) { (void)hipStreamSynchronize(sc->stream); return rc; }
        HIP_TRY(hipStreamWaitEvent(sc->copy_stream, sc->ev_primaries, 0));
        HIP_TRY(hipMemcpyAsync(depth, st.depth, npix * 4, hipMemcpyDeviceToHost, sc->copy_stream));
        if (object_id) HIP_TRY(hipMemcpyAsync(object_id, st.ids, npix * 8, hipMemcpyDeviceToHost, sc->copy_stream));
        HIP_TRY(hipMemcpyAsync(rgba, st.rgba, npix * 16, hipMemcpyDeviceToHost, sc->stream));
        rc = fetch_scalars(sc, st);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(sc->stream));
        HIP_TRY(hipStreamSynchronize(sc->copy_stream));
        rc = status_code(sc, (int)(uint32_t)sc->h_pin[2]);
        if (rc != RT_E_OVERFLOW || !sc->use_bvh || attempt >= kOverflowRetries) break;
    }
    if (rc) return rc;
    if (ray_count) *ray_count = sc->h_pin[0];
    if (max_depth) { uint32_t bits = (uint32_t)sc->h_pin[1]; memcpy(max_depth, &bits, 4); }
    g_err[0] = 0;
    return RT_OK;
}

// Scene i of n renders the columns i, i+n, ... on its own device and stream; a placement kernel on that device then
// stores its compact block at stride n into the frame that lives in the HBM of scenes[0]'s device (peer stores over
// xGMI when the devices differ), and the gathered frame is copied to the host once.
int rt_render_multi(rt_scene* const* scenes, int32_t nscenes, const rt_camera* cam, const rt_params* p,
                    float* rgba, float* depth, uint64_t* object_id, float* max_depth, uint64_t* ray_count)
{
    if (!scenes || nscenes <= 0) return fail(RT_E_INVALID, "no scenes");
    for (int i = 0; i < nscenes; i++) if (!scenes[i]) return fail(RT_E_INVALID, "scene %d is null", i);
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (!rgba || !depth) return fail(RT_E_INVALID, "rgba and depth buffers are required");
    if (p->x_begin != 0 || p->x_end != p->width || p->x_step != 1)
        return fail(RT_E_INVALID, "rt_render_multi partitions the whole frame itself: x_begin/x_end/x_step must be 0/width/1");
    for (int i = 0; i < nscenes; i++)
        for (int j = 0; j < i; j++)
            if (scenes[i] == scenes[j]) return fail(RT_E_INVALID, "scene %d given twice (a handle renders one frame at a time)", i);
    rt_scene* root = scenes[0];
    const int W = p->width, H = p->height;
    const size_t fpix = (size_t)W * (size_t)H;
    // the gathered frame on the root device: rgba | ids | depth
    HIP_TRY(hipSetDevice(root->device));
    rc = ensure_buffer(reinterpret_cast<void**>(&root->d_frame), &root->frame_bytes, fpix * 28);
    if (rc) return rc;
    float* f_rgba = root->d_frame;
    uint64_t* f_ids = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(root->d_frame) + fpix * 16);
    float* f_depth = reinterpret_cast<float*>(reinterpret_cast<char*>(root->d_frame) + fpix * 24);
    std::vector<Staging> st((size_t)nscenes);
    std::vector<int> ncols((size_t)nscenes, 0);
    uint64_t rays = 0; float maxd = 0.0f;
    // Every way out of the frame loop below goes through `drain`: work already queued on the scenes' streams writes into
    // root->d_frame and into the scenes' staging, which a follow-up call may free or regrow.
    auto drain = [&](int upto) {
        for (int i = 0; i <= upto && i < nscenes; i++) {
            if (hipSetDevice(scenes[i]->device) == hipSuccess) (void)hipStreamSynchronize(scenes[i]->stream);
        }
    };
    // 1. every scene starts its share (asynchronous: the devices render concurrently) and queues its placement
    auto start_share = [&](int i) -> int {
        rt_scene* sc = scenes[i];
        rt_params pi = *p;
        pi.x_begin = i; pi.x_step = nscenes;
        ncols[(size_t)i] = share_columns(W, i, nscenes);
        if (ncols[(size_t)i] <= 0) return RT_OK;
        HIP_TRY(hipSetDevice(sc->device));
        if (sc->device != root->device && !sc->peer_ok[root->device]) {
            // once per pair of handles' devices: the query and the enable cost host time on every frame otherwise
            int can = 0;
            HIP_TRY(hipDeviceCanAccessPeer(&can, sc->device, root->device));
            if (!can) return fail(RT_E_UNSUPPORTED, "device %d cannot store to device %d (no peer access)", sc->device, root->device);
            const hipError_t pe = hipDeviceEnablePeerAccess(root->device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) return fail(RT_E_HIP, "hipDeviceEnablePeerAccess: %s", hipGetErrorString(pe));
            (void)hipGetLastError();
            sc->peer_ok[root->device] = true;
        }
        const size_t npix = (size_t)ncols[(size_t)i] * (size_t)H;
        int r = staging_of(sc, npix, &st[(size_t)i]);
        if (r) return r;
        r = render_on_stream(sc, cam, &pi, sc->stream, st[(size_t)i].rgba, st[(size_t)i].depth, object_id ? st[(size_t)i].ids : nullptr,
                             st[(size_t)i].stats, false);
        if (r) return r;
        hipLaunchKernelGGL(place_columns_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, sc->stream,
                           st[(size_t)i].rgba, st[(size_t)i].depth,
                           object_id ? reinterpret_cast<const unsigned long long*>(st[(size_t)i].ids) : nullptr,
                           f_rgba, f_depth, object_id ? reinterpret_cast<unsigned long long*>(f_ids) : nullptr,
                           ncols[(size_t)i], H, i, nscenes);
        HIP_TRY(hipGetLastError());
        return fetch_scalars(sc, st[(size_t)i]);
    };
    // 2. wait for a share, take its scalars (maxFrameBufferDepth: max, Scene.fs:47-48; ray count: sum)
    auto finish_share = [&](int i, int* status) -> int {
        rt_scene* sc = scenes[i];
        HIP_TRY(hipSetDevice(sc->device));
        HIP_TRY(hipStreamSynchronize(sc->stream));
        *status = status_code(sc, (int)(uint32_t)sc->h_pin[2]);     // (doubles the capacity of a BVH handle that overflowed)
        rays += sc->h_pin[0];
        uint32_t bits = (uint32_t)sc->h_pin[1]; float m; memcpy(&m, &bits, 4);
        maxd = std::max(maxd, m);
        return RT_OK;
    };
    for (int attempt = 0;; attempt++) {
        for (int i = 0; i < nscenes; i++) {
            rc = start_share(i);
            if (rc) { char keep[sizeof g_err]; memcpy(keep, g_err, sizeof keep); drain(i); memcpy(g_err, keep, sizeof keep); return rc; }
        }
        rays = 0; maxd = 0.0f;
        int first_err = RT_OK;
        char first_msg[sizeof g_err] = "";
        for (int i = 0; i < nscenes; i++) {
            if (ncols[(size_t)i] <= 0) continue;
            int status = RT_OK;
            rc = finish_share(i, &status);
            if (rc) { char keep[sizeof g_err]; memcpy(keep, g_err, sizeof keep); drain(nscenes - 1); memcpy(g_err, keep, sizeof keep); return rc; }
            if (status && !first_err) { first_err = status; memcpy(first_msg, g_err, sizeof first_msg); }
        }
        if (!first_err) break;
        // every share has been waited for.  A BVH scene's pool capacity is a heuristic: the handles that overflowed have
        // had theirs doubled (status_code); render the frame again
        bool any_bvh = false;
        for (int i = 0; i < nscenes; i++) any_bvh = any_bvh || scenes[i]->use_bvh;
        if (first_err != RT_E_OVERFLOW || !any_bvh || attempt >= kOverflowRetries) { memcpy(g_err, first_msg, sizeof first_msg); return first_err; }
    }
    // 3. one copy of the gathered frame to the host
    HIP_TRY(hipSetDevice(root->device));
    HIP_TRY(hipMemcpyAsync(rgba, f_rgba, fpix * 16, hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(hipMemcpyAsync(depth, f_depth, fpix * 4, hipMemcpyDeviceToHost, root->stream));
    if (object_id) HIP_TRY(hipMemcpyAsync(object_id, f_ids, fpix * 8, hipMemcpyDeviceToHost, root->stream));
    HIP_TRY(hipStreamSynchronize(root->stream));
    if (ray_count) *ray_count = rays;
    if (max_depth) *max_depth = maxd;
    g_err[0] = 0;
    return RT_OK;
}

int rt_find_closest(rt_scene* sc, int64_t nrays, const float* origins, const float* directions, float* t, int32_t* index)
{
    if (!sc) return fail(RT_E_INVALID, "scene is null");
    if (nrays < 0 || (nrays > 0 && (!origins || !directions || !t || !index))) return fail(RT_E_INVALID, "bad ray arrays");
    if (nrays == 0) return RT_OK;
    HIP_TRY(hipSetDevice(sc->device));
    const size_t n = (size_t)nrays;
    int rc = ensure_tmp(sc, n * 32);
    if (rc) return rc;
    float* d_o = sc->d_tmp; float* d_d = d_o + 3 * n; float* d_t = d_d + 3 * n; int* d_i = reinterpret_cast<int*>(d_t + n);
    HIP_TRY(hipMemcpyAsync(d_o, origins, n * 12, hipMemcpyHostToDevice, sc->stream));
    HIP_TRY(hipMemcpyAsync(d_d, directions, n * 12, hipMemcpyHostToDevice, sc->stream));
    const long long blocks = ((long long)n + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(RT_E_INVALID, "too many rays for one launch");
    hipLaunchKernelGGL(find_closest_kernel, dim3((unsigned)blocks), dim3(256), 0, sc->stream, sc->d_geom, (int)sc->nloose,
                       bvh_ref(sc), (long long)n, d_o, d_d, d_t, d_i);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(t, d_t, n * 4, hipMemcpyDeviceToHost, sc->stream));
    HIP_TRY(hipMemcpyAsync(index, d_i, n * 4, hipMemcpyDeviceToHost, sc->stream));
    HIP_TRY(hipStreamSynchronize(sc->stream));
    g_err[0] = 0;
    return RT_OK;
}

}  // extern "C"
