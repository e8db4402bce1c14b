// rt_kernels.h — the render kernels: persistent waves that trace whole ray trees out of a wave-private
// LDS pool ("wavefront path tracing inside a wave"), plus the resolve pass.
//
// Reference path: Scene.renderScene -> renderPass -> rayTraceBase -> rayTrace (Scene.fs:51-121).
// The reference's estimator is a branching recursion  L = E + (1/n) * sum_i shading_i * L(child_i)
// (Scene.fs:62-70).  It is evaluated here in its throughput form (cf. notes.txt:21-54): every
// pending ray carries beta = prod shading/n along its path and adds beta*E to its pixel-sample
// when it hits an emitter.  Per-ray arithmetic (intersection, hit point, scatter direction) is
// bit-identical to the recursion; only the final additions differ, and those are done in 32.32
// fixed point, so the image does not depend on the order in which rays happen to be traced
// (run-to-run, across wave counts and across multi-GPU partitions the output is bit-identical).
//
// Work distribution: pixel-samples are handed out by position in a fixed order — pixel-major for large frames (the samples
// of a pixel arrive together), sample-major in one strip of columns per XCD for medium frames, two passes over a list of
// primary hit records for small ones (`PHASE`, below) — with consecutive positions mapped to pixels far apart in the
// image.  A wave takes exactly as many positions as it has free lanes (at most 32) with one returning atomic whenever its
// pool runs dry, so all waves finish together whatever the cost of individual pixels (1 ray for a light, thousands for
// the floor).
//
// Pool: a LIFO of HIT RECORDS (hit point, incoming direction, beta, surface, tree node).  A record
// with n children is expanded lazily: each iteration takes the top records whose children fit in
// 64 lanes (ballot + prefix sum), every lane generates ONE child ray, intersects it against the
// scene and, if the child hits a scattering surface, pushes one new record (ballot + prefix-sum
// compaction).  Lanes left over when the pool is drained start new pixel-samples (primary rays).
// Records stay sorted by tree level (deepest on top), which bounds the pool to 64 records per
// level — see DESIGN.md "Pool bound".
#pragma once
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_partition.h"

namespace rt {

// Cold per-frame constants: only the primary-ray branch, the flush and the epilogue read them.  The
// struct is the FIRST kernel parameter and is read through an opaque pointer to the kernarg segment
// (cold_args() below), so that the compiler neither keeps its values in SGPRs across the closest-hit
// loop nor spills them to VGPR lanes there (measured: 14 v_readlane/v_writelane per surface test
// before this split).
struct ColdArgs {
    float rot[9];                   // Rotation(cameraWS) rows (Scene.fs:100)
    float origin[3];                // cameraWS.Translation (Scene.fs:102)
    float look_n[3];                // Normalize(lookAt) (Scene.fs:76)
    float fw, fh, aspect, tan_half; // Camera.fs:20-32
    float bg[3];                    // backgroundColor (Scene.fs:32)
    int x_begin, x_step, jitter;
    int H, ncols;
    unsigned long long total;       // pixel-samples of this launch = ncols * H * spp
    unsigned int npix, scatter;     // hand-out order: pixel (i * scatter) mod npix, gcd(scatter, npix) = 1
    unsigned long long* counter;    // hand-out counter(s).  One for the pixel-samples of a frame; kCounterParts for the list
                                    // entries of the second pass of a small frame, part p dealing the p-th share of the
                                    // list: tens of thousands of hand-outs within half a millisecond queue at a single
                                    // address (~12 ns per same-address atomic; measured 1.00 -> 0.80 ms on 640x480x1,
                                    // while large frames are 6 % FASTER with the single counter)
    unsigned long long* accum;      // [ncols*H*4] per-pixel radiance sums (r,g,b,pad), 32.32 fixed point
    float* depth;                   // [ncols*H]
    unsigned long long* ids;        // [ncols*H] or null
    unsigned long long* stats;      // [kStatsParts][RT_STATS_WORDS] partial sums (wave w adds to part w % kStatsParts;
                                    // resolve_kernel folds them into the caller's array), or null
    int* status;                    // overflow flag
    uint32_t* spill;                // [waves][max_spill][NF*64] pool blocks spilled to HBM
    uint32_t* gstack;               // [waves][kBvhStack][64] backing store of the traversal stacks (BVH)
    // two-pass rendering of small frames (RenderArgs::phase): the records of the primary hits, field-major
    uint32_t* list;                 // [NF + 3][list_cap]: the record fields, then RNG counter (lo, hi), compact pixel index;
                                    // entry = (child of the primary) * total + position of the pixel-sample in the hand-out
                                    // order: the four subtrees of one expensive sample lie far apart in the list
    unsigned int list_cap;          // = 4 * total
    unsigned int scatter_inv;       // inverse of `scatter` modulo npix
    int max_spill;                  // blocks one wave may spill to HBM (= record levels); read on the (rare) spill path only
    int group_shift;                // PHASE 3 (one-pass frame handed out sample-major): hand-out position q =
                                    // ((sample / G) * npix + pixel position) * G + sample % G with G = 1 << group_shift (a divisor
                                    // of spp) instead of pixel position * spp + sample (see the primary ray in trace_kernel)
    // PHASE 3 again: the frame is cut into kXcdParts strips of part_pixels consecutive compact pixel positions (whole
    // columns side by side), each with its own hand-out counter and the order above INSIDE the strip; a wave starts
    // on the strip of the XCD it runs on (see take() in trace_kernel)
    unsigned int part_pixels;       // pixels of a strip (the last non-empty strip may hold fewer)
    unsigned int part_items;        // = part_pixels * spp: hand-out positions of a strip
    unsigned int scatter_tail;      // the hand-out multiplier of the last, shorter strip (`scatter` is the full strips')
};

// Hot constants: live in SGPRs for the whole kernel.
struct RenderArgs {
    int nsurf, spp, max_depth, bg_nonzero;
    int nblk;                       // 64-record blocks of the pool kept in LDS
    uint32_t seed_lo, seed_hi;
    int min_active;                 // BVH: leave the traversal when fewer lanes than this are still walking
    int lamb;                       // SampleCount of Lambertian surfaces that do not override it (Surface.fs:63: 4)
};

// The pointer stays in the constant address space (the kernarg segment): loads through it are scalar loads of
// wave-uniform values.  (Through a generic pointer every load counts as divergent for the compiler — a generic address
// might be lane-private scratch — and with it everything computed from it: the hand-out bounds, the loop conditions,
// and so the whole render loop was compiled as a divergent loop with its wave-uniform state in vector registers.)
typedef const ColdArgs __attribute__((address_space(4))) * ColdPtr;
RT_DEV ColdPtr cold_args()
{
    uint64_t p = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));     // opaque: plain loads at the point of use, never hoisted
    return (ColdPtr)p;
}

// bits of the status word a launch leaves behind (rt_scene_status)
constexpr uint32_t kNoPixel = 0xffffffffu;   // (frames have fewer than 2^31 pixels)
constexpr int kPixelCacheMinSpp = 8;         // samples per pixel from which finished samples go through the pixel cache
constexpr int kStatusOverflow = 1;      // a wave's record pool overflowed
constexpr int kStatusRange = 2;         // a radiance contribution was NaN, infinite or >= 2^31 and was dropped

// Waves add their statistics to one of kStatsParts copies of the array: thousands of waves leaving a small frame at
// the same moment otherwise queue at one cache line (measured on 640x480x1: 5632 waves x 10 atomics = 0.4 ms per launch).
constexpr int kCounterParts = 16;       // each counter 128 bytes from the next (its own cache line)
constexpr int kCounterStride = 16;      // in u64
constexpr int kListChunk = 32;          // list entries a wave takes at a time in the second pass of a small frame ...
constexpr int kListLow = 16;            // ... when its pool holds at most this many records (measured on 640x480x1: chunks of
                                        // 8 / 16 entries 2.0 / 1.2 ms, thresholds 0 / 48 records 0.83 / 1.09 ms against 0.83 ms)
#ifndef RT_XCD_PARTS
#define RT_XCD_PARTS 8
#endif
constexpr int kXcdParts = RT_XCD_PARTS; // strips of a sample-major one-pass frame = XCDs of the chip (each has its own L2); 1 = one strip (A/B builds)
static_assert(kXcdParts <= kCounterParts, "one hand-out counter per strip");
constexpr int kStatsParts = 128;
constexpr int kStatsWords = 20;         // == RT_STATS_WORDS (include/rt_abi.h)

// meta word of a record
//   bits 0..7 slot | 8..12 level | 13..15 children | 16..17 kind | 18..19 geometry type | 20..31 surface index
constexpr int kMaxPoolSurfaces = 4096;   // surface index bits of the meta word (linear-scan scenes)
RT_DEV uint32_t make_meta(int slot, int level, int n, int kind, int geom, int surf)
{
    return (uint32_t)slot | ((uint32_t)level << 8) | ((uint32_t)n << 13) | ((uint32_t)kind << 16) |
           ((uint32_t)geom << 18) | ((uint32_t)surf << 20);
}

// A 64-bit zero made where it is stored: as a plain constant the compiler keeps it in a register pair for the whole render
// loop (and, when registers run out, spills THAT to scratch rather than making it again).
RT_DEV unsigned long long local_zero() { unsigned long long z = 0ull; asm volatile("" : "+v"(z)); return z; }

RT_DEV int mbcnt64(unsigned long long m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// radiance contribution -> signed 32.32 fixed point (two's complement in a u64).  Defined for finite |x| < 2^31
// (the callers check that and flag anything else, see add_radiance): floor(x) in the high word, the fraction
// x - floor(x), scaled by 2^32 and truncated, in the low word.  The fraction is exact in [0,1) except for a tiny
// negative x (x = -1e-10: -1e-10 - (-1) rounds to 1.0f), where it is held at the largest float below 1: the value
// converted is then off by less than 2^-24, always the same for the same x, so sums stay order-independent.
RT_DEV unsigned long long to_fixed(float x)
{
    const float fl = floorf(x);
    const float fr = fminf(x - fl, 0x1.fffffep-1f);
    const uint32_t lo = (uint32_t)(fr * 4294967296.0f);         // exact scaling, truncating convert, < 2^32
    const int32_t hi = (int32_t)fl;
    return ((unsigned long long)(uint32_t)hi << 32) | lo;
}
// |x| < 2^31 and not NaN
RT_DEV bool fixed_ok(f3 v) { return fabsf(v.x) < 2147483648.0f && fabsf(v.y) < 2147483648.0f && fabsf(v.z) < 2147483648.0f; }

// pixel-samples a wave can have in flight: 32 (ray trees widen fast enough to fill the lanes from 32 samples, and
// the smaller slot table buys waves: linear-scan kernel 8.4 -> 7.2 KB of LDS, 19 -> 22 waves/CU, c4s +6 %; small-
// tree BVH kernel 16 waves/CU; 16 slots were measured 30-45 % slower); 64 in the large-tree BVH variant
constexpr int slots_of(bool bvh, int ring) { return (!bvh || ring == 8) ? 32 : 64; }

// LDS of one wave, in dwords: slot table, lane map, record pool
struct LdsLayout {
    int stack, acc, ps, pix, pend, map, pool;
    // `ring` = entries of a lane's traversal stack kept in LDS (0: linear-scan kernels).  The stack comes FIRST: its rows
    // are 256 bytes and the 8-entry ring is addressed as (column base) | (row << 8) (rt_bvh.h RING2), which needs bits
    // 8..10 of every column base to be zero.
    constexpr LdsLayout(int slots, int ring)
        : stack(0),                              // u32 [ring][64]  traversal stacks, one column per lane
          acc(ring * 64),                        // u64 [slots*3]  radiance of the pixel-sample, 32.32
          ps(ring * 64 + slots * 6),             // u64 [slots]    global pixel-sample index (RNG counter)
          pix(ring * 64 + slots * 8),            // u32 [slots]    compact pixel index (k*H + py)
          pend(ring * 64 + slots * 9),           // u32 [slots]    records of this slot in the pool or in flight
          map(ring * 64 + slots * 10),           // u32 [64]       lane map (children / slot allocation)
          pool(ring * 64 + slots * 10 + 64) {}   // records, in blocks of 64
};

// NODE64 = keep the ray-tree node index in 64 bits (needed when max_trace_depth > 13).
// BVH    = large scene: `geom` holds only the loose surfaces (planes), spheres and triangles are reached
//          through `bvh` (rt_bvh.h); surface indices no longer fit the meta word and get their own field.
// RING   = entries of a lane's traversal stack kept in LDS (BVH only).  8 entries and a 128-register budget put
//          14 waves on a CU instead of 12 (the traversal is latency-bound: measured +10 % on the 4K mixed scene)
//          at the price of a few spilled registers and of more stack traffic to HBM, which is why scenes with
//          deep, crowded trees (the 1M-sphere field: -11 %) use the 16-entry variant at 3 waves per SIMD.
// COUNT  = all words of the statistics array (rays by kind, iterations, BVH visits and rounds); without it a frame counts
//          its rays, its largest depth and the wave clocks only (rt_scene_counters: the counters cost 1-3 % of a frame).
template <bool NODE64, bool BVH, int RING, int PHASE, bool SINGLE, bool COUNT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BVH ? (RING == 8 ? 4 : 3) : 6))) void trace_kernel(const ColdArgs cold_unused, const RenderArgs a,
                                                   const GeomRec* __restrict__ geom,
                                                   const ShadeRec* __restrict__ shade, const BvhRef bvh)
{
    extern __shared__ uint32_t smem[];
    const int lane = threadIdx.x;
    constexpr int NF = 11 + (NODE64 ? 1 : 0) + (BVH ? 1 : 0);
    constexpr int SLOTS = slots_of(BVH, RING);
    constexpr LdsLayout L(SLOTS, BVH ? RING : 0);
    unsigned long long* sl_acc = reinterpret_cast<unsigned long long*>(smem + L.acc);
    unsigned long long* sl_ps = reinterpret_cast<unsigned long long*>(smem + L.ps);
    uint32_t* sl_pix = smem + L.pix;
    uint32_t* sl_pend = smem + L.pend;
    uint32_t* map = smem + L.map;
    uint32_t* pool = smem + L.pool;
    // Record r lives in block r/64: field f of record r is the dword pool[(r/64)*NF*64 + f*64 + r%64], so a
    // field is an immediate offset (f*256 bytes) from the record's base address and lanes reading
    // consecutive records hit consecutive banks.
    // The LDS holds the top nblk blocks of the pool as a ring (base_blk = physical block of logical block 0);
    // when a push would not fit, the bottom block (the shallowest 64 records, needed last) is spilled to
    // HBM and reloaded once everything above it has been consumed.  Typical depth is ~100 records, the
    // proven bound 64 per level: the ring keeps LDS per wave small (more waves per CU) without giving up
    // the bound.
    int base_blk = 0, nspill = 0;
    auto rec_base = [&](int r) -> uint32_t* {
        int b = (r >> 6) + base_blk;
        b = (b >= a.nblk) ? b - a.nblk : b;
        return pool + (b * (NF * 64) + (r & 63));
    };
    // beta*E into the pixel-sample's fixed-point sum; a contribution the format cannot hold (NaN, infinite or
    // >= 2^31) is dropped and flagged in the status word at once (it never happens with the reference's materials)
    auto add_radiance = [&](int sl, f3 v) {
        if (fixed_ok(v)) {
            atomicAdd(&sl_acc[sl * 3 + 0], to_fixed(v.x));
            atomicAdd(&sl_acc[sl * 3 + 1], to_fixed(v.y));
            atomicAdd(&sl_acc[sl * 3 + 2], to_fixed(v.z));
        } else atomicOr(cold_args()->status, kStatusRange);
    };
    enum { F_PX = 0, F_PY = 64, F_PZ = 128, F_DX = 192, F_DY = 256, F_DZ = 320, F_BX = 384, F_BY = 448, F_BZ = 512,
           F_META = 576, F_NODE = 640, F_NODEHI = 704, F_SURF = NODE64 ? 768 : 704 };
    uint32_t* bvh_stack = smem + L.stack + lane;                     // [RING][64], only when BVH
    if (BVH && RING == 8 && RT_BVH_RING2 && (lds_addr_of(smem + L.stack) & 0x7ffu) != 0u) {
        // (cannot happen: the kernels have no static LDS, so the dynamic segment starts at byte 0)
        if (lane == 0) atomicOr(cold_args()->status, kStatusOverflow);
        return;
    }

    const int spp = a.spp;
    const unsigned long long t_start = __builtin_readcyclecounter();
#ifdef RT_WAVE_TIMES
    const unsigned long long wt_start = wall_clock64();     // 100 MHz, common to the whole device
#endif

    int top = 0;                                // records in the pool (wave-uniform)
    unsigned long long freemask = ~0ull >> (64 - SLOTS);   // free slots (wave-uniform)
    unsigned long long res_next = 0;            // first pixel-sample of the current refill
    bool exhausted = false;                     // the global counter has run past the end
    uint32_t nrays = 0;                         // closest-hit queries by this wave
    uint32_t n_nodes = 0, n_prims = 0;          // BVH nodes visited / primitives tested by this wave
    uint32_t n_rin = 0, n_rlf = 0;              // BVH traversal rounds with an inner / a leaf step (wave-uniform)
    float maxd = 0.0f;
    constexpr bool kPixelCache = !BVH && PHASE == 0;
    uint32_t cpix = kNoPixel; unsigned long long cacc = 0ull;   // this lane's share of the pixel cache (see the flush below)
    bool overflow = false;
    uint32_t n_iter = 0;                        // loop iterations (wave-uniform)
    uint32_t n_lamb = 0, n_metal = 0, n_diel = 0;   // child rays intersected, by the kind that scattered them
    uint32_t n_refill = 0;                      // hand-outs this wave has taken (two-pass frames)
#ifdef RT_PHASE_TIMING
    unsigned long long ph_gen = 0, ph_scan = 0, ph_trav = 0, ph_shade = 0, ph_t = __builtin_readcyclecounter();
#define RT_PHASE(acc) { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - ph_t; ph_t = now_; }
#ifdef RT_PHASE_FINE
    unsigned long long ph_fine[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#define RT_FINE(i) { const unsigned long long now_ = __builtin_readcyclecounter(); ph_fine[i] += now_ - ph_t; ph_gen += now_ - ph_t; ph_t = now_; }
#else
#define RT_FINE(i)
#endif
#else
#define RT_PHASE(acc)
#define RT_FINE(i)
#endif

    // Hand-out: `want` consecutive items of [0, total) from the wave's current part of the index space; when that
    // part is used up the wave moves to the next part that has items left.  Returns the number taken (0: none left).
    // XCD-aware hand-out (PHASE 3).  Each XCD has its own 4 MB L2 and every XCD used to walk the whole scene: the 87 MB
    // tree of the 1M-sphere frame ran at a 65 % L2 hit rate, and that frame's time follows the lines it misses (with a
    // node spread over two cache lines instead of one it takes 68 % longer, the mesh frames 1.5 %: profiles/
    // r03_node_bytes.txt).  So the frame is cut into one strip of columns per XCD and a wave starts on the strip of the
    // XCD it runs on (HW_REG_XCC_ID; which blocks share an XCD is not promised by HIP, so it is read, not computed):
    // the primaries and first bounces of one XCD's waves then stay in one part of the scene and of the tree.  A wave
    // whose strip is used up moves to the next strip that has items left, so the balance stays dynamic.
    // (pixel-major 4K frames keep one counter: strips measured +-0 on the 4K sphere scene, +0.5 % on the 4K mixed scene)
    constexpr bool kStrips = PHASE == 3;
    constexpr int PARTS = (PHASE == 2) ? kCounterParts : kStrips ? kXcdParts : 1;
    int cur_part = kStrips ? (int)(xcc_id() % (uint32_t)PARTS) : (int)(blockIdx.x % PARTS);
    auto take = [&](int want, unsigned long long total, unsigned long long per, unsigned long long& base_out) -> int {
        const ColdPtr ca = cold_args();
        while (true) {
            const unsigned long long lo = (unsigned long long)cur_part * per;
            const unsigned long long hi = (lo + per < total) ? lo + per : total;
            unsigned long long b = 0;
            if (lane == 0) b = atomicAdd(ca->counter + cur_part * kCounterStride, (unsigned long long)want);
            // (readfirstlane, not a shuffle: the compiler then knows the value to be wave-uniform, and with it the loop
            // conditions that depend on it — otherwise the whole render loop is compiled as a divergent loop)
            b = uniform64(b) + lo;
            if (b < hi) { base_out = b; return __builtin_amdgcn_readfirstlane((b + (unsigned long long)want <= hi) ? want : (int)(hi - b)); }
            if (PARTS == 1) return 0;
            // this part is used up: look at all the counters at once for a part with items left, nearest after this one
            unsigned long long cv = ~0ull;
            if (lane < PARTS) cv = __hip_atomic_load(ca->counter + lane * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long llo = (unsigned long long)lane * per;
            const unsigned long long lhi = (llo + per < total) ? llo + per : total;
            const uint32_t avail = (uint32_t)__ballot(lane < PARTS && llo < lhi && cv < lhi - llo);
            if (avail == 0u) return 0;
            const uint32_t rot = ((avail >> cur_part) | (avail << (PARTS - cur_part))) & ((1u << PARTS) - 1u);
            cur_part = (cur_part + __builtin_ctz(rot)) % PARTS;
        }
    };

    // Per-lane ray state.  In BVH mode a lane may keep its ray over several iterations: when most lanes of the
    // wave have finished their traversal, the wave leaves the traversal loop, handles the finished rays,
    // refills the free lanes with new rays and goes back in; the unfinished lanes ("busy") keep their
    // registers and their traversal stack (measured before this: 16 % of the lanes active per VALU
    // instruction, the wave waiting for its longest traversal).
    f3 O = F3(0, 0, 0), Dn = F3(0, 0, 0), beta = F3(1.0f, 1.0f, 1.0f);
    int slot = 0, clevel = 0; uint64_t cnode = 0;
    bool alive = false, owner = false, is_prim = false;
    bool prim_first = false;
    float tcur = 0.0f; int hidx = -1;
    uint32_t trav_cur = kBvhDone; int trav_sp = 0, trav_lo = 0;
    bool busy = false;                          // BVH: this lane's traversal is suspended, not finished
    unsigned long long busymask = 0ull;         // ballot(busy), wave-uniform

    while (true) {
        // The wave-uniform state is passed through readfirstlane once per iteration: free where the compiler already holds
        // it in scalar registers, and where its analysis had given up (one value it cannot prove uniform makes the exit
        // condition "divergent", and with it every value carried around the loop) it restores scalar branches and scalar
        // arithmetic for the bookkeeping of the whole loop (measured: 640x480x1 +6 %, 1M spheres +4 %; the 4K frames of the
        // linear-scan kernel lose 2 % to the longer scalar dependency chains, so that variant keeps the vector form).
        if (BVH || PHASE != 0) {
            top = __builtin_amdgcn_readfirstlane(top); nspill = __builtin_amdgcn_readfirstlane(nspill); base_blk = __builtin_amdgcn_readfirstlane(base_blk);
            freemask = uniform64(freemask); busymask = uniform64(busymask); res_next = uniform64(res_next);
            exhausted = __builtin_amdgcn_readfirstlane((int)exhausted) != 0; overflow = __builtin_amdgcn_readfirstlane((int)overflow) != 0;
            nrays = __builtin_amdgcn_readfirstlane(nrays); n_iter = __builtin_amdgcn_readfirstlane(n_iter); n_refill = __builtin_amdgcn_readfirstlane(n_refill);
            if (BVH) {
                n_nodes = __builtin_amdgcn_readfirstlane(n_nodes); n_prims = __builtin_amdgcn_readfirstlane(n_prims);
                n_rin = __builtin_amdgcn_readfirstlane(n_rin); n_rlf = __builtin_amdgcn_readfirstlane(n_rlf);
                n_lamb = __builtin_amdgcn_readfirstlane(n_lamb); n_metal = __builtin_amdgcn_readfirstlane(n_metal); n_diel = __builtin_amdgcn_readfirstlane(n_diel);
            }
        }
        if (top == 0 && nspill > 0) {
            // ---- the LDS part of the pool is consumed: bring back the most recently spilled block
            nspill--;
            const ColdPtr cs = cold_args();
            const uint32_t* src = cs->spill + ((size_t)blockIdx.x * cs->max_spill + nspill) * (NF * 64) + lane;
            uint32_t* dst = pool + base_blk * (NF * 64) + lane;
#pragma unroll
            for (int f = 0; f < NF; f++) dst[f * 64] = src[f * 64];
            top = 64;
            __syncthreads();
        }
        RT_FINE(0)   // spill reload
        // ---- second pass of a small frame: when the pool runs low, hit records of the first pass are taken from the
        // list in HBM and put on top of it (here, at the top of the loop, no lane still reads a popped record).  The
        // list has four entries per pixel-sample, in hand-out order (scattered over the image, so neighbours in the
        // list are not neighbours in cost), valid where a child of the primary hit a scattering surface.
        if (PHASE == 2 && nspill == 0 && top <= kListLow && !exhausted && popc_mask(freemask) >= 8) {
            const ColdPtr ca = cold_args();
            const int want = min(kListChunk, popc_mask(freemask));
            unsigned long long base = 0;
            const int n = take(want, (unsigned long long)ca->list_cap, ((unsigned long long)ca->list_cap + PARTS - 1) / PARTS, base);
            if (n == 0) exhausted = true;
            else {
                const size_t cap = ca->list_cap;
                const uint32_t* lr = ca->list + (size_t)base + lane;
                const uint32_t lmeta = (lane < n) ? lr[(F_META / 64) * cap] : 0u;
                const bool valid = ((lmeta >> 13) & 7u) != 0u;
                const unsigned long long vm = __ballot(valid);
                const int nv = popc_mask(vm);
                const bool offer = ((freemask >> lane) & 1ull) != 0 && mbcnt64(freemask) < nv;
                if (offer) map[mbcnt64(freemask)] = (uint32_t)lane;
                freemask &= ~__ballot(offer);
                __syncthreads();
                if (valid) {
                    const int sl = (int)map[mbcnt64(vm)];
                    uint32_t* wb = rec_base(top + mbcnt64(vm));
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        uint32_t v = (f == F_META / 64) ? ((lmeta & ~0xffu) | (uint32_t)sl) : lr[f * cap];
                        wb[f * 64] = v;
                    }
                    sl_ps[sl] = (unsigned long long)lr[NF * cap] | ((unsigned long long)lr[(NF + 1) * cap] << 32);
                    sl_pix[sl] = lr[(NF + 2) * cap];
                    sl_pend[sl] = 1u;
                    { const unsigned long long z = local_zero(); sl_acc[sl * 3 + 0] = z; sl_acc[sl * 3 + 1] = z; sl_acc[sl * 3 + 2] = z; }
                }
                top += nv;
                __syncthreads();
            }
        }
        // ---- take the top records whose children fit in 64 lanes (prefix sum over ballots)
        int nchild = 0;
        uint32_t mp = 0u;
        const int top0 = top;
        const int nfree = BVH ? 64 - popc_mask(busymask) : 64;       // lanes that can take a new ray
        const int frank = BVH ? mbcnt64(~busymask) : lane;          // rank of this lane among the free lanes
        if (top > 0 && nfree > 0) {
            const int ridx = top - 1 - lane;
            const uint32_t rmeta = (ridx >= 0) ? rec_base(ridx)[F_META] : 0u;
            const int rn = (rmeta >> 13) & 7;
            const unsigned long long b0 = __ballot(rn & 1), b1 = __ballot(rn & 2), b2 = __ballot(rn & 4);
            const int excl = mbcnt64(b0) + 2 * mbcnt64(b1) + 4 * mbcnt64(b2);
            const bool take = rn > 0 && excl + rn <= nfree;
            const unsigned long long tk = __ballot(take);
            nchild = popc_mask(b0 & tk) + 2 * popc_mask(b1 & tk) + 4 * popc_mask(b2 & tk);
            if (take)
                for (int c = 0; c < rn; c++) map[excl + c] = (uint32_t)(lane * 4 + c);
            __syncthreads();
            if (!busy && frank < nchild) mp = map[frank];
            top -= popc_mask(tk);
            __syncthreads();                    // map is reused for the slot allocation below
        }
        RT_FINE(1)   // list install + pop
        // ---- the pool is drained: start new pixel-samples in the free lanes
        int nprim = 0;
        if (PHASE != 2 && top == 0 && nspill == 0 && nchild < nfree && !exhausted) {
            // pull exactly as many pixel-samples as there are free lanes (and free slots): one returning
            // atomic per refill keeps the hand-out fine-grained, so no wave is left holding a queue of
            // expensive pixels while the others have finished
            const int want = min(nfree - nchild, popc_mask(freemask));
            if (want > 0) {
                const ColdPtr ca = cold_args();
                unsigned long long base = 0;
                unsigned long long limit = ca->total;
                if (PHASE == 1) {
                    // primaries cost about the same everywhere: every wave owns one contiguous range of the hand-out
                    // order, no shared counter (all the waves of a small frame would hit it in the same instant)
                    const unsigned long long per = (limit + gridDim.x - 1) / gridDim.x;
                    base = (unsigned long long)blockIdx.x * per;
                    limit = (base + per < limit) ? base + per : limit;
                    base += (unsigned long long)n_refill;
                } else {
                    const int n = take(want, limit, kStrips ? (unsigned long long)ca->part_items : limit, base);
                    limit = base + (unsigned long long)n;       // (n == 0: nothing left)
                    if (n == 0) base = limit = 0;
                }
                if (base >= limit) exhausted = true;
                else {
                    res_next = base;
                    nprim = (base + (unsigned long long)want <= limit) ? want : (int)(limit - base);
                    if (PHASE == 1) n_refill += (uint32_t)nprim;
                }
            }
        }
        if (nchild + nprim == 0 && busymask == 0ull) {
            if (nspill > 0) continue;           // (cannot happen: a reloaded block always yields children)
            if (exhausted) break;               // nothing in the pool, nothing left to start
            if (PHASE == 2) continue;         // a hand-out without a valid entry (sky, lights): take the next one
            overflow = true; break;             // cannot happen (all slots are free here and the hand-out is not used up): never spin
        }
        if (COUNT) n_iter++;
        RT_FINE(2)   // refill atomic
        const bool is_child = !busy && frank < nchild;
        const bool fresh_prim = !busy && !is_child && frank < nchild + nprim;
        int ckind = 0;                          // kind of the surface this lane's child ray leaves (0 = primary)
        if (!busy) {
            O = F3(0, 0, 0); Dn = F3(0, 0, 0); beta = F3(1.0f, 1.0f, 1.0f);
            slot = 0; clevel = 0; cnode = 0;
            alive = false;                      // this lane intersects a ray
            owner = false;                      // this lane retires one pending unit of its slot
            is_prim = fresh_prim;
            prim_first = false;
        }

        if (nprim > 0) {
            // ---- slot allocation: the lane whose index is a free slot number offers it by rank
            const bool offer = ((freemask >> lane) & 1ull) != 0 && mbcnt64(freemask) < nprim;
            if (offer) map[mbcnt64(freemask)] = (uint32_t)lane;
            freemask &= ~__ballot(offer);
            __syncthreads();
            if (fresh_prim) {
                // ============ rayTraceBase: primary ray (Camera.fs:8-39, Scene.fs:98-102) ============
                const ColdPtr ca = cold_args();
                slot = (int)map[frank - nchild];
                const unsigned long long q = res_next + (unsigned long long)(frank - nchild);  // compact index
                // Which pixel-sample position q is.  Large frames: consecutive positions are the samples of one pixel
                // (they meet in the pixel cache of the flush).  Frames with few refills per wave (PHASE 3): consecutive
                // positions are the same sample of pixels far apart — a refill of 32 neighbouring samples is 32 ray trees
                // of the same size, and when they are floor pixels (thousands of rays each) that one refill is as much
                // work as a wave's whole fair share of an 800x640x16 frame: measured with tools/wave_times.py, the last
                // wave of that frame ended 0.77 ms after the counter ran out (10 % of the frame; 9 % on the mesh frame,
                // 15 % on the 1M-sphere frame).
                // Between the two, G samples of a pixel stay together (the BVH kernels: neighbouring primaries walk the
                // tree together, which is worth more there than the last bit of balance).
                constexpr bool smajor = PHASE == 3;
                int s; uint32_t o;
                if (smajor) {
                    // position j inside the strip this refill came from (cur_part: take() has just set it); a PHASE 3 frame
                    // has fewer than 2^27 pixel-samples, so 32-bit arithmetic
                    const int gs = ca->group_shift;
                    const uint32_t first = (uint32_t)cur_part * ca->part_pixels;            // first compact pixel position of the strip
                    const uint32_t np = min(ca->part_pixels, ca->npix - first);
                    const uint32_t j = (uint32_t)(q - (unsigned long long)cur_part * (unsigned long long)ca->part_items);
                    const uint32_t qt = j >> gs, sg = qt / np, l = qt - sg * np;
                    s = (int)(sg << gs) + (int)(j & ((1u << gs) - 1u));
                    const uint32_t m = (np == ca->part_pixels) ? ca->scatter : ca->scatter_tail;
                    o = first + (uint32_t)(((unsigned long long)l * (unsigned long long)m) % (unsigned long long)np);
                } else {
                    const unsigned long long qq = q / (unsigned long long)spp;
                    s = (int)(q - qq * (unsigned long long)spp);
                    // Consecutive hand-outs go to pixels far apart in the image (a fixed bijection), so that every
                    // refill mixes cheap and expensive pixels: with row-major hand-out one wave would start on 64
                    // neighbouring floor pixels (thousands of rays each) and finish long after the others.
                    o = (uint32_t)((qq * (unsigned long long)ca->scatter) % (unsigned long long)ca->npix);
                }
                const int k = (int)(o / (uint32_t)ca->H), py = (int)(o - (uint32_t)k * (uint32_t)ca->H);
                const int px = ca->x_begin + k * ca->x_step;
                const uint64_t pixel = (uint64_t)px * (uint64_t)ca->H + (uint64_t)py;
                sl_ps[slot] = pixel * (uint64_t)spp + (uint64_t)s;
                sl_pix[slot] = o;
                sl_pend[slot] = 1u;
                { const unsigned long long z = local_zero(); sl_acc[slot * 3 + 0] = z; sl_acc[slot * 3 + 1] = z; sl_acc[slot * 3 + 2] = z; }
                f3 rv = F3(0.0f, 0.0f, 0.0f);
                if (ca->jitter) rv = unit_sphere_sample(a.seed_lo, a.seed_hi, pixel);
                const float fw = ca->fw, fh = ca->fh, tan_half = ca->tan_half;
                float xoff = rv.x / fw, yoff = rv.y / fh;
                float pxp = ((float)px + 0.5f) + xoff, pyp = ((float)py + 0.5f) + yoff;
                float xn = pxp / fw, yn = pyp / fh;
                float xs = 2.0f * xn - 1.0f, ys = 1.0f - 2.0f * yn;
                float xc = xs * ca->aspect * tan_half, yc = ys * tan_half;
                f3 dirCS = normalize(F3(xc, yc, -1.0f));
                f3 dirT = F3((dirCS.x * ca->rot[0] + dirCS.y * ca->rot[3]) + dirCS.z * ca->rot[6],
                             (dirCS.x * ca->rot[1] + dirCS.y * ca->rot[4]) + dirCS.z * ca->rot[7],
                             (dirCS.x * ca->rot[2] + dirCS.y * ca->rot[5]) + dirCS.z * ca->rot[8]);
                Dn = make_ray_dir(normalize(dirT));
                O = F3(ca->origin[0], ca->origin[1], ca->origin[2]);
                alive = true; owner = true;
                prim_first = (s == 0);
            }
        }

        RT_FINE(3)   // primary generation
        if (is_child) {
            // ============ rayTrace: one child of a popped record (Scene.fs:62-70) ============
            const int rec = top0 - 1 - (int)(mp >> 2);
            const int child = (int)(mp & 3);
            const uint32_t* rb = rec_base(rec);
            O = F3(__uint_as_float(rb[F_PX]), __uint_as_float(rb[F_PY]), __uint_as_float(rb[F_PZ]));
            const f3 Din = F3(__uint_as_float(rb[F_DX]), __uint_as_float(rb[F_DY]), __uint_as_float(rb[F_DZ]));
            const f3 pbeta = F3(__uint_as_float(rb[F_BX]), __uint_as_float(rb[F_BY]), __uint_as_float(rb[F_BZ]));
            const uint32_t meta = rb[F_META];
            uint64_t node = rb[F_NODE];
            if (NODE64) node |= (uint64_t)rb[F_NODEHI] << 32;
            slot = meta & 0xff;
            const int level = (meta >> 8) & 31, n = (meta >> 13) & 7;
            const int kind = (meta >> 16) & 3, gty = (meta >> 18) & 3;
            const uint32_t surf = BVH ? rb[F_SURF] : (meta >> 20);
            clevel = level + 1;
            ckind = kind;
            cnode = NODE64 ? 4ull * node + (uint64_t)child + 1ull : (uint64_t)(4u * (uint32_t)node + (uint32_t)child + 1u);
            const uint64_t ps = sl_ps[slot];                      // the RNG counter
            if (BVH) {
                // every child lane holds a pending unit of its own (child 0 inherits the record's), because
                // the children of one record may finish in different iterations
                owner = true;
                if (child == 0 && n > 1) atomicAdd(&sl_pend[slot], (uint32_t)(n - 1));
            } else owner = (child == 0);

            const ShadeRec* sr = shade + surf;
            const float4 al = *reinterpret_cast<const float4*>(&sr->albedo[0]);
            const float4 nc = *reinterpret_cast<const float4*>(&sr->nc[0]);
            const f3 albedo = F3(al.x, al.y, al.z);
#ifdef RT_PHASE_FINE
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            RT_FINE(4)   // record read + ShadeRec gather
#endif

            // The three Scatter/Fresnel routines are interleaved so that their common steps (one
            // Vector3.Normalize each) run once for all lanes with per-lane operands: every lane still
            // performs exactly the operations of its own kind, in the reference's order.
            // Hitable.NormalForSurfacePoint: Sphere.fs:37-38, Plane.fs:50-51
            f3 n0 = F3(nc.x, nc.y, nc.z);
            if (gty == kGeomSphere) n0 = normalize(O - n0);
            // Metal (Surface.fs:94) and Dielectric (Surface.fs:136) renormalise it, Lambertian does not
            f3 n1 = n0;
            if (kind != kKindLambert) n1 = normalize(n0);

            alive = true;
            f3 w = n1;              // the vector whose Normalize gives the scattered / modified / refracted direction
            float cos_l = 0.0f;     // Lambertian: cos(theta) = sample.Y (Surface.fs:73)
            // one Philox block per child ray: the hemisphere sample of Lambertian / Metal and, under the single-path
            // estimator, the RandomFloat of Dielectric.Scatter (Surface.fs:157); the branching Dielectric draws nothing
            f3 rn = F3(0.0f, 1.0f, 0.0f);
            float rand_float = 0.0f;
            if ((kind != kKindDielectric) | SINGLE) rn = hemisphere_sample(a.seed_lo, a.seed_hi, ps, cnode, rand_float);
            if (kind != kKindDielectric) {
                // Lambertian.Scatter (Surface.fs:66-83) / Metal.Scatter (Surface.fs:93-109): hemisphere
                // sample in the frame around the normal
                f3 nt, nb;
                frame_around(n1, nt, nb);
                // Vector4.Transform(rn, ChangeOfBase nt n nb): rows nt, n, nb; M4x = 0
                const f3 ns = F3(((rn.x * nt.x + rn.y * n1.x) + rn.z * nb.x) + 0.0f,
                                 ((rn.x * nt.y + rn.y * n1.y) + rn.z * nb.y) + 0.0f,
                                 ((rn.x * nt.z + rn.y * n1.z) + rn.z * nb.z) + 0.0f);
                const float fuzz = al.w;
                w = (kind == kKindLambert) ? ns : n1 * (1.0f - fuzz) + ns * fuzz;
                cos_l = rn.y;
            }
            float F = 1.0f; bool single = true;
            bool refr_pick = (child == 1);      // this lane follows the refracted ray
            if (kind == kKindDielectric) {
                // Surface.fs:124-153 CalcFresnel + :168-181 GenerateSamples (branching) / :155-166 Scatter (single path)
                float eta_i, eta_t; f3 fn;
                if (dot(Din, n1) > 0.0f) { eta_i = al.w; eta_t = 1.0f; fn = neg(n1); }
                else                     { eta_i = 1.0f; eta_t = al.w; fn = n1; }
                const float cos_i = dot(Din, neg(fn));
                const float eta = div_rn(eta_i, eta_t);
                const float disc = 1.0f - (eta * eta) * (1.0f - (cos_i * cos_i));
                if (disc > 0.0f) {
                    const float q = div_rn(eta_t - eta_i, eta_t + eta_i);
                    const float R0 = q * q;
                    F = R0 + (1.0f - R0) * pow5(1.0f - cos_i);
                }
                single = rounds_to(F, 1e3f, 1e3f);                  // Round(F,3) = 1: reflection only
                w = (Din + fn * cos_i) * eta - fn * sqrt_rn(disc);  // refraction direction before Normalize
                if (SINGLE) refr_pick = !(rand_float <= F);    // Surface.fs:158: reflect when RandomFloat <= reflectProb (F = 1 under TIR)
                else if (child == 1) alive = !single;
            }
            const f3 u = normalize(w);
            // Metal reflects about the fuzzed normal u, Dielectric about n1; Lambertian keeps u
            f3 r = u;
            if (kind != kKindLambert) r = normalize(reflect(Din, (kind == kKindMetal) ? u : n1));
            const bool use_u = (kind == kKindLambert) | ((kind == kKindDielectric) & refr_pick);
            Dn = make_ray_dir(use_u ? u : r);

            f3 shd;                 // shading_i; folded into beta below
            float inv_n = 1.0f;
            if (kind == kKindLambert) {
                const float pdf = 1.0f / (2.0f * 3.14159265358979323846f);
                const f3 brdf = albedo * (1.0f / 3.14159265358979323846f);   // Vector3 / float = v * (1/f)
                shd = brdf * div_rn(cos_l, pdf);
            } else if (kind == kKindMetal) {
                shd = albedo * (1.0f / 1.0f);
            } else {
                // branching: (2*albedo*F)/2 == albedo*F exactly; single path: BRDF*(cos/PDF) = albedo*(1/1) (Surface.fs:28,162)
                shd = albedo * (SINGLE ? (1.0f / 1.0f) : (child == 0) ? F : (1.0f - F));
            }
            if (kind != kKindDielectric)
                inv_n = (n == 1) ? 1.0f : (n == 2) ? 0.5f : (n == 3) ? (1.0f / 3.0f) : 0.25f;   // 1/n, correctly rounded
            beta = (pbeta * shd) * inv_n;

            if (alive && clevel > a.max_depth + 1) {
                // depth guard (Scene.fs:52-54): the child returns backgroundColor unintersected
                alive = false;
                const ColdPtr ca = cold_args();
                add_radiance(slot, beta * F3(ca->bg[0], ca->bg[1], ca->bg[2]));
            }
        }

        // rays by kind (wave-uniform counts)
        if (COUNT) {
        n_lamb += (uint32_t)popc_mask(__ballot(!busy && alive && ckind == kKindLambert));
        n_metal += (uint32_t)popc_mask(__ballot(!busy && alive && ckind == kKindMetal));
        n_diel += (uint32_t)popc_mask(__ballot(!busy && alive && ckind == kKindDielectric));
        }

        // ============ findClosestIntersection + hit handling, all 64 lanes ============
        bool push = false;
        f3 P = F3(0, 0, 0);
        uint32_t hinfo = 0;
        int hn = 0;                             // children of the hit this lane found
        RT_PHASE(ph_gen)
        if (alive && !busy) {
            closest_hit(geom, a.nsurf, O, Dn, hidx, tcur);
            trav_cur = (BVH && bvh.nnodes) ? 0u : kBvhDone; trav_sp = 0; trav_lo = 0;   // BVH: start at the root (an empty tree has none)
        }
        nrays += (uint32_t)popc_mask(__ballot(alive && !busy));
        RT_PHASE(ph_scan)
        if (BVH) {
            // new and suspended rays walk the tree together; leave early only when other work can fill the lanes
            // (an iteration that could not start a single new ray — no free slot, nothing poppable — must not
            // leave early either, or the wave would spin without progress)
            const bool refill = (nchild + nprim > 0) && (top > 0 || nspill > 0 || !exhausted);
            if (!alive) trav_cur = kBvhDone;
            auto backing = [&]() { return cold_args()->gstack + (size_t)blockIdx.x * (kBvhStack * 64) + lane; };
            // (the threshold is wave-uniform by construction, but arrives in a vector register, and the comparison with it
            // makes the round loop a divergent loop for the compiler — exec-mask bookkeeping, loop-carried scalars in vector
            // registers.  Handed over through readfirstlane: 4K mixed frame +0.7 %; the sample-major frames lost 1 % to it
            // while they carried the statistics through the loop, and gain 0.5 - 1 % since they do not, rt_scene_counters.)
            const int ma = __builtin_amdgcn_readfirstlane(refill ? a.min_active : 0);
            closest_hit_bvh<RING, true, kBvhInnerMin, kBvhLeafMin, COUNT>(bvh, O, Dn, bvh_stack, 64, map + lane, backing, tcur, hidx, trav_cur, trav_sp, trav_lo,
                                            ma, n_nodes, n_prims, n_rin, n_rlf);
            busy = alive && trav_cur != kBvhDone;
            busymask = __ballot(busy);
        }
        RT_PHASE(ph_trav)
        if (alive && !busy) {
            const float t = tcur;
            bool hit = hidx >= 0;
            float4 em = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (hit) {
                em = *reinterpret_cast<const float4*>(&shade[hidx].emission[0]);
                hinfo = __float_as_uint(em.w);
                // Scene.fs:79: the primary's closest candidate is re-tested with dotView; planes then
                // cull at TMax/dotView (Plane.fs:46).  A failed re-test is a miss, not the next hit.
                const uint32_t hg = (hinfo >> 5) & 3;
                if (is_prim && (hg == kGeomPlaneInf || hg == kGeomPlaneRect)) {
                    const ColdPtr ca = cold_args();
                    const float dot_view = dot(F3(ca->look_n[0], ca->look_n[1], ca->look_n[2]), Dn);   // Scene.fs:76
                    hit = t <= (kTMax / dot_view);
                }
            }
            if (is_prim && prim_first) {                               // Scene.fs:80 / :29
                const ColdPtr ca = cold_args();
                const uint32_t prim_o = sl_pix[slot];
                ca->depth[prim_o] = hit ? t : kFltMax;
                if (ca->ids) ca->ids[prim_o] = hit ? shade[hidx].id : 0ull;
                if (hit) maxd = fmaxf(maxd, t);
            }
            if (hit) {
                if (em.x != 0.0f || em.y != 0.0f || em.z != 0.0f) add_radiance(slot, beta * F3(em.x, em.y, em.z));
                // SampleCount of the surface hit (Surface.fs:20,63,92,115); the single-path estimator takes one child
                hn = (hinfo >> 2) & 7;
                if (hinfo & 0x80u) hn = a.lamb;
                if (SINGLE) hn = min(hn, 1);
                // children of a hit at level max_depth+1 are all cut by the depth guard; with a black
                // background they contribute exactly 0 and are skipped
                push = hn > 0 && (clevel <= a.max_depth || a.bg_nonzero);
                P = O + Dn * t;                                        // Core.fs:59
            } else if (a.bg_nonzero) {
                const ColdPtr ca = cold_args();
                add_radiance(slot, beta * F3(ca->bg[0], ca->bg[1], ca->bg[2]));
            }
        }
        // ---- compaction: hits on scattering surfaces become records, deepest on top
        unsigned long long pm = __ballot(push);
        int npush = popc_mask(pm);
        if (PHASE == 1 && __ballot(alive && !busy && clevel == 1) != 0ull) {
            // ---- first pass of a small frame: every child of a primary that has finished its closest-hit query fills
            // its entry of the list in HBM — the hit record if its subtree is to be traced, else stays without children
            // (the meta plane of the list is cleared before the launch: entries of children that never existed)
            if (alive && !busy && clevel == 1 && push) {
                const ColdPtr ca = cold_args();
                const size_t cap = ca->list_cap;
                const unsigned long long ps = sl_ps[slot];
                const uint32_t o = sl_pix[slot];
                const unsigned long long o64 = ((unsigned long long)o * (unsigned long long)ca->scatter_inv) % (unsigned long long)ca->npix;
                const unsigned long long q = o64 * (unsigned long long)spp + ps % (unsigned long long)spp;
                uint32_t* lw = ca->list + (size_t)((unsigned long long)(cnode - 1) * ca->total + q);
                lw[(F_PX / 64) * cap] = __float_as_uint(P.x); lw[(F_PY / 64) * cap] = __float_as_uint(P.y); lw[(F_PZ / 64) * cap] = __float_as_uint(P.z);
                lw[(F_DX / 64) * cap] = __float_as_uint(Dn.x); lw[(F_DY / 64) * cap] = __float_as_uint(Dn.y); lw[(F_DZ / 64) * cap] = __float_as_uint(Dn.z);
                lw[(F_BX / 64) * cap] = __float_as_uint(beta.x); lw[(F_BY / 64) * cap] = __float_as_uint(beta.y); lw[(F_BZ / 64) * cap] = __float_as_uint(beta.z);
                lw[(F_META / 64) * cap] = make_meta(0, clevel, hn, hinfo & 3, (hinfo >> 5) & 3, BVH ? 0 : hidx);
                if (BVH) lw[(F_SURF / 64) * cap] = (uint32_t)hidx;
                lw[(F_NODE / 64) * cap] = (uint32_t)cnode;
                if (NODE64) lw[(F_NODEHI / 64) * cap] = (uint32_t)(cnode >> 32);
                lw[NF * cap] = (uint32_t)ps; lw[(NF + 1) * cap] = (uint32_t)(ps >> 32); lw[(NF + 2) * cap] = o;
            }
            if (clevel == 1) push = false;
            pm = __ballot(push);
            npush = popc_mask(pm);
        }
        if (top + npush > a.nblk * 64) {
            // ---- no room: spill the bottom block of the ring (top >= 64 here, so it is full)
            const ColdPtr cs = cold_args();
            if (nspill < cs->max_spill) {
                uint32_t* dst = cs->spill + ((size_t)blockIdx.x * cs->max_spill + nspill) * (NF * 64) + lane;
                const uint32_t* src = pool + base_blk * (NF * 64) + lane;
#pragma unroll
                for (int f = 0; f < NF; f++) dst[f * 64] = src[f * 64];
                nspill++;
                base_blk = (base_blk + 1 == a.nblk) ? 0 : base_blk + 1;
                top -= 64;
                __syncthreads();
            } else overflow = true;             // beyond 64 records per level: a bug
        }
        if (top + npush <= a.nblk * 64) {
            if (push) {
                const int pos = top + (npush - 1 - mbcnt64(pm));
                uint32_t* wb = rec_base(pos);
                wb[F_PX] = __float_as_uint(P.x); wb[F_PY] = __float_as_uint(P.y); wb[F_PZ] = __float_as_uint(P.z);
                wb[F_DX] = __float_as_uint(Dn.x); wb[F_DY] = __float_as_uint(Dn.y); wb[F_DZ] = __float_as_uint(Dn.z);
                wb[F_BX] = __float_as_uint(beta.x); wb[F_BY] = __float_as_uint(beta.y); wb[F_BZ] = __float_as_uint(beta.z);
                wb[F_META] = make_meta(slot, clevel, hn, hinfo & 3, (hinfo >> 5) & 3, BVH ? 0 : hidx);
                if (BVH) wb[F_SURF] = (uint32_t)hidx;
                wb[F_NODE] = (uint32_t)cnode;
                if (NODE64) wb[F_NODEHI] = (uint32_t)(cnode >> 32);
                atomicAdd(&sl_pend[slot], 1u);
            }
            top += npush;
        }
        // ---- retire: every popped record and every primary gives back one pending unit of its slot
        //      (after this iteration's pushes, so the count reaches 0 only when the whole tree is done)
        bool done = false;
        if (owner && !busy) done = atomicSub(&sl_pend[slot], 1u) == 1u;
        unsigned long long dm = __ballot(done);
        if (dm) {
            // Finished pixel-samples are added to their pixels (order-independent integer adds).
            const ColdPtr ca = cold_args();
            if (kPixelCache && spp >= kPixelCacheMinSpp) {
                // (One-pass linear-scan kernel only: the BVH kernels have no three registers to spare and their HBM traffic
                // is the tree, not the output; two-pass frames are small.)  Frames with many samples per pixel: through a 16-entry direct-mapped cache held in registers — lanes
                // 4e..4e+2 keep the three channel sums (and the pixel index) of entry e = pixel & 15.  The hand-out gives
                // a wave consecutive samples of the same pixel, so most finished samples meet their pixel in the cache and
                // cost no memory traffic; an entry goes to HBM when another pixel claims it (and at the end of the
                // kernel), its three 8-byte adds travelling as one 32-byte atomic request.  (Before: one request per
                // finished pixel-SAMPLE — 17.7 GB of HBM writes per 4K 64-spp frame for 166 MB of output, 530 M atomics.)
                const uint32_t mypix = done ? sl_pix[slot] : 0u;
                const int ch = lane & 3;
                while (dm) {
                    const int l = __builtin_ctzll(dm);
                    dm &= dm - 1;
                    const int sl = __builtin_amdgcn_readlane(slot, l);
                    const uint32_t pix = (uint32_t)__builtin_amdgcn_readlane((int)mypix, l);
                    freemask |= 1ull << sl;
                    if ((uint32_t)(lane >> 2) == (pix & 15u) && ch < 3) {
                        const unsigned long long v = sl_acc[sl * 3 + ch];
                        if (cpix == pix) cacc += v;
                        else {
                            if (cpix != kNoPixel) atomicAdd(ca->accum + (size_t)cpix * 4 + ch, cacc);
                            cacc = v; cpix = pix;
                        }
                    }
                }
            } else {
                // Few samples per pixel (nothing to merge): the wave does it cooperatively, 4 lanes per finished slot, one
                // per channel, so the three 8-byte adds of a pixel (32-byte aligned) travel as one atomic request instead
                // of three wave-instructions with one lane each (measured: 33 B of HBM write per 8-byte add when issued
                // separately).
                const int ndone = popc_mask(dm);
                if (done) map[mbcnt64(dm)] = (uint32_t)slot;
                __syncthreads();
                int fl = lane;
                asm volatile("" : "+v"(fl));        // (made here: hoisted to the kernel's entry, lane >> 2 and friends cost registers for the whole loop)
                for (int base = 0; base < ndone; base += 16) {
                    const int k = base + (fl >> 2), ch = fl & 3;
                    if (k < ndone && ch < 3) {
                        const int sl = (int)map[k];
                        atomicAdd(ca->accum + (size_t)sl_pix[sl] * 4 + ch, sl_acc[sl * 3 + ch]);
                    }
                }
                while (dm) {
                    const int l = __builtin_ctzll(dm);
                    dm &= dm - 1;
                    freemask |= 1ull << __builtin_amdgcn_readlane(slot, l);
                }
            }
        }
        __syncthreads();
        RT_PHASE(ph_shade)
    }

    const ColdPtr ca = cold_args();
    if (kPixelCache && cpix != kNoPixel) atomicAdd(ca->accum + (size_t)cpix * 4 + (lane & 3), cacc);     // what is left in the pixel cache
    // ================= statistics ================================================================
    unsigned long long* stats = ca->stats ? ca->stats + (size_t)(blockIdx.x % kStatsParts) * kStatsWords : nullptr;
    if (stats) {
        const unsigned long long r = nrays, nn = n_nodes, np = n_prims;      // wave-uniform counts
        float m = maxd;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) {
            atomicAdd(stats + 0, r);
            atomicMax(reinterpret_cast<unsigned int*>(stats + 1), __float_as_uint(m));
            if (COUNT) {
                atomicAdd(stats + 2, (unsigned long long)n_iter);
                atomicAdd(stats + 3, (unsigned long long)n_lamb);
                atomicAdd(stats + 4, (unsigned long long)n_metal);
                atomicAdd(stats + 5, (unsigned long long)n_diel);
            }
            atomicAdd(stats + 6, __builtin_readcyclecounter() - t_start);
            atomicAdd(stats + 7, 1ull);
            atomicMax(stats + 18, __builtin_readcyclecounter() - t_start);
#ifdef RT_WAVE_TIMES
            if (PHASE == RT_WAVE_TIMES) {   // debug build (tools/wave_times.py): when the waves of one pass start and end
                const unsigned long long wt_end = wall_clock64();
                atomicAdd(stats + 12, wt_start); atomicAdd(stats + 13, wt_end);
                atomicMax(stats + 14, ~wt_start); atomicMax(stats + 15, wt_start);
                atomicMax(stats + 16, ~wt_end); atomicMax(stats + 17, wt_end);
                atomicAdd(stats + 19, 1ull);
            }
#endif
            if (BVH && COUNT) {
                atomicAdd(stats + 8, nn); atomicAdd(stats + 9, np);
                atomicAdd(stats + 10, (unsigned long long)n_rin); atomicAdd(stats + 11, (unsigned long long)n_rlf);
#ifdef RT_PHASE_TIMING
                atomicAdd(stats + 12, ph_gen); atomicAdd(stats + 13, ph_scan); atomicAdd(stats + 14, ph_trav); atomicAdd(stats + 15, ph_shade);
#endif
#ifdef RT_PHASE_FINE
                atomicAdd(stats + 16, ph_fine[0]); atomicAdd(stats + 17, ph_fine[1]); atomicAdd(stats + 19, ph_fine[2]);
                atomicAdd(stats + 8, ph_fine[3]); atomicAdd(stats + 9, ph_fine[4]);    // (debug build: on top of the node / primitive counts)
#endif
            }
        }
    }
    if (overflow && lane == 0) atomicOr(ca->status, kStatusOverflow);
}

// Scene.fs:109,114: mean over samples, gamma 2, alpha 1.  One thread per pixel.  The blocks after the last pixel block
// fold the kStatsParts partial statistics arrays into the caller's (words 1 and 18 are maxima, the others sums), one
// wave per word: a launch of its own costs a small frame more than the work does.
__global__ __launch_bounds__(256) void resolve_kernel(const unsigned long long* __restrict__ accum, float* __restrict__ rgba,
                                                      long long npix, int spp, unsigned int pixel_blocks,
                                                      const unsigned long long* __restrict__ parts, unsigned long long* __restrict__ stats_out)
{
    if (blockIdx.x >= pixel_blocks) {
        const int w = (int)(blockIdx.x - pixel_blocks) * 4 + (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63u);
        if (w >= kStatsWords) return;
#ifdef RT_WAVE_TIMES
        const bool is_max = (w == 1 || w == 18 || (w >= 14 && w <= 17));
#else
        const bool is_max = (w == 1 || w == 18);
#endif
        unsigned long long v = 0ull;
        for (int p = l; p < kStatsParts; p += 64) {
            const unsigned long long x = parts[(size_t)p * kStatsWords + w];
            v = is_max ? (x > v ? x : v) : v + x;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long x = __shfl_xor(v, off);
            v = is_max ? (x > v ? x : v) : v + x;
        }
        if (l == 0) stats_out[w] = v;
        return;
    }
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= npix) return;
    float c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        const float sum = (float)((double)(long long)accum[o * 4 + ch] * 0x1p-32);   // one rounding
        c[ch] = sum;
    }
    const f3 avg = divs(F3(c[0], c[1], c[2]), (float)spp);
    *reinterpret_cast<float4*>(rgba + 4 * o) = make_float4(sqrtf(avg.x), sqrtf(avg.y), sqrtf(avg.z), 1.0f);
}

// rt_render_multi: the compact [ncols][H] block of partition `r` of `n` (columns r, r+n, ...) goes to its columns of
// the whole [W][H] frame (Array2D layout px*H + py, Scene.fs:28).  The frame may live in the HBM of another GPU: the
// stores then travel over xGMI as 16-byte peer writes, consecutive threads to consecutive addresses along py.
__global__ __launch_bounds__(256) void place_columns_kernel(const float* __restrict__ rgba, const float* __restrict__ depth,
                                                            const unsigned long long* __restrict__ ids,
                                                            float* __restrict__ f_rgba, float* __restrict__ f_depth,
                                                            unsigned long long* __restrict__ f_ids, int ncols, int H, int r, int n)
{
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= (long long)ncols * H) return;
    const int k = (int)(o / H), py = (int)(o - (long long)k * H);
    const long long dst = frame_index(r, n, k, H, py);
    *reinterpret_cast<float4*>(f_rgba + 4 * dst) = *reinterpret_cast<const float4*>(rgba + 4 * o);
    f_depth[dst] = depth[o];
    if (ids) f_ids[dst] = ids[o];
}

// findClosestIntersection as a batch query (Surface.fs:49-58): one ray per thread.
__global__ __launch_bounds__(256) void find_closest_kernel(const GeomRec* __restrict__ geom, int nsurf, const BvhRef bvh,
                                                           long long nrays,
                                                           const float* __restrict__ o, const float* __restrict__ d,
                                                           float* __restrict__ t_out, int* __restrict__ idx_out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrays) return;
    int idx; float t;
    const f3 O = F3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), D = F3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    closest_hit(geom, nsurf, O, D, idx, t);
    uint32_t stack[kBvhStack];
    uint32_t nn = 0, np = 0;
    uint32_t cur = bvh.nnodes ? 0u : kBvhDone; int sp = 0, lo = 0;
    auto backing = []() -> uint32_t* { return nullptr; };
    closest_hit_bvh<kBvhStack, false, 1, 1, false>(bvh, O, D, stack, 1, stack, backing, t, idx, cur, sp, lo, 0, nn, np, nn, np);
    t_out[i] = t; idx_out[i] = idx;
}

}  // namespace rt
