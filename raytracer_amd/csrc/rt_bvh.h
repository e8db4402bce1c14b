// rt_bvh.h — closest hit through a flattened 4-wide BVH in HBM (large scenes: triangle meshes, many spheres).
//
// NOT in the reference: Geometry/AABB.fs is an empty module and findClosestIntersection is a linear scan
// (Surface.fs:49-58).  The BVH only prunes: every primitive that the linear scan would accept is still
// tested with exactly the same arithmetic, and ties in t are resolved to the lowest array index, which is
// what "first in array order wins" (strict '<', Surface.fs:53) means.  So the result equals the linear
// scan's bit for bit; tests compare against the brute-force oracle.
//
// Layout (built on the host, rt_bvh_build.h).  Traversal is bound by dependent, divergent memory requests
// (measured with a binary tree of 32-byte nodes: 28 visits per ray, ~1.6 TB/s of 64-byte random reads,
// 0.65 Grays/s on the 100k-triangle mesh), so a node holds FOUR child boxes in exactly one 128-byte line:
//   nodes[]   8 float4 per node: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] ref[4] pad
//             ref: 0 = empty slot (its box lies at +infinity and is never hit); bit 31 set = leaf,
//             (ref & 0x7fffffff) >> 3 = first primitive slot, ref & 7 = primitive count (1..kBvhLeafMax);
//             otherwise the index of an inner node (never 0 = root)
//   pdata[]   3 float4 per primitive slot, in BVH order so a leaf's records are contiguous:
//             sphere   = (centre, r^2) (0,0,0,0)       (0,    surface index, 0, 0)
//             triangle = (v0, e1.x)    (e1.yz, e2.xy)  (e2.z, surface index, 1, 0)
// Boxes are padded on the host so that the (inexact) slab test never rejects a primitive the exact test hits.
#pragma once
#include "rt_device.h"

namespace rt {

struct BvhRef {
    const float4* __restrict__ nodes;     // 8 float4 per node, followed in the same allocation by the primitive records
    uint32_t pdata_off;                   // byte offset of the records (3 float4 per primitive slot) from `nodes`
    int nnodes;
    // spheres in the tree: the ray is walked as a cone (closest_hit_bvh).  cone_k = bound of the sphere test's rounding error
    // per squared distance (0: no spheres), rmin / rmax = smallest / largest radius, reach = sum over the axes of the
    // largest |coordinate| of the root box (|c - O| <= reach + |O|_1 for every centre c)
    float cone_k, rmin, rmax, reach;
    // triangles in the tree: their test is off by what a shift of the ray by up to ~lin_s |O - v0| would do; the boxes'
    // own padding covers that for rays that cannot get further than lin_far from a primitive (FLT_MAX: no triangles)
    float lin_s, lin_far;
};

constexpr int kBvhLeafMax = 2;           // primitives per leaf (3 bits in a leaf ref)
// References as the device sees them (rt_scene_create rewrites the builder's): HALF THE BYTE OFFSET of the child from the
// start of the allocation — node index << 6 for an inner node, (offset of the leaf's first record / 16) << 3 | count with
// bit 31 set for a leaf — so that (ref & 0x7ffffff8) << 1 is the address of either, without a branch, a multiply by 48 or
// the records' base (0: A/B build, node indices and record indices as the builder writes them).
#ifndef RT_BVH_LATE_SORT
#define RT_BVH_LATE_SORT 1    // 0: A/B build, the whole sort before the next round's loads
#endif
#ifndef RT_BVH_ISSUE_ASM
#define RT_BVH_ISSUE_ASM 1    // 0: A/B build, the addresses of a round's loads as the compiler writes them
#endif
#ifndef RT_BVH_REF_OFFSETS
#define RT_BVH_REF_OFFSETS 1
#endif
static_assert(!RT_BVH_REF_OFFSETS || kBvhLeafMax == 2, "(ref & 2) * 24 is the offset of a leaf's second record");
// lanes that must have gathered at an inner node / at a leaf for that group to step (measured on the 100k-triangle, mixed
// and 1M-sphere scenes; re-tuned in round 2, when the rounds had become cheaper: 16/16 -> 16/8 is +3 % on the mixed scene)
#ifndef RT_BVH_INNER_MIN
#define RT_BVH_INNER_MIN 16
#endif
#ifndef RT_BVH_LEAF_MIN
#define RT_BVH_LEAF_MIN 8
#endif
constexpr int kBvhInnerMin = RT_BVH_INNER_MIN, kBvhLeafMin = RT_BVH_LEAF_MIN;
constexpr int kBvhStack = 64;             // host guarantees 3 * (tree depth) + 1 <= kBvhStack
// EXPERIMENT (profiles/r03_node_bytes.txt): log2 of the bytes from one node to the next.  7 = the 128-byte node itself;
// 8 = the same nodes 256 bytes apart (twice the node footprint in the caches at exactly the same instruction stream),
// which measures how much of a frame's time depends on the bytes a node occupies — the question behind quantised
// 64-byte nodes.
#ifndef RT_BVH_NODE_SHIFT
#define RT_BVH_NODE_SHIFT 7
#endif
constexpr int kBvhNodeShift = RT_BVH_NODE_SHIFT;
static_assert(kBvhNodeShift >= 7 && kBvhNodeShift <= 9, "a node is 128 bytes");
// EXPERIMENT, second half: RT_BVH_NODE_SPLIT (needs a node shift of 8) puts rows 4..6 of a node into the NEXT 128-byte
// line, so that every node visit touches two cache lines instead of one — twice the working set in lines, same
// instruction stream: the mirror image of a 64-byte node, which would halve it.
#ifndef RT_BVH_NODE_SPLIT
#define RT_BVH_NODE_SPLIT 0
#endif
static_assert(!RT_BVH_NODE_SPLIT || RT_BVH_NODE_SHIFT >= 8, "a split node needs 256 bytes");
// byte offset of row r (lo.x lo.y lo.z hi.x hi.y hi.z ref) inside a node
constexpr uint32_t node_row(int r) { return (RT_BVH_NODE_SPLIT && r >= 4) ? 64u + 16u * (uint32_t)r : 16u * (uint32_t)r; }

// (t, idx) is a better hit than (tmin, best): closer, or equally close with a lower array index
RT_DEV bool better_hit(bool ok, float t, int idx, float tmin, int best)
{
    return ok & ((t < tmin) | ((t == tmin) & (idx < best)));
}

typedef float v2f __attribute__((ext_vector_type(2)));
// Round 3, last session: the 8-entry stack ring addressed as raw LDS bytes (RT_BVH_RING2, closest_hit_bvh below).
#ifndef RT_BVH_RING2
#define RT_BVH_RING2 1        // 0: A/B build, the ring through generic pointers and a dump word
#endif
#ifndef RT_BVH_FOLD_SHRINK
#define RT_BVH_FOLD_SHRINK 1  // 0: A/B build, the entry parameter of every box is shrunk by a multiply of its own
#endif
typedef __attribute__((address_space(3))) uint32_t lds_u32;
RT_DEV uint32_t lds_addr_of(uint32_t* p) { return (uint32_t)(uintptr_t)(lds_u32*)p; }
RT_DEV void lds_store(uint32_t a, uint32_t v) { *(lds_u32*)(uintptr_t)a = v; }
RT_DEV uint32_t lds_load(uint32_t a) { return *(lds_u32*)(uintptr_t)a; }
#ifndef RT_BVH_PACK_OFF
#define RT_BVH_PACK_OFF 1     // 0: A/B build, six offset registers and seven selects per round
#endif

// Slab tests of two boxes at once (v_pk_fma_f32); returns the entry parameters, kFltMax when missed.
// `near*` / `far*` are the box planes the ray meets first / last on each axis (chosen by the sign of the
// direction when the node is loaded), so no per-axis min/max is needed.  Inexact on purpose (boxes are
// padded): (plane - O) / D is evaluated as fma(plane, 1/D, -O/D).  max/min ignore NaNs (0 * inf when a direction
// component is 0), which only ever widens the test.  '<= tmax' keeps boxes that can only tie with the
// current hit: a tie may still win on index.  Empty child slots are boxes at +infinity and always miss.
// The rounding error of a plane parameter grows with the ray, not with the scene: ~2.4e-7 * (|t| + 2 |O_k / D_k|) on
// axis k (one ulp of the reciprocal, the roundings of O * inv and of the FMA).  The host pads the boxes for the
// primitives' own roundings (scene scale); the ray's share is applied here, per axis: the near planes are evaluated with
// -O/D lowered by 1e-6 |O/D| and the far planes with it raised (two addends per axis, prepared once per ray: no
// instruction per box), and the entry is shrunk by 2e-6 relative before it is compared with the exit.  So a far camera,
// or a bounce ray that starts 500 units from a unit-scale mesh, cannot lose a hit
// (tests/test_gpu_bvh.py::test_find_closest_unit_scale_mesh_from_far_origins).  Per axis on purpose: a single slack of
// max_k |O_k / D_k| accepts every box within reach for a ray with one small direction component (measured: 1M spheres
// -22 %).
// (v_pk_fma_f32 reads a 64-bit register pair per operand; op_sel / op_sel_hi pick, per result half, which half of the pair
// is used.  The compiler does not use them for a broadcast: it builds a pair {c, c} per constant — 24 registers for the
// twelve per-ray constants of the slab test.  Written by hand, the near constant of an axis lives in the low half of a
// pair and the far constant in the high half: 12 registers.)
#ifndef RT_BVH_PK_ASM
#define RT_BVH_PK_ASM 1       // 0: A/B build, the compiler's own pairs
#endif
RT_DEV v2f pk_fma_bcast_lo(v2f a, v2f m, v2f c)
{
    v2f r;
    if (!RT_BVH_PK_ASM) return __builtin_elementwise_fma(a, v2f{ m.x, m.x }, v2f{ c.x, c.x });
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(m), "v"(c));
    return r;
}
RT_DEV v2f pk_fma_bcast_hi(v2f a, v2f m, v2f c)
{
    v2f r;
    if (!RT_BVH_PK_ASM) return __builtin_elementwise_fma(a, v2f{ m.y, m.y }, v2f{ c.y, c.y });
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(m), "v"(c));
    return r;
}
// m* = { reciprocal for the near planes, reciprocal for the far planes } of an axis, c* = { addend near, addend far }
RT_DEV void box_entry2(v2f nearx, v2f neary, v2f nearz, v2f farx, v2f fary, v2f farz, v2f mx, v2f my, v2f mz, v2f cx, v2f cy, v2f cz, float tmax, float& ta, float& tb)
{
    const v2f x0 = pk_fma_bcast_lo(nearx, mx, cx), x1 = pk_fma_bcast_hi(farx, mx, cx);
    const v2f y0 = pk_fma_bcast_lo(neary, my, cy), y1 = pk_fma_bcast_hi(fary, my, cy);
    const v2f z0 = pk_fma_bcast_lo(nearz, mz, cz), z1 = pk_fma_bcast_hi(farz, mz, cz);
    float tna, tfa, tnb, tfb;
    if (RT_BVH_PK_ASM) {
        // (the compiler does not know that the result of an asm statement is not a signalling NaN and would put a
        // canonicalising v_max x, x in front of every fmaxf / fminf that takes one: the maxima and minima are written out too)
        float m;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x0.x), "v"(y0.x), "v"(z0.x)); asm("v_max_f32_e32 %0, 0, %1" : "=v"(tna) : "v"(m));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x0.y), "v"(y0.y), "v"(z0.y)); asm("v_max_f32_e32 %0, 0, %1" : "=v"(tnb) : "v"(m));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x1.x), "v"(y1.x), "v"(z1.x)); asm("v_min_f32_e32 %0, %1, %2" : "=v"(tfa) : "v"(m), "v"(tmax));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x1.y), "v"(y1.y), "v"(z1.y)); asm("v_min_f32_e32 %0, %1, %2" : "=v"(tfb) : "v"(m), "v"(tmax));
    } else {
        tna = fmaxf(fmaxf(x0.x, y0.x), fmaxf(z0.x, 0.0f)); tfa = fminf(fminf(x1.x, y1.x), fminf(z1.x, tmax));
        tnb = fmaxf(fmaxf(x0.y, y0.y), fmaxf(z0.y, 0.0f)); tfb = fminf(fminf(x1.y, y1.y), fminf(z1.y, tmax));
    }
    // the |t| share of the rounding error, both sides at once: the entry shrunk by 2e-6 relative before it is compared
    // with the exit.  RT_BVH_FOLD_SHRINK: the exit is widened instead, once per ray — the far planes' reciprocals and
    // addends and `tmax` arrive multiplied by kFarWiden = 1 + 2.5e-6 (tn (1 - 2e-6) <= X implies tn <= X (1 + 2.5e-6) for
    // X >= 0, and for X < 0 both forms reject since tn >= 0) — so a box costs a compare and a select, no multiply.
    if (RT_BVH_FOLD_SHRINK) {
        ta = (tna <= tfa) ? tna : kFltMax;
        tb = (tnb <= tfb) ? tnb : kFltMax;
    } else {
        ta = (tna * 0.999998f <= tfa) ? tna : kFltMax;
        tb = (tnb * 0.999998f <= tfb) ? tnb : kFltMax;
    }
}
constexpr float kFarWiden = RT_BVH_FOLD_SHRINK ? 1.0000025f : 1.0f;

// a x b and a . b with explicit FMAs, in a fixed order (the CPU checker mirrors them with fmaf)
RT_DEV f3 cross_fma(f3 a, f3 b)
{
    return F3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
RT_DEV float dot_fma(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }

// One primitive record against the ray: the same operations as the linear scan (Sphere.fs:18-43) / the
// build's triangle Hitable (Moeller-Trumbore with explicit FMAs, include/rt_abi.h RT_GEOM_TRIANGLE).
RT_DEV void prim_test(float4 a, float4 bb, float4 cc, f3 O, f3 D, float& t, bool& ok)
{
    if (__float_as_uint(cc.z) == 0u) {
        const f3 ctr = O - F3(a.x, a.y, a.z);
        const float b = dot(ctr, D);
        const float C = dot(ctr, ctr) - a.w;
        const float d = b * b - C;
        const bool has = d * 1e3f > 0.125f;          // == !(d < 0) & !(|1000 d| <= 0.125) for every d that can still give a hit
        const float sq = sqrt_rn(d);
        const float i1 = -b + sq, i2 = -b - sq;
        // sq >= 0, so i1 >= i2: 'both >= TMin' is 'i2 >= TMin' and the smaller root is i2; otherwise the reference takes i1
        // unless i1 < 0, and then either choice is below TMin and rejected (Sphere.fs:28-36 in 2 instead of 6 selects)
        t = (i2 >= kTMin) ? i2 : i1;
        ok = has & (t > kTMin);
    } else {
        const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, bb.x, bb.y), e2 = F3(bb.z, bb.w, cc.x);
        // the triangle Hitable is this build's own (the reference has none): it is DEFINED with fused
        // multiply-adds, written out one by one (fmaf on a CPU gives the same roundings)
        const f3 pv = cross_fma(D, e2);
        const float det = dot_fma(e1, pv);
        const float inv = div_rn(1.0f, det);
        const f3 tv = O - v0;
        const float u = dot_fma(tv, pv) * inv;
        const f3 q = cross_fma(tv, e1);
        const float v = dot_fma(D, q) * inv;
        t = dot_fma(e2, q) * inv;
        const bool has = !(fabsf(det) < 1e-12f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f);
        ok = has & (t > kTMin);
    }
}

RT_DEV void sort2(float& ta, uint32_t& ra, float& tb, uint32_t& rb)
{
    const bool sw = tb < ta;
    const float t0 = sw ? tb : ta, t1 = sw ? ta : tb;
    const uint32_t r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ta = t0; tb = t1; ra = r0; rb = r1;
}

constexpr uint32_t kBvhDone = 0xffffffffu;

// Depth-first traversal, nearest child first, the other hit children on a per-lane stack.  (tmin, idx) come
// in from the scan of the loose surfaces (planes) and go out updated.
//
// Stack: the top K entries live at stack[(i % K) * stride] (LDS, one column per lane); with SPILL, older
// entries are moved to the lane's column of a backing store in HBM (`backing()` returns its base, stride 64)
// — LDS per wave decides how many waves a CU holds, and a stack deeper than K is rare.  Entries [lo, sp) are
// in the ring, [0, lo) in the backing store.
//
// Shape of the loop: every round, each of the two groups of lanes — those at an inner node, those at a leaf —
// takes one step if at least IN_MIN / LEAF_MIN lanes have gathered in it, the larger group in any case: the two
// steps are different code, too much to run for a handful of lanes.  The loads of both groups are issued
// before either is used, so a round costs one memory latency.
//
// `cur`/`sp`/`lo` are the lane's traversal state: cur = kBvhDone: no ray, 0: at the root.  The call returns
// when no lane is walking or fewer than `min_active` are; those continue in the next call.
// COUNT: keep the statistics (nodes visited, primitives tested, rounds); a scene renders without them unless
// rt_scene_counters asked for them (the counters cost the mesh frames 2-3 %).
template <int K, bool SPILL, int IN_MIN, int LEAF_MIN, bool COUNT, typename Backing>
RT_DEV void closest_hit_bvh(const BvhRef& bvh, f3 O, f3 D, uint32_t* stack, int stride, uint32_t* dump, Backing backing,
                            float& tmin, int& idx, uint32_t& cur, int& sp, int& lo, int min_active,
                            uint32_t& nodes_visited, uint32_t& prims_tested, uint32_t& rounds_inner, uint32_t& rounds_leaf)
{
    static_assert((K & (K - 1)) == 0, "ring size must be a power of two");

    // The ray as a cone.  The reference's sphere test (Sphere.fs:18-43, prim_test) computes d = b^2 - C from two numbers
    // of size L^2, L = |O - centre|: 15 roundings of relative 2^-24 on terms <= L^2, and |D|^2 - 1 (<= 2.4e-7 for a
    // direction normalised in f32), put up to k L^2, k = 1.2e-6, between d and its exact value.  So the linear scan accepts
    // a sphere that the ray passes at up to sqrt(r^2 + k L^2) from its centre, and reports the hit up to that far in
    // front of it: 500 units away a sphere of radius 0.2 is "hit" out to 0.58.  (Measured with exact boxes: 14 of 40 000
    // far, grazing rays over the 1M-sphere scene and 1 of 40 000 random ones got a different answer from the tree than
    // from the scan.)  A tree that holds spheres is therefore walked with boxes that grow along the ray: by
    // s (t + t0) on every side at parameter t, where
    //   s  = (sqrt(rmin^2 + k Lc^2) - rmin) / Lc   the largest growth any sphere needs, per unit of distance, up to the
    //        distance Lc beyond which this ray needs no sphere (its current hit, or the far end of the tree): the growth
    //        sqrt(r^2 + k L^2) - r is convex in L and largest for the smallest radius;
    //   t0 = 2.5 sqrt(rmax^2 + k Lc^2)             (the hit may lie that far before the closest approach).
    // A near plane is then reached at (plane -+ s t0 - O) / (|D| + s) and a far plane left at (plane +- s t0 - O) /
    // (|D| - s): the same two multiply-adds per plane pair as the plain slab test, with two sets of constants.  On an
    // axis where |D| <= s the cone's edge never passes the far plane in the ray's direction: it moves the other way, at
    // most (s - |D|) tcap by the end of the ray, so that edge is held at its final position: boxes beyond it are
    // missed, the others never left on this axis.  With k = 0 (a tree of triangles: their test is off by ~1e-7 L,
    // inside the host's padding) all of this IS the plain slab test.
    // reciprocals for the slab test only (not on the exact path): v_rcp_f32 is enough
    // A ray with a NaN or an infinity in it hits nothing in the scan (every comparison of its tests fails).  Here it would
    // pass every slab test (max / min ignore NaNs), the empty child slots included — whose reference is 0, the root — and
    // never come back: it does not enter the tree (nor does a ray whose coordinates add up beyond FLT_MAX).
    f3 inv_n, inv_f, noi_lo, noi_hi;
    const float far = bvh.reach + (fabsf(O.x) + fabsf(O.y) + fabsf(O.z));       // no primitive is further from O than this
    if (!(far + (fabsf(D.x) + fabsf(D.y) + fabsf(D.z)) < kFltMax)) { cur = kBvhDone; sp = 0; lo = 0; }      // (false for a NaN, too)
    if (bvh.cone_k == 0.0f && __ballot(far > bvh.lin_far) == 0ull) {
        // the plain slab test: one reciprocal per axis (a zero component gives infinities and NaNs, which max / min ignore)
        inv_n = F3(__builtin_amdgcn_rcpf(D.x), __builtin_amdgcn_rcpf(D.y), __builtin_amdgcn_rcpf(D.z));
        inv_f = inv_n;
        const f3 noi = F3(-(O.x * inv_n.x), -(O.y * inv_n.y), -(O.z * inv_n.z));
        noi_lo = F3(noi.x - 1e-6f * fabsf(noi.x), noi.y - 1e-6f * fabsf(noi.y), noi.z - 1e-6f * fabsf(noi.z));
        noi_hi = F3(noi.x + 1e-6f * fabsf(noi.x), noi.y + 1e-6f * fabsf(noi.y), noi.z + 1e-6f * fabsf(noi.z));
    } else {
        const float tcap = fminf(tmin, far * 1.001f + 4.0f * bvh.rmax);
        const float lc = tcap * 1.01f + 4.0f * bvh.rmax + 1e-20f;
        const float e = bvh.cone_k * lc * lc;
        float cs = (far > bvh.lin_far) ? bvh.lin_s : 0.0f;
        if (bvh.cone_k != 0.0f) cs = fmaxf(cs, 1.02f * e * __builtin_amdgcn_rcpf((__builtin_sqrtf(bvh.rmin * bvh.rmin + e) + bvh.rmin) * lc));
        const float ct = cs * 2.5f * __builtin_sqrtf(bvh.rmax * bvh.rmax + e);
        auto axis = [&](float o, float dk, float& in, float& jf, float& nlo, float& nhi) {
            const float ak = fabsf(dk);
            const bool leaves = ak > cs;                    // the cone's trailing edge moves in the ray's direction on this axis
            const float sh = __builtin_copysignf(ct, dk);
            in = __builtin_copysignf(__builtin_amdgcn_rcpf(ak + cs), dk);
            const float nn = -((o + sh) * in);
            nlo = nn - 1e-6f * fabsf(nn);
            if (leaves) {
                jf = __builtin_copysignf(__builtin_amdgcn_rcpf(ak - cs), dk);
                const float nf = -((o - sh) * jf);
                nhi = nf + 1e-6f * fabsf(nf);
            } else {
                // far value = (plane - edge) * 1e30 in the ray's direction: +-huge, i.e. never left / missed
                const float out = ct + tcap * (cs - ak);
                const float edge = o - __builtin_copysignf(out + 1e-5f * (fabsf(o) + out), dk);
                jf = __builtin_copysignf(1e30f, dk);
                nhi = -(edge * jf);
            }
        };
        axis(O.x, D.x, inv_n.x, inv_f.x, noi_lo.x, noi_hi.x);
        axis(O.y, D.y, inv_n.y, inv_f.y, noi_lo.y, noi_hi.y);
        axis(O.z, D.z, inv_n.z, inv_f.z, noi_lo.z, noi_hi.z);
    }
    if (RT_BVH_FOLD_SHRINK) { inv_f = inv_f * kFarWiden; noi_hi = noi_hi * kFarWiden; }
    const v2f mx = { inv_n.x, inv_f.x }, my = { inv_n.y, inv_f.y }, mz = { inv_n.z, inv_f.z };
    const v2f cx = { noi_lo.x, noi_hi.x }, cy = { noi_lo.y, noi_hi.y }, cz = { noi_lo.z, noi_hi.z };
    const f3 inv = inv_n;
    // byte offsets, inside a node, of the planes met first / last on each axis (lo.x lo.y lo.z hi.x hi.y hi.z), one byte
    // each, in the order of the seven loads: near x, far x, near y | far y, near z, far z, references.  (Two registers
    // instead of six, and issue() selects between the node's and the leaf's offsets with two selects instead of seven.)
    static_assert(node_row(6) < 256u, "offsets inside a node are packed into bytes");
    const uint32_t onx = (inv.x < 0.0f) ? node_row(3) : node_row(0), ony = (inv.y < 0.0f) ? node_row(4) : node_row(1), onz = (inv.z < 0.0f) ? node_row(5) : node_row(2);
    const uint32_t ofx = (node_row(0) + node_row(3)) - onx, ofy = (node_row(1) + node_row(4)) - ony, ofz = (node_row(2) + node_row(5)) - onz;
    const uint32_t in_off0 = onx | (ofx << 8) | (ony << 16), in_off1 = ofy | (onz << 8) | (ofz << 16) | (node_row(6) << 24);
    constexpr uint32_t kLeafOff0 = 0u | (16u << 8) | (32u << 16), kLeafOff1 = 0u | (16u << 8) | (32u << 16) | (32u << 24);
    const char* const node_bytes = reinterpret_cast<const char*>(bvh.nodes);
    // RING2 (the 8-entry LDS ring of the render kernels): `sp` and `lo` count in BYTES of ring rows (256 per entry) and
    // the ring is addressed as raw LDS bytes, l4 | (sp & 0x700) — one v_and_or_b32 per address where the generic form cost
    // a shift, a mask and a three-operand add (the caller puts the ring where bits 8..10 of every lane's column base are
    // zero: trace_kernel checks it).  The pop is a ds_read (the compiler had merged the LDS read and the rare read of the
    // backing store into ONE flat load through a selected pointer: flat latency on the load -> decide -> load chain).
    constexpr bool RING2 = RT_BVH_RING2 && SPILL && K <= 8;
    const uint32_t l4 = RING2 ? lds_addr_of(stack) : 0u;
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return kBvhDone;
        if (RING2) {
            sp -= 256;
            uint32_t v = lds_load(l4 | ((uint32_t)sp & 0x700u));
            if (sp < lo) { lo = sp; v = backing()[sp >> 2]; }
            return v;
        }
        sp--;
        uint32_t v = stack[(sp & (K - 1)) * stride];
        if (SPILL && sp < lo) { lo = sp; v = backing()[sp * 64]; }
        return v;
    };
    // The loop is software-pipelined: the loads of round r+1 are issued as soon as round r has produced every lane's next
    // node (`cur`), BEFORE round r's stack pushes and statistics, which then run in the shadow of the memory latency.
    // (The traversal is latency-bound at 4 waves per SIMD: frame rate is proportional to the waves per CU, so every
    // instruction taken off the load -> decide -> load chain counts.)
    bool do_in = false, do_leaf = false;
    unsigned long long m_in = 0ull, m_lf = 0ull;
    float4 d[7];
    // decide(): which groups step in the coming round (false: leave the traversal); issue(): their loads
    int st_in = 0, st_lf = 0;       // lanes stepping at an inner node / at a leaf in the round just decided
    auto decide = [&]() -> bool {
        // (a leaf reference is 0x80000000 | first << 3 | count with count >= 1 and first < 2^28: as an int it is below -1 = kBvhDone)
        const bool in = (int)cur >= 0, lf = (int)cur < -1;
        const int n_in = popc_mask(__ballot(in)), n_lf = popc_mask(__ballot(lf));
        if (n_in + n_lf == 0 || n_in + n_lf < min_active) return false;
        // a group steps when it is at least IN_MIN / LEAF_MIN lanes strong, the larger one in any case (wave-uniform decisions)
        const bool step_in = (n_in >= IN_MIN) | (n_in >= n_lf);
        const bool step_lf = (n_lf >= LEAF_MIN) | (n_lf > n_in);
        do_in = in & step_in;
        do_leaf = lf & step_lf;
        m_in = step_in ? __ballot(in) : 0ull;        // the same as wave-uniform masks, for issue() (not carried around the loop)
        m_lf = step_lf ? __ballot(lf) : 0ull;
        st_in = step_in ? n_in : 0; st_lf = step_lf ? n_lf : 0;      // for account(), once the loads are under way
        return true;
    };
    // wave-uniform statistics of the round just decided (scalar arithmetic), off the load -> decide -> load chain
    auto account = [&]() {
        if (!COUNT) return;
        rounds_inner += (st_in != 0) ? 1u : 0u;
        rounds_leaf += (st_lf != 0) ? 1u : 0u;
        nodes_visited += (uint32_t)st_in;
        if (st_lf != 0) prims_tested += (uint32_t)st_lf + (uint32_t)popc_mask(__ballot(do_leaf & ((cur & 6u) != 0u)));   // leaves hold 1 or 2
    };
    // Both groups load through the SAME seven 16-byte requests, each lane from its own addresses: a lane at an inner
    // node the six plane vectors and the references of its node, a lane at a leaf its (up to) two records of three
    // float4 — nodes and records share one allocation, so an address is a 32-bit offset from the uniform base.
    // One set of 28 registers instead of two sets (28 + 24) in flight, seven VMEM instructions per round instead of
    // thirteen when both groups step.
    auto issue = [&]() {
        if (RT_BVH_ISSUE_ASM ? __builtin_amdgcn_inverse_ballot_w64(m_in | m_lf) : (do_in | do_leaf)) {
            uint32_t o0, o1;        // inner: the node; leaf: first and second record
            if (RT_BVH_REF_OFFSETS) {
                o0 = (cur & 0x7ffffff8u) << 1;
                o1 = o0 + (cur & 2u) * 24u;                     // the second record of a two-record leaf (an inner reference has no low bits)
            }
            else if (do_in) { o0 = cur << kBvhNodeShift; o1 = o0; }             // 128 bytes per node; the allocation is < 4 GB (host check)
            else {
                const uint32_t first = (cur & 0x7fffffffu) >> 3;
                o0 = bvh.pdata_off + first * 48u;              // 48-byte records
                o1 = o0 + (((cur & 6u) != 0u) ? 48u : 0u);
            }
#if RT_BVH_PACK_OFF
            // (the selects read the step decision as the scalar mask decide() has just made, not the 0/1 flag carried
            // around the loop, which costs a v_and and a v_cmp per test)
            const bool sel_in = RT_BVH_ISSUE_ASM ? __builtin_amdgcn_inverse_ballot_w64(m_in) : do_in;
            const uint32_t f0 = sel_in ? in_off0 : kLeafOff0, f1 = sel_in ? in_off1 : kLeafOff1;
#if RT_BVH_ISSUE_ASM
            // one add with a byte select per address (the compiler takes three instructions for four of the seven)
            uint32_t a0, a1, a2, a3, a4, a5, a6;
#define RT_ADD_BYTE(dst, base, packed, sel) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" sel : "=v"(dst) : "v"(base), "v"(packed))
            RT_ADD_BYTE(a0, o0, f0, "BYTE_0"); RT_ADD_BYTE(a1, o0, f0, "BYTE_1"); RT_ADD_BYTE(a2, o0, f0, "BYTE_2");
            RT_ADD_BYTE(a3, o1, f1, "BYTE_0"); RT_ADD_BYTE(a4, o1, f1, "BYTE_1"); RT_ADD_BYTE(a5, o1, f1, "BYTE_2"); RT_ADD_BYTE(a6, o1, f1, "BYTE_3");
#undef RT_ADD_BYTE
#else
            const uint32_t a0 = o0 + (f0 & 0xffu), a1 = o0 + ((f0 >> 8) & 0xffu), a2 = o0 + ((f0 >> 16) & 0xffu);
            const uint32_t a3 = o1 + (f1 & 0xffu), a4 = o1 + ((f1 >> 8) & 0xffu), a5 = o1 + ((f1 >> 16) & 0xffu), a6 = o1 + (f1 >> 24);
#endif
#else
            const uint32_t a0 = o0 + (do_in ? onx : 0u), a1 = o0 + (do_in ? ofx : 16u), a2 = o0 + (do_in ? ony : 32u);
            const uint32_t a3 = o1 + (do_in ? ofy : 0u), a4 = o1 + (do_in ? onz : 16u), a5 = o1 + (do_in ? ofz : 32u), a6 = o1 + (do_in ? node_row(6) : 32u);
#endif
            d[0] = *reinterpret_cast<const float4*>(node_bytes + a0); d[1] = *reinterpret_cast<const float4*>(node_bytes + a1);
            d[2] = *reinterpret_cast<const float4*>(node_bytes + a2); d[3] = *reinterpret_cast<const float4*>(node_bytes + a3);
            d[4] = *reinterpret_cast<const float4*>(node_bytes + a4); d[5] = *reinterpret_cast<const float4*>(node_bytes + a5);
            // (a leaf lane's seventh request fetches the last 16 bytes of its second record again.  Leaving it out — one
            // exec-mask toggle — was measured: c4 -0.2 %, c3 -0.4 %, although the L1 serves only four 16-byte lane-requests
            // per clock whatever their addresses, tools/gather_rate.hip: the request rate is not what bounds a round.)
            d[6] = *reinterpret_cast<const float4*>(node_bytes + a6);
        }
    };
    if (!decide()) return;
    issue();
    account();
    while (true) {
        // what this round's inner lanes still have to push once the next loads are under way
        bool pushing = do_in;
        uint32_t p0 = 0u, p1 = 0u, p2 = 0u; int nh = 0;
        float lt1 = kFltMax, lt2 = kFltMax, lt3 = kFltMax;
        if (do_in) {
            // ---- inner node: four slab tests
            const float4 nx4 = d[0], fx4 = d[1], ny4 = d[2], fy4 = d[3], nz4 = d[4], fz4 = d[5], rf = d[6];
            uint32_t r0 = __float_as_uint(rf.x), r1 = __float_as_uint(rf.y), r2 = __float_as_uint(rf.z), r3 = __float_as_uint(rf.w);
            float t0, t1, t2, t3;
            const float tmaxw = tmin * kFarWiden;
            box_entry2(v2f{ nx4.x, nx4.y }, v2f{ ny4.x, ny4.y }, v2f{ nz4.x, nz4.y }, v2f{ fx4.x, fx4.y }, v2f{ fy4.x, fy4.y }, v2f{ fz4.x, fz4.y },
                       mx, my, mz, cx, cy, cz, tmaxw, t0, t1);
            box_entry2(v2f{ nx4.z, nx4.w }, v2f{ ny4.z, ny4.w }, v2f{ nz4.z, nz4.w }, v2f{ fx4.z, fx4.w }, v2f{ fy4.z, fy4.w }, v2f{ fz4.z, fz4.w },
                       mx, my, mz, cx, cy, cz, tmaxw, t2, t3);
            // sort the four by entry distance (5 compare-exchanges); misses (kFltMax) go last
            sort2(t0, r0, t1, r1); sort2(t2, r2, t3, r3); sort2(t0, r0, t2, r2);
            if (RING2 && RT_BVH_LATE_SORT) {
                // the nearest child is known after three of the five exchanges, and it is all the next round's loads wait
                // for: the last two, the hit count and the pushes run after those loads are issued (below)
                lt1 = t1; lt2 = t2; lt3 = t3; p0 = r3; p1 = r2; p2 = r1;
            } else {
                sort2(t1, r1, t3, r3); sort2(t1, r1, t2, r2);
                // the hits are a prefix of the sorted four: descend into the nearest, push the others far to near (below).
                // A lane pushes (it has a hit and descends) or pops (it has none), never both in one round.
                const bool h2 = t2 < kFltMax, h3 = t3 < kFltMax;
                nh = (t1 < kFltMax ? 1 : 0) + (h2 ? 1 : 0) + (h3 ? 1 : 0);
                if (RING2) { p0 = r3; p1 = r2; p2 = r1; pushing = nh != 0; nh <<= 8; }   // the sorted three as they are (below)
                else { p0 = h3 ? r3 : (h2 ? r2 : r1); p1 = h3 ? r2 : r1; p2 = r1; }
            }
            cur = (t0 < kFltMax) ? r0 : pop();
        }
        if (do_leaf) {
            // ---- leaf: test in array order
            const int cnt = (int)(cur & 7u);
#pragma unroll
            for (int k = 0; k < kBvhLeafMax; k++) {
                float t; bool ok;
                prim_test(d[3 * k], d[3 * k + 1], d[3 * k + 2], O, D, t, ok);
                const int si = (int)__float_as_uint(d[3 * k + 2].y);
                const bool bt = better_hit(ok & (k < cnt), t, si, tmin, idx);
                tmin = bt ? t : tmin;
                idx = bt ? si : idx;
            }
            cur = pop();
        }
        // ---- next round: decision and loads
        const bool more = decide();
        if (more) { issue(); account(); }
        // ---- this round's pushes.  Branch-free: three entries are always stored, those beyond the new top are garbage in
        // free slots.
        if (RING2) {
            // Far to near from the old top: the farthest hit lands at sp, the nearest at the new top.  All three sorted
            // references are stored, unconditionally and in this order — r3 at the old top, r2 one above it when there are
            // three hits (else at the old top too), r1 at the new top: one wave's LDS operations complete in order, so with
            // fewer than three hits the later, valid stores overwrite the earlier ones and nothing is written outside
            // [sp, sp + nh).  (Before: three selects for the values, three for dump-or-ring addresses, nine address
            // operations: 22 vector instructions per inner round, now 9.)
            if (RT_BVH_LATE_SORT && pushing) {
                sort2(lt1, p2, lt3, p0); sort2(lt1, p2, lt2, p1);      // (p2, p1, p0) = the second, third and fourth nearest
                nh = ((lt1 < kFltMax ? 1 : 0) + (lt2 < kFltMax ? 1 : 0) + (lt3 < kFltMax ? 1 : 0)) << 8;
                pushing = nh != 0;
            }
            if (pushing) {
                while (sp + nh - lo > K * 256) { backing()[lo >> 2] = lds_load(l4 | ((uint32_t)lo & 0x700u)); lo += 256; }
                const int nsp = sp + nh;
                lds_store(l4 | ((uint32_t)sp & 0x700u), p0);
                lds_store(l4 | ((uint32_t)max(nsp - 512, sp) & 0x700u), p1);
                lds_store(l4 | ((uint32_t)(nsp - 256) & 0x700u), p2);
                sp = nsp;
            }
        } else if (pushing) {
            if (SPILL && K <= 8) {
                // (a 16-entry ring has room to spare and its kernel no LDS to spare: it keeps the simpler rule below)
                // room for exactly the nh entries pushed; the stores beyond the new top go to the lane's `dump` word (any
                // scratch word of LDS: the render kernel lends its lane map, which is rewritten before every use), so
                // that they cannot land on a live entry of a full ring.  (Making room for three whatever nh is cost the
                // 8-entry ring three of its entries: 0.22 TB of spill writes per 4K frame.)
                while (sp + nh - lo > K) { backing()[lo * 64] = stack[(lo & (K - 1)) * stride]; lo++; }
                *((0 < nh) ? stack + (sp & (K - 1)) * stride : dump) = p0;
                *((1 < nh) ? stack + ((sp + 1) & (K - 1)) * stride : dump) = p1;
                *((2 < nh) ? stack + ((sp + 2) & (K - 1)) * stride : dump) = p2;
            } else {
                if (SPILL) while (sp + 3 - lo > K) { backing()[lo * 64] = stack[(lo & (K - 1)) * stride]; lo++; }
                stack[(sp & (K - 1)) * stride] = p0;
                stack[((sp + 1) & (K - 1)) * stride] = p1;
                stack[((sp + 2) & (K - 1)) * stride] = p2;
            }
            sp += nh;
        }
        if (!more) break;
    }
}

}  // namespace rt
