// rt_bvh.h — closest hit through a flattened BVH in HBM (large scenes: triangle meshes, many spheres).
//
// NOT in the reference: Geometry/AABB.fs is an empty module and findClosestIntersection is a linear scan
// (Surface.fs:49-58).  The BVH only prunes: every primitive that the linear scan would accept is still
// tested with exactly the same arithmetic, and ties in t are resolved to the lowest array index, which is
// what "first in array order wins" (strict '<', Surface.fs:53) means.  So the result equals the linear
// scan's bit for bit; tests compare against the brute-force oracle.
//
// Layout (built on the host, rt_bvh_build.h):
//   nodes[]   32 B: lo.xyz, a | hi.xyz, b     inner: a = index of the left child (right = a+1), b = -1
//                                             leaf : a = first primitive slot, b = primitive count (> 0)
//             the two children of a node are adjacent, so one visit fetches both boxes as 64 contiguous bytes
//   prims[]   uint2 per primitive slot (BVH order): x = type << 31 | float4 offset into pdata, y = surface index
//   pdata[]   float4: sphere = (centre, r^2); triangle = (v0, e1.x) (e1.yz, e2.xy) (e2.z, 0, 0, 0)
// Boxes are padded on the host so that the (inexact) slab test never rejects a primitive the exact test hits.
#pragma once
#include "rt_device.h"

namespace rt {

struct BvhNode { float lo[3]; int a; float hi[3]; int b; };
static_assert(sizeof(BvhNode) == 32, "BvhNode layout");

struct BvhRef {
    const float4* __restrict__ nodes;     // 2 float4 per node
    const uint2* __restrict__ prims;
    const float4* __restrict__ pdata;
    int nnodes;
};

constexpr int kBvhStack = 32;             // host guarantees tree depth <= kBvhStack - 2

// (t, idx) is a better hit than (tmin, best): closer, or equally close with a lower array index
RT_DEV bool better_hit(bool ok, float t, int idx, float tmin, int best)
{
    return ok & ((t < tmin) | ((t == tmin) & (idx < best)));
}

// slab test with NaN-ignoring min/max; returns the entry parameter, or a value > tmax when missed
RT_DEV float box_entry(float4 lo, float4 hi, f3 O, f3 inv, float tmax)
{
    const float x0 = (lo.x - O.x) * inv.x, x1 = (hi.x - O.x) * inv.x;
    const float y0 = (lo.y - O.y) * inv.y, y1 = (hi.y - O.y) * inv.y;
    const float z0 = (lo.z - O.z) * inv.z, z1 = (hi.z - O.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
    return (tn <= tf) ? tn : kFltMax;
}

// One primitive against the ray: the same operations as the linear scan (Sphere.fs:18-43) / the
// build's triangle Hitable (Moeller-Trumbore, see oracle/rt_oracle.c triangle_intersect).
RT_DEV void prim_test(const float4* __restrict__ pdata, uint32_t ref, f3 O, f3 D, float& t, bool& ok)
{
    const float4* p = pdata + (ref & 0x7fffffffu);
    const float4 a = p[0];
    if ((ref >> 31) == 0u) {
        const f3 ctr = O - F3(a.x, a.y, a.z);
        const float b = dot(ctr, D);
        const float C = dot(ctr, ctr) - a.w;
        const float d = b * b - C;
        const bool has = !(d < 0.0f) & !(fabsf(d * 1e3f) <= 0.125f);
        const float sq = sqrt_rn(d);
        const float i1 = -b + sq, i2 = -b - sq;
        const float tboth = (i1 < i2) ? i1 : i2;
        const float tone = (i1 < 0.0f) ? i2 : i1;
        t = ((i1 >= kTMin) & (i2 >= kTMin)) ? tboth : tone;
        ok = has & (t > kTMin);
    } else {
        const float4 bb = p[1], cc = p[2];
        const f3 v0 = F3(a.x, a.y, a.z), e1 = F3(a.w, bb.x, bb.y), e2 = F3(bb.z, bb.w, cc.x);
        const f3 pv = cross(D, e2);
        const float det = dot(e1, pv);
        const float inv = div_rn(1.0f, det);
        const f3 tv = O - v0;
        const float u = dot(tv, pv) * inv;
        const f3 q = cross(tv, e1);
        const float v = dot(D, q) * inv;
        t = dot(e2, q) * inv;
        const bool has = !(fabsf(det) < 1e-12f) & (u >= 0.0f) & (v >= 0.0f) & ((u + v) <= 1.0f);
        ok = has & (t > kTMin);
    }
}

// Depth-first traversal, near child first, far child on a per-lane stack (in `stack`, strided by
// `stride` dwords so that lanes hit different LDS banks).  (tmin, idx) come in from the scan of the
// loose surfaces (planes) and go out updated.
RT_DEV void closest_hit_bvh(const BvhRef& bvh, f3 O, f3 D, uint32_t* stack, int stride, float& tmin, int& idx,
                            uint32_t& nodes_visited, uint32_t& prims_tested)
{
    if (bvh.nnodes == 0) return;
    // reciprocal direction for the slab test only (not on the exact path): v_rcp_f32 is enough
    const f3 inv = F3(__builtin_amdgcn_rcpf(D.x), __builtin_amdgcn_rcpf(D.y), __builtin_amdgcn_rcpf(D.z));
    int sp = 0;
    int node = 0;
    // the root is visited unconditionally (its children are tested below)
    while (true) {
        const float4 n0 = bvh.nodes[2 * node], n1 = bvh.nodes[2 * node + 1];
        const int a = __float_as_int(n0.w), b = __float_as_int(n1.w);
        nodes_visited++;
        if (b > 0) {
            for (int k = 0; k < b; k++) {
                const uint2 pr = bvh.prims[a + k];
                float t; bool ok;
                prim_test(bvh.pdata, pr.x, O, D, t, ok);
                prims_tested++;
                const bool bt = better_hit(ok, t, (int)pr.y, tmin, idx);
                tmin = bt ? t : tmin;
                idx = bt ? (int)pr.y : idx;
            }
            node = -1;
        } else {
            const float4 l0 = bvh.nodes[2 * a], l1 = bvh.nodes[2 * a + 1];
            const float4 r0 = bvh.nodes[2 * a + 2], r1 = bvh.nodes[2 * a + 3];
            // '<= tmin' inside box_entry keeps boxes that can only tie: a tie may still win on index
            const float tl = box_entry(l0, l1, O, inv, tmin);
            const float tr = box_entry(r0, r1, O, inv, tmin);
            const bool hl = tl < kFltMax, hr = tr < kFltMax;
            if (hl & hr) {
                const bool left_first = tl <= tr;
                stack[sp * stride] = (uint32_t)(left_first ? a + 1 : a);
                sp++;
                node = left_first ? a : a + 1;
            } else node = hl ? a : (hr ? a + 1 : -1);
        }
        if (node < 0) {
            if (sp == 0) break;
            sp--;
            node = (int)stack[sp * stride];
        }
    }
}

}  // namespace rt
