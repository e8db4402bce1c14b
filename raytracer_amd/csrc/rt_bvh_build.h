// rt_bvh_build.h — host-side BVH construction for large scenes (spheres and triangles; planes stay in the
// linear list).  Not in the reference (Geometry/AABB.fs is empty).  Object-median splits along the longest
// axis of the centroid bounds: perfectly balanced, so the per-lane traversal stack is bounded by
// ceil(log2(n / leaf)) + 1, checked against kBvhStack.
#pragma once
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace rt {

struct BuildPrim {
    float lo[3], hi[3], c[3];
    uint32_t ref;       // type << 31 | float4 offset into pdata
    uint32_t surf;      // index in the caller's surface array
};

struct HostBvh {
    std::vector<BvhNode> nodes;
    std::vector<uint2> prims;
    std::vector<float4> pdata;
    int depth = 0;
};

constexpr int kBvhLeaf = 4;

inline void bvh_bounds(const std::vector<BuildPrim>& p, size_t b, size_t e, float pad, BvhNode& n)
{
    for (int k = 0; k < 3; k++) { n.lo[k] = INFINITY; n.hi[k] = -INFINITY; }
    for (size_t i = b; i < e; i++)
        for (int k = 0; k < 3; k++) { n.lo[k] = std::min(n.lo[k], p[i].lo[k]); n.hi[k] = std::max(n.hi[k], p[i].hi[k]); }
    // pad: the slab test is not exact (approximate reciprocal, one rounding per step) while the primitive
    // tests are; a box must never reject a ray whose exact test hits, so boxes are grown by far more than
    // the slab test's error at this scene's scale
    for (int k = 0; k < 3; k++) {
        n.lo[k] = std::nextafter(n.lo[k] - pad, -INFINITY);
        n.hi[k] = std::nextafter(n.hi[k] + pad, INFINITY);
    }
}

inline void bvh_build_range(std::vector<BuildPrim>& p, size_t b, size_t e, int node, int depth, float pad, HostBvh& out)
{
    out.depth = std::max(out.depth, depth);
    bvh_bounds(p, b, e, pad, out.nodes[node]);
    const size_t n = e - b;
    if (n <= (size_t)kBvhLeaf) {
        out.nodes[node].a = (int)b;
        out.nodes[node].b = (int)n;
        return;
    }
    float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = b; i < e; i++)
        for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], p[i].c[k]); chi[k] = std::max(chi[k], p[i].c[k]); }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    const size_t mid = b + n / 2;
    std::nth_element(p.begin() + b, p.begin() + mid, p.begin() + e,
                     [axis](const BuildPrim& x, const BuildPrim& y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.surf < y.surf); });
    const int left = (int)out.nodes.size();
    out.nodes.emplace_back();
    out.nodes.emplace_back();
    out.nodes[node].a = left;
    out.nodes[node].b = -1;
    bvh_build_range(p, b, mid, left, depth + 1, pad, out);
    bvh_build_range(p, mid, e, left + 1, depth + 1, pad, out);
}

// prims: one entry per sphere / triangle with its exact bounds; pdata already filled.
inline void bvh_build(std::vector<BuildPrim>& prims, HostBvh& out)
{
    out.nodes.clear(); out.prims.clear(); out.depth = 0;
    if (prims.empty()) return;
    float scale = 0.0f;
    for (const BuildPrim& p : prims)
        for (int k = 0; k < 3; k++) scale = std::max(scale, std::max(std::fabs(p.lo[k]), std::fabs(p.hi[k])));
    const float pad = 1e-5f * (scale + 1.0f);
    out.nodes.reserve(prims.size());
    out.nodes.emplace_back();
    bvh_build_range(prims, 0, prims.size(), 0, 1, pad, out);
    out.prims.resize(prims.size());
    for (size_t i = 0; i < prims.size(); i++) { out.prims[i].x = prims[i].ref; out.prims[i].y = prims[i].surf; }
}

}  // namespace rt
