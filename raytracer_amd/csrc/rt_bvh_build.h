// rt_bvh_build.h — host-side BVH construction for large scenes (spheres and triangles; planes stay in the
// linear list).  Not in the reference (Geometry/AABB.fs is empty).  A binary tree is built with binned
// surface-area-heuristic splits (object-median splits as the fallback), then collapsed into 4-wide nodes
// (rt_bvh.h): the collapse with the least summed node area (bvh_plan_collapse).
#pragma once
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace rt {

struct BuildPrim {
    float lo[3], hi[3], c[3];
    uint32_t ref;       // type << 31 | float4 offset into the caller-order pdata
    uint32_t surf;      // index in the caller's surface array
};

struct HostBvh {
    std::vector<float4> nodes;      // 8 per node
    std::vector<float4> pdata;      // 3 per primitive slot, BVH order
    int depth = 0;                  // depth of the 4-wide tree
    float pad = 0.0f;               // what every box was grown by (bvh_pad)
};

constexpr int kBvhLeaf = kBvhLeafMax;

struct BinNode { float lo[3], hi[3]; int left, right, first, count; };

struct Box3 {
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    void grow(const float* l, const float* h) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], l[k]); hi[k] = std::max(hi[k], h[k]); } }
    void grow(const Box3& o) { grow(o.lo, o.hi); }
    float area() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kSahBins = 16;
constexpr int kSahMaxLevel = 40;     // SAH splits below this binary depth are replaced by median splits

// Binary tree over p[b, e).  sah = true: the split with the least surface-area-heuristic cost
// area(L) * |L| + area(R) * |R| over 16 centroid bins on each axis (measured on the bench scenes against median
// splits: up to -23 % expected node visits and -55 % expected primitive tests); sah = false, or no usable
// split: object median along the longest axis of the centroid bounds (balanced, depth = ceil(log2(n / 4)) + 1).
inline int bvh_bin_build(std::vector<BuildPrim>& p, size_t b, size_t e, float pad, bool sah, std::vector<BinNode>& out, int level = 0)
{
    if (level > kSahMaxLevel) sah = false;      // a degenerate distribution: finish this subtree balanced
    const int me = (int)out.size();
    out.emplace_back();
    BinNode n;
    Box3 bounds, cb;
    for (size_t i = b; i < e; i++) { bounds.grow(p[i].lo, p[i].hi); cb.grow(p[i].c, p[i].c); }
    // pad: the slab test is not exact (approximate reciprocal, one rounding per step) while the primitive
    // tests are; a box must never reject a ray whose exact test hits, so boxes are grown by far more than
    // the slab test's error at this scene's scale
    for (int k = 0; k < 3; k++) {
        n.lo[k] = std::nextafter(bounds.lo[k] - pad, -INFINITY);
        n.hi[k] = std::nextafter(bounds.hi[k] + pad, INFINITY);
    }
    n.left = n.right = -1; n.first = (int)b; n.count = (int)(e - b);
    if (e - b > (size_t)kBvhLeaf) {
        size_t mid = 0;
        if (sah) {
            float best = INFINITY; int best_axis = -1, best_k = 0;
            for (int axis = 0; axis < 3; axis++) {
                const float ext = cb.hi[axis] - cb.lo[axis];
                if (!(ext > 0.0f)) continue;
                const float scale = (float)kSahBins / ext;
                Box3 bin_box[kSahBins]; size_t bin_n[kSahBins] = { 0 };
                for (size_t i = b; i < e; i++) {
                    const int k = std::min(kSahBins - 1, (int)((p[i].c[axis] - cb.lo[axis]) * scale));
                    bin_box[k].grow(p[i].lo, p[i].hi); bin_n[k]++;
                }
                float right_area[kSahBins]; size_t right_n[kSahBins];
                Box3 acc; size_t cnt = 0;
                for (int k = kSahBins - 1; k > 0; k--) { acc.grow(bin_box[k]); cnt += bin_n[k]; right_area[k] = acc.area(); right_n[k] = cnt; }
                acc = Box3(); cnt = 0;
                for (int k = 1; k < kSahBins; k++) {                 // split: bins < k go left
                    acc.grow(bin_box[k - 1]); cnt += bin_n[k - 1];
                    if (cnt == 0 || right_n[k] == 0) continue;
                    const float cost = acc.area() * (float)cnt + right_area[k] * (float)right_n[k];
                    if (cost < best) { best = cost; best_axis = axis; best_k = k; }
                }
            }
            if (best_axis >= 0) {
                const float lo = cb.lo[best_axis], scale = (float)kSahBins / (cb.hi[best_axis] - cb.lo[best_axis]);
                const int axis = best_axis, kk = best_k;
                auto it = std::partition(p.begin() + b, p.begin() + e, [=](const BuildPrim& x) {
                    return std::min(kSahBins - 1, (int)((x.c[axis] - lo) * scale)) < kk;
                });
                mid = (size_t)(it - p.begin());
            }
        }
        if (mid <= b || mid >= e) {
            int axis = 0;
            if (cb.hi[1] - cb.lo[1] > cb.hi[axis] - cb.lo[axis]) axis = 1;
            if (cb.hi[2] - cb.lo[2] > cb.hi[axis] - cb.lo[axis]) axis = 2;
            mid = b + (e - b) / 2;
            std::nth_element(p.begin() + b, p.begin() + mid, p.begin() + e, [axis](const BuildPrim& x, const BuildPrim& y) {
                return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.surf < y.surf);
            });
        }
        n.count = 0;
        n.left = bvh_bin_build(p, b, mid, pad, sah, out, level + 1);
        n.right = bvh_bin_build(p, mid, e, pad, sah, out, level + 1);
    }
    out[me] = n;
    return me;
}

inline float bvh_area(const BinNode& n)
{
    const float dx = n.hi[0] - n.lo[0], dy = n.hi[1] - n.lo[1], dz = n.hi[2] - n.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// Which binary nodes become 4-wide nodes.  Greedy (round 1-2): a node adopts grandchildren, largest area first, until it
// has four children — which fills the upper node and leaves its children half empty (16 primitives split 8|8, 4|4 ...:
// the root adopts four grandchildren of two leaves each, five nodes where three would do; measured on the 1M-sphere
// tree: 2.95 children per node).  Now: the collapse that minimises the summed area of the wide nodes (= the expected
// number of node visits of a random ray), by dynamic programming over the binary tree (Ylitie, Karras, Laine 2017,
// section 3.1): cost[n][i] = least cost of the subtree of n when it may occupy up to i child slots of the node above.
#ifndef RT_BVH_DP_COLLAPSE
#define RT_BVH_DP_COLLAPSE 1
#endif
struct CollapsePlan {
    std::vector<float> cost;            // [4 * n + i - 1]
    std::vector<unsigned char> take;    // [4 * n + i - 1]: slots really used for (n, i); 1 = n is a child (a leaf, or a node of its own)
    std::vector<unsigned char> left;    // [4 * n + j - 1]: of j slots passed through n, how many go to its left child
};
inline void bvh_plan_collapse(const std::vector<BinNode>& bin, CollapsePlan& pl)
{
    const size_t nb = bin.size();
    pl.cost.assign(4 * nb, 0.0f); pl.take.assign(4 * nb, 1); pl.left.assign(4 * nb, 0);
    // children are created after their parent (bvh_bin_build): a reverse sweep sees every child before its parent
    for (size_t n = nb; n-- > 0;) {
        const BinNode& b = bin[n];
        if (b.count > 0) continue;                                  // a leaf costs the same wherever it hangs: 0 here
        const float* cl = &pl.cost[4 * (size_t)b.left]; const float* cr = &pl.cost[4 * (size_t)b.right];
        float dist[5]; int dk[5];
        for (int j = 2; j <= 4; j++) {
            dist[j] = INFINITY; dk[j] = 1;
            for (int k = 1; k < j; k++) { const float c = cl[k - 1] + cr[j - k - 1]; if (c < dist[j]) { dist[j] = c; dk[j] = k; } }
            pl.left[4 * n + j - 1] = (unsigned char)dk[j];
        }
        pl.cost[4 * n] = bvh_area(b) + dist[4]; pl.take[4 * n] = 1;           // one slot: a node of its own with four slots to fill
        for (int i = 2; i <= 4; i++) {
            if (dist[i] < pl.cost[4 * n + i - 2]) { pl.cost[4 * n + i - 1] = dist[i]; pl.take[4 * n + i - 1] = (unsigned char)i; }
            else { pl.cost[4 * n + i - 1] = pl.cost[4 * n + i - 2]; pl.take[4 * n + i - 1] = pl.take[4 * n + i - 2]; }
        }
    }
}
// the children of the wide node made from binary node x, which may use `slots` slots
inline void bvh_gather(const std::vector<BinNode>& bin, const CollapsePlan& pl, int x, int slots, int* kids, int& nk)
{
    const int j = bin[x].count > 0 ? 1 : pl.take[4 * (size_t)x + slots - 1];
    if (j == 1) { kids[nk++] = x; return; }
    const int kl = pl.left[4 * (size_t)x + j - 1];
    bvh_gather(bin, pl, bin[x].left, kl, kids, nk);
    bvh_gather(bin, pl, bin[x].right, j - kl, kids, nk);
}

// emits the 4-wide node for binary node `bn` at index `me` (already reserved); returns the subtree depth
inline int bvh_collapse(const std::vector<BinNode>& bin, const CollapsePlan& pl, int bn, int me, HostBvh& out)
{
    int kids[4]; int nk = 0;
    if (bin[bn].count > 0) kids[nk++] = bn;                 // a leaf root: one leaf child
    else if (RT_BVH_DP_COLLAPSE) {
        const int kl = pl.left[4 * (size_t)bn + 3];
        bvh_gather(bin, pl, bin[bn].left, kl, kids, nk);
        bvh_gather(bin, pl, bin[bn].right, 4 - kl, kids, nk);
    } else {
        kids[nk++] = bin[bn].left; kids[nk++] = bin[bn].right;
        while (nk < 4) {
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < nk; i++)
                if (bin[kids[i]].count == 0 && bvh_area(bin[kids[i]]) > ba) { ba = bvh_area(bin[kids[i]]); best = i; }
            if (best < 0) break;
            const int k = kids[best];
            kids[best] = bin[k].left;
            kids[nk++] = bin[k].right;
        }
    }
    float f[32]; memset(f, 0, sizeof f);
    uint32_t refs[4] = { 0, 0, 0, 0 };
    int child_node[4] = { -1, -1, -1, -1 };
    for (int i = 0; i < nk; i++) {
        const BinNode& c = bin[kids[i]];
        for (int k = 0; k < 3; k++) { f[4 * k + i] = c.lo[k]; f[12 + 4 * k + i] = c.hi[k]; }
        if (c.count > 0) refs[i] = 0x80000000u | ((uint32_t)c.first << 3) | (uint32_t)c.count;
        else {
            child_node[i] = (int)(out.nodes.size() / 8);
            refs[i] = (uint32_t)child_node[i];
            out.nodes.resize(out.nodes.size() + 8);
        }
    }
    for (int i = nk; i < 4; i++)                            // empty slots: a box at +infinity (never hit), ref 0
        for (int k = 0; k < 3; k++) { f[4 * k + i] = INFINITY; f[12 + 4 * k + i] = INFINITY; }
    memcpy(&f[24], refs, sizeof refs);
    memcpy(&out.nodes[(size_t)me * 8], f, sizeof f);
    int depth = 1;
    for (int i = 0; i < nk; i++)
        if (child_node[i] >= 0) depth = std::max(depth, 1 + bvh_collapse(bin, pl, kids[i], child_node[i], out));
    return depth;
}

// what every box is grown by on every side: the primitives' own roundings at the scale of the scene (rt_bvh.h: what the
// widened ray of a far origin takes over from)
// (A/B builds, profiles/r03_experiments.txt 11b: 1e-5 sends more waves of the 4K mixed frame through the widened ray, -1.1 %;
// 1e-4 adds node visits, -3 %)
#ifndef RT_BVH_PAD
#define RT_BVH_PAD 2e-5f
#endif
inline float bvh_pad(float scale) { return RT_BVH_PAD * (scale + 1.0f); }

// prims: one entry per sphere / triangle with its exact bounds and its data in `src_pdata` (caller order).
inline void bvh_build(std::vector<BuildPrim>& prims, const std::vector<float4>& src_pdata, HostBvh& out)
{
    out.nodes.clear(); out.pdata.clear(); out.depth = 0;
    if (prims.empty()) return;
    float scale = 0.0f;
    for (const BuildPrim& p : prims)
        for (int k = 0; k < 3; k++) scale = std::max(scale, std::max(std::fabs(p.lo[k]), std::fabs(p.hi[k])));
    const float pad = bvh_pad(scale);
    out.pad = pad;
    // SAH tree first; if it comes out too deep for the traversal stack (3 entries per level), the balanced
    // median tree, whose depth is known to fit, is built instead
    for (int attempt = 0; attempt < 2; attempt++) {
        std::vector<BinNode> bin;
        bin.reserve(prims.size());
        bvh_bin_build(prims, 0, prims.size(), pad, attempt == 0, bin);
        out.nodes.clear();
        out.nodes.reserve(prims.size() * 4);
        out.nodes.resize(8);
        CollapsePlan pl;
        if (RT_BVH_DP_COLLAPSE) bvh_plan_collapse(bin, pl);
        out.depth = bvh_collapse(bin, pl, 0, 0, out);
        if (3 * out.depth + 1 <= kBvhStack) break;
    }
    out.pdata.resize(3 * prims.size());
    for (size_t i = 0; i < prims.size(); i++) {
        const uint32_t src = prims[i].ref & 0x7fffffffu, type = prims[i].ref >> 31;
        float4 a = src_pdata[src], b = make_float4(0, 0, 0, 0), c = make_float4(0, 0, 0, 0);
        if (type) { b = src_pdata[src + 1]; c.x = src_pdata[src + 2].x; }
        memcpy(&c.y, &prims[i].surf, 4);
        memcpy(&c.z, &type, 4);
        out.pdata[3 * i] = a; out.pdata[3 * i + 1] = b; out.pdata[3 * i + 2] = c;
    }
}

}  // namespace rt
