// tools/bvh_stats.cpp — offline quality numbers of the host BVH builder (no GPU): expected visits of a random ray that
// hits the root box, by the surface-area heuristic.  Input: a binary file of N x 6 floats (lo.xyz hi.xyz per primitive),
// written by tools/bvh_stats.py.   hipcc -O2 -std=c++17 -I raytracer_amd/csrc -o /tmp/bvh_stats tools/bvh_stats.cpp
#include "rt_bvh_build.h"
#include <chrono>
#include <cstdio>
using namespace rt;
static double area(const float* lo, const float* hi) { const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; }
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); const size_t n = ftell(f) / 24; fseek(f, 0, SEEK_SET);
    std::vector<float> raw(n * 6); if (fread(raw.data(), 24, n, f) != n) return 1; fclose(f);
    std::vector<BuildPrim> prims(n); std::vector<float4> pd(n);
    float rlo[3] = { INFINITY, INFINITY, INFINITY }, rhi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { prims[i].lo[k] = raw[6 * i + k]; prims[i].hi[k] = raw[6 * i + 3 + k]; prims[i].c[k] = 0.5f * (prims[i].lo[k] + prims[i].hi[k]);
                                      rlo[k] = std::min(rlo[k], prims[i].lo[k]); rhi[k] = std::max(rhi[k], prims[i].hi[k]); }
        prims[i].ref = (uint32_t)i; prims[i].surf = (uint32_t)i; pd[i] = make_float4(0, 0, 0, 0);
    }
    HostBvh hb;
    const auto t0 = std::chrono::steady_clock::now();
    bvh_build(prims, pd, hb);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const double aroot = area(rlo, rhi);
    const size_t nn = hb.nodes.size() / 8;
    double e_inner = 1.0, e_leaf = 0.0, e_prims = 0.0; size_t leaves = 0, kids = 0, one = 0;
    for (size_t i = 0; i < nn; i++) {
        const float* fl = reinterpret_cast<const float*>(&hb.nodes[i * 8]);
        uint32_t refs[4]; memcpy(refs, fl + 24, 16);
        for (int c = 0; c < 4; c++) {
            if (refs[c] == 0) continue;
            kids++;
            const float lo[3] = { fl[c], fl[4 + c], fl[8 + c] }, hi[3] = { fl[12 + c], fl[16 + c], fl[20 + c] };
            const double p = area(lo, hi) / aroot;
            if (refs[c] & 0x80000000u) { e_leaf += p; e_prims += p * (refs[c] & 7u); leaves++; one += (refs[c] & 7u) == 1; }
            else e_inner += p;
        }
    }
    printf("prims %zu  nodes %zu  leaves %zu (%.1f %% with one primitive)  children/node %.2f  depth %d  build %.0f ms\n", n, nn, leaves, 100.0 * one / leaves, (double)kids / nn, hb.depth, ms);
    printf("expected per random ray through the root box:  inner visits %.3f  leaf visits %.3f  primitive tests %.3f\n", e_inner, e_leaf, e_prims);
    return 0;
}
