// gather_rate.hip — how fast does a CU serve the BVH traversal's kind of load?  (round 3, last session)
// A traversal round is, per lane, SEVEN 16-byte loads from ONE 128-byte line (a node) whose address depends on the
// previous round's data, every lane at a different line.  bench.py reports 15-22 bytes per clock per CU of such requests
// for the BVH frames (bvh.l1_request_GBps / 256 CUs / 2.39 GHz); this tool measures what the memory pipeline gives when
// NOTHING else happens: each lane chases pointers through a table of `nnodes` 128-byte nodes (the next index comes out
// of the loaded data), R requests of 16 bytes per visit, W waves per CU.  Variants:
//   rows = 7 / 4 / 2 / 1   16-byte requests per visit (7 = today's node, 4 = a 64-byte node)
//   same = 1               all 64 lanes at the same node (coalesced), 0: every lane its own
// Output: ns per visit per wave, lane-requests per clock per CU, bytes per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int ROWS>
__global__ __launch_bounds__(64) void chase(const float4* __restrict__ nodes, uint32_t mask, int visits, int same, float* out)
{
    uint32_t cur = (same ? blockIdx.x * 7919u : (blockIdx.x * 64u + threadIdx.x) * 2654435761u) & mask;
    float acc = 0.0f;
    for (int v = 0; v < visits; v++) {
        float4 d[ROWS];
        const char* base = reinterpret_cast<const char*>(nodes) + ((size_t)cur << 7);
#pragma unroll
        for (int r = 0; r < ROWS; r++) d[r] = *reinterpret_cast<const float4*>(base + 16 * ((r * 3) & 7));   // rows 0 3 6 1 4 7 2
        uint32_t nxt = 0;
#pragma unroll
        for (int r = 0; r < ROWS; r++) { nxt ^= __float_as_uint(d[r].x); acc += d[r].y; }
        cur = (nxt + (uint32_t)v * 0x9e3779b9u) & mask;
        if (same) cur = __builtin_amdgcn_readfirstlane(cur);
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + (float)cur;
}

template <int ROWS>
void run(const float4* d_nodes, uint32_t mask, int waves_per_cu, int visits, int same, float* d_out, size_t lds)
{
    const int grid = 256 * waves_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chase<ROWS>, dim3(grid), dim3(64), lds, 0, d_nodes, mask, visits, same, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double clk = 2.39e9 * best * 1e-3;                  // shader clocks of the launch (profiles/r03_clock_probe.txt)
    const double req = (double)grid * 64 * visits * ROWS;     // lane-requests
    printf("rows %d  same %d  waves/CU %2d  nodes %8u (%6.1f MB)  %8.3f ms  %7.1f ns per visit per wave  %6.2f lane-requests/clk/CU  %6.1f B/clk/CU\n",
           ROWS, same, waves_per_cu, mask + 1, (mask + 1) * 128.0 / 1048576.0, best, best * 1e6 / visits,
           req / clk / 256.0, req * 16.0 / clk / 256.0);
}

int main(int argc, char** argv)
{
    const int visits = argc > 1 ? atoi(argv[1]) : 4000;
    const uint32_t sizes[] = { 1u << 11, 1u << 15, 1u << 18, 1u << 20 };      // 0.25 MB (L1-ish), 4 MB (L2), 32 MB, 128 MB (Infinity Cache)
    float* d_out; hipMalloc(&d_out, 256 * 32 * 64 * sizeof(float));
    for (uint32_t n : sizes) {
        std::vector<float4> h((size_t)n * 8);
        uint32_t s = 12345u;
        for (auto& x : h) { s = s * 1664525u + 1013904223u; uint32_t a = s; s = s * 1664525u + 1013904223u; x = make_float4(__builtin_bit_cast(float, a >> 3), 1e-9f, 0.f, 0.f); }
        float4* d_nodes; hipMalloc(&d_nodes, h.size() * sizeof(float4));
        hipMemcpy(d_nodes, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
        for (int w : { 16, 32 }) {
            // 16 waves per CU are forced the way the render kernel's are: by LDS (10 KB per single-wave workgroup)
            const size_t lds = (w == 16) ? 10240 : 4096;
            run<7>(d_nodes, n - 1, w, visits, 0, d_out, lds);
            run<4>(d_nodes, n - 1, w, visits, 0, d_out, lds);
            run<2>(d_nodes, n - 1, w, visits, 0, d_out, lds);
            run<1>(d_nodes, n - 1, w, visits, 0, d_out, lds);
        }
        run<7>(d_nodes, n - 1, 16, visits, 1, d_out, 10240);
        // latency: few waves per CU (one 64 KB workgroup per wave so that they spread over the CUs)
        for (int w : { 1, 2, 4, 8 }) { run<7>(d_nodes, n - 1, w, visits, 0, d_out, 10240); run<1>(d_nodes, n - 1, w, visits, 0, d_out, 10240); }
        hipFree(d_nodes);
    }
    return 0;
}
