// How long does it take to get N single-wave workgroups onto the chip?  (The first pass of a 640x480x1 frame launches
// 5 632 waves that live 45 us each, and its last wave starts 35 us after the first: tools/wave_times.py.)
// Empty kernels with the render kernel's LDS footprint, as 64-thread and as 256-thread workgroups; event-timed.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/launch_ramp tools/launch_ramp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void touch(unsigned long long* out, int spin)
{
    extern __shared__ unsigned int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = wall_clock64(), t = t0;
    while ((long long)(t - t0) < spin) t = wall_clock64();     // spin: ticks of the 100 MHz clock the wave stays alive
    if (threadIdx.x == 0 && out) out[blockIdx.x] = t0;
}

int main()
{
    unsigned long long* d = nullptr;
    hipMalloc(&d, sizeof(unsigned long long) * 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int waves = 5632;
    struct Cfg { int threads; size_t lds; int spin; const char* name; };
    const Cfg cfgs[] = {
        { 64, 7200, 0, "64-thread groups, 7.2 KB LDS, empty" }, { 256, 4 * 7200, 0, "256-thread groups, 28.8 KB LDS, empty" },
        { 64, 0, 0, "64-thread groups, no LDS, empty" }, { 256, 0, 0, "256-thread groups, no LDS, empty" },
        { 64, 7200, 4500, "64-thread groups, 7.2 KB LDS, 45 us alive" }, { 256, 4 * 7200, 4500, "256-thread groups, 28.8 KB LDS, 45 us alive" },
    };
    for (const Cfg& c : cfgs) {
        const int grid = waves * 64 / c.threads;
        hipFuncSetAttribute(reinterpret_cast<const void*>(&touch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(touch, dim3(grid), dim3(c.threads), c.lds, 0, d, c.spin);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        const int reps = 20;
        for (int i = 0; i < reps; i++) hipLaunchKernelGGL(touch, dim3(grid), dim3(c.threads), c.lds, 0, d, c.spin);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0; for (auto v : h) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        printf("%-48s %7.1f us per launch, last group starts %6.1f us after the first\n", c.name, ms * 1e3 / reps, (hi - lo) / 100.0);
    }
    // LDS allocation granularity: groups per CU that really are resident together, by LDS bytes per group
    for (int lds = 6144; lds <= 8192; lds += 256) {
        for (int per_cu = 19; per_cu <= 26; per_cu++) {
            const int grid = per_cu * 256;
            hipFuncSetAttribute(reinterpret_cast<const void*>(&touch), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL(touch, dim3(grid), dim3(64), lds, 0, d, 2000);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), d, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
            unsigned long long lo = ~0ull, hi = 0; for (auto v : h) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
            if ((hi - lo) > 1000) { printf("LDS %5d B per wave: %d waves per CU resident (the %dth waits)\n", lds, per_cu - 1, per_cu); break; }
            if (per_cu == 26) printf("LDS %5d B per wave: 26 or more waves per CU resident\n", lds);
        }
    }
    return 0;
}
