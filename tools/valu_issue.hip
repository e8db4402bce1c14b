// valu_issue.hip — how many cycles does a SIMD need per wave64 VALU instruction?  
// Each wave runs N independent v_add_f32 chains (ILP 8) for a fixed instruction count and reads s_memtime around it;
// launched with 1, 2, 4 and 8 waves per SIMD (one workgroup of 64 lanes per wave, every SIMD of the chip loaded).
// cycles per instruction per SIMD = (mean wave cycles) / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(64) void loop_kernel(float* out, unsigned long long* cycles, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float k = 1.0001f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) {        // v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else if (KIND == 1) { // v_mul_f32 + v_add_f32 pairs (what -ffp-contract=off code looks like)
                asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %0\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %2\n"
                             "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %4\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %6\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));
            } else {                // one dependent chain (latency)
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                             "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                             : "+v"(a0) : "v"(k));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int iters = 2000;
    const double instr = (double)iters * 16 * 8;
    const char* names[3] = { "independent v_add_f32 (8 chains)", "v_mul_f32 + dependent v_add_f32 (4 pairs)", "one dependent v_add_f32 chain" };
    std::printf("CUs %d; %g VALU instructions per wave\n", cus, instr);
    for (int kind = 0; kind < 3; kind++)
        for (int wps : { 1, 2, 4, 8 }) {
            const int grid = cus * 4 * wps;
            float* out; unsigned long long* cyc;
            hipMalloc(&out, sizeof(float) * 64 * grid); hipMalloc(&cyc, 8 * grid);
            for (int rep = 0; rep < 2; rep++) {
                if (kind == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
                else if (kind == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
                else hipLaunchKernelGGL(loop_kernel<2>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), cyc, 8 * grid, hipMemcpyDeviceToHost);
            double sum = 0; for (auto v : h) sum += (double)v;
            const double mean = sum / grid;
            // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: convert with the shader clock if so (print both)
            std::printf("%-44s waves/SIMD %d: wave ticks %.0f -> ticks per instruction per wave %.3f, per SIMD %.3f\n", names[kind], wps, mean,
                        mean / instr, mean / (instr * wps));
            hipFree(out); hipFree(cyc);
        }
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    std::printf("shader clock %d kHz, wall clock %d kHz\n", clk, wclk);
    // wall-clock cross-check of the 8 waves/SIMD case with events
    {
        const int wps = 8, grid = cus * 4 * wps;
        float* out; unsigned long long* cyc;
        hipMalloc(&out, sizeof(float) * 64 * grid); hipMalloc(&cyc, 8 * grid);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(loop_kernel<0>, dim3(grid), dim3(64), 0, 0, out, cyc, iters * 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop_kernel<0>, dim3(grid), dim3(64), 0, 0, out, cyc, iters * 10);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double total_instr = instr * 10 * grid;
        std::printf("events: %.3f ms for %.3g wave-instructions on %d SIMDs -> %.2f ns per instruction per SIMD = %.2f cycles at 2.4 GHz; %.1f TFLOP/s of adds\n",
                    ms, total_instr, cus * 4, ms * 1e6 / (total_instr / (cus * 4)), ms * 1e6 / (total_instr / (cus * 4)) * 2.4, total_instr * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
