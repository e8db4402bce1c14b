// clock_probe.hip — which clock do the "cycles" of DESIGN.md section 6 refer to?  
// Every wave runs a fixed number of independent VALU instructions and reads BOTH device counters around them:
// s_memtime (__builtin_readcyclecounter) and the constant 100 MHz wall clock (wall_clock64).  Their ratio is the
// frequency of the s_memtime counter; if it moved with the load it would be the shader clock.  The cost of an
// instruction is reported in wall-clock ns (no assumption) for: one wave on an idle chip, then 1 / 2 / 4 / 8 waves per
// SIMD on every SIMD.  Opcode rows: v_add_f32, v_fma_f32 with three VGPR operands, with an SGPR multiplicand, with an
// inline-constant multiplicand, v_pk_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define OP8(str) asm volatile(str(0) str(1) str(2) str(3) str(4) str(5) str(6) str(7) \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k), "s"(ks));
#define S_ADD(i)     "v_add_f32 %" #i ", %" #i ", %8\n"
#define S_FMA(i)     "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define S_FMA_S(i)   "v_fma_f32 %" #i ", %" #i ", %9, %8\n"
#define S_FMA_C(i)   "v_fma_f32 %" #i ", %" #i ", 0.5, %8\n"
#define S_FMAC_S(i)  "v_fmac_f32 %" #i ", %9, %8\n"
#define S_MAX(i)     "v_max_f32 %" #i ", %" #i ", %8\n"

struct Sample { unsigned long long ticks, wall, w0, w1; };

template <int KIND>
__global__ __launch_bounds__(64) void probe(float* out, Sample* samples, int iters)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = (float)(threadIdx.x + i);
    const float k = 1.0001f;
    float ks = 0.9999f;
    asm volatile("" : "+s"(ks));
    float2 pk[4] = { { a[0], a[1] }, { a[2], a[3] }, { a[4], a[5] }, { a[6], a[7] } };
    const double kk = 1.0;
    const unsigned long long w0 = wall_clock64(), t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) OP8(S_ADD)
            else if (KIND == 1) OP8(S_FMA)
            else if (KIND == 2) OP8(S_FMA_S)
            else if (KIND == 3) OP8(S_FMA_C)
            else if (KIND == 4) OP8(S_FMAC_S)
            else if (KIND == 5) OP8(S_MAX)
            else {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]) : "v"(kk));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s + pk[0].x + pk[1].y + pk[2].x + pk[3].y;
    if (threadIdx.x == 0) { samples[blockIdx.x].ticks = t1 - t0; samples[blockIdx.x].wall = w1 - w0; samples[blockIdx.x].w0 = w0; samples[blockIdx.x].w1 = w1; }
}

template <int KIND>
void run(const char* name, int grid, int wps, int iters, float* out, Sample* d_s)
{
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(64), 0, 0, out, d_s, iters); hipDeviceSynchronize(); }
    std::vector<Sample> h(grid);
    hipMemcpy(h.data(), d_s, sizeof(Sample) * grid, hipMemcpyDeviceToHost);
    double ticks = 0, wall = 0;
    unsigned long long first = ~0ull, last = 0;
    for (auto& s : h) { ticks += (double)s.ticks; wall += (double)s.wall; first = std::min(first, s.w0); last = std::max(last, s.w1); }
    const double span = (double)(last - first);                  // first wave in to last wave out, 10 ns ticks
    const double simds = grid == 1 ? 1.0 : (double)grid / wps;
    const double resident = wall / span / simds;                 // waves that were resident per SIMD, averaged over the span
    const double mhz = ticks / wall * 100.0;
    ticks /= grid; wall /= grid;
    const double instr = (double)iters * 16 * 8;
    const double ns = span * 10.0 * simds / (instr * grid);      // what a SIMD needed per instruction over the whole launch
    std::printf("%-38s %d waves/SIMD launched%s: s_memtime at %6.1f MHz; resident per SIMD %.2f; %.3f ns per wave64 instruction per SIMD = %.2f cycles of that clock"
                " (a wave alone sees %.2f cycles between its instructions)\n",
                name, wps, grid == 1 ? " (ONE wave, idle chip)" : "", mhz, resident, ns, ns * mhz * 1e-3, ticks / instr);
}

int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    std::printf("CUs %d; hipDeviceAttributeClockRate %d kHz, WallClockRate %d kHz\n", cus, clk, wclk);
    const int max_grid = cus * 4 * 8;
    float* out; Sample* d_s;
    hipMalloc(&out, sizeof(float) * 64 * max_grid); hipMalloc(&d_s, sizeof(Sample) * max_grid);
    const int iters = 20000;      // 2.56 M instructions per wave: tens of milliseconds, long enough for the clock to settle
    run<0>("v_add_f32 v,v,v", 1, 1, iters, out, d_s);
    run<1>("v_fma_f32 v,v,v,v", 1, 1, iters, out, d_s);
    for (int wps : { 1, 2, 4, 8 }) {
        const int grid = cus * 4 * wps;
        run<0>("v_add_f32 v,v,v", grid, wps, iters, out, d_s);
        run<1>("v_fma_f32 v,v,v,v", grid, wps, iters, out, d_s);
        run<2>("v_fma_f32 v,v,s,v (SGPR multiplicand)", grid, wps, iters, out, d_s);
        run<3>("v_fma_f32 v,v,0.5,v (inline const)", grid, wps, iters, out, d_s);
        run<4>("v_fmac_f32 v,s,v", grid, wps, iters, out, d_s);
        run<5>("v_max_f32 v,v,v", grid, wps, iters, out, d_s);
        run<6>("v_pk_fma_f32", grid, wps, iters, out, d_s);
    }
    return 0;
}
