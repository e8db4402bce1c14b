// valu_ops.hip — issue cost of individual wave64 VALU opcodes on gfx950, relative to v_add_f32.
// Eight independent chains per wave, 8 waves per SIMD, every SIMD loaded; timed with events.
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP8(str) asm volatile(str(0) str(1) str(2) str(3) str(4) str(5) str(6) str(7) \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k), "v"(kk) : "vcc", "s20", "s21");
#define S_ADD(i)    "v_add_f32 %" #i ", %" #i ", %8\n"
#define S_FMA(i)    "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define S_XOR(i)    "v_xor_b32 %" #i ", %" #i ", %8\n"
#define S_MULLO(i)  "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define S_MULHI(i)  "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define S_MUL24(i)  "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define S_SQRT(i)   "v_sqrt_f32 %" #i ", %" #i "\n"
#define S_RCP(i)    "v_rcp_f32 %" #i ", %" #i "\n"
#define S_CND(i)    "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define S_ALIGN(i)  "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 13\n"
#define S_CVT(i)    "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define S_MOV(i)    "v_mov_b32 %" #i ", %8\n"
#define S_ADD3(i)   "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define S_SUB(i)    "v_sub_f32 %" #i ", %" #i ", %8\n"
#define S_MULF(i)   "v_mul_f32 %" #i ", %" #i ", %8\n"
#define S_FMAC(i)   "v_fmac_f32 %" #i ", %8, %8\n"
#define S_AND(i)    "v_and_b32 %" #i ", %" #i ", %8\n"
#define S_OR(i)     "v_or_b32 %" #i ", %" #i ", %8\n"
#define S_LSHL(i)   "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define S_LSHR(i)   "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define S_ADDU(i)   "v_add_u32 %" #i ", %" #i ", %8\n"
#define S_SUBU(i)   "v_sub_u32 %" #i ", %" #i ", %8\n"
#define S_MIN(i)    "v_min_f32 %" #i ", %" #i ", %8\n"
#define S_MAX3(i)   "v_max3_f32 %" #i ", %" #i ", %8, %8\n"
#define S_MED3(i)   "v_med3_f32 %" #i ", %" #i ", %8, %8\n"
#define S_BFE(i)    "v_bfe_u32 %" #i ", %" #i ", 3, 8\n"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define S_CMP(i)    "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define S_CMPS(i)   "v_cmp_lt_f32 s[20:21], %" #i ", %8\n"
#define S_CND2(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define S_CNDS(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define S_FLOOR(i)  "v_floor_f32 %" #i ", %" #i "\n"
#define S_CVTI(i)   "v_cvt_u32_f32 %" #i ", %" #i "\n"
#define S_MBCNT(i)  "v_mbcnt_lo_u32_b32 %" #i ", -1, %" #i "\n"
#define S_ADDCO(i)  "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define S_PERM(i)   "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define S_SDWA(i)   "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define S_DPP(i)    "v_add_f32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define S_ADDABS(i) "v_add_f32 %" #i ", |%" #i "|, %8\n"
#define S_MULLIT(i) "v_mul_f32 %" #i ", 0x3fc90fdb, %" #i "\n"
#define S_RSQ(i)    "v_rsq_f32 %" #i ", %" #i "\n"
#define S_CNDE64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define S_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define S_CMPCNDS(i) "v_cmp_lt_f32 s[20:21], %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define S_ADDCND(i) "v_add_f32 %" #i ", %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define S_ADDC(i)   "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define S_CMPADD(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_add_f32 %" #i ", %" #i ", %8\n"
#define S_MAX(i)    "v_max_f32 %" #i ", %" #i ", %8\n"

template <int KIND>
__global__ __launch_bounds__(64) void op_kernel(float* out, int iters)
{
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = (float)(threadIdx.x + i);
    const float k = 1.0001f;
    const double kk = 1.0;
    unsigned long long w[4] = { threadIdx.x, threadIdx.x + 1, threadIdx.x + 2, threadIdx.x + 3 };
    float2 pk[4] = { { a[0], a[1] }, { a[2], a[3] }, { a[4], a[5] }, { a[6], a[7] } };
    double dd[4] = { a[0], a[1], a[2], a[3] };
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) OP8(S_ADD)
            else if (KIND == 1) OP8(S_FMA)
            else if (KIND == 2) OP8(S_XOR)
            else if (KIND == 3) OP8(S_MULLO)
            else if (KIND == 4) OP8(S_MULHI)
            else if (KIND == 5) OP8(S_MUL24)
            else if (KIND == 6) OP8(S_SQRT)
            else if (KIND == 7) OP8(S_RCP)
            else if (KIND == 8) OP8(S_CND)
            else if (KIND == 9) OP8(S_ALIGN)
            else if (KIND == 10) OP8(S_CVT)
            else if (KIND == 11) OP8(S_MOV)
            else if (KIND == 12) OP8(S_ADD3)
            else if (KIND == 13) OP8(S_MAX)
            else if (KIND == 20) OP8(S_SUB)
            else if (KIND == 21) OP8(S_MULF)
            else if (KIND == 22) OP8(S_FMAC)
            else if (KIND == 23) OP8(S_AND)
            else if (KIND == 24) OP8(S_OR)
            else if (KIND == 25) OP8(S_LSHL)
            else if (KIND == 26) OP8(S_LSHR)
            else if (KIND == 27) OP8(S_ADDU)
            else if (KIND == 28) OP8(S_SUBU)
            else if (KIND == 29) OP8(S_MIN)
            else if (KIND == 30) OP8(S_MAX3)
            else if (KIND == 31) OP8(S_MED3)
            else if (KIND == 32) OP8(S_BFE)
            else if (KIND == 33) OP8(S_LSHLADD)
            else if (KIND == 34) OP8(S_CMP)
            else if (KIND == 35) OP8(S_CMPS)
            else if (KIND == 36) OP8(S_CND2)
            else if (KIND == 37) OP8(S_CNDS)
            else if (KIND == 38) OP8(S_FLOOR)
            else if (KIND == 39) OP8(S_CVTI)
            else if (KIND == 40) OP8(S_MBCNT)
            else if (KIND == 41) OP8(S_ADDCO)
            else if (KIND == 42) OP8(S_PERM)
            else if (KIND == 43) OP8(S_SDWA)
            else if (KIND == 44) OP8(S_DPP)
            else if (KIND == 45) OP8(S_ADDABS)
            else if (KIND == 46) OP8(S_MULLIT)
            else if (KIND == 47) OP8(S_RSQ)
            else if (KIND == 60) OP8(S_CNDE64)
            else if (KIND == 61) OP8(S_CMPCND)
            else if (KIND == 62) OP8(S_CMPCNDS)
            else if (KIND == 63) OP8(S_ADDCND)
            else if (KIND == 64) OP8(S_ADDC)
            else if (KIND == 65) OP8(S_CMPADD)
            else if (KIND == 14) {      // v_mad_u64_u32: 4 chains x 2 = 8 instructions
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3\n"
                             "v_mad_u64_u32 %0, vcc, %4, %4, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %4, %4, %2\n v_mad_u64_u32 %3, vcc, %4, %4, %3\n"
                             : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "v"(k) : "vcc");
            } else if (KIND == 15) {    // v_pk_fma_f32
                asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                             : "+v"(pk[0]), "+v"(pk[1]), "+v"(pk[2]), "+v"(pk[3]) : "v"(kk));
            } else if (KIND == 16) {    // v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                             "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                             : "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]) : "v"(kk));
            } else if (KIND == 17) {    // v_lshl_add_u64
                asm volatile("v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3\n"
                             "v_lshl_add_u64 %0, %0, 1, %0\n v_lshl_add_u64 %1, %1, 1, %1\n v_lshl_add_u64 %2, %2, 1, %2\n v_lshl_add_u64 %3, %3, 1, %3\n"
                             : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s + (float)(w[0] + w[1] + w[2] + w[3]) + pk[0].x + pk[1].x + pk[2].y + pk[3].y + (float)(dd[0] + dd[1] + dd[2] + dd[3]);
}

template <int KIND>
double run(const char* name, int cus, double base)
{
    const int wps = 8, grid = cus * 4 * wps, iters = 4000;
    float* out; hipMalloc(&out, sizeof(float) * 64 * grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(op_kernel<KIND>, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(op_kernel<KIND>, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 16 * 8 * wps;
    const double ns = ms * 1e6 / per_simd;
    std::printf("%-16s %.3f ns per wave64 instruction per SIMD (%.2f cycles at 2.4 GHz)%s\n", name, ns, ns * 2.4, base > 0 ? "" : "  [baseline]");
    if (base > 0) std::printf("%-16s   = %.2f x v_add_f32\n", "", ns / base);
    hipFree(out);
    return ns;
}

int main()
{
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const double b = run<0>("v_add_f32", cus, 0);
    run<1>("v_fma_f32", cus, b); run<2>("v_xor_b32", cus, b); run<3>("v_mul_lo_u32", cus, b); run<4>("v_mul_hi_u32", cus, b);
    run<5>("v_mul_u32_u24", cus, b); run<6>("v_sqrt_f32", cus, b); run<7>("v_rcp_f32", cus, b); run<8>("v_cndmask_b32", cus, b);
    run<9>("v_alignbit_b32", cus, b); run<10>("v_cvt_f32_u32", cus, b); run<11>("v_mov_b32", cus, b); run<12>("v_add3_u32", cus, b);
    run<13>("v_max_f32", cus, b); run<14>("v_mad_u64_u32", cus, b); run<15>("v_pk_fma_f32", cus, b); run<16>("v_fma_f64", cus, b);
    run<17>("v_lshl_add_u64", cus, b);
    run<20>("v_sub_f32", cus, b);
    run<21>("v_mul_f32", cus, b);
    run<22>("v_fmac_f32", cus, b);
    run<23>("v_and_b32", cus, b);
    run<24>("v_or_b32", cus, b);
    run<25>("v_lshlrev_b32", cus, b);
    run<26>("v_lshrrev_b32", cus, b);
    run<27>("v_add_u32", cus, b);
    run<28>("v_sub_u32", cus, b);
    run<29>("v_min_f32", cus, b);
    run<30>("v_max3_f32", cus, b);
    run<31>("v_med3_f32", cus, b);
    run<32>("v_bfe_u32", cus, b);
    run<33>("v_lshl_add_u32", cus, b);
    run<34>("v_cmp_lt_f32 vcc", cus, b);
    run<35>("v_cmp_lt_f32 sgpr", cus, b);
    run<36>("v_cndmask vcc", cus, b);
    run<37>("v_cndmask sgpr", cus, b);
    run<38>("v_floor_f32", cus, b);
    run<39>("v_cvt_u32_f32", cus, b);
    run<40>("v_mbcnt_lo", cus, b);
    run<41>("v_add_co_u32", cus, b);
    run<42>("v_perm_b32", cus, b);
    run<43>("v_add_u32_sdwa", cus, b);
    run<44>("v_add_f32_dpp", cus, b);
    run<45>("v_add_f32 |abs|", cus, b);
    run<46>("v_mul_f32 literal", cus, b);
    run<47>("v_rsq_f32", cus, b);
    run<60>("v_cndmask_e64 vcc", cus, b);
    run<61>("[cmp vcc + cndmask vcc] x1 pair", cus, b);
    run<62>("[cmp sgpr + cndmask sgpr] pair", cus, b);
    run<63>("[add + cndmask vcc] pair", cus, b);
    run<64>("v_addc_co_u32 vcc", cus, b);
    run<65>("[cmp vcc + add] pair", cus, b);
    return 0;
}
