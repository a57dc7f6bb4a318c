// per-opcode wave64 issue cost on gfx950, inline asm so the compiler cannot fold anything.
// 8 independent accumulators per lane, 8 waves per SIMD -> latency is hidden, issue rate is measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 1024;
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DEFK(NAME, ASM, NOPS)                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {             \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,   \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed ^ 0x55, c = seed + 9;        \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;        \
        for (int it = 0; it < ITER; it++) { ASM ASM ASM ASM }                                    \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7); \
    }                                                                                           \
    static const int nops_##NAME = NOPS;
#define A32(INS) asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define A64(INS) asm volatile(INS : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(b), "v"(c));
#define X8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)

DEFK(sub, A32("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8"), 8)
DEFK(min3, A32("v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n v_min3_u32 %3, %3, %8, %9\n v_min3_u32 %4, %4, %8, %9\n v_min3_u32 %5, %5, %8, %9\n v_min3_u32 %6, %6, %8, %9\n v_min3_u32 %7, %7, %8, %9"), 8)
DEFK(min, A32("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8"), 8)
DEFK(and_, A32("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"), 8)
DEFK(lshl_or, A32("v_lshl_or_b32 %0, %0, 6, %8\n v_lshl_or_b32 %1, %1, 6, %8\n v_lshl_or_b32 %2, %2, 6, %8\n v_lshl_or_b32 %3, %3, 6, %8\n v_lshl_or_b32 %4, %4, 6, %8\n v_lshl_or_b32 %5, %5, 6, %8\n v_lshl_or_b32 %6, %6, 6, %8\n v_lshl_or_b32 %7, %7, 6, %8"), 8)
DEFK(and_or, A32("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9"), 8)
DEFK(alignbyte, A32("v_alignbyte_b32 %0, %0, %8, 1\n v_alignbyte_b32 %1, %1, %8, 1\n v_alignbyte_b32 %2, %2, %8, 1\n v_alignbyte_b32 %3, %3, %8, 1\n v_alignbyte_b32 %4, %4, %8, 1\n v_alignbyte_b32 %5, %5, %8, 1\n v_alignbyte_b32 %6, %6, %8, 1\n v_alignbyte_b32 %7, %7, %8, 1"), 8)
DEFK(bfrev, A32("v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1\n v_bfrev_b32 %2, %2\n v_bfrev_b32 %3, %3\n v_bfrev_b32 %4, %4\n v_bfrev_b32 %5, %5\n v_bfrev_b32 %6, %6\n v_bfrev_b32 %7, %7"), 8)
DEFK(ffbl, A32("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7"), 8)
DEFK(bfe, A32("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n v_bfe_u32 %4, %4, 3, 9\n v_bfe_u32 %5, %5, 3, 9\n v_bfe_u32 %6, %6, 3, 9\n v_bfe_u32 %7, %7, 3, 9"), 8)
DEFK(cmp_cnd, A32("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc"), 8)
DEFK(sad_u8, A32("v_sad_u8 %0, %0, %8, %9\n v_sad_u8 %1, %1, %8, %9\n v_sad_u8 %2, %2, %8, %9\n v_sad_u8 %3, %3, %8, %9\n v_sad_u8 %4, %4, %8, %9\n v_sad_u8 %5, %5, %8, %9\n v_sad_u8 %6, %6, %8, %9\n v_sad_u8 %7, %7, %8, %9"), 8)
DEFK(sad_hi_u8, A32("v_sad_hi_u8 %0, %0, %8, %9\n v_sad_hi_u8 %1, %1, %8, %9\n v_sad_hi_u8 %2, %2, %8, %9\n v_sad_hi_u8 %3, %3, %8, %9\n v_sad_hi_u8 %4, %4, %8, %9\n v_sad_hi_u8 %5, %5, %8, %9\n v_sad_hi_u8 %6, %6, %8, %9\n v_sad_hi_u8 %7, %7, %8, %9"), 8)
DEFK(dot4, A32("v_dot4_u32_u8 %0, %0, %8, %9\n v_dot4_u32_u8 %1, %1, %8, %9\n v_dot4_u32_u8 %2, %2, %8, %9\n v_dot4_u32_u8 %3, %3, %8, %9\n v_dot4_u32_u8 %4, %4, %8, %9\n v_dot4_u32_u8 %5, %5, %8, %9\n v_dot4_u32_u8 %6, %6, %8, %9\n v_dot4_u32_u8 %7, %7, %8, %9"), 8)
DEFK(mad_u24, A32("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9"), 8)
DEFK(add3, A32("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9"), 8)
DEFK(pk_sub_u16, A32("v_pk_sub_u16 %0, %0, %8\n v_pk_sub_u16 %1, %1, %8\n v_pk_sub_u16 %2, %2, %8\n v_pk_sub_u16 %3, %3, %8\n v_pk_sub_u16 %4, %4, %8\n v_pk_sub_u16 %5, %5, %8\n v_pk_sub_u16 %6, %6, %8\n v_pk_sub_u16 %7, %7, %8"), 8)
DEFK(pk_min_u16, A32("v_pk_min_u16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_min_u16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n v_pk_min_u16 %4, %4, %8\n v_pk_min_u16 %5, %5, %8\n v_pk_min_u16 %6, %6, %8\n v_pk_min_u16 %7, %7, %8"), 8)
DEFK(lshr64, A64("v_lshrrev_b64 %0, %8, %0\n v_lshrrev_b64 %1, %8, %1\n v_lshrrev_b64 %2, %8, %2\n v_lshrrev_b64 %3, %8, %3\n v_lshrrev_b64 %4, %8, %4\n v_lshrrev_b64 %5, %8, %5\n v_lshrrev_b64 %6, %8, %6\n v_lshrrev_b64 %7, %8, %7"), 8)
DEFK(lshl64, A64("v_lshlrev_b64 %0, 4, %0\n v_lshlrev_b64 %1, 4, %1\n v_lshlrev_b64 %2, 4, %2\n v_lshlrev_b64 %3, 4, %3\n v_lshlrev_b64 %4, 4, %4\n v_lshlrev_b64 %5, 4, %5\n v_lshlrev_b64 %6, 4, %6\n v_lshlrev_b64 %7, 4, %7"), 8)
DEFK(fma_f32, A32("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"), 8)
DEFK(mov_dpp, A32("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf"), 8)
DEFK(perm, A32("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9"), 8)
DEFK(xor3, A32("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x96"), 8)

template <class K>
void run(const char* name, K kern, int nops) {
    const int blocks = 256 * 8;
    uint32_t* d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u + r);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double waveinstr = (double)blocks * 4 * ITER * 4 * nops;
    double cyc = (best * 1e-3) * 2.4e9 * 1024.0 / waveinstr;
    printf("%-14s %8.3f ms  %7.2f T lane-ops/s  %5.2f cycles/wave64-instr @2.4GHz\n", name, best,
           waveinstr * 64 / (best * 1e-3) / 1e12, cyc);
    hipFree(d);
}
#define RUN(NAME) run(#NAME, k_##NAME, nops_##NAME);
int main() {
    RUN(sub) RUN(min) RUN(min3) RUN(and_) RUN(lshl_or) RUN(and_or) RUN(add3) RUN(xor3) RUN(mad_u24) RUN(alignbyte) RUN(perm) RUN(bfrev)
    RUN(ffbl) RUN(bfe) RUN(cmp_cnd) RUN(sad_u8) RUN(sad_hi_u8) RUN(dot4) RUN(pk_sub_u16) RUN(pk_min_u16) RUN(lshr64) RUN(lshl64)
    RUN(fma_f32) RUN(mov_dpp)
    return 0;
}
