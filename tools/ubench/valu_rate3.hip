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

DEFK(add, A32("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"), 8)
DEFK(or_, A32("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8"), 8)
DEFK(xor_, A32("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"), 8)
DEFK(lshl, A32("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7"), 8)
DEFK(lshr, A32("v_lshrrev_b32 %0, %8, %0\n v_lshrrev_b32 %1, %8, %1\n v_lshrrev_b32 %2, %8, %2\n v_lshrrev_b32 %3, %8, %3\n v_lshrrev_b32 %4, %8, %4\n v_lshrrev_b32 %5, %8, %5\n v_lshrrev_b32 %6, %8, %6\n v_lshrrev_b32 %7, %8, %7"), 8)
DEFK(max, A32("v_max_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_max_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n v_max_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_max_u32 %7, %7, %8"), 8)
DEFK(mov, A32("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"), 8)
DEFK(cnd, A32("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"), 8)
DEFK(cmp, A32("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8"), 8)
DEFK(subrev, A32("v_subrev_u32 %0, %8, %0\n v_subrev_u32 %1, %8, %1\n v_subrev_u32 %2, %8, %2\n v_subrev_u32 %3, %8, %3\n v_subrev_u32 %4, %8, %4\n v_subrev_u32 %5, %8, %5\n v_subrev_u32 %6, %8, %6\n v_subrev_u32 %7, %8, %7"), 8)
DEFK(sub_k, A32("v_sub_u32 %0, %0, 32\n v_sub_u32 %1, %1, 32\n v_sub_u32 %2, %2, 32\n v_sub_u32 %3, %3, 32\n v_sub_u32 %4, %4, 32\n v_sub_u32 %5, %5, 32\n v_sub_u32 %6, %6, 32\n v_sub_u32 %7, %7, 32"), 8)
DEFK(sub_e64, A32("v_sub_u32_e64 %0, %0, %8\n v_sub_u32_e64 %1, %1, %8\n v_sub_u32_e64 %2, %2, %8\n v_sub_u32_e64 %3, %3, %8\n v_sub_u32_e64 %4, %4, %8\n v_sub_u32_e64 %5, %5, %8\n v_sub_u32_e64 %6, %6, %8\n v_sub_u32_e64 %7, %7, %8"), 8)
DEFK(mul_u24, A32("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8"), 8)
DEFK(and_k, A32("v_and_b32 %0, 0xffffff, %0\n v_and_b32 %1, 0xffffff, %1\n v_and_b32 %2, 0xffffff, %2\n v_and_b32 %3, 0xffffff, %3\n v_and_b32 %4, 0xffffff, %4\n v_and_b32 %5, 0xffffff, %5\n v_and_b32 %6, 0xffffff, %6\n v_and_b32 %7, 0xffffff, %7"), 8)
DEFK(min_i, A32("v_min_i32 %0, %0, %8\n v_min_i32 %1, %1, %8\n v_min_i32 %2, %2, %8\n v_min_i32 %3, %3, %8\n v_min_i32 %4, %4, %8\n v_min_i32 %5, %5, %8\n v_min_i32 %6, %6, %8\n v_min_i32 %7, %7, %8"), 8)
DEFK(sub_dpp, A32("v_sub_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_sub_u32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf"), 8)
DEFK(add_sdwa, A32("v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"), 8)
DEFK(lshl_add, A32("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8"), 8)
DEFK(sub2x, A32("v_sub_u32 %0, %8, %9\n v_sub_u32 %1, %8, %9\n v_sub_u32 %2, %8, %9\n v_sub_u32 %3, %8, %9\n v_sub_u32 %4, %8, %9\n v_sub_u32 %5, %8, %9\n v_sub_u32 %6, %8, %9\n v_sub_u32 %7, %8, %9"), 8)

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
    RUN(add) RUN(or_) RUN(xor_) RUN(lshl) RUN(lshr) RUN(max) RUN(mov) RUN(cnd) RUN(cmp) RUN(subrev) RUN(sub_k) RUN(sub_e64) RUN(mul_u24) RUN(and_k) RUN(min_i) RUN(sub_dpp) RUN(add_sdwa) RUN(lshl_add) RUN(sub2x)
    return 0;
}
