// round 2: does gfx950 have a 64-bit integer add that issues faster than two 32-bit adds?  (two key differences per instruction
// in k_compress's match search: a pair of own keys minus a pair of candidate keys, the carry into the high half is harmless)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 1024;
#define DEFK(NAME, BODY, NOPS)                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {             \
        uint32_t a0 = seed + threadIdx.x, b = seed ^ 0x55, c = seed + 9;                         \
        uint64_t q0 = a0, q1 = a0 * 3, q2 = a0 * 5, q3 = a0 * 7, q4 = a0 * 11, q5 = a0 * 13, q6 = a0 * 17, q7 = a0 * 19; \
        uint64_t s64 = ((uint64_t)a0 << 32) | b;                                                 \
        for (int it = 0; it < ITER; it++) { BODY BODY BODY BODY }                                \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q3 ^ q5) >> 32); \
    }                                                                                           \
    static const int nops_##NAME = NOPS;
#define A64_8(F) asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(b), "v"(c), "v"(s64));
#define F_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %10\n"
#define F_ADDCO(i) "v_add_co_u32 %L" #i ", vcc, %L" #i ", %8\n v_addc_co_u32 %H" #i ", vcc, %H" #i ", %9, vcc\n"
#define F_PKADDF32(i) "v_pk_add_f32 %" #i ", %" #i ", %10\n"
#define F_PKMOV(i) "v_pk_mov_b32 %" #i ", %10, %" #i "\n"
#define F_SUB2(i) "v_sub_u32 %L" #i ", %L" #i ", %8\n v_sub_u32 %H" #i ", %H" #i ", %9\n"
DEFK(lshl_add_u64, A64_8(F_LSHLADD64), 8) DEFK(pk_add_f32, A64_8(F_PKADDF32), 8) DEFK(pk_mov_b32, A64_8(F_PKMOV), 8)
template <class K>
void run(const char* name, K kern, int nops) {
    const int blocks = 256 * 8;
    uint32_t* d;
    (void)hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u + r);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double waveinstr = (double)blocks * 4 * ITER * 4 * nops;
    printf("%-18s %8.3f ms  %5.2f cycles/wave64-instr @2.4GHz\n", name, best, (best * 1e-3) * 2.4e9 * 1024.0 / waveinstr);
    (void)hipFree(d);
}
#define RUN(NAME) run(#NAME, k_##NAME, nops_##NAME);
int main() { RUN(lshl_add_u64) RUN(pk_add_f32) RUN(pk_mov_b32) return 0; }
