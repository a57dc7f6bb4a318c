// valu_cycles.hip -- issue cost of wave64 VALU instructions in SHADER CYCLES (s_memtime), independent of the clock the chip happens
// to run at: valu_rate*.hip convert a wall time with an assumed 2.4 GHz, which is how DESIGN.md came to "2.5 / 4.2 cycles" where the
// guide says 2 (full rate) -- this one counts ticks.  W waves per SIMD run the same unrolled body of independent instructions;
// cycles per instruction per SIMD = ticks of a wave / (instructions of a wave * W).  Also prints ticks / wall time = the clock.
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_cycles tools/ubench/valu_cycles.hip && tools/ubench/valu_cycles
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define BODY(op) REP8(REP8(op))       /* 64 instructions on 8 independent destinations */

template <int OP>
__global__ void k(uint64_t* ticks, uint32_t* sink, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    uint32_t b = seed * 2654435761u + 12345u, c = seed | 1u;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#define R8(ins) asm volatile(ins(%0) ins(%1) ins(%2) ins(%3) ins(%4) ins(%5) ins(%6) ins(%7) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define I_SUB(r) "v_sub_u32 " #r ", " #r ", %8\n\t"
#define I_ADD(r) "v_add_u32 " #r ", " #r ", %8\n\t"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %8\n\t"
#define I_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n\t"
#define I_MOV(r) "v_mov_b32 " #r ", %8\n\t"
#define I_LSHR(r) "v_lshrrev_b32 " #r ", 1, " #r "\n\t"
#define I_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n\t"
#define I_MIN(r) "v_min_u32 " #r ", " #r ", %8\n\t"
#define I_MIN3(r) "v_min3_u32 " #r ", " #r ", %8, %9\n\t"
#define I_ALIGN(r) "v_alignbyte_b32 " #r ", " #r ", %8, %9\n\t"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n\t"
#define I_CMP(r) "v_cmp_lt_u32 vcc, " #r ", %8\n\t"
#define I_CND(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n\t"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 9\n\t"
#define I_FFBL(r) "v_ffbl_b32 " #r ", " #r "\n\t"
#define I_MIN16(r) "v_min_u16 " #r ", " #r ", %8\n\t"
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n\t"
#define I_LSHL64 "v_lshlrev_b64 %0, 1, %0\n\t"
        if (OP == 0) { REP8(R8(I_SUB)) }
        if (OP == 1) { REP8(R8(I_ADD)) }
        if (OP == 2) { REP8(R8(I_AND)) }
        if (OP == 3) { REP8(R8(I_XOR)) }
        if (OP == 4) { REP8(R8(I_MOV)) }
        if (OP == 5) { REP8(R8(I_LSHR)) }
        if (OP == 6) { REP8(R8(I_LSHL)) }
        if (OP == 7) { REP8(R8(I_MIN)) }
        if (OP == 8) { REP8(R8(I_MIN3)) }
        if (OP == 9) { REP8(R8(I_ALIGN)) }
        if (OP == 10) { REP8(R8(I_MAD24)) }
        if (OP == 11) { REP8(R8(I_CMP)) }
        if (OP == 12) { REP8(R8(I_BFE)) }
        if (OP == 13) { REP8(R8(I_FFBL)) }
        if (OP == 14) { REP8(R8(I_MIN16)) }
        if (OP == 15) { REP8(R8(I_FMA)) }
        if (OP == 16) { REP8(R8(I_CND)) }
        // the compare of the match search: one v_sub + half a v_min3 -- 2 : 1 mix
        if (OP == 17) { REP8(asm volatile(
            "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_min3_u32 %2, %2, %0, %1\n\t"
            "v_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_min3_u32 %5, %5, %3, %4\n\t"
            "v_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63u) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) sink[0] = a0;
}

template <int OP>
void run(const char* name, int waves_per_simd, int insts_per_iter) {
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;
    uint64_t* d; uint32_t* sink;
    hipMalloc(&d, sizeof(uint64_t) * blocks * 16); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, sink, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, sink, iters, 7u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * threads / 64);
    hipMemcpy(h.data(), d, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double per = med / ((double)iters * insts_per_iter * waves_per_simd);
    printf("%-22s waves/SIMD %d: %8.0f ticks per wave (median)  %5.2f ticks per wave64 instr per SIMD   [%.3f ms wall -> %.2f G ticks/s]\n",
           name, waves_per_simd, med, per, ms, med / (ms * 1e6));
    hipFree(d); hipFree(sink);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_sub_u32", w, 64); run<1>("v_add_u32", w, 64); run<2>("v_and_b32", w, 64); run<3>("v_xor_b32", w, 64);
        run<4>("v_mov_b32", w, 64); run<5>("v_lshrrev_b32", w, 64); run<14>("v_min_u16", w, 64);
        run<6>("v_lshlrev_b32", w, 64); run<7>("v_min_u32", w, 64); run<8>("v_min3_u32", w, 64); run<9>("v_alignbyte_b32", w, 64);
        run<10>("v_mad_u32_u24", w, 64); run<11>("v_cmp_lt_u32", w, 64); run<16>("v_cndmask_b32", w, 64); run<12>("v_bfe_u32", w, 64);
        run<13>("v_ffbl_b32", w, 64); run<15>("v_fma_f32", w, 64);
        run<17>("2 v_sub + 1 v_min3 mix", w, 64);
    }
    return 0;
}
