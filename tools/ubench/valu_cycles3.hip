// valu_cycles3.hip -- valu_cycles2.hip found v_cndmask_b32 with an implicit VCC (VOP2) at 12.7 cycles per wave64 instruction against 3.2 for
// the VOP3 form with an SGPR pair.  Is that the encoding, the VCC read, or the missing producer?  Pairs as compilers emit them.
// build + run:  hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/valu_cycles3 tools/ubench/valu_cycles3.hip && tools/ubench/valu_cycles3
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP4(x) x x x x

template <int OP>
__global__ void k(uint64_t* ticks, uint32_t* sink, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    uint32_t b = seed * 2654435761u + 12345u + threadIdx.x, c = seed | 1u;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#define ASM8(txt) asm volatile(txt : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) \
        : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        // 0: cmp -> vcc, cndmask e32 (vcc), 4 pairs = 8 VALU
        if (OP == 0) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
            "v_cmp_lt_u32 vcc, %4, %8\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t")) }
        // 1: cmp -> sgpr pair, cndmask e64 (sgpr pair)
        if (OP == 1) { REP8(ASM8(
            "v_cmp_lt_u32_e64 s[20:21], %0, %8\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\tv_cmp_lt_u32_e64 s[22:23], %2, %8\n\tv_cndmask_b32_e64 %3, %3, %9, s[22:23]\n\t"
            "v_cmp_lt_u32_e64 s[24:25], %4, %8\n\tv_cndmask_b32_e64 %5, %5, %9, s[24:25]\n\tv_cmp_lt_u32_e64 s[26:27], %6, %8\n\tv_cndmask_b32_e64 %7, %7, %9, s[26:27]\n\t")) }
        // 2: cmp -> vcc, cndmask e64 with vcc as the mask operand
        if (OP == 2) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32_e64 %1, %1, %9, vcc\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cndmask_b32_e64 %3, %3, %9, vcc\n\t"
            "v_cmp_lt_u32 vcc, %4, %8\n\tv_cndmask_b32_e64 %5, %5, %9, vcc\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32_e64 %7, %7, %9, vcc\n\t")) }
        // 3: one cmp -> vcc, seven cndmask e32
        if (OP == 3) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
            "v_cndmask_b32 %4, %4, %9, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_cndmask_b32 %6, %6, %9, vcc\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t")) }
        // 4: cndmask e32 with an inline constant as src0 (as hipcc writes `x ? v : 0`)
        if (OP == 4) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, 0, %1, vcc\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cndmask_b32 %3, 0, %3, vcc\n\t"
            "v_cmp_lt_u32 vcc, %4, %8\n\tv_cndmask_b32 %5, 0, %5, vcc\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32 %7, 0, %7, vcc\n\t")) }
        // 5: v_addc_co_u32 (carry in from vcc) chains
        if (OP == 5) { REP8(ASM8(
            "v_add_co_u32 %0, vcc, %0, %8\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\tv_add_co_u32 %2, vcc, %2, %8\n\tv_addc_co_u32 %3, vcc, %3, %9, vcc\n\t"
            "v_add_co_u32 %4, vcc, %4, %8\n\tv_addc_co_u32 %5, vcc, %5, %9, vcc\n\tv_add_co_u32 %6, vcc, %6, %8\n\tv_addc_co_u32 %7, vcc, %7, %9, vcc\n\t")) }
        // 6: the same 8 destinations, cndmask e32 only, VCC written once per iteration by a v_cmp in front
        if (OP == 6) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc");
            REP8(ASM8(
            "v_cndmask_b32 %0, %0, %9, vcc\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
            "v_cndmask_b32 %4, %4, %9, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_cndmask_b32 %6, %6, %9, vcc\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t")) }
        // 7: v_sub only, reference
        if (OP == 7) { REP8(ASM8(
            "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\t"
            "v_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t")) }
        // 8: cmp e32 -> vcc + s_and_b64 into an SGPR pair + cndmask e64: what make_tokens does
        if (OP == 8) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\ts_and_b64 s[20:21], vcc, s[26:27]\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\t"
            "v_cmp_lt_u32 vcc, %2, %8\n\ts_and_b64 s[22:23], vcc, s[26:27]\n\tv_cndmask_b32_e64 %3, %3, %9, s[22:23]\n\t"
            "v_cmp_lt_u32 vcc, %4, %8\n\ts_and_b64 s[20:21], vcc, s[26:27]\n\tv_cndmask_b32_e64 %5, %5, %9, s[20:21]\n\t"
            "v_cmp_lt_u32 vcc, %6, %8\n\ts_and_b64 s[22:23], vcc, s[26:27]\n\tv_cndmask_b32_e64 %7, %7, %9, s[22:23]\n\t")) }
        // 9: min3 pairs of the search with DPP-free subs: 2 v_sub + 1 v_min3, 24 VALU per ASM8 (reference for the mix)
        if (OP == 9) { REP8(ASM8(
            "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_min3_u32 %2, %2, %0, %1\n\tv_sub_u32 %3, %3, %8\n\tv_sub_u32 %4, %4, %8\n\tv_min3_u32 %5, %5, %3, %4\n\t"
            "v_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t")) }

        // 10: one cmp, two cndmask e32 (a 64-bit select), 12 VALU per ASM8... written as 8: cmp c c cmp c c cmp c
        if (OP == 10) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\t"
            "v_cmp_lt_u32 vcc, %3, %8\n\tv_cndmask_b32 %4, %4, %9, vcc\n\tv_cndmask_b32 %5, %5, %9, vcc\n\t"
            "v_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t")) }
        // 11: the same with the e64 encoding and vcc as the explicit mask operand
        if (OP == 11) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32_e64 %1, %1, %9, vcc\n\tv_cndmask_b32_e64 %2, %2, %9, vcc\n\t"
            "v_cmp_lt_u32 vcc, %3, %8\n\tv_cndmask_b32_e64 %4, %4, %9, vcc\n\tv_cndmask_b32_e64 %5, %5, %9, vcc\n\t"
            "v_cmp_lt_u32 vcc, %6, %8\n\tv_cndmask_b32_e64 %7, %7, %9, vcc\n\t")) }
        // 12: cndmask e32 separated by an unrelated VALU instruction
        if (OP == 12) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_sub_u32 %2, %2, %8\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
            "v_sub_u32 %4, %4, %8\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_sub_u32 %6, %6, %8\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t")) }
        // 13: eight cndmask e64 with vcc as the explicit operand, back to back
        if (OP == 13) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(a0), "v"(b) : "vcc");
            REP8(ASM8(
            "v_cndmask_b32_e64 %0, %0, %9, vcc\n\tv_cndmask_b32_e64 %1, %1, %9, vcc\n\tv_cndmask_b32_e64 %2, %2, %9, vcc\n\tv_cndmask_b32_e64 %3, %3, %9, vcc\n\t"
            "v_cndmask_b32_e64 %4, %4, %9, vcc\n\tv_cndmask_b32_e64 %5, %5, %9, vcc\n\tv_cndmask_b32_e64 %6, %6, %9, vcc\n\tv_cndmask_b32_e64 %7, %7, %9, vcc\n\t")) }
        // 14: cmp_e64 -> sgpr pair, then two cndmask e64 on it
        if (OP == 14) { REP8(ASM8(
            "v_cmp_lt_u32_e64 s[20:21], %0, %8\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\tv_cndmask_b32_e64 %2, %2, %9, s[20:21]\n\t"
            "v_cmp_lt_u32_e64 s[22:23], %3, %8\n\tv_cndmask_b32_e64 %4, %4, %9, s[22:23]\n\tv_cndmask_b32_e64 %5, %5, %9, s[22:23]\n\t"
            "v_cmp_lt_u32_e64 s[24:25], %6, %8\n\tv_cndmask_b32_e64 %7, %7, %9, s[24:25]\n\t")) }
        // 15: v_cmp e32 back to back (all write vcc)
        if (OP == 15) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cmp_lt_u32 vcc, %1, %8\n\tv_cmp_lt_u32 vcc, %2, %8\n\tv_cmp_lt_u32 vcc, %3, %8\n\t"
            "v_cmp_lt_u32 vcc, %4, %8\n\tv_cmp_lt_u32 vcc, %5, %8\n\tv_cmp_lt_u32 vcc, %6, %8\n\tv_cmp_lt_u32 vcc, %7, %8\n\t")) }
        // 16: cndmask e32 vcc followed by a dependent op, alternating registers (latency exposure?)
        if (OP == 16) { REP8(ASM8(
            "v_cmp_lt_u32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
            "v_cmp_lt_u32 vcc, %4, %8\n\tv_cndmask_b32 %5, %5, %9, vcc\n\tv_cndmask_b32 %6, %6, %9, vcc\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t")) }
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
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess) printf("launch error %s\n", hipGetErrorString(err));
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * threads / 64);
    hipMemcpy(h.data(), d, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double per = med / ((double)iters * insts_per_iter * waves_per_simd);
    printf("%-52s waves/SIMD %d: %5.2f ticks per wave64 VALU instr per SIMD   [%.3f ms wall]\n", name, waves_per_simd, per, ms);
    hipFree(d); hipFree(sink);
}

int main() {
    for (int w : {1, 4}) {
        run<7>("v_sub_u32 (reference)", w, 64);
        run<0>("v_cmp -> vcc ; v_cndmask_b32_e32 vcc", w, 64);
        run<1>("v_cmp_e64 -> s[n:n+1] ; v_cndmask_b32_e64 s[n:n+1]", w, 64);
        run<2>("v_cmp -> vcc ; v_cndmask_b32_e64 vcc", w, 64);
        run<3>("1 v_cmp -> vcc ; 7 v_cndmask_b32_e32 vcc", w, 64);
        run<4>("v_cmp -> vcc ; v_cndmask_b32_e32 v, 0, v, vcc", w, 64);
        run<5>("v_add_co ; v_addc_co (vcc carry)", w, 64);
        run<6>("v_cndmask_b32_e32 vcc only (vcc set per iteration)", w, 64);
        run<8>("v_cmp -> vcc ; s_and_b64 ; v_cndmask_b32_e64 sgpr", w, 64);
        run<9>("2 v_sub + 1 v_min3 mix", w, 64);
        run<10>("cmp ; 2 cndmask e32 vcc (64-bit select)", w, 64);
        run<11>("cmp ; 2 cndmask e64 vcc", w, 64);
        run<12>("cndmask e32 vcc ; v_sub alternating", w, 64);
        run<13>("8 cndmask e64 with vcc operand back to back", w, 64);
        run<14>("cmp_e64 sgpr ; 2 cndmask e64 sgpr", w, 64);
        run<15>("v_cmp -> vcc back to back", w, 64);
        run<16>("cmp ; 3 cndmask e32 vcc", w, 64);
    }
    return 0;
}
