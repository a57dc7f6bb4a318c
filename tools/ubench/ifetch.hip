// ifetch.hip -- does the VALU rate of straight-line code depend on the SIZE of the loop body?  k_compress executes ~4600 instructions
// (~28 KB) per tile, straight-line, on 20 waves per CU, and measures ~3.8 cycles per wave64 VALU instruction where tools/ubench/valu_cycles.hip
// (a 64-instruction body) gives 1.96 (full rate) / 3.25.  Here: the same independent v_sub_u32 / v_min3_u32 instructions, loop bodies of 64 .. 16384
// instructions, W waves per SIMD on every CU.  cycles = ticks of a wave (s_memtime) / (instructions of a wave * W).
// build + run:  hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/ifetch tools/ubench/ifetch.hip && tools/ubench/ifetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define R2(x) x x
#define R4(x) R2(R2(x))
#define R16(x) R4(R4(x))
#define R64(x) R16(R4(x))
#define R256(x) R64(R4(x))

#define SUB8 "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_sub_u32 %2, %2, %8\n\tv_sub_u32 %3, %3, %8\n\t" \
             "v_sub_u32 %4, %4, %8\n\tv_sub_u32 %5, %5, %8\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t"
// the search mix, VOP2 + VOP3 (8-byte) encodings: 2 v_sub + 1 v_min3
#define MIX8 "v_sub_u32 %0, %0, %8\n\tv_sub_u32 %1, %1, %8\n\tv_min3_u32 %2, %2, %0, %1\n\tv_sub_u32 %3, %3, %8\n\t" \
             "v_sub_u32 %4, %4, %8\n\tv_min3_u32 %5, %5, %3, %4\n\tv_sub_u32 %6, %6, %8\n\tv_sub_u32 %7, %7, %8\n\t"
#define ASM8(txt) asm volatile(txt : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));

template <int BODY, int MIX>     // BODY = instructions per loop iteration: 64, 512, 4096, 16384
__global__ void k(uint64_t* ticks, uint32_t* sink, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    uint32_t b = seed * 2654435761u + 12345u, c = seed | 1u;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MIX == 0) {
            if (BODY == 64) { R4(R2(ASM8(SUB8))) }
            if (BODY == 512) { R64(ASM8(SUB8)) }
            if (BODY == 4096) { R256(R2(ASM8(SUB8))) }
            if (BODY == 16384) { R256(R4(R2(ASM8(SUB8)))) }
        } else {
            if (BODY == 64) { R4(R2(ASM8(MIX8))) }
            if (BODY == 512) { R64(ASM8(MIX8)) }
            if (BODY == 4096) { R256(R2(ASM8(MIX8))) }
            if (BODY == 16384) { R256(R4(R2(ASM8(MIX8)))) }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63u) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) sink[0] = a0;
}

template <int BODY, int MIX>
void run(int waves_per_simd) {
    const int total = 1 << 21;                        // instructions per wave
    const int iters = total / BODY, blocks = 256, threads = 256 * waves_per_simd;
    uint64_t* d; uint32_t* sink;
    hipMalloc(&d, sizeof(uint64_t) * blocks * 32); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BODY, MIX>), dim3(blocks), dim3(threads), 0, 0, d, sink, 2, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BODY, MIX>), dim3(blocks), dim3(threads), 0, 0, d, sink, iters, 7u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * threads / 64);
    hipMemcpy(h.data(), d, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("%-22s body %6d instructions (%3d KB)  waves/SIMD %d: %5.2f cycles per wave64 instr per SIMD   [%.3f ms wall, %.2f G instr-slots/s/SIMD]\n",
           MIX ? "2 v_sub + 1 v_min3" : "v_sub_u32", BODY, BODY * (MIX ? 16 : 12) / 3 / 1024, waves_per_simd, med / ((double)total * waves_per_simd), ms,
           (double)total * waves_per_simd / (ms * 1e6));
    hipFree(d); hipFree(sink);
}

int main() {
    for (int w : {2, 4}) {                      // (256 threads per SIMD-wave: more than 4 waves per SIMD exceed the block size limit)
        run<64, 0>(w); run<512, 0>(w); run<4096, 0>(w); run<16384, 0>(w);
        run<64, 1>(w); run<512, 1>(w); run<4096, 1>(w); run<16384, 1>(w);
    }
    return 0;
}
