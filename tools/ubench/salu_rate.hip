// micro-benchmark: SALU issue rate on gfx950 as a function of waves per SIMD (is a scalar-heavy, wave-uniform
// decoder such as k_inflate_dyn bound by scalar issue?), plus the s_load-free LDS->SGPR round trip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;

// MODE 0: 16 independent s_add per iteration; MODE 1: one dependent chain of 16 (s_add, s_lshr, s_and ...);
// MODE 2: dependent chain with an LDS read + v_readfirstlane every 16 scalar ops
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t seed) {
    __shared__ uint32_t tab[256];
    tab[threadIdx.x] = threadIdx.x * 7u + seed; tab[threadIdx.x + 64] = seed; tab[threadIdx.x + 128] = 3; tab[threadIdx.x + 192] = 1;
    __syncthreads();
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    for (int it = 0; it < ITER; it++) {
        if (MODE == 0) {
            asm volatile(
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 1) {
            asm volatile(
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0xffffff\n"
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0xffffff\n"
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0xffffff\n"
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0xffffff\n"
                : "+s"(s0) : "s"(s1) : "scc");
        } else {
            asm volatile(
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0xffffff\n"
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0xffffff\n"
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0xffffff\n"
                "s_add_u32 %0, %0, 1\n s_lshr_b32 %0, %0, 1\n s_add_u32 %0, %0, %1\n s_and_b32 %0, %0, 0x3fc\n"
                : "+s"(s0) : "s"(s1) : "scc");
            uint32_t v = *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(tab) + s0);
            s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s0 ^ s1 ^ s2 ^ s3;
}

template <int MODE>
int run(const char* name, int wps) {
    const int blocks = 256 * 4 * wps;
    uint32_t* d;
    CHECK(hipMalloc(&d, blocks * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 3u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)blocks * ITER * 16;
    const double per_cu_cycle = instr / (ms * 1e-3) / (256.0 * 2.4e9);
    printf("%-28s waves/SIMD %d: %8.3f ms  %.3f SALU instr/cycle/CU  (%.2f cycles per instr per wave)\n", name, wps, ms,
           per_cu_cycle, (ms * 1e-3 * 2.4e9) / (ITER * 16.0));
    CHECK(hipFree(d));
    return 0;
}
int main() {
    for (int w : {1, 2, 4, 8}) run<0>("independent s_add", w);
    for (int w : {1, 2, 4, 8}) run<1>("dependent scalar chain", w);
    for (int w : {1, 2, 4, 8}) run<2>("chain + LDS->SGPR per 16", w);
    return 0;
}
