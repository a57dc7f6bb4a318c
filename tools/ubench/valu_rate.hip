// micro-benchmark: wave64 issue cost of the integer VALU ops k_compress is made of (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2048, ACC = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[ACC], b = seed + threadIdx.x;
    uint64_t q = seed * 0x9876543210ull + threadIdx.x;
#pragma unroll
    for (int i = 0; i < ACC; i++) a[i] = seed * (i + 3) + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ACC; i++) {
            if (OP == 0) a[i] = a[i] - b;                                   // v_sub_u32
            if (OP == 1) { uint32_t t = a[i] < b ? a[i] : b; a[i] = t < (uint32_t)it ? t : (uint32_t)it; }  // v_min3_u32
            if (OP == 2) a[i] = __builtin_amdgcn_alignbyte(a[i], b, 1);     // v_alignbyte
            if (OP == 3) a[i] = __builtin_bitreverse32(a[i]) + 1;           // v_bfrev + add
            if (OP == 4) a[i] = (a[i] << 6) | b;                            // v_lshl_or
            if (OP == 5) { q = (q << 4) | (a[i] & 15); a[i] += (uint32_t)(q >> (a[i] & 28)); }  // 64-bit shifts
            if (OP == 6) a[i] = a[i] > b ? a[i] : (b + i);                  // cmp+cndmask
            if (OP == 7) a[i] = __builtin_amdgcn_sad_u8(a[i], b, a[i]);     // v_sad_u8
            if (OP == 8) a[i] = (uint32_t)__builtin_ctz(a[i] | 0x80000000u) + a[i];  // v_ffbl + add
            if (OP == 9) a[i] = fmaf(__uint_as_float(a[i]), 1.0001f, 0.5f) > 0 ? a[i] + 1 : a[i];  // reference fp32
        }
    }
    uint32_t r = (uint32_t)q;
#pragma unroll
    for (int i = 0; i < ACC; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
int run(const char* name, int ops_per) {
    const int blocks = 256 * 8, threads = 256;        // 8 blocks/CU = 32 waves/CU = 8 waves/SIMD
    uint32_t* d;
    CHECK(hipMalloc(&d, blocks * threads * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 3u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double waveinstr = (double)blocks * (threads / 64) * ITER * ACC * ops_per;
    double per_simd_per_cycle = waveinstr / (ms * 1e-3) / (1024.0 * 2.4e9);
    printf("%-22s %8.3f ms  %6.2f T lane-ops/s  => %.2f cycles per wave64 instr @2.4GHz (assuming %d instr/op)\n",
           name, ms, waveinstr * 64 / (ms * 1e-3) / 1e12, 1.0 / per_simd_per_cycle, ops_per);
    CHECK(hipFree(d));
    return 0;
}
int main() {
    run<0>("v_sub_u32", 1); run<1>("v_min3_u32", 1); run<2>("v_alignbyte_b32", 1); run<3>("v_bfrev+v_add", 2);
    run<4>("v_lshl_or_b32", 1); run<5>("64b shl/shr mix", 5); run<6>("v_cmp+v_cndmask", 2); run<7>("v_sad_u8", 1);
    run<8>("v_ffbl+or+add", 3); run<9>("fp32 fma+cmp+cnd", 3);
    return 0;
}
