// micro-benchmark: LDS operation throughput per CU on gfx950 for the access patterns of the hash match finder
// (hdlz_compress_common.h: match_search_hash): random-address gathers, atomics (ds_max_u32, ds_or_b64), unaligned dword reads at
// byte-consecutive lane addresses, byte reads / writes.  Reported: CU cycles per wave64 instruction with W waves per SIMD (the
// LDS is shared by the CU's four SIMDs) -- the issue cost that matters for a kernel that is LDS-instruction bound.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 2048;
constexpr int NT = 1024;

template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint32_t T[NT * 2 + 1024];
    uint8_t* T8 = reinterpret_cast<uint8_t*>(T);
    const uint32_t lane = threadIdx.x;
    for (uint32_t k = lane; k < NT * 2 + 1024; k += 64) T[k] = k * 2654435761u;
    __syncthreads();
    uint32_t x = lane * 2654435761u + seed, acc = 0;
    const uint64_t bit = 1ull << lane;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t h = (x >> 12) & (NT - 1);
            if (MODE == 0) acc += T[h];                                                        // random ds_read_b32
            else if (MODE == 1) __hip_atomic_fetch_max(&T[h], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // random ds_max_u32
            else if (MODE == 2) __hip_atomic_fetch_or(reinterpret_cast<uint64_t*>(T) + h, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // random ds_or_b64
            else if (MODE == 3) { const uint64_t v = reinterpret_cast<uint64_t*>(T)[h]; acc += (uint32_t)v + (uint32_t)(v >> 32); }             // random ds_read_b64
            else if (MODE == 4) reinterpret_cast<uint64_t*>(T)[h] = 0ull;                      // random ds_write_b64
            else if (MODE == 5) { uint32_t v; __builtin_memcpy(&v, T8 + ((it * 8 + u) & 1023) + lane, 4); acc += v; }   // unaligned dword, lanes one byte apart
            else if (MODE == 6) reinterpret_cast<volatile uint8_t*>(T8)[((it * 8 + u) & 1023) + lane] = (uint8_t)x;                 // consecutive ds_write_b8
            else if (MODE == 7) { uint32_t v; __builtin_memcpy(&v, T8 + (x >> 20), 4); acc += v; }      // unaligned dword, random
            else if (MODE == 8) acc += T8[x >> 20];                                            // random ds_read_u8
            else if (MODE == 9) acc += T[(it * 8 + u) * 64 % NT + lane];                       // conflict-free ds_read_b32 (reference)
            else if (MODE == 10) { uint32_t old = __hip_atomic_fetch_max(&T[h], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); acc += old; }   // ds_max_rtn_u32
            else if (MODE == 12) reinterpret_cast<volatile uint16_t*>(T8)[((it * 8 + u) & 511) + lane] = (uint16_t)x;     // consecutive ds_write_b16
            else if (MODE == 13) acc += reinterpret_cast<const uint16_t*>(T8)[x >> 21];                           // random ds_read_u16
            else if (MODE == 14) { const uint32_t a = (x >> 20) & ~3u; acc += T[a >> 2] ^ T[(a >> 2) + 1]; }      // random aligned dword PAIR (ds_read2_b32)
            else if (MODE == 15) reinterpret_cast<volatile uint32_t*>(T)[((it * 8 + u) & 255) + lane] = x;         // consecutive ds_write_b32
            else if (MODE == 11) { uint32_t old = __hip_atomic_exchange(&T[h], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); acc += old; }   // ds_wrxchg_rtn_b32
        }
    }
    out[blockIdx.x * 64 + lane] = acc + T[lane];
}

template <int MODE>
int run(const char* name, int wps) {
    const int blocks = 256 * 4 * wps;
    uint32_t* d;
    CHECK(hipMalloc(&d, blocks * 64 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, 3u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_cu = (double)(4 * wps) * ITER * 8;          // LDS instructions per CU
    printf("%-44s waves/SIMD %d: %8.3f ms  %6.1f CU cycles per wave64 instruction\n", name, wps, ms, ms * 1e-3 * 2.4e9 / per_cu);
    CHECK(hipFree(d));
    return 0;
}
int main() {
    for (int w : {1, 4}) {
        run<9>("ds_read_b32 conflict-free", w);
        run<0>("ds_read_b32 random", w);
        run<1>("ds_max_u32 random", w);
        run<10>("ds_max_rtn_u32 random", w);
        run<11>("ds_wrxchg_rtn_b32 random", w);
        run<2>("ds_or_b64 random", w);
        run<3>("ds_read_b64 random", w);
        run<4>("ds_write_b64 random", w);
        run<5>("unaligned b32, lanes 1 byte apart", w);
        run<7>("unaligned b32, random", w);
        run<6>("ds_write_b8 consecutive", w);
        run<8>("ds_read_u8 random", w);
        run<12>("ds_write_b16 consecutive", w);
        run<15>("ds_write_b32 consecutive", w);
        run<13>("ds_read_u16 random", w);
        run<14>("ds_read2_b32 random pair", w);
    }
    return 0;
}
