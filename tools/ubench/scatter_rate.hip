// scatter_rate.hip -- what does a VMEM instruction cost when every lane touches its own 2 KiB-strided region (one stream per
// lane, as in k_inflate_tok)?  Loads of 8 / 16 bytes, aligned or not, LDS-DMA, and the flush stores (16 bytes per lane, or the
// same bytes with four adjacent lanes writing one 64-byte line).  Prints ns per wave-instruction per CU and the lane-request rate.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/scatter_rate tools/ubench/scatter_rate.hip && /tmp/scatter_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint64_t __attribute__((aligned(1))) u64u;
struct __attribute__((packed, aligned(1))) u128u { uint32_t a, b, c, d; };

template <int MODE>
__global__ __launch_bounds__(256) void k(uint8_t* buf, uint32_t pitch, int iters, uint32_t* sink) {
    __shared__ uint32_t slot[256 * 4];
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t* mine = buf + (uint64_t)gid * pitch;
    uint32_t acc = 0;
    uint32_t off = (gid * 40u) & (pitch - 1u);
    for (int i = 0; i < iters; i++) {
        off = (off + 328u) & (pitch - 64u);        // a 16-aligned walk through the lane's own region
        if (MODE == 0) { acc += (uint32_t)*reinterpret_cast<const u64u*>(mine + off + 5u); }
        if (MODE == 1) { acc += (uint32_t)*reinterpret_cast<const uint64_t*>(mine + off + 8u); }
        if (MODE == 2) { const uint4 v = *reinterpret_cast<const uint4*>(mine + off); acc += v.x ^ v.w; }
        if (MODE == 3) { const u128u v = *reinterpret_cast<const u128u*>(mine + off + 5u); acc += v.a ^ v.d; }
        if (MODE == 4 || MODE == 5) {
            const uint8_t* g = mine + off + (MODE == 5 ? 5u : 0u);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(&slot[(threadIdx.x >> 6) * 256]));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(base) : "memory");
            if ((i & 7) == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += slot[threadIdx.x * 4]; }
        }
        if (MODE == 6) { *reinterpret_cast<uint4*>(mine + off) = make_uint4(i, gid, acc, off); }
        if (MODE == 7) {                            // four adjacent lanes write one 64-byte line of stream (lane / 4) of this quarter
            uint8_t* tgt = buf + (uint64_t)((gid & ~63u) + (lane >> 2) + 16u * (i & 3)) * pitch + off + 16u * (lane & 3u);
            *reinterpret_cast<uint4*>(tgt) = make_uint4(i, gid, acc, off);
        }
        if (MODE == 8) {                            // the flush as it is: 4 x 16 bytes per lane
#pragma unroll
            for (int q = 0; q < 4; q++) *reinterpret_cast<uint4*>(mine + off + 16 * q) = make_uint4(i, gid, acc, q);
        }
        if (MODE == 9) {                            // the same 4 KiB per wave as four full-line instructions
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint8_t* tgt = buf + (uint64_t)((gid & ~63u) + (lane >> 2) + 16u * q) * pitch + off + 16u * (lane & 3u);
                *reinterpret_cast<uint4*>(tgt) = make_uint4(i, gid, acc, q);
            }
        }
        if (MODE == 10) { acc += *reinterpret_cast<const u32u*>(mine + off + 5u); }
        if (MODE == 11) { acc += *reinterpret_cast<const uint32_t*>(mine + off + 4u); }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
void run(const char* name, uint8_t* buf, uint32_t pitch, uint32_t lanes, uint32_t* sink, int per_iter) {
    const int iters = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(lanes / 256), dim3(256), 0, 0, buf, pitch, 8, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(lanes / 256), dim3(256), 0, 0, buf, pitch, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)(lanes / 64) * iters * per_iter;      // wave-instructions
    const double per_cu_ns = ms * 1e6 / (winstr / 256.0);
    printf("%-44s %8.3f ms  %7.1f ns / wave-instr / CU  %7.2f G lane-requests/s\n", name, ms, per_cu_ns, winstr * 64 / ms / 1e6);
}

int main(int argc, char** argv) {
    // (round 5: optional argument = the stride between the lanes' regions; 262144 lanes x 2048 B = 512 MiB is the working set of
    //  k_inflate_tok, x 512 B = 128 MiB fits the 256 MiB Infinity Cache, x 128 B = 32 MiB the eight L2s together)
    const uint32_t pitch = argc > 1 ? (uint32_t)atoi(argv[1]) : 2048, lanes = 262144;        // 16 waves per CU x 256 CUs
    printf("== region stride %u bytes: %u MiB touched\n", pitch, (unsigned)(((uint64_t)lanes * pitch) >> 20));
    uint8_t* buf; uint32_t* sink;
    hipMalloc(&buf, (size_t)lanes * pitch + 4096);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, (size_t)lanes * pitch + 4096);
    run<10>("load dword, unaligned", buf, pitch, lanes, sink, 1);
    run<11>("load dword, aligned", buf, pitch, lanes, sink, 1);
    run<0>("load dwordx2, unaligned (+5)", buf, pitch, lanes, sink, 1);
    run<1>("load dwordx2, aligned 8", buf, pitch, lanes, sink, 1);
    run<2>("load dwordx4, aligned 16", buf, pitch, lanes, sink, 1);
    run<3>("load dwordx4, unaligned (+5)", buf, pitch, lanes, sink, 1);
    run<4>("LDS-DMA dwordx4, aligned 16", buf, pitch, lanes, sink, 1);
    run<5>("LDS-DMA dwordx4, unaligned (+5)", buf, pitch, lanes, sink, 1);
    run<6>("store dwordx4, one line per lane", buf, pitch, lanes, sink, 1);
    run<7>("store dwordx4, four lanes per 64-byte line", buf, pitch, lanes, sink, 1);
    run<8>("flush: 4 x store dwordx4 per lane (64 B)", buf, pitch, lanes, sink, 4);
    run<9>("flush: 4 x full-line stores (same bytes)", buf, pitch, lanes, sink, 4);
    return 0;
}
