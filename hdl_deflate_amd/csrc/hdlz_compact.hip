// hdlz_compact.hip -- gather the variable-length per-block outputs into one contiguous archive.
//
// the step immediately after the path.  The reference has no counterpart: it
// drains its output memory one byte per READ (/root/reference/deflate.py:601).  Here block b's
// out_len[b] bytes at rows + b*pitch are copied to archive + off[b], where off[] is the exclusive scan
// of the lengths (computed by the caller; across GPUs after the RCCL length all-gather, shard.py).
// One wave per block; head/tail bytes singly, the body as aligned dword stores fed by unaligned loads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"

namespace hdlz {

typedef uint32_t __attribute__((aligned(1))) u32_any;

__global__ __launch_bounds__(64) void k_compact(const uint8_t* __restrict__ rows, uint64_t pitch,
                                                const uint32_t* __restrict__ len, const uint64_t* __restrict__ off,
                                                uint64_t nblocks, uint8_t* __restrict__ archive) {
    const uint32_t lane = threadIdx.x;
    for (uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const uint8_t* src = rows + b * pitch;
        uint8_t* dst = archive + off[b];
        const uint32_t n = len[b];
        const uint32_t head = min(n, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));
        if (lane < head) dst[lane] = src[lane];
        const uint32_t body = (n - head) >> 2;                    // whole dwords at an aligned destination
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
        for (uint32_t k = lane; k < body; k += 64u) d32[k] = *reinterpret_cast<const u32_any*>(src + head + 4u * k);
        const uint32_t done = head + 4u * body;
        if (done + lane < n) dst[done + lane] = src[done + lane];
    }
}

// The same gather when the destination is PINNED HOST memory (device-mapped: Engine.inflate_host delivers its rows this way, the
// stores cross PCIe).  Such a copy is bound by the link (~57 GB/s), not by the GPU: a grid that fills every CU with waves parked on
// posted PCIe writes starves whatever runs beside it (measured: the next chunk's inflate kernel 0.63 -> 3.2 ms, profiles/r04_d2h_paths.txt),
// so this form runs FOUR waves per CU and moves 16 bytes per lane and store where source and destination allow it.
__global__ __launch_bounds__(64) void k_compact_host(const uint8_t* __restrict__ rows, uint64_t pitch,
                                                     const uint32_t* __restrict__ len, const uint64_t* __restrict__ off,
                                                     uint64_t nblocks, uint8_t* __restrict__ archive) {
    const uint32_t lane = threadIdx.x;
    for (uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const uint8_t* src = rows + b * pitch;
        uint8_t* dst = archive + off[b];
        const uint32_t n = len[b];
        if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0) {
            const uint32_t body = n >> 4;
            const uint4* s16 = reinterpret_cast<const uint4*>(src);
            uint4* d16 = reinterpret_cast<uint4*>(dst);
            for (uint32_t k = lane; k < body; k += 64u) d16[k] = s16[k];
            const uint32_t done = 16u * body;
            if (done + lane < n) dst[done + lane] = src[done + lane];
        } else {
            const uint32_t head = min(n, (uint32_t)((4u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));
            if (lane < head) dst[lane] = src[lane];
            const uint32_t body = (n - head) >> 2;
            uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
            for (uint32_t k = lane; k < body; k += 64u) d32[k] = *reinterpret_cast<const u32_any*>(src + head + 4u * k);
            const uint32_t done = head + 4u * body;
            if (done + lane < n) dst[done + lane] = src[done + lane];
        }
    }
}

hipError_t launch_compact(const uint8_t* rows, uint64_t pitch, const uint32_t* len, const uint64_t* off,
                          uint64_t nblocks, uint8_t* archive, hipStream_t stream) {
    if (nblocks == 0) return hipSuccess;
    hipPointerAttribute_t at;
    const bool to_host = hipPointerGetAttributes(&at, archive) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();                                   // (an unregistered pointer is an error of the query, not of this call)
    if (to_host) {
        const uint64_t g = nblocks < 1024u ? nblocks : 1024u;
        hipLaunchKernelGGL(k_compact_host, dim3((unsigned)g), dim3(64), 0, stream, rows, pitch, len, off, nblocks, archive);
        return hipGetLastError();
    }
    const uint64_t g = nblocks < 65536u ? nblocks : 65536u;
    hipLaunchKernelGGL(k_compact, dim3((unsigned)g), dim3(64), 0, stream, rows, pitch, len, off, nblocks, archive);
    return hipGetLastError();
}

}  // namespace hdlz
