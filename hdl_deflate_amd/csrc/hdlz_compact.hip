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

// ---- scan + gather in ONE launch (round 5): rows -> archive at the exclusive scan of the lengths, and the offset index.
// hdlz_compact_batch needs the offsets from the caller -- three small scan launches in front of a copy kernel that moved four bytes per
// lane and store (0.83 ms behind the 4.6 ms compress of BASELINE configs[1]).  Here a workgroup takes a TILE of 256 consecutive rows:
// tiles are handed out by a ticket (so every tile in front of a running one has started: the look-back below cannot dead-lock), the
// tile scans its 256 lengths, publishes its sum as ONE 64-bit word {state, value} -- aggregate first, inclusive prefix once known --
// and finds its base by summing the aggregates of the tiles in front of it up to the nearest published prefix (decoupled look-back,
// 64 tiles per step; a single word carries state and value, so no fence orders anything).  Then it writes the 256 offsets and copies
// its rows with 16-byte stores to 16-byte aligned destinations (unaligned 16-byte loads; head and tail bytes singly).
// (A compress kernel that writes straight into the archive was weighed again and not built: a block's offset needs ALL lengths in
// front of it, so every flush would wait for the slowest of the ~5000 blocks in flight ahead of it -- with one LDS output buffer per
// wave the wave can do nothing else meanwhile, and the VALU pipe this kernel lives on needs its five waves per SIMD issuing.)
constexpr uint32_t AT = 256;                         // rows per tile
constexpr uint64_t ST_AGG = 1ull << 62, ST_PFX = 2ull << 62, ST_MASK = 3ull << 62;

__global__ __launch_bounds__(256) void k_archive(const uint8_t* __restrict__ rows, uint64_t pitch, const uint32_t* __restrict__ len,
                                                 uint64_t nblocks, uint8_t* __restrict__ archive, uint64_t cap, uint64_t* __restrict__ off,
                                                 unsigned long long* __restrict__ desc, uint32_t* __restrict__ ticket) {
    __shared__ uint32_t s_tile;
    __shared__ uint64_t s_wsum[4], s_base;
    __shared__ uint64_t s_off[AT];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0u) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t b = (uint64_t)tile * AT + tid;
    const uint32_t n = b < nblocks ? len[b] : 0u;
    // exclusive scan of the tile's lengths
    uint64_t v = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t t = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    if (lane == 63u) s_wsum[wave] = v;
    __syncthreads();
    const uint64_t w0 = s_wsum[0], w1 = s_wsum[1], w2 = s_wsum[2], w3 = s_wsum[3];
    const uint64_t local = v - n + (wave > 0u ? w0 : 0ull) + (wave > 1u ? w1 : 0ull) + (wave > 2u ? w2 : 0ull);
    const uint64_t tsum = w0 + w1 + w2 + w3;
    if (wave == 0u) {
        uint64_t base = 0;
        if (tile != 0u) {
            if (lane == 0u) __hip_atomic_store(&desc[tile], ST_AGG | tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t t = (int64_t)tile - 1 - (int64_t)lane;        // lane 0 looks at the nearest tile
            for (;;) {
                uint64_t d = ST_PFX;                               // (tiles in front of tile 0: an empty prefix)
                if (t >= 0) {
                    do { d = __hip_atomic_load(&desc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((d & ST_MASK) == 0ull);
                }
                const uint64_t pm = ballot64((d & ST_MASK) == ST_PFX);
                const uint32_t first = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;      // the nearest published prefix among these 64
                uint64_t x = lane <= first ? (d & ~ST_MASK) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
                base += x;
                if (pm) break;
                t -= 64;
            }
        }
        if (lane == 0u) {
            __hip_atomic_store(&desc[tile], ST_PFX | (base + tsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base = base;
            if ((uint64_t)(tile + 1u) * AT >= nblocks) off[nblocks] = base + tsum;      // the last tile: the archive's length
        }
    }
    __syncthreads();
    const uint64_t mine = s_base + local;
    if (b < nblocks) off[b] = mine;
    s_off[tid] = mine;
    __syncthreads();
    // the copy: a wave per row, 64 rows each
    for (uint32_t r = wave * 64u; r < wave * 64u + 64u; r++) {
        const uint64_t rb = (uint64_t)tile * AT + r;
        if (rb >= nblocks) break;
        const uint64_t o0 = s_off[r];
        const uint32_t rn = (uint32_t)((r + 1u < AT ? s_off[r + 1u] : s_base + tsum) - o0);     // (= len[rb])
        if (o0 + rn > cap) continue;                          // (the caller reads the total from off[nblocks] and sees that it did not fit)
        const uint8_t* src = rows + rb * pitch;
        uint8_t* dst = archive + o0;
        const uint32_t head = min(rn, (uint32_t)((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u));
        if (lane < head) dst[lane] = src[lane];
        const uint32_t body = (rn - head) >> 4;
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        typedef v4 __attribute__((aligned(1))) v4u;
        for (uint32_t k = lane; k < body; k += 64u)
            *reinterpret_cast<v4*>(dst + head + 16u * k) = *reinterpret_cast<const v4u*>(src + head + 16u * k);
        const uint32_t done = head + 16u * body;
        if (done + lane < rn) dst[done + lane] = src[done + lane];
    }
}

size_t archive_work_bytes(uint64_t nblocks) {                  // ticket (+ pad), then one 64-bit descriptor per tile
    const uint64_t ntiles = (nblocks + AT - 1u) / AT;
    return ntiles ? (sizeof(uint32_t) * (2u + 2u * (size_t)ntiles) + 255u) & ~(size_t)255u : 0u;
}

hipError_t launch_archive(const uint8_t* rows, uint64_t pitch, const uint32_t* len, uint64_t nblocks, uint8_t* archive, uint64_t cap,
                          uint64_t* off, hipStream_t stream, const Work& w) {
    const uint64_t ntiles = (nblocks + AT - 1u) / AT;
    if (ntiles == 0) return zero_words(reinterpret_cast<uint32_t*>(off), 2u, stream);      // off[0] = 0: an empty archive
    uint32_t* ws = nullptr;                                    // ticket (+ pad), then one 64-bit descriptor per tile
    const size_t words = 2u + 2u * (size_t)ntiles;
    hipError_t e = w.get(sizeof(uint32_t) * words, stream, reinterpret_cast<uint8_t**>(&ws));
    if (e != hipSuccess) return e;
    e = zero_words(ws, (uint32_t)words, stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_archive, dim3((unsigned)ntiles), dim3(256), 0, stream, rows, pitch, len, nblocks, archive, cap, off,
                           reinterpret_cast<unsigned long long*>(ws + 2), ws);
        e = hipGetLastError();
    }
    const hipError_t e2 = w.put(reinterpret_cast<uint8_t*>(ws), stream);
    return e != hipSuccess ? e : e2;
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
