// hdlz_device.h -- shared declarations of the HIP side of libhdlz.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include "../../include/hdlz.h"

namespace hdlz {

struct CompressArgs {
    const uint8_t* in;
    const uint64_t* in_off;   // nullable: then fixed pitch / length
    uint64_t in_pitch;
    uint32_t in_len;
    uint64_t nblocks;
    int cwindow;
    int maxmatch;
    uint8_t* out;
    uint64_t out_pitch;
    uint32_t* out_len;
    uint32_t* status;
};

struct InflateArgs {
    const uint8_t* in;
    const uint64_t* in_off;
    uint64_t in_pitch;
    uint32_t in_len;
    uint64_t nstreams;
    uint32_t flags;
    uint32_t obsize;
    uint8_t* out;
    uint64_t out_pitch;
    uint32_t* out_len;
    uint32_t* status;
};

__host__ __device__ inline uint32_t out_bound(uint32_t n) {
    return 6u + (uint32_t)((9ull * n + 10ull + 7ull) >> 3);
}

// wave64 ballot straight from the compare.  (HIP's __ballot(int) takes the predicate through a 0/1 VGPR: v_cndmask + v_cmp_ne per
// call -- 19 such pairs in a round of k_inflate_tok.)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// LDS-DMA: every active lane's 16 bytes at gptr (any alignment) go to LDS byte address lds_base + 16 * lane, no VGPR in between
// (global_load_lds_dwordx4, M0 = the wave-uniform LDS base; semantics checked by tools/ubench/lds_dma.hip).  The request is a VMEM
// load: it retires in order with the wave's other loads and stores, and only an s_waitcnt vmcnt covers it.
__device__ __forceinline__ void lds_dma16(const uint8_t* gptr, uint32_t lds_base) {
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(gptr), "s"(lds_base) : "memory");
}
// ordering point between the LDS accesses of ONE wave (single-wave workgroups): the LDS executes a wave's instructions in order, so
// all it takes is that the compiler keeps them in order.  (__syncthreads() also waits vmcnt(0) -- for output stores issued a moment ago.)
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }

// compile-time counted loop: body(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

hipError_t launch_compress(const CompressArgs& a, hipStream_t stream);
hipError_t launch_compress_small(const CompressArgs& a, hipStream_t stream, int ncu);
// The scratch of ONE call.  caller == true (the hdlz_*_ws entry points): [base, base + bytes) is the caller's device buffer and
// nothing is allocated -- a request that does not fit FAILS and the call takes a path that needs less (same results).  caller == false
// (the entry points without d_work): requests go to the library's stream-ordered pool (scratch_alloc / hipFreeAsync).
struct Work {
    uint8_t* base;
    size_t bytes;
    bool caller;
    hipError_t get(size_t need, hipStream_t stream, uint8_t** p) const;
    hipError_t put(uint8_t* p, hipStream_t stream) const;
};
size_t inflate_tok_work_bytes(uint64_t nstreams, bool ragged);              // pass 1's ordered list / pass 2's lists (max of the two)
size_t inflate_par_work_bytes(uint32_t in_len, uint64_t nstreams, uint64_t out_pitch, uint32_t flags);    // all streams at once (0: the path does not apply)
size_t archive_work_bytes(uint64_t nblocks);
hipError_t launch_inflate_tok(const InflateArgs& a, hipStream_t stream, const Work& w);
hipError_t launch_inflate_grp(const InflateArgs& a, hipStream_t stream);
hipError_t launch_inflate_tok_dyn(const InflateArgs& a, hipStream_t stream, bool all, const Work& w);
hipError_t launch_inflate_par(const InflateArgs& a, hipStream_t stream, bool* used, const Work& w);
hipError_t launch_inflate_dyn(const InflateArgs& a, hipStream_t stream, bool all, const uint32_t* few_n = nullptr, uint32_t lane_min = 0);
hipError_t launch_inflate_dyn_flagged(const InflateArgs& a, hipStream_t stream);
size_t stream_work_bytes(uint32_t n, uint32_t nblocks);
hipError_t launch_compress_streams(const uint8_t* in, uint64_t in_pitch, uint32_t n, uint32_t nblocks, int cwindow, int maxmatch,
                                   uint8_t* out, uint64_t out_pitch, uint32_t* out_len, uint32_t* status, void* work,
                                   hipStream_t stream);
typedef hdlz_cstate ChunkState;     // resumable compress session (include/hdlz.h)
hipError_t launch_compress_chunk(const uint8_t* in, uint32_t n, uint32_t q_end, int final_, int cwindow, int maxmatch, uint8_t* out,
                                 uint64_t out_cap, void* state, hipStream_t stream);
hipError_t launch_inflate_chunk(const uint8_t* in, uint32_t in_len, int final_, uint32_t flags, uint32_t obsize, uint8_t* out,
                                uint64_t out_cap, uint32_t out_limit, void* state, hipStream_t stream);
// stream-ordered scratch memory from the library's OWN per-device memory pool (release threshold 256 MiB -- with the default
// pool's threshold of 0 every call paid a fresh device allocation: 10..40 ms for the 8 MB of a 1 MiB single-stream inflate; with
// "keep everything" one large single-stream inflate held gigabytes of HBM for the life of the process)
hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t stream);
hipError_t scratch_release();          // give the cached scratch of the current device back (hdlz_release_scratch)
// p[0 .. n) = 0 by a kernel.  (Not hipMemsetAsync: captured into a HIP graph, a memset node on memory that a mem-alloc node of the
// same graph hands out was seen to leave the words unchanged on ROCm 7.2 -- the device-side counters then started from garbage.)
hipError_t zero_words(uint32_t* p, uint32_t n, hipStream_t stream);

hipError_t launch_compact(const uint8_t* rows, uint64_t pitch, const uint32_t* len, const uint64_t* off,
                          uint64_t nblocks, uint8_t* archive, hipStream_t stream);

hipError_t launch_archive(const uint8_t* rows, uint64_t pitch, const uint32_t* len, uint64_t nblocks, uint8_t* archive, uint64_t cap,
                          uint64_t* off, hipStream_t stream, const Work& w);

}  // namespace hdlz
