// hdlz_inflate_two.hip -- STARTD for a batch of SMALL independent zlib streams in TWO phases (round 3).
//
// k_inflate_tok (one lane per stream, hdlz_inflate_tok.hip) pays one 64-byte HBM sector per copy whose source lies more than 112
// bytes back: the history of 2.6e5 streams in flight fits no cache (8.9x the algorithmic traffic).  When the whole
// OUTPUT of a stream fits a slice of LDS the history can stay on chip instead:
//   phase A  k_inflate_tok built with HDLZ_TOK_PHASE_A (this file includes hdlz_inflate_tok.hip a second time): the same bit reader,
//            tables, slow path, checks and status codes (/root/reference/deflate.py:635-732 HEADER, :1402-1445 NEXT, :1519-1591
//            INFLATE, :1593-1659 COPY -- the copy's CHECKS; its bytes are phase B's), one lane per stream, but a round appends ONE
//            RECORD -- up to three literals and a match -- to the stream's token list instead of moving bytes: no history is read,
//            a 64-byte ring per lane is enough (seven waves per SIMD instead of four).
//   phase B  k_emit: G lanes per stream replay the records with the stream's output in LDS (128 G bytes), one byte per lane and
//            step (deflate.py:1627-1659: the byte-by-byte copy, overlap included: byte i of a match comes from o - dist + i mod dist),
//            and write the finished output with full 16-byte stores.
// Streams whose token list does not fit (tpitch) come back marked HDLZ_E_TOK_OVERFLOW and go through the one-pass kernel, as do
// -- by the caller's usual second pass -- the streams with dynamic-tree blocks.  Results are identical by construction: phase A IS
// the one-pass kernel's decode; tests/test_gpu_inflate.py runs every inflate case through this mapping too.
#define HDLZ_TOK_PHASE_A 1
#define HDLZ_TOK_RING 64
#define HDLZ_TOK_MOVES 1
#define HDLZ_TOK_NS tokA
#include "hdlz_inflate_tok.hip"

namespace hdlz {
namespace two {

// a record: header (bits 0-1 literals, 2-10 match length or 0, 11-25 distance - 1) + the literal bytes
template <uint32_t G>
__global__ __launch_bounds__(64) void k_emit(InflateArgs a, const uint8_t* __restrict__ tokbuf, uint32_t tpitch,
                                             const uint32_t* __restrict__ tok_len) {
    constexpr uint32_t NG = 64u / G, S = 128u * G, CH = 8u * G, FIFO = 2u * CH;
    __shared__ __attribute__((aligned(16))) uint8_t outb[NG * S];
    __shared__ __attribute__((aligned(16))) uint8_t fifo[NG * FIFO];
    const uint32_t lane = threadIdx.x, g = lane / G, l = lane % G;
    const uint64_t sid = (uint64_t)blockIdx.x * NG + g;
    const bool mine = sid < a.nstreams && a.status[sid] == HDLZ_OK;
    const uint32_t tlen = mine ? tok_len[sid] : 0u;
    const uint32_t olen = mine ? a.out_len[sid] : 0u;
    const uint8_t* __restrict__ tk = tokbuf + sid * tpitch;
    uint8_t* ob = outb + g * S;
    uint8_t* ff = fifo + g * FIFO;
    // the token list goes through a FIFO of two chunks; the chunk after those is on its way in `nxt`
    uint32_t loaded = 0;
    uint64_t nxt = 0;
    if (mine) {
        *reinterpret_cast<uint64_t*>(ff + l * 8u) = *reinterpret_cast<const uint64_t*>(tk + l * 8u);
        loaded = CH;
        if (loaded < tpitch) nxt = *reinterpret_cast<const uint64_t*>(tk + loaded + l * 8u);
    }
    uint32_t tp = 0, o = 0, len = 0, dist = 1, mi = 0;
    bool active = mine && tlen != 0u;
    uint32_t w0 = 0, w1 = 0, w2 = 0;
#define EMIT_FETCH() do {                                                                                   \
        if (tp + 12u > loaded) {                                                                               \
            *reinterpret_cast<uint64_t*>(ff + (loaded & (FIFO - 1u)) + l * 8u) = nxt;                          \
            loaded += CH;                                                                                      \
            if (loaded < tpitch) nxt = *reinterpret_cast<const uint64_t*>(tk + loaded + l * 8u);               \
        }                                                                                                      \
        w0 = *reinterpret_cast<const uint32_t*>(ff + (tp & (FIFO - 4u)));                                      \
        w1 = *reinterpret_cast<const uint32_t*>(ff + ((tp + 4u) & (FIFO - 4u)));                               \
        w2 = *reinterpret_cast<const uint32_t*>(ff + ((tp + 8u) & (FIFO - 4u)));                               \
    } while (0)
    if (active) EMIT_FETCH();
    while (__ballot(active) != 0ull) {
        if (active) {
            if (mi >= len) {                                   // the next record (its words were requested a step ago)
                const uint32_t hdr = __builtin_amdgcn_alignbyte(w1, w0, tp), lits = __builtin_amdgcn_alignbyte(w2, w1, tp);
                const uint32_t nl = hdr & 3u;
                len = (hdr >> 2) & 511u;
                dist = ((hdr >> 11) & 32767u) + 1u;
                tp += 4u + nl;
                if (l < nl) ob[o + l] = (uint8_t)(lits >> (8u * l));
                o += nl;
                mi = 0;
                if (tp < tlen) EMIT_FETCH();
            }
            const uint32_t i = mi + l;
            if (i < len) {
                uint32_t r = i;
                if (i >= dist) {                               // overlapping copy: the pattern of `dist` bytes repeats
                    const uint32_t q = (uint32_t)((float)i * __builtin_amdgcn_rcpf((float)dist));
                    r = i - q * dist;
                    if ((int32_t)r < 0) r += dist;
                    if (r >= dist) r -= dist;
                }
                ob[o + i] = ob[o - dist + r];
            }
            mi += G;
            if (mi >= len) {
                o += len;
                len = 0; mi = 0;
                active = tp < tlen;
            }
        }
    }
#undef EMIT_FETCH
    // the finished output: full 16-byte stores, then the tail
    if (mine) {
        uint8_t* out = a.out + sid * a.out_pitch;
        uint32_t p = l * 16u;
        for (; p + 16u <= olen; p += G * 16u) {
            const tok::u32x4 v = *reinterpret_cast<const tok::u32x4*>(ob + p);
            // (d_out / out_pitch are 4-byte aligned, all gfx950 asks of a dwordx4 store; s_nop: see TOK_FLUSH)
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(out + p), "v"(v) : "memory");
        }
        if (p < olen) for (uint32_t q = p; q < olen; q++) out[q] = ob[q];
    }
}

}  // namespace two

// eligibility: every stream's output fits the LDS slice of its lane group (out_pitch bounds it), and there are enough streams for a
// lane-per-stream pass.  Scratch: tpitch + 8 bytes per stream.
hipError_t launch_inflate_two(const InflateArgs& a, hipStream_t stream, bool* used) {
    *used = false;
    constexpr uint32_t G = 16u, S = 128u * G;
    if (a.nstreams == 0 || a.nstreams > 0xFFFFFFFFull || a.out_pitch > S) return hipSuccess;
    const uint32_t tpitch = (uint32_t)((a.out_pitch + 64u + 255u) & ~255ull);     // (a multiple of the FIFO chunk and of the flush chunk)
    uint8_t* tokbuf = nullptr;
    uint32_t* ws = nullptr;            // ws[0]: number of streams handed back; tok_len from ws + 64, the list behind it
    if (scratch_alloc(reinterpret_cast<void**>(&tokbuf), (size_t)tpitch * a.nstreams + 256u, stream) != hipSuccess) {
        (void)hipGetLastError();
        return hipSuccess;
    }
    if (scratch_alloc(reinterpret_cast<void**>(&ws), sizeof(uint32_t) * (2u * a.nstreams + 64u), stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFreeAsync(tokbuf, stream);
        return hipSuccess;
    }
    uint32_t* tok_len = ws + 64;
    uint32_t* list = ws + 64 + a.nstreams;
    hipError_t e = zero_words(ws, 1u, stream);
    if (e == hipSuccess) {
        typedef tokA::Lds<false, tokA::CAP_FULL> L;
        const uint64_t per_wg = 64u * L::WAVES;
        hipLaunchKernelGGL((tokA::k_inflate_tok<false, tokA::CAP_FULL>), dim3((unsigned)((a.nstreams + per_wg - 1u) / per_wg)),
                           dim3(64 * L::WAVES), 0, stream, a, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, tokbuf, tpitch, tok_len);
        constexpr uint32_t NG = 64u / G;
        hipLaunchKernelGGL((two::k_emit<G>), dim3((unsigned)((a.nstreams + NG - 1u) / NG)), dim3(64), 0, stream, a,
                           (const uint8_t*)tokbuf, tpitch, (const uint32_t*)tok_len);
        hipLaunchKernelGGL(tokA::k_collect_dyn, dim3((unsigned)((a.nstreams + 255u) / 256u)), dim3(256), 0, stream,
                           (const uint32_t*)a.status, a.nstreams, list, ws, HDLZ_E_TOK_OVERFLOW);
        e = hipGetLastError();
        if (e == hipSuccess) e = launch_inflate_tok_list(a, list, ws, a.nstreams, stream);
    }
    const hipError_t e1 = hipFreeAsync(tokbuf, stream), e2 = hipFreeAsync(ws, stream);
    *used = true;
    return e != hipSuccess ? e : e1 != hipSuccess ? e1 : e2;
}

}  // namespace hdlz
