// hdlz_compress_chunk.hip -- STARTC for a stream that ARRIVES IN PIECES.
//
// The reference compresses while the caller is still WRITE-ing: position di is encoded as soon as ten more bytes are
// known (`di >= isize - 10 and i_mode != IDLE` stalls, /root/reference/deflate.py:768-770) and the output appears in oram
// as it is produced (put / do_flush, :535-567).  The batch kernels need the whole stream.  This kernel is the resumable
// form: ONE deflate block is produced over any number of calls, bit-identical to the one-shot kernels, by carrying
//   * the greedy-parse state (the entry skip, deflate.py:960,1008), * the bit position and the partial output word
//   (put's ob1/doo), * the Adler-32 sums (deflate.py:826-831)
// in a 64-byte device-resident state between the calls.  A call encodes the positions [state.pos, q_end): the caller
// promises q_end <= n - 11 unless `final` (then q_end = n), which is the reference's own stall margin: every position
// below q_end then has its full 10-byte look-ahead and cannot be touched by the tail rules R3/R5, whatever the final
// length turns out to be.  (q_end - pos) is a multiple of 32 for non-final calls, so that a call ends on a lane boundary
// of the wave-tile.  The tile phases are the shared ones of hdlz_compress_common.h.
// One wave: this is the port adapter's path (one stream, one byte per clock on the host side), not a throughput path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"
#include "hdlz_compress_common.h"

namespace hdlz {

struct ChunkArgs {
    const uint8_t* in;      // the stream from its first byte on; bytes [0, n) are valid
    uint32_t n;             // bytes known so far (final: the stream length)
    uint32_t q_end;         // encode positions [state.pos, q_end)
    uint32_t final_;
    int cwindow, maxmatch;
    uint8_t* out;           // the whole output stream, linear
    uint64_t out_cap;
    ChunkState* st;
};

template <int NCH>
__global__ __launch_bounds__(64) void k_compress_chunk(ChunkArgs a) {
    constexpr bool FULLWIN = false;
    __shared__ WaveLds lds;
    const uint32_t lane = threadIdx.x;
    fill_luts<NCH>(lds.lut, lane);
    __syncthreads();
    const uint32_t cw4 = 4u * (uint32_t)a.cwindow;
    const uint32_t kmax = (uint32_t)a.maxmatch;
    uint8_t* lin8 = reinterpret_cast<uint8_t*>(lds.in);
    const uint8_t* lut8 = reinterpret_cast<const uint8_t*>(lds.lut);
    uint8_t* out8 = reinterpret_cast<uint8_t*>(lds.out);
    const uint32_t n = a.n;
    ChunkState st = *a.st;                                   // (uniform load)
    if (st.done || st.status != HDLZ_OK) return;             // a finished or failed session stays as it is
    if (!st.started) {                                       // R1: 78 9C + bits 1,1,0
        st.started = 1; st.base_bits = 19; st.carry_word = 0x78u | (0x9Cu << 8) | (0x3u << 16);
    }
    const bool final_ = a.final_ != 0;
    uint32_t fail = HDLZ_OK;
    if (final_ && n < 5u) fail = HDLZ_E_SHORT_INPUT;         // R0: the reference never starts
    if (a.q_end <= st.pos || a.q_end > n || (final_ ? a.q_end != n : (a.q_end + 11u > n || ((a.q_end - st.pos) & 31u) != 0u)))
        fail = fail ? fail : HDLZ_E_BAD_PARAM;
    if (fail) {
        if (lane == 0) { a.st->status = fail; a.st->out_len = 0; }
        return;
    }
    const uint8_t* __restrict__ src = a.in;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
    const bool aligned16 = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
    uint32_t* __restrict__ outw = reinterpret_cast<uint32_t*>(a.out);
    uint32_t gw = st.out_words, base_bits = st.base_bits, carry_word = st.carry_word, skip_in = st.skip;
    uint32_t ad_a = 0, ad_c = 0;                              // this call's per-lane Adler partials: sum x, sum p * x (mod 65521)
    bool finished = false;

    // (a final call always has positions left: the non-final calls before it stopped at least 11 short of the bytes known then)
    for (uint32_t t0 = st.pos; t0 < a.q_end; t0 += TILE) {
        const uint32_t span = a.q_end - t0;                   // positions of this call from t0 on
        const bool last = final_ && span <= (uint32_t)TILE;   // the stream ends inside this tile
        const uint32_t lq = (!last && span < (uint32_t)TILE) ? span / RUN : 64u;   // non-final cut: lanes >= lq emit nothing
        if ((uint64_t)gw * 4u + (uint64_t)OUT_WORDS * 4u + 8u > a.out_cap) { fail = HDLZ_E_OUT_CAPACITY; break; }
        stage_tile(lin8, src, t0, n, aligned16, mis, lane);
        for (uint32_t w = lane; w < OUT_WORDS; w += 64) lds.out[w] = (w == 0) ? carry_word : 0u;
        __syncthreads();
        const uint32_t p_run = t0 + lane * RUN;
        const uint32_t nrem = n - min(p_run, n);
        const uint32_t run_dw = (HALO / 4) + lane * (RUN / 4);
        uint32_t best[RUN], tok[RUN], code[RUN];
        match_search<NCH>(lds.in, run_dw, best);
        {
            uint32_t ow[12];
            load_own(lds.in, run_dw, ow);
            uint32_t sa, sc;
            adler_run(ow, sa, sc);                            // bytes at p >= n are zero; lanes behind a non-final cut do not count
            if (lane < lq) {
                ad_a = (ad_a + sa) % ADLER_MOD;
                ad_c = (ad_c + (p_run % ADLER_MOD) * sa + sc) % ADLER_MOD;      // sum p * x_p = p_run * sa + sum i * x
            }
            make_tokens<NCH, FULLWIN>(lds.in, HALO + lane * RUN, ow, best, cw4, kmax, 4u * min(p_run, 32u * NCH), nrem, tok);
        }
        pin(tok);
        PHASE_FENCE();
        const uint64_t P = run_transfer(tok);
        uint32_t myskip = chain_skips(P, lane, skip_in);      // skip_in: now the exit skip of lane 63
        if (lq < 64u) skip_in = (uint32_t)__builtin_amdgcn_readlane((int)myskip, (int)lq);   // the call ends in front of lane lq
        uint32_t c0 = lane < lq ? myskip : 64u;               // a lane behind the cut starts no token
        pin(tok); asm volatile("" : "+v"(c0));
        PHASE_FENCE();
        uint32_t lane_bits = token_codes<NCH, false>(lut8, tok, c0, 0u, code);
        pin(code);
        PHASE_FENCE();
        uint32_t incl = wave_scan_incl(lane_bits, lane);
        const uint32_t tile_bits_all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        pin(code); asm volatile("" : "+v"(incl), "+v"(lane_bits));
        PHASE_FENCE();
        scatter_codes(out8, code, base_bits + incl - lane_bits);
        __syncthreads();
        if (!last) {
            const uint32_t end_bits = base_bits + tile_bits_all;
            const uint32_t full = end_bits >> 5;
            for (uint32_t w = lane; w < full; w += 64) outw[gw + w] = lds.out[w];
            carry_word = lds.out[full];
            gw += full;
            base_bits = end_bits & 31u;
        } else {
            // positions >= N of this tile were emitted as one 8-bit literal each (see hdlz_compress.hip): wipe them
            const uint32_t ninv = t0 + TILE - n;
            const uint32_t end_bits = base_bits + tile_bits_all - 8u * ninv;
            {
                const uint32_t ew = end_bits >> 5, rb = end_bits & 31u;
                for (uint32_t w = ew + lane; w < OUT_WORDS; w += 64)
                    lds.out[w] = (w == ew) ? (lds.out[w] & ((1u << rb) - 1u)) : 0u;
            }
            // R8: EOB = 7 zero bits, zero pad to a byte, Adler-32 big-endian (s2 then s1)
            uint32_t sA = ad_a, sC = ad_c;
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) { sA += __shfl_xor(sA, ofs, 64); sC += __shfl_xor(sC, ofs, 64); }
            const uint64_t A = ((uint64_t)st.adler_a + sA) % ADLER_MOD, C = ((uint64_t)st.adler_c + sC) % ADLER_MOD;
            const uint64_t nm = n % ADLER_MOD;
            const uint32_t s1 = (uint32_t)((A + 1u) % ADLER_MOD);
            const uint32_t s2 = (uint32_t)((nm + nm * A + ADLER_MOD - C) % ADLER_MOD);   // N + sum (N - p) x_p
            const uint32_t nbytes = (end_bits + 7u + 7u) >> 3;
            __syncthreads();
            if (lane == 0) {
                out8[nbytes] = (uint8_t)(s2 >> 8);
                out8[nbytes + 1] = (uint8_t)s2;
                out8[nbytes + 2] = (uint8_t)(s1 >> 8);
                out8[nbytes + 3] = (uint8_t)s1;
            }
            __syncthreads();
            const uint32_t total = nbytes + 4u;
            const uint32_t words = (total + 3u) >> 2;
            for (uint32_t w = lane; w < words; w += 64) outw[gw + w] = lds.out[w];
            if (lane == 0) {
                ChunkState o = st;
                o.pos = n; o.done = 1; o.out_len = gw * 4u + total; o.status = HDLZ_OK;      // R9
                o.out_words = gw; o.base_bits = 0; o.carry_word = 0; o.skip = 0;
                *a.st = o;
            }
            finished = true;
            break;
        }
    }
    if (finished) return;
    if (fail) {
        if (lane == 0) { a.st->status = fail; a.st->out_len = 0; }
        return;
    }
    // not the end of the stream: park the state.  The partial word goes out too, so that every COMPLETE byte produced so
    // far is readable (the next call rewrites that word when it is full)
    uint32_t sA = ad_a, sC = ad_c;
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) { sA += __shfl_xor(sA, ofs, 64); sC += __shfl_xor(sC, ofs, 64); }
    if (lane == 0) {
        outw[gw] = carry_word;
        ChunkState o = st;
        o.pos = a.q_end; o.skip = skip_in; o.out_words = gw; o.base_bits = base_bits; o.carry_word = carry_word;
        o.adler_a = (uint32_t)(((uint64_t)st.adler_a + sA) % ADLER_MOD);
        o.adler_c = (uint32_t)(((uint64_t)st.adler_c + sC) % ADLER_MOD);
        o.out_len = gw * 4u + (base_bits >> 3);
        *a.st = o;
    }
}

template __global__ void k_compress_chunk<1>(ChunkArgs);
template __global__ void k_compress_chunk<2>(ChunkArgs);
template __global__ void k_compress_chunk<8>(ChunkArgs);

hipError_t launch_compress_chunk(const uint8_t* in, uint32_t n, uint32_t q_end, int final_, int cwindow, int maxmatch, uint8_t* out,
                                 uint64_t out_cap, void* state, hipStream_t stream) {
    ChunkArgs a{in, n, q_end, (uint32_t)(final_ != 0), cwindow, maxmatch, out, out_cap, static_cast<ChunkState*>(state)};
    if (cwindow <= 32) hipLaunchKernelGGL(k_compress_chunk<1>, dim3(1), dim3(64), 0, stream, a);
    else if (cwindow <= 64) hipLaunchKernelGGL(k_compress_chunk<2>, dim3(1), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(k_compress_chunk<8>, dim3(1), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
