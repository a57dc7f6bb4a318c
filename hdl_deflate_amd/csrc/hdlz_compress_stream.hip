// hdlz_compress_stream.hip -- STARTC for ONE large stream, spread over the whole GPU.
//
// The tile phases (match search, extension, backward parse, token lookup, bit scatter) are the shared device functions of
// hdlz_compress_common.h -- the same source k_compress and k_compress_small are built from.
//
// The reference processes exactly one stream per START (deflate.py:616-633); the batch kernel gives a
// stream to ONE wave, which is right for millions of blocks but leaves a single 16 MiB stream (the LMAX
// limit of the reference, deflate.py:73-76) on one wave for ~50 ms.  The output must stay ONE deflate block,
// bit-identical to the reference, so tiles cannot simply be compressed independently: the greedy parse
// (deflate.py:960,1008) and the bit position chain through the whole stream.  Three parallel passes resolve it:
//   A' k_stream_tails: one lane per tile, the tile's last 32 positions only -> if that piece of the parse transfer
//      function is constant it is the whole tile's function (the parse re-synchronises); else the tile is marked
//   A  k_stream_xfer : marked tiles -> the parse transfer function "entry skip 0..9 -> exit skip" (40 bits)
//      k_stream_skips_chunk<0> / _top / _chunk<1>: two-level scan under function composition -> the entry
//      skip of every tile
//   C  k_stream_tile: every tile, now with its entry skip -> tokens, bit count, Adler partials; the tile's bit
//      string (from local bit 0) goes to its slot of a scratch buffer
//      k_stream_offsets_top / _chunk: two-level prefix sum -> bit offset of every tile, Adler-32 of the stream
//   E  k_stream_place: every tile's bits are funnel-shifted from the slot to their final bit offset in the
//      output (one writer per word); tile 0 adds the header, the last tile EOB, padding, Adler-32, length (R1/R8/R9).
// On ordinary data the match search runs once per tile (pass C; pass A' costs 1/64 of it); scratch is 32 bytes +
// one 2368-byte slot per tile (1.16x the input).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"
#include "hdlz_compress_common.h"

namespace hdlz {

struct StreamArgs {
    const uint8_t* in;          // block b at in + b * in_pitch
    uint32_t n;                 // bytes per block (one block: the stream)
    uint32_t nblocks, tpb;      // blocks, tiles per block (ntiles = nblocks * tpb)
    uint64_t in_pitch, out_pitch;
    int cwindow, maxmatch;
    uint8_t* out;
    uint32_t* out_len;
    uint32_t* status;
    uint32_t ntiles;
    uint64_t* xfer;      // [ntiles] transfer function of tile t (10 nibbles)
    uint32_t* skip;      // [ntiles] entry skip of tile t
    uint32_t* bits;      // [ntiles] token bits of tile t
    uint64_t* bitoff;    // [ntiles] bit offset of tile t in the output stream
    uint32_t* totals;    // [2 * nblocks] s1, s2 of every block
    uint32_t nchunks;    // chunks of 256 tiles
    uint64_t* cxfer;     // [nchunks] composed transfer function of the chunk
    uint32_t* centry;    // [nchunks] entry skip of the chunk
    uint64_t* csum;      // [nchunks][3] bits, sum bytes, weighted sum of the chunk; [..][0] becomes its bit offset
    uint32_t* tmp;       // [ntiles][OUT_WORDS] the tiles' bit strings, each from local bit 0
    uint2* ad;           // [ntiles] Adler partials of the tile: sum of bytes, sum (N - p) x_p mod 65521
};

constexpr bool FULLWIN = false;      // the stream passes keep the per-position window compare (one build per NCH)
constexpr uint64_t XF_IDENT = 0x9876543210ull;                     // transfer function: identity
constexpr uint64_t XF_MARK = 0xFFFFFFFFFFFFFFFFull;                  // "not known yet: needs the full pass"

#define HDLZ_STREAM_PROLOGUE()                                                                         \
    constexpr bool HASH = wide_hash<NCH>();           /* wide windows: the window-independent finder */ \
    __shared__ typename std::conditional<HASH, WaveLdsNoOut, WaveLds>::type lds;                       \
    __shared__ typename std::conditional<HASH, HashLds<NCH>, uint32_t>::type hl;                       \
    uint32_t* const lout = lds_out(lds, hl);          /* HASH: the bit buffer overlays the dead tables */ \
    const uint32_t lane = threadIdx.x;                                                                 \
    fill_luts<NCH>(lds.lut, lane);                                                                     \
    __syncthreads();                                                                                   \
    const uint32_t cw4 = 4u * (uint32_t)a.cwindow;                                                     \
    const uint32_t kmax = (uint32_t)a.maxmatch;                                                        \
    const uint32_t n = a.n;                                                                            \
    uint8_t* lin8 = reinterpret_cast<uint8_t*>(lds.in);                                                \
    const uint8_t* lut8 = reinterpret_cast<const uint8_t*>(lds.lut);                                   \
    uint8_t* out8 = reinterpret_cast<uint8_t*>(lout);                                                  \
    (void)lut8; (void)out8; (void)kmax; (void)cw4;
// tile t -> its block, its first position inside the block, the block's bytes
#define HDLZ_TILE_COORDS(t)                                                                            \
    const uint32_t blk = (t) / a.tpb;                                                                  \
    const uint32_t t0 = ((t) - blk * a.tpb) * (uint32_t)TILE;                                          \
    const uint8_t* __restrict__ src = a.in + (size_t)blk * a.in_pitch;                                 \
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);                            \
    const bool aligned16 = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;

// ------------------------------------------------------------------------------------------------ pass A
template <int NCH>
__global__ __launch_bounds__(64) void k_stream_xfer(StreamArgs a) {
    HDLZ_STREAM_PROLOGUE()
    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        if (a.xfer[t] != XF_MARK) continue;                          // k_stream_tails already has this tile's function
        HDLZ_TILE_COORDS(t)

        stage_tile(lin8, src, t0, n, aligned16, mis, lane);
        __syncthreads();
        const uint32_t p_run = t0 + lane * RUN;
        const uint32_t nrem = n - min(p_run, n);
        const uint32_t run_dw = (HALO / 4) + lane * (RUN / 4);       // dword index of the run in lds.in
        uint32_t best[RUN], tok[RUN];
        if constexpr (HASH) match_search_hash<NCH>(lds.in, hl, lane, (uint32_t)a.cwindow, best);   // 2. R3/R4, wide windows
        else match_search<NCH>(lds.in, run_dw, best);                                              // 2. R3/R4
        {
            uint32_t ow[12];
            load_own(lds.in, run_dw, ow);
            make_tokens<NCH, FULLWIN>(lds.in, HALO + lane * RUN, ow, best, cw4, kmax, 4u * min(p_run, 32u * NCH), nrem, tok);   // 3. R5
        }
        pin(tok);
        PHASE_FENCE();
        const uint64_t P = run_transfer(tok);                                                      // 4. greedy parse of the run
        // transfer function of the whole tile: lanes 0..9 each push one entry skip through the 64 runs
        {
            const uint32_t plo = (uint32_t)P, phi = (uint32_t)(P >> 32);
            uint32_t s = lane < 10u ? lane : 0u;
#pragma unroll 4
            for (int l = 0; l < 64; l++) {
                const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)phi, l) << 32) |
                                   (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)plo, l);
                s = (uint32_t)(f >> (4u * s)) & 15u;
            }
            uint64_t T = 0;
#pragma unroll
            for (int k = 0; k < 10; k++) T |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)s, k) << (4 * k);
            if (lane == 0) a.xfer[t] = T;
        }
    }
}

// ---- scans over the tiles: two levels (chunks of 256 tiles = one wave x 4 tiles per lane, then one workgroup over
// the <= 4096 chunks), so that a 2 GiB stream's 1M tiles never sit in one serial loop
__device__ __forceinline__ uint64_t xf_compose(uint64_t first, uint64_t then) {
    uint64_t r = 0;
#pragma unroll
    for (int s = 0; s < 10; s++) r |= ((then >> (4u * ((uint32_t)(first >> (4 * s)) & 15u))) & 15ull) << (4 * s);
    return r;
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int ofs) {
    return ((uint64_t)(uint32_t)__shfl_up((int)(v >> 32), ofs, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)v, ofs, 64);
}
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int ofs) {
    return ((uint64_t)(uint32_t)__shfl_xor((int)(v >> 32), ofs, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, ofs, 64);
}
constexpr uint32_t CHUNK_TILES = 256;

// APPLY = 0: chunk c -> its composed function (cxfer[c]);  APPLY = 1: with the chunk's entry skip known, every tile's
template <int APPLY>
__global__ __launch_bounds__(64) void k_stream_skips_chunk(StreamArgs a) {
    const uint32_t lane = threadIdx.x, c = blockIdx.x;
    const uint32_t b = c * CHUNK_TILES + lane * 4u;
    uint64_t T[4];
#pragma unroll
    for (int k = 0; k < 4; k++) T[k] = (b + k < a.ntiles) ? a.xfer[b + k] : XF_IDENT;
    uint64_t F = xf_compose(xf_compose(xf_compose(T[0], T[1]), T[2]), T[3]);
    const uint64_t own = F;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const uint64_t o = shfl_up64(F, ofs);
        if (lane >= (uint32_t)ofs) F = xf_compose(o, F);
    }
    if (!APPLY) {
        if (lane == 63) a.cxfer[c] = F;
        return;
    }
    (void)own;
    uint64_t ex = shfl_up64(F, 1);                 // function of the tiles before this lane's four
    if (lane == 0) ex = XF_IDENT;
    uint32_t s = (uint32_t)(ex >> (4u * a.centry[c])) & 15u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (b + k < a.ntiles) a.skip[b + k] = s;
        s = (uint32_t)(T[k] >> (4u * s)) & 15u;
    }
}

// one workgroup: entry skip of every chunk
__global__ __launch_bounds__(64) void k_stream_skips_top(StreamArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t K = (a.nchunks + 63u) / 64u;
    const uint32_t b = lane * K, e = min(b + K, a.nchunks);
    uint64_t F = XF_IDENT;
    for (uint32_t c = b; c < e; c++) F = xf_compose(F, a.cxfer[c]);
    uint32_t s = 0, mine = 0;
    const uint32_t flo = (uint32_t)F, fhi = (uint32_t)(F >> 32);
    for (int l = 0; l < 64; l++) {
        if ((int)lane == l) mine = s;
        const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)fhi, l) << 32) |
                           (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)flo, l);
        s = (uint32_t)(f >> (4u * s)) & 15u;
    }
    s = mine;
    for (uint32_t c = b; c < e; c++) {
        a.centry[c] = s;
        s = (uint32_t)(a.cxfer[c] >> (4u * s)) & 15u;
    }
}

// one workgroup: bit offset of every chunk (exclusive scan of the chunk bit sums), Adler-32 of the stream
__global__ __launch_bounds__(64) void k_stream_offsets_top(StreamArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t K = (a.nchunks + 63u) / 64u;
    const uint32_t b = lane * K, e = min(b + K, a.nchunks);
    uint64_t sb = 0, sa = 0, sw = 0;
    for (uint32_t c = b; c < e; c++) { sb += a.csum[3 * c]; sa += a.csum[3 * c + 1]; sw += a.csum[3 * c + 2]; }
    uint64_t incl = sb;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const uint64_t o = shfl_up64(incl, ofs);
        if (lane >= (uint32_t)ofs) incl += o;
    }
    uint64_t off = 19u + incl - sb;                                 // R1: 78 9C + bits 1,1,0
    for (uint32_t c = b; c < e; c++) { const uint64_t v = a.csum[3 * c]; a.csum[3 * c] = off; off += v; }
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) { sa += shfl_xor64(sa, ofs); sw += shfl_xor64(sw, ofs); }
    if (lane == 0) {
        a.totals[0] = (uint32_t)((sa + 1u) % ADLER_MOD);
        a.totals[1] = (uint32_t)((sw + a.n % ADLER_MOD) % ADLER_MOD);
    }
}

// APPLY = 0: chunk c -> its sums (bits, Adler partials) in csum[c];  APPLY = 1: bit offset of each of its tiles
template <int APPLY>
__global__ __launch_bounds__(64) void k_stream_offsets_chunk(StreamArgs a) {
    const uint32_t lane = threadIdx.x, c = blockIdx.x;
    const uint32_t b = c * CHUNK_TILES + lane * 4u;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (b + k < a.ntiles) ? a.bits[b + k] : 0u;
    const uint32_t own = v[0] + v[1] + v[2] + v[3];                  // <= 4 * 2048 * 24 bits
    if (!APPLY) {
        uint64_t sb = own, sa = 0, sw = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (b + k < a.ntiles) { const uint2 p = a.ad[b + k]; sa += p.x; sw += p.y; }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) { sb += shfl_xor64(sb, ofs); sa += shfl_xor64(sa, ofs); sw += shfl_xor64(sw, ofs); }
        if (lane == 0) { a.csum[3 * c] = sb; a.csum[3 * c + 1] = sa; a.csum[3 * c + 2] = sw; }
        return;
    }
    uint32_t incl = own;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const uint32_t o = __shfl_up(incl, ofs, 64);
        if (lane >= (uint32_t)ofs) incl += o;
    }
    uint64_t off = a.csum[3 * c] + (incl - own);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (b + k < a.ntiles) a.bitoff[b + k] = off;
        off += v[k];
    }
}

// ---- streams of at most one chunk (256 tiles = 512 KiB): the three scan kernels of each kind collapse into one wave
// (every launch costs ~4.5 us, and ten of them were the 45 us floor of the whole path)
__global__ __launch_bounds__(64) void k_stream_skips_one(StreamArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t b = lane * 4u;
    uint64_t T[4];
#pragma unroll
    for (int k = 0; k < 4; k++) T[k] = (b + k < a.ntiles) ? a.xfer[b + k] : XF_IDENT;
    uint64_t F = xf_compose(xf_compose(xf_compose(T[0], T[1]), T[2]), T[3]);
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const uint64_t o = shfl_up64(F, ofs);
        if (lane >= (uint32_t)ofs) F = xf_compose(o, F);
    }
    uint64_t ex = shfl_up64(F, 1);
    if (lane == 0) ex = XF_IDENT;
    uint32_t s = (uint32_t)ex & 15u;                                 // the stream starts with entry skip 0
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (b + k < a.ntiles) a.skip[b + k] = s;
        s = (uint32_t)(T[k] >> (4u * s)) & 15u;
    }
}

__global__ __launch_bounds__(64) void k_stream_offsets_one(StreamArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t b = lane * 4u;
    uint32_t v[4];
    uint64_t sa = 0, sw = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = 0;
        if (b + k < a.ntiles) { v[k] = a.bits[b + k]; const uint2 p = a.ad[b + k]; sa += p.x; sw += p.y; }
    }
    const uint32_t own = v[0] + v[1] + v[2] + v[3];
    uint32_t incl = own;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const uint32_t o = __shfl_up(incl, ofs, 64);
        if (lane >= (uint32_t)ofs) incl += o;
    }
    uint64_t off = 19u + (incl - own);                               // R1: 78 9C + bits 1,1,0
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (b + k < a.ntiles) a.bitoff[b + k] = off;
        off += v[k];
    }
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) { sa += shfl_xor64(sa, ofs); sw += shfl_xor64(sw, ofs); }
    if (lane == 0) {
        a.totals[0] = (uint32_t)((sa + 1u) % ADLER_MOD);
        a.totals[1] = (uint32_t)((sw + a.n % ADLER_MOD) % ADLER_MOD);
    }
}

// several blocks: one wave per block walks its tiles 64 at a time (bit offsets restart at 19 in every block)
__global__ __launch_bounds__(64) void k_stream_offsets_blocks(StreamArgs a) {
    const uint32_t lane = threadIdx.x, b = blockIdx.x;
    uint64_t off = 19u, sa = 0, sw = 0;                              // R1: 78 9C + bits 1,1,0
    for (uint32_t base = 0; base < a.tpb; base += 64u) {
        const uint32_t lt = base + lane, t = b * a.tpb + lt;
        uint32_t v = 0;
        if (lt < a.tpb) { v = a.bits[t]; const uint2 p = a.ad[t]; sa += p.x; sw += p.y; }
        uint32_t incl = v;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) {
            const uint32_t o = __shfl_up(incl, ofs, 64);
            if (lane >= (uint32_t)ofs) incl += o;
        }
        if (lt < a.tpb) a.bitoff[t] = off + (incl - v);
        off += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) { sa += shfl_xor64(sa, ofs); sw += shfl_xor64(sw, ofs); }
    if (lane == 0) {
        a.totals[2 * b] = (uint32_t)((sa + 1u) % ADLER_MOD);
        a.totals[2 * b + 1] = (uint32_t)((sw + a.n % ADLER_MOD) % ADLER_MOD);
    }
}

// ------------------------------------------------------------------------------------------------ pass C
template <int NCH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NCH == 1 ? HDLZ_W1 : wide_hash<NCH>() ? HDLZ_WH : 4, NCH == 1 ? HDLZ_W1 : wide_hash<NCH>() ? HDLZ_WH : 4))) void k_stream_tile(StreamArgs a) {
    HDLZ_STREAM_PROLOGUE()
    for (uint32_t t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        HDLZ_TILE_COORDS(t)
        uint32_t skip_in = a.skip[t];
        const uint32_t base_bits = 0;                                // the tile's bits are built from local bit 0

        stage_tile(lin8, src, t0, n, aligned16, mis, lane);
        if constexpr (!HASH) for (uint32_t w = lane; w < OUT_WORDS; w += 64) lout[w] = 0u;
        __syncthreads();
        const uint32_t p_run = t0 + lane * RUN;
        const uint32_t nrem = n - min(p_run, n);
        const uint32_t run_dw = (HALO / 4) + lane * (RUN / 4);       // dword index of the run in lds.in
        uint32_t best[RUN], tok[RUN], code[RUN];
        uint32_t sa, sc;                                             // Adler partials of the run
        if constexpr (HASH) {
            match_search_hash<NCH>(lds.in, hl, lane, (uint32_t)a.cwindow, best);                   // 2. R3/R4, wide windows
            for (uint32_t w = lane; w < OUT_WORDS; w += 64) lout[w] = 0u;                          // (the bit buffer: in the finder's dead tables)
        } else {
            __builtin_amdgcn_s_setprio(0);                       // (the search at the lowest priority, every other phase above it: hdlz_compress.hip)
            match_search<NCH>(lds.in, run_dw, best);                                               // 2. R3/R4
            __builtin_amdgcn_s_setprio(1);
        }
        {
            uint32_t ow[12];
            load_own(lds.in, run_dw, ow);
            adler_run(ow, sa, sc);
            make_tokens<NCH, FULLWIN>(lds.in, HALO + lane * RUN, ow, best, cw4, kmax, 4u * min(p_run, 32u * NCH), nrem, tok);   // 3. R5
        }
        pin(tok);
        PHASE_FENCE();
        const uint64_t P = run_transfer(tok);                                                      // 4. greedy parse of the run
        uint32_t myskip = chain_skips(P, lane, skip_in);
        pin(tok); asm volatile("" : "+v"(myskip));
        PHASE_FENCE();
        uint32_t lane_bits = token_codes<NCH, false>(lut8, tok, myskip, 0u, code);                 // 5. R6/R7
        pin(code);
        PHASE_FENCE();
        uint32_t incl = wave_scan_incl(lane_bits, lane);
        const uint32_t tile_bits_all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        // positions >= N of the last tile were parsed as one 8-bit literal each (see hdlz_compress.hip); their
        // bits sit behind the real end and are cut off by k_stream_place, which only takes bits[t] bits
        const bool last = (t0 + (uint32_t)TILE >= n);
        const uint32_t ninv = last ? t0 + (uint32_t)TILE - n : 0u;
        const uint32_t tile_bits = tile_bits_all - 8u * ninv;
        {
            // bits and Adler partials of the tile
            uint32_t wsum = ((nrem % ADLER_MOD) * sa) % ADLER_MOD + ADLER_MOD * 8u - (sc % ADLER_MOD);   // < 10 * 65521
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) {
                sa += __shfl_xor(sa, ofs, 64);
                wsum += __shfl_xor(wsum, ofs, 64);
            }
            // (per-tile partials: 256 same-address atomics per chunk sum serialised in L2 and doubled this pass)
            if (lane == 0) { a.bits[t] = tile_bits; a.ad[t] = make_uint2(sa, wsum % ADLER_MOD); }
        }
        pin(code); asm volatile("" : "+v"(incl), "+v"(lane_bits));
        PHASE_FENCE();
        scatter_codes(out8, code, base_bits + incl - lane_bits);                               // bit writer

        __syncthreads();
        // the tile's bit string goes to its slot of the scratch buffer; k_stream_place shifts it to its final position
        {
            uint32_t* __restrict__ slot = a.tmp + (size_t)t * OUT_WORDS;
            const uint32_t nw = (tile_bits_all + 31u) >> 5;
            for (uint32_t w = lane; w < nw; w += 64) slot[w] = lout[w];
        }
    }
}

// ------------------------------------------------------------------------------------------------ pass E
// one wave per tile: its bits[t] bits move from the scratch slot to bit offset bitoff[t] of the output (a funnel
// shift per word).  Every output word has exactly ONE writer, so there are no atomics and no pre-zeroing (two
// same-word atomicOr per tile cost 0.18 of 0.26 ms at 256 MiB): the word in which tile t starts also holds the last
// bits of tile t-1 -- tile t fetches them from its predecessor's slot and writes the whole word; tile t does not write
// the partial word it ends in (tile t+1 owns it), except the last tile, which goes on with EOB (7 zero bits), the
// zero padding and the Adler-32 (R8) and sets the length and the status (R9).  Tile 0 starts with the header (R1).
__device__ __forceinline__ uint32_t slot_bits(const uint32_t* __restrict__ slot, uint32_t pos, uint32_t cnt) {
    // cnt (1..31) bits of a slot from bit `pos` on
    const uint32_t k = pos >> 5, sh = pos & 31u;
    uint64_t v = slot[k];
    if (sh + cnt > 32u) v |= (uint64_t)slot[k + 1u] << 32;
    return (uint32_t)(v >> sh) & ((1u << cnt) - 1u);
}

__global__ __launch_bounds__(256) void k_stream_place(StreamArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    constexpr int NW = (OUT_WORDS + 63) / 64;                       // local words per lane
    for (uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6); t < a.ntiles; t += gridDim.x * 4u) {
        const uint32_t blk = t / a.tpb, lt = t - blk * a.tpb;
        uint32_t* __restrict__ outw = reinterpret_cast<uint32_t*>(a.out + (size_t)blk * a.out_pitch);
        const uint32_t* __restrict__ slot = a.tmp + (size_t)t * OUT_WORDS;
        const uint64_t B = a.bitoff[t];
        const uint32_t nb = a.bits[t];
        const uint32_t s = (uint32_t)B & 31u;
        const uint64_t w0 = B >> 5;
        const bool lastt = (lt == a.tpb - 1u);
        const uint32_t nd_all = (s + nb + 31u) >> 5;                 // destination words touched
        const uint32_t nd = lastt ? nd_all - 1u : (s + nb) >> 5;     // ... and stored here: not the word the tile ends in
                                                                     // (the next tile owns it; the last tile adds the trailer to it below)
        const uint32_t fullw = nb >> 5, rb = nb & 31u;               // whole local words, bits in the partial one
        // the bits below this tile in its first word: header (tile 0) or the tail of the previous tile
        uint32_t below = 0;
        if (s != 0u) {
            if (lt == 0u) below = 0x78u | (0x9Cu << 8) | (0x3u << 16);          // s == 19
            else below = slot_bits(slot - OUT_WORDS, a.bits[t - 1u] - s, s);
        }
        // all loads first (the kernel is a copy: what counts is bytes in flight), local words cut to the tile's nb bits
        uint32_t h[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const uint32_t k = lane + 64u * (uint32_t)i;
            const uint32_t raw = slot[min(k, fullw)];                 // unconditional (in bounds): no branch, no wait between loads
            const uint32_t m = k < fullw ? 0xFFFFFFFFu : (k == fullw ? ((1u << rb) - 1u) : 0u);
            h[i] = raw & m;
        }
        uint32_t tailword = 0;                                       // the last tile's final (partial) token word
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const uint32_t j = lane + 64u * (uint32_t)i;             // destination word w0 + j = local words j-1, j
            uint32_t lo = __shfl_up(h[i], 1, 64);
            if (lane == 0u) lo = i ? (uint32_t)__builtin_amdgcn_readlane((int)h[i ? i - 1 : 0], 63) : 0u;
            uint32_t v = s ? ((h[i] << s) | (lo >> (32u - s))) : h[i];
            if (j == 0u) v |= below;
            if (j < nd) outw[w0 + j] = v;
            // (uniform) pick up the word that contains the end of the last tile
            const uint32_t je = nd_all - 1u;
            if (lastt && (je >> 6) == (uint32_t)i) tailword = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(je & 63u));
        }
        if (lastt && lane == 0u) {
            const uint64_t E = B + nb;                               // first bit behind the last token
            const uint64_t nbytes = (E + 7u + 7u) >> 3;              // EOB = 7 zero bits, then pad to a byte
            const uint64_t total = nbytes + 4u;
            const uint32_t s1 = a.totals[2u * blk], s2 = a.totals[2u * blk + 1u];
            const uint32_t tr[4] = {s2 >> 8, s2 & 255u, s1 >> 8, s1 & 255u};
            // words from the one holding bit E-1 (already stored above, but without the trailer) to the last one
            const uint64_t wa = (E - 1u) >> 5, wb = (total - 1u) >> 2;
            for (uint64_t w = wa; w <= wb; w++) {
                uint32_t v = (w == wa) ? tailword : 0u;
                for (uint32_t k = 0; k < 4u; k++) {
                    const uint64_t bpos = nbytes + k;
                    if ((bpos >> 2) == w) v |= tr[k] << (8u * (uint32_t)(bpos & 3u));
                }
                outw[w] = v;
            }
            a.out_len[blk] = (uint32_t)total;
            a.status[blk] = HDLZ_OK;
        }
    }
}

// ------------------------------------------------------------------------------------------------ pass A, cheap form
// The exit skip of a tile is almost always decided by its last few tokens: the greedy parse re-synchronises at every
// position that is a token start whatever the entry skip was.  k_stream_tails runs the tile phases on the LAST 32
// positions of 64 tiles at once (one lane per tile, the lane's bytes and its CWINDOW of history in a private LDS
// region) and gets each tile's transfer function over those 32 positions.  If it is constant, that constant IS the
// tile's exit skip for every entry skip, i.e. the whole tile's transfer function; only tiles where it is not
// (back-to-back maximal matches across the whole window) are marked and go through the full k_stream_xfer.
template <int NCH> struct TailLds {
    static constexpr uint32_t HALO_DW = 8u * NCH;                    // CWINDOW bytes of history, in dwords
    static constexpr uint32_t STRIDE_DW = HALO_DW + 20u;             // + 32 own bytes + 16 look-ahead + 32 bytes of padding / bank spread
    static constexpr uint32_t GMASK = NCH == 1 ? 0x7FFu : 0x3FFFu;      // (NCH == 1: 64 * 28 + 8 = 1800 dwords, padded to the mask's span)
    static constexpr uint32_t IN_DW = 64u * STRIDE_DW + 8u;
    uint32_t in[NCH == 1 ? (IN_DW > GMASK + 3u ? IN_DW : GMASK + 3u) : IN_DW];
    static_assert(NCH != 1 || IN_DW <= GMASK + 1u, "make_tokens: every real gather address below the mask");
};

template <int NCH>
__global__ __launch_bounds__(64) void k_stream_tails(StreamArgs a) {
    __shared__ TailLds<NCH> lds;
    const uint32_t lane = threadIdx.x;
    const uint32_t cw4 = 4u * (uint32_t)a.cwindow;
    const uint32_t kmax = (uint32_t)a.maxmatch;
    const uint32_t n = a.n;
    typedef uint32_t __attribute__((aligned(1))) u32u;
    const uint32_t t = blockIdx.x * 64u + lane;                      // this lane's tile
    const uint32_t blk = min(t, a.ntiles - 1u) / a.tpb, lt = min(t, a.ntiles - 1u) - blk * a.tpb;
    const bool mine = t < a.ntiles && lt + 1u < a.tpb;               // (the last tile of a block has no successor)
    const uint32_t tc = mine ? lt : 0u;                              // idle lanes compute on a tile 0 and drop the result
    const uint8_t* __restrict__ src = a.in + (size_t)blk * a.in_pitch;
    const uint32_t p_run = (tc + 1u) * (uint32_t)TILE - (uint32_t)RUN;      // first of the tile's last 32 positions
    const uint32_t run_dw = TailLds<NCH>::HALO_DW + lane * TailLds<NCH>::STRIDE_DW;
    const uint32_t lds_run = 4u * run_dw;
    const uint32_t nrem = n - min(p_run, n);
    // stage: history, the 32 positions, 16 bytes of look-ahead (zero beyond N)
    {
        const uint32_t first = p_run - 4u * TailLds<NCH>::HALO_DW;
        for (uint32_t k = 0; k < TailLds<NCH>::HALO_DW + 12u; k++) {
            const uint32_t pos = first + 4u * k;
            uint32_t v = 0;
            if (pos + 4u <= n) v = *reinterpret_cast<const u32u*>(src + pos);
            else for (uint32_t b = 0; b < 4u; b++) if (pos + b < n) v |= (uint32_t)src[pos + b] << (8u * b);
            lds.in[run_dw - TailLds<NCH>::HALO_DW + k] = v;
        }
    }
    __syncthreads();
    uint32_t best[RUN], tok[RUN];
    match_search<NCH>(lds.in, run_dw, best);                                                       // 2. R3/R4
    {
        uint32_t ow[12];
        load_own(lds.in, run_dw, ow);
        make_tokens<NCH, FULLWIN, false, TailLds<NCH>::GMASK>(lds.in, lds_run, ow, best, cw4, kmax, 4u * min(p_run, 32u * NCH), nrem, tok);   // 3. R5
    }
    pin(tok);
    PHASE_FENCE();
    const uint64_t P = run_transfer(tok);                                                          // 4. greedy parse of the run
    // constant?  (nibbles 0..9 of P = exit skip for entry skip 0..9 over these 32 positions)
    {
        const uint64_t P40 = P & 0xFFFFFFFFFFull;
        const uint64_t c = P40 & 15ull;
        if (mine) a.xfer[t] = (P40 == c * 0x1111111111ull) ? P40 : XF_MARK;
        else if (t < a.ntiles) a.xfer[t] = 0;                        // last tile of a block: "-> 0", the next block starts with entry skip 0
    }
}

template __global__ void k_stream_tails<1>(StreamArgs);
template __global__ void k_stream_tails<2>(StreamArgs);
template __global__ void k_stream_tails<8>(StreamArgs);
template __global__ void k_stream_xfer<1>(StreamArgs);
template __global__ void k_stream_xfer<2>(StreamArgs);
template __global__ void k_stream_xfer<8>(StreamArgs);
template __global__ void k_stream_tile<1>(StreamArgs);
template __global__ void k_stream_tile<2>(StreamArgs);
template __global__ void k_stream_tile<8>(StreamArgs);

size_t stream_work_bytes(uint32_t n, uint32_t nblocks) {
    const size_t tpb = ((size_t)n + TILE - 1) / TILE;
    const size_t nt = tpb * nblocks;
    const size_t nc = (nt + CHUNK_TILES - 1) / CHUNK_TILES;
    return nt * 32 + nc * 40 + 64 + (size_t)nblocks * 8 + nt * (size_t)OUT_WORDS * 4;
}

template <int NCH>
static void launch_passes(const StreamArgs& a, hipStream_t stream) {
    const dim3 g(a.ntiles < 16384u ? a.ntiles : 16384u), c(a.nchunks), one(1), b(64);
    hipLaunchKernelGGL(k_stream_tails<NCH>, dim3((a.ntiles + 63u) / 64u), b, 0, stream, a);
    hipLaunchKernelGGL(k_stream_xfer<NCH>, g, b, 0, stream, a);
    if (a.nchunks == 1u) {
        hipLaunchKernelGGL(k_stream_skips_one, one, b, 0, stream, a);
    } else {
        hipLaunchKernelGGL(k_stream_skips_chunk<0>, c, b, 0, stream, a);
        hipLaunchKernelGGL(k_stream_skips_top, one, b, 0, stream, a);
        hipLaunchKernelGGL(k_stream_skips_chunk<1>, c, b, 0, stream, a);
    }
    hipLaunchKernelGGL(k_stream_tile<NCH>, g, b, 0, stream, a);
    if (a.nblocks > 1u) {
        hipLaunchKernelGGL(k_stream_offsets_blocks, dim3(a.nblocks), b, 0, stream, a);
    } else if (a.nchunks == 1u) {
        hipLaunchKernelGGL(k_stream_offsets_one, one, b, 0, stream, a);
    } else {
        hipLaunchKernelGGL(k_stream_offsets_chunk<0>, c, b, 0, stream, a);
        hipLaunchKernelGGL(k_stream_offsets_top, one, b, 0, stream, a);
        hipLaunchKernelGGL(k_stream_offsets_chunk<1>, c, b, 0, stream, a);
    }
    hipLaunchKernelGGL(k_stream_place, dim3((a.ntiles + 3u) / 4u < 4096u ? (a.ntiles + 3u) / 4u : 4096u), dim3(256), 0, stream, a);
}

// nblocks blocks of n bytes each (block b at in + b * in_pitch -> out + b * out_pitch); one block = one stream
hipError_t launch_compress_streams(const uint8_t* in, uint64_t in_pitch, uint32_t n, uint32_t nblocks, int cwindow, int maxmatch,
                                   uint8_t* out, uint64_t out_pitch, uint32_t* out_len, uint32_t* status, void* work,
                                   hipStream_t stream) {
    StreamArgs a;
    a.in = in; a.n = n; a.cwindow = cwindow; a.maxmatch = maxmatch; a.out = out; a.out_len = out_len; a.status = status;
    a.nblocks = nblocks; a.in_pitch = in_pitch; a.out_pitch = out_pitch;
    a.tpb = (uint32_t)(((uint64_t)n + TILE - 1) / TILE);
    a.ntiles = a.tpb * nblocks;
    a.nchunks = (a.ntiles + CHUNK_TILES - 1) / CHUNK_TILES;
    uint8_t* w = static_cast<uint8_t*>(work);
    const size_t nt = a.ntiles, nc = a.nchunks;
    a.xfer = reinterpret_cast<uint64_t*>(w);                 w += nt * 8;     // 8-byte arrays first
    a.bitoff = reinterpret_cast<uint64_t*>(w);               w += nt * 8;
    a.ad = reinterpret_cast<uint2*>(w);                      w += nt * 8;
    a.cxfer = reinterpret_cast<uint64_t*>(w);                w += nc * 8;
    a.csum = reinterpret_cast<uint64_t*>(w);                 w += nc * 24;
    a.skip = reinterpret_cast<uint32_t*>(w);                 w += nt * 4;
    a.bits = reinterpret_cast<uint32_t*>(w);                 w += nt * 4;
    a.centry = reinterpret_cast<uint32_t*>(w);               w += nc * 4;
    a.totals = reinterpret_cast<uint32_t*>(w);               w += 64 + (size_t)nblocks * 8;
    a.tmp = reinterpret_cast<uint32_t*>(w);
    if (cwindow <= 32) launch_passes<1>(a, stream);
    else if (cwindow <= 64) launch_passes<2>(a, stream);
    else launch_passes<8>(a, stream);
    return hipGetLastError();
}

}  // namespace hdlz
