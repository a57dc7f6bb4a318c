// hdlz_compress_small.hip -- STARTC for batches of SMALL blocks (N <= 1024), uniform or ragged.
//
// The general kernel (hdlz_compress.hip) gives every block a whole 2048-position wave-tile, so a 256-byte
// block keeps 8 of 64 lanes busy (38 GB/s measured).  Sub-KiB inputs are the reference's own scale
// (IBSIZE = 512 in the FAST build, deflate.py:64-68; the test bench compresses ~500 bytes,
// test_deflate.py:329), so this variant PACKS G = floor(64 / ceil(N/32)) blocks into one wave-tile:
// lane l works on run r = l mod Rb of block g = l div Rb.  Match search, extension, parse and token lookup
// are the general kernel's code, unchanged -- positions are simply block-relative (a block's first run has
// no history: d <= p), and the parse needs no reset because no token ever crosses a block end (the last two
// bytes of a block are always literals, R5).  What is per block here: bit offsets (segmented scan), the
// LDS bit-buffer region, Adler-32, trailer, length and the flush.  Output is bit-identical to the general
// kernel (and to the reference).  Fixed-pitch, 16-byte aligned batches take two 16-byte loads per lane; ragged batches
// (in_off, with in_len = an upper bound on the block lengths: every block gets ceil(bound/32) lanes) re-align their
// bytes from aligned dword loads.
// (Round 5: single-wave workgroups order their LDS accesses with wave_lds_order() -- __syncthreads() also waited vmcnt(0) for the
// output stores of the previous group, three times per group.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "hdlz_device.h"
#include "hdlz_compress_common.h"

namespace hdlz {

#ifndef HDLZ_WS
#define HDLZ_WS 4
#endif
constexpr int SMALL_OUT_WORDS = 704;          // G * ceil(out_bound(N)/4) is largest for N = 32: 64 * 11

struct __attribute__((aligned(16))) SmallLds {
    uint32_t in[IN_BYTES / 4];
    uint32_t out[SMALL_OUT_WORDS];
    uint32_t lut[LUT_LIT + LUT_MATCH + LUT_LEN];      // fill_luts writes all three tables
    uint32_t ad[2][64];                       // per-lane Adler partials, summed per block by its first lane
};

static_assert(sizeof(SmallLds) >= GATHER_SPAN && offsetof(SmallLds, in) == 0, "make_tokens: masked gather inside the LDS block");
// windows above 32: the bit buffer lives in the finder's table / transposition buffer, which is dead once best[] is in registers (as in
// k_compress: 13.6 instead of 16.4 KB per wave -- the LDS, not the registers, bounded the waves per CU)
struct __attribute__((aligned(16))) SmallLdsNoOut {
    uint32_t in[IN_BYTES / 4];
    uint32_t lut[LUT_LIT + LUT_MATCH + LUT_LEN];
    uint32_t ad[2][64];
};
static_assert(sizeof(SmallLdsNoOut) >= GATHER_SPAN && offsetof(SmallLdsNoOut, in) == 0, "make_tokens: masked gather inside the LDS block");
__device__ __forceinline__ uint32_t* small_out(SmallLds& l, uint32_t&) { return l.out; }
template <int NCH> __device__ __forceinline__ uint32_t* small_out(SmallLdsNoOut&, HashLds<NCH>& h) {
    static_assert(sizeof(h.T) >= sizeof(uint32_t) * SMALL_OUT_WORDS, "the bit buffer overlays T / D");
    return h.T;
}

// NCH = ceil(cwindow / 32) as in k_compress: 1, or 2 / 8 with the window-independent finder (round 6: up to then a small block with a
// window above 32 had a whole tile of the general kernel to itself -- 12.5 ns per block whatever its size: 256-byte blocks at
// CWINDOW 256 20 GB/s against 390 at CWINDOW 32).  The finder works on the packed tile as on any other: it returns the NEAREST
// earlier position with the same three bytes, and if that one lies in the block in front (distance > position: make_tokens drops
// it) no nearer one exists in the block itself.
template <int NCH> constexpr int small_waves() { return NCH == 1 ? HDLZ_WS : waves_eu<NCH>(); }
template <bool RAGGED, bool FULLWIN, int NCH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(small_waves<NCH>(), small_waves<NCH>()))) void k_compress_small(CompressArgs a) {
    __shared__ typename std::conditional<(NCH > 1), SmallLdsNoOut, SmallLds>::type lds;
    __shared__ typename std::conditional<(NCH > 1), HashLds<NCH>, uint32_t>::type hl;
    uint32_t* const lout = small_out(lds, hl);
    const uint32_t lane = threadIdx.x;
    fill_luts<NCH>(lds.lut, lane);
    __syncthreads();

    const uint32_t cw4 = 4u * (uint32_t)a.cwindow;
    const uint32_t kmax = (uint32_t)a.maxmatch;
    const uint32_t n = a.in_len;                                  // block length (fixed pitch) or its upper bound (ragged)
    constexpr bool ragged = RAGGED;                                // (a template parameter: the uniform path stays as lean as it was)
    const uint32_t Rb = (n + 31u) >> 5;                           // runs (lanes) per block
    const uint32_t G = 64u / Rb;                                  // blocks per wave-tile
    const uint32_t Wb = (out_bound(n) + 3u) >> 2;                 // bit-buffer words per block
    const uint32_t g = lane / Rb, r = lane - g * Rb;              // this lane: run r of block g
    const uint32_t p_run = r * RUN;                               // block-relative position of the run
    uint8_t* lin8 = reinterpret_cast<uint8_t*>(lds.in);
    const uint8_t* lut8 = reinterpret_cast<const uint8_t*>(lds.lut);
    uint8_t* out8 = reinterpret_cast<uint8_t*>(lout);
    const uint64_t ngroups = (a.nblocks + G - 1u) / G;

    for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint64_t blk = grp * G + g;
        const bool has_blk = (g < G) && (blk < a.nblocks);
        uint64_t boff = 0;
        uint32_t nb = n;                                          // this lane's block: offset and length
        if (has_blk) {
            if (ragged) {
                boff = a.in_off[blk];
                const uint64_t len64 = a.in_off[blk + 1] - boff;          // descending offsets wrap to a huge value
                nb = len64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)len64;   // -> BAD_PARAM below (longer than the bound)
            }
            else boff = blk * a.in_pitch;
        }
        const bool lane_ok = has_blk && nb >= 5u && nb <= n;      // R0: shorter blocks never start (status below)
        if (has_blk && !lane_ok && r == 0u) {                     // (nb > n: the caller's bound on the lengths was wrong)
            a.out_len[blk] = 0;
            a.status[blk] = nb < 5u ? HDLZ_E_SHORT_INPUT : HDLZ_E_BAD_PARAM;
        }
        const uint32_t nrem = lane_ok ? nb - min(p_run, nb) : 0u; // positions of the block from this run on
        // -------------------------------------------------------------- 1. stage: every lane loads its own run
        wave_lds_order();
        {
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
            if (nrem != 0u) {
                const uint8_t* src = a.in + boff + p_run;
                if (!ragged) {
                    v0 = *reinterpret_cast<const uint4*>(src);                // in_pitch % 16 == 0: whole 16-B pieces are readable
                    if (nrem > 16u) v1 = *reinterpret_cast<const uint4*>(src + 16);
                } else {
                    // any alignment: aligned dwords that hold valid bytes only, re-aligned with v_alignbyte
                    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(src - mis);
                    const uint32_t nd = (min(nrem, (uint32_t)RUN) + mis + 3u) >> 2;      // dwords with valid bytes (<= 9)
                    uint32_t dwd[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) dwd[k] = (uint32_t)k < nd ? q[k] : 0u;
                    v0.x = alignbyte(dwd[1], dwd[0], mis); v0.y = alignbyte(dwd[2], dwd[1], mis);
                    v0.z = alignbyte(dwd[3], dwd[2], mis); v0.w = alignbyte(dwd[4], dwd[3], mis);
                    v1.x = alignbyte(dwd[5], dwd[4], mis); v1.y = alignbyte(dwd[6], dwd[5], mis);
                    v1.z = alignbyte(dwd[7], dwd[6], mis); v1.w = alignbyte(dwd[8], dwd[7], mis);
                }
                // bytes at or beyond N must read as zero
                uint32_t* vv = reinterpret_cast<uint32_t*>(&v0);
                uint32_t* ww = reinterpret_cast<uint32_t*>(&v1);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t lo = 4u * k;
                    vv[k] &= nrem <= lo ? 0u : (nrem >= lo + 4u ? 0xFFFFFFFFu : ((1u << (8u * (nrem - lo))) - 1u));
                    const uint32_t hi = 16u + lo;
                    ww[k] &= nrem <= hi ? 0u : (nrem >= hi + 4u ? 0xFFFFFFFFu : ((1u << (8u * (nrem - hi))) - 1u));
                }
            }
            *reinterpret_cast<uint4*>(lin8 + HALO + lane * RUN) = v0;
            *reinterpret_cast<uint4*>(lin8 + HALO + lane * RUN + 16) = v1;
            lds.in[lane] = 0;                                                  // halo in front of lane 0
            if (lane < LOOKAHEAD / 4) lds.in[(HALO + TILE) / 4 + lane] = 0;    // look-ahead behind lane 63
        }
        static_assert(SMALL_OUT_WORDS % 4 == 0, "16-byte stores");
        auto clear_out = [&]() {
            for (uint32_t q = lane; q < (uint32_t)SMALL_OUT_WORDS / 4u; q += 64) *reinterpret_cast<uint4*>(&lout[4u * q]) = make_uint4(0, 0, 0, 0);
            wave_lds_order();
            if (lane_ok && r == 0u) lout[g * Wb] = 0x78u | (0x9Cu << 8) | (0x3u << 16);   // R1 per block
            wave_lds_order();
        };
        if constexpr (NCH == 1) clear_out();
        else wave_lds_order();

        // -------------------------------------------------------------- 2..5: the shared tile phases (hdlz_compress_common.h);
        // positions are block-relative: a block's first run has no history (d <= p)
        const uint32_t run_dw = (HALO / 4) + lane * (RUN / 4);   // dword index of the run in lds.in
        uint32_t best[RUN], tok[RUN], code[RUN];
        __builtin_amdgcn_s_setprio(0);                         // (the search at the lowest priority, every other phase above it: hdlz_compress.hip)
        if constexpr (NCH > 1) {
            match_search_hash<NCH>(lds.in, hl, lane, (uint32_t)a.cwindow, best);                         // 2. R3/R4, windows above 32
            pin(best);
            wave_lds_order();
            clear_out();                                                                                 // (the finder's buffers are dead: the bit buffer)
        }
        else match_search<NCH, NCH == 1>(lds.in, run_dw, best);    // 2. R3/R4 (candidate keys by DPP: a run in front of a block's first run belongs to
                                                                   //    another block -- or is lane 63 -- and only yields distances beyond the position)
        {
            uint32_t ow[12];                                          // own 32 bytes + 16 look-ahead (reloaded: see match_search)
            load_own(lds.in, run_dw, ow);
            {                                                         // 6. Adler partials per lane -> LDS
                uint32_t sa, sc;
                adler_run(ow, sa, sc);
                lds.ad[0][lane] = sa;                                 // <= 8160
                lds.ad[1][lane] = nrem * sa - sc;                     // sum (N - p) x_p over the run, < 2^24 for N <= 1024
            }
            __builtin_amdgcn_s_setprio(1);
            make_tokens<NCH, FULLWIN>(lds.in, HALO + lane * RUN, ow, best, cw4, kmax, 4u * min(p_run, 32u * NCH), nrem, tok);   // 3. R5
        }
        pin(tok);
        PHASE_FENCE();
        const uint64_t P = run_transfer(tok);                                                      // 4. greedy parse
        uint32_t s_chain = 0;            // no token ever crosses a block end (R5), so skips reset by themselves
        uint32_t myskip = chain_skips(P, lane, s_chain);
        pin(tok); asm volatile("" : "+v"(myskip));
        PHASE_FENCE();
        // lanes without a block never start a token; padding positions of a block emit nothing
        uint32_t lane_bits = token_codes<NCH, true>(lut8, tok, lane_ok ? myskip : 0xFFFFu, nrem, code);   // 5. R6/R7
        pin(code);
        PHASE_FENCE();
        // wave scan of lane_bits; bit offsets are per block (segmented by the block's first lane)
        uint32_t incl = wave_scan_incl(lane_bits, lane);
        const uint32_t first_lane = g * Rb;
        // (shuffles must be executed by ALL lanes: a source lane that skipped it reads as garbage)
        const uint32_t prev_incl = (uint32_t)__shfl((int)incl, (int)((first_lane - 1u) & 63u), 64);
        const uint32_t before_blk = first_lane == 0u ? 0u : prev_incl;
        pin(code); asm volatile("" : "+v"(incl), "+v"(lane_bits));
        PHASE_FENCE();
        // OR every token into the block's region of the LDS bit buffer (a lane without a block emits nothing: all codes are zero)
        scatter_codes(out8, code, lane_ok ? 32u * g * Wb + 19u + (incl - lane_bits - before_blk) : 0u);
        wave_lds_order();
        // -------------------------------------------------------------- 7. per block: trailer, length, flush
        // the block's first lane finishes its block (R8/R9)
        const uint32_t last_lane = min(first_lane + Rb - 1u, 63u);
        const uint32_t blk_bits = (uint32_t)__shfl((int)incl, (int)last_lane, 64) - before_blk;
        uint32_t total = 0;
        if (lane_ok && r == 0u) {
            uint32_t s1 = 1u, s2 = nb;
            for (uint32_t k = 0; k < Rb; k++) { s1 += lds.ad[0][first_lane + k]; s2 += lds.ad[1][first_lane + k]; }
            s1 %= ADLER_MOD; s2 %= ADLER_MOD;
            const uint32_t end_bits = 19u + blk_bits;
            const uint32_t nbytes = (end_bits + 7u + 7u) >> 3;     // EOB = 7 zero bits, then zero padding
            uint8_t* ob = out8 + 4u * g * Wb;
            ob[nbytes] = (uint8_t)(s2 >> 8);
            ob[nbytes + 1] = (uint8_t)s2;
            ob[nbytes + 2] = (uint8_t)(s1 >> 8);
            ob[nbytes + 3] = (uint8_t)s1;
            total = nbytes + 4u;
            a.out_len[blk] = total;
            a.status[blk] = HDLZ_OK;
        }
        wave_lds_order();
        for (uint32_t gg = 0; gg < G; gg++) {                       // flush block by block, coalesced dwords
            const uint64_t b2 = grp * G + gg;
            if (b2 >= a.nblocks) break;
            const uint32_t words = ((uint32_t)__builtin_amdgcn_readlane((int)total, (int)(gg * Rb)) + 3u) >> 2;
            uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(a.out + b2 * a.out_pitch);
            for (uint32_t w = lane; w < words; w += 64) dst[w] = lout[gg * Wb + w];
        }
    }
}

hipError_t launch_compress_small(const CompressArgs& a, hipStream_t stream, int ncu) {
    const uint32_t Rb = (a.in_len + 31u) >> 5;
    const uint64_t G = 64u / Rb;
    uint64_t groups = (a.nblocks + G - 1u) / G;
#ifndef HDLZ_SMALL_GRID
#define HDLZ_SMALL_GRID 256
#endif
    uint64_t grid = (uint64_t)ncu * HDLZ_SMALL_GRID;       // (see launch_compress: short-lived waves, short tail)
    if (grid > groups) grid = groups;
    const dim3 g((unsigned)grid), b(64);
    const bool rag = a.in_off != nullptr;
#define SMALL_LAUNCH(R, F, N) hipLaunchKernelGGL((k_compress_small<R, F, N>), g, b, 0, stream, a)
#define SMALL_BY_NCH(N, full) do { if (rag) { if (full) SMALL_LAUNCH(true, true, N); else SMALL_LAUNCH(true, false, N); } \
                                   else { if (full) SMALL_LAUNCH(false, true, N); else SMALL_LAUNCH(false, false, N); } } while (0)
    if (a.cwindow <= 32) SMALL_BY_NCH(1, a.cwindow == 32);
    else if (a.cwindow <= 64) SMALL_BY_NCH(2, a.cwindow == 64);
    else SMALL_BY_NCH(8, a.cwindow == 256);
#undef SMALL_BY_NCH
#undef SMALL_LAUNCH
    return hipGetLastError();
}

}  // namespace hdlz
