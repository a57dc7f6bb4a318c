// hdlz_compress.hip -- STARTC for a batch of independent blocks on gfx950 (CDNA4, wave64).
//
// Replaces the reference's compress FSM (/root/reference/deflate.py:734-1082 + :407-515 +
// :535-567) with a data-parallel formulation.  Rule names R0..R9 are SURVEY.md 8(a)'s.
//
// Mapping: persistent single-wave workgroups; a wave takes blocks blockIdx.x, +gridDim.x, ... and
// walks each block in tiles of 2048 positions; lane l owns the RUN of 32 consecutive positions
// [32l, 32l+32) of the tile.  Per tile:
//   1. tile + 256-byte look-back halo + 16-byte look-ahead staged in LDS (coalesced 16-B loads)
//   2. match search (R3/R4): each lane builds keys K = (3-byte-string << 8) | 4*window_index for
//      its own 32 positions and the 32..256 positions before them, all in VGPRs.  For an own key
//      Ko and a candidate key Kc,  Ko - Kc  equals 4*distance (<= 128) iff the three bytes are
//      equal and is > 256 (as u32) otherwise, so the MIN over the 32 candidates IS four times the
//      nearest matching distance: one v_sub + half a v_min3 per compare, no branches.
//   3. extension (R5): 8-byte LDS gather at p-d+3, xor with the own bytes, count-trailing-zeros.
//   4. greedy parse ("di += m / di += 1", deflate.py:960,1008): every lane folds its run into a
//      transfer function "entry skip (0..9) -> exit skip", 10 nibbles packed in 40 bits, by a
//      backward pass; a 64-step scalar readlane chain composes them across the wave.
//   5. token bits (R6/R7) from per-wave LDS look-up tables (literal: [byte] -> code|nbits, match:
//      [len][dist] -> code|nbits), in-lane prefix sums + wave scan -> bit offsets, then every
//      token is OR-ed into an LDS bit buffer at its own bit offset (ds_or_b32), coalesced dword
//      flush to HBM; the partial word is carried to the next tile.
//   6. Adler-32 (R8) by per-lane byte sums / index-weighted sums (v_sad_u8 / v_dot4_u32_u8).
// Positions >= N in the last tile are zero bytes: they can never match (R3) and each of them is
// parsed as one 8-bit literal; their bits land behind the real end of the stream and are wiped
// once before the trailer is written -- so the hot loop carries no per-position validity mask.
//
// Cost model (measured, tools/ubench): v_add/sub/and/or/xor/lshr ~2.5 cycles per wave64
// instruction, every other VALU op (min3, cmp, cndmask, alignbyte, lshl, 64-bit shifts, ...) ~4.2.
// The kernel is VALU-issue bound, so the design minimises VALU instructions and moves
// table work to LDS.  No MFMA: nothing here is a dense contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"
#include "hdlz_compress_common.h"

namespace hdlz {

// wave priorities of the wide-window (hash finder) kernels by phase: the finder is a chain of LDS round trips -- it goes first
#ifndef HDLZ_HP_SEARCH
#define HDLZ_HP_SEARCH 1
#define HDLZ_HP_EXTEND 0
#define HDLZ_HP_REST 0
#endif
#ifdef HDLZ_TILE_TIMING       // diagnostic build (tools/exp_tile_timing.py): s_memtime at the phase boundaries of the tile; the j-th block a wave
                              // processed reports the wave's total of part j (cycles, 32 bits) in out_len INSTEAD of its result
#define TT_DECL() uint32_t tacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = (uint32_t)__builtin_readcyclecounter(); \
        const uint32_t tt_c0 = tlast, tt_r0 = (uint32_t)__builtin_amdgcn_s_memrealtime()
#define TT(k) do { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#define TT_TILE() tacc[12] += 1u
#else
#define TT_DECL() do {} while (0)
#define TT(k) do {} while (0)
#define TT_TILE() do {} while (0)
#endif

// lsrc[0 .. words) (LDS, 16-byte aligned) -> dst (HBM, 4-byte aligned): 16-byte stores, then the one to three words that are left
__device__ __forceinline__ void store_words(uint32_t* __restrict__ dst, const uint32_t* lsrc, uint32_t words, uint32_t lane) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const uint32_t nq = words >> 2;
    for (uint32_t q0 = 0; q0 < nq; q0 += 64u) {
        const uint32_t q = q0 + lane;
        if (q < nq) {
            const v4 v = *reinterpret_cast<const v4*>(lsrc + 4u * q);
            // (one 16-byte store whatever the alignment of the row: gfx950 asks for 4 bytes; s_nop: the data registers of a store wider
            //  than 8 bytes must not be written in the next two wait states, and the hazard recognizer does not look into inline asm)
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst + 4u * q), "v"(v) : "memory");
        }
    }
    if (lane < (words & 3u)) dst[4u * nq + lane] = lsrc[4u * nq + lane];
}

// NCH = ceil(cwindow / 32): 1, 2 or 8 chunks of 32 candidate distances; FULLWIN: cwindow == 32 * NCH (the reference's
// own windows 32 and 256, and 64), which spares the per-position window compare; ONE_TILE: every block of the batch fits one
// wave-tile (N <= 2048: BASELINE configs[1]'s block size and the reference's own IBSIZE scale) -- no tile loop, no halo
// carried from a previous tile, no carried bit / Adler state
template <int NCH, bool FULLWIN, bool ONE_TILE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(waves_eu<NCH>(), waves_eu<NCH>()))) void k_compress(CompressArgs a) {
    constexpr bool HASH = wide_hash<NCH>();         // windows > 32: the window-independent finder
    __shared__ typename std::conditional<HASH, WaveLdsNoOut, WaveLds>::type lds;
    __shared__ typename std::conditional<HASH, HashLds<NCH>, uint32_t>::type hl;
    // the bit buffer of a tile: HASH kernels keep it in the finder's transposition buffer, which is dead once best[] is in registers
    // (28 KB of LDS per wave left ONE wave per SIMD: VALU 31 %, LDS 39 % busy -- this kernel lives on overlapping the two)
    uint32_t* const lout = lds_out(lds, hl);
    const uint32_t lane = threadIdx.x;

    fill_luts<NCH>(lds.lut, lane);                  // per-wave look-up tables (once per wave lifetime)
    __syncthreads();

    const uint32_t cw4 = 4u * (uint32_t)a.cwindow;
    const uint32_t kmax = (uint32_t)a.maxmatch;
    uint8_t* lin8 = reinterpret_cast<uint8_t*>(lds.in);
    const uint8_t* lut8 = reinterpret_cast<const uint8_t*>(lds.lut);
    uint8_t* out8 = reinterpret_cast<uint8_t*>(lout);

    // ---- a block: where it starts, how long it is, whether the reference would run it at all (R0) and whether its output fits
    // (fixed-pitch batches: the three checks are the same for every block and made once)
    const uint32_t fixed_st = a.in_len < 5u ? (uint32_t)HDLZ_E_SHORT_INPUT                           // R0: the reference never starts
                            : (ONE_TILE && a.in_len > (uint32_t)TILE) ? (uint32_t)HDLZ_E_BAD_PARAM   // (the caller's bound on the lengths was wrong)
                            : (uint64_t)out_bound(a.in_len) > a.out_pitch ? (uint32_t)HDLZ_E_OUT_CAPACITY : (uint32_t)HDLZ_OK;
    auto block_params = [&](uint64_t blk, const uint8_t*& src, uint32_t& n) -> uint32_t {
        if (!a.in_off) {
            src = a.in + blk * a.in_pitch;
            n = a.in_len;
            return fixed_st;
        }
        const uint64_t off = a.in_off[blk];
        const uint64_t len64 = a.in_off[blk + 1] - off;          // in_off must ascend; a block is limited to 2 GiB - 1
        if (len64 >= 0x80000000ull) return HDLZ_E_BAD_PARAM;      // (descending offsets wrap to a huge value)
        n = (uint32_t)len64;
        src = a.in + off;
        if (n < 5u) return HDLZ_E_SHORT_INPUT;
        if (ONE_TILE && n > (uint32_t)TILE) return HDLZ_E_BAD_PARAM;
        if ((uint64_t)out_bound(n) > a.out_pitch) return HDLZ_E_OUT_CAPACITY;
        return HDLZ_OK;
    };
    // ---- round 5: the tile's input comes through LDS-DMA -- no VGPR round trip, and the zeroing of the tile's tail and of the bit buffer
    // runs while the bytes are on their way.  Only whole 16-byte chunks inside the block are requested (a source of any alignment: the
    // bytes land where they belong); the chunk the block ends in comes through registers, masked.
    // (Requesting tile i+1 behind make_tokens of tile i -- lds.in has no reader left there -- with a COUNTED vmcnt at the stage that leaves
    // the flush stores in flight was built and measured: 4.786 against 4.773 ms on configs[1], profiles/r05_compress_ab.txt.  Five waves
    // per SIMD hide the stage; what counted was the serial skip chain.  Not kept: a counted wait is a hazard for no gain.)
    constexpr uint32_t NCHUNK = (TILE + LOOKAHEAD) / 16;        // 129 16-byte chunks behind the halo
    const uint32_t in_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(lin8)) + (uint32_t)HALO;
    auto request_tile = [&](const uint8_t* tsrc, uint32_t nfull) {       // chunks [0, nfull) of the tile that starts at tsrc
        if (lane < nfull) lds_dma16(tsrc + 16u * lane, in_base);
        if (lane + 64u < nfull) lds_dma16(tsrc + 16u * (lane + 64u), in_base + 1024u);
        if (nfull > 128u) { if (lane == 0u) lds_dma16(tsrc + 2048u, in_base + 2048u); }
    };

    TT_DECL();
    for (uint64_t blk = blockIdx.x; blk < a.nblocks; blk += gridDim.x) {
        const uint8_t* src;
        uint32_t n;
        const uint32_t bst = block_params(blk, src, n);
        if (bst != HDLZ_OK) {
            if (lane == 0) { a.out_len[blk] = 0; a.status[blk] = bst; }
            continue;
        }
        uint32_t* __restrict__ outw = reinterpret_cast<uint32_t*>(a.out + blk * a.out_pitch);
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
        const bool aligned16 = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;

        uint32_t gw = 0;            // output words already flushed to HBM
        uint32_t base_bits = 19;    // R1: 78 9C + bits 1,1,0
        uint32_t carry_word = 0x78u | (0x9Cu << 8) | (0x3u << 16);
        uint32_t skip_in = 0;       // positions at the tile start covered by the previous tile's last match
        // per-lane Adler partials: ad_a = sum x (plain: reduced once per GiB and at the end), ad_w = sum (N-p) x mod 65521 with
        // ONE modulo per tile -- the weight N - p_run is carried mod 65521 from tile to tile instead of being reduced every time
        uint32_t ad_a = 0, ad_w = 0;
        uint32_t wm = (n - min(lane * (uint32_t)RUN, n)) % ADLER_MOD;

        for (uint32_t t0 = 0; t0 < n; t0 += TILE) {       // (ONE_TILE: one iteration; left as a loop -- hipcc spills when it is peeled)
            // -------------------------------------------------------------- 1. stage the tile
            if (blk == blockIdx.x) TT(10); else TT(0);            // block prologue / loop overhead (slot 10: the wave's first -- ~45 us in the
                                                                  // timing build of <1,true,true>, whose prologue spills SGPRs to scratch (the
                                                                  // shipped kernel has none); 140 cycles in the wide-window instantiations)
            HDLZ_MARK("stage");
            const uint32_t nfull = min((n - t0) >> 4, NCHUNK);    // whole chunks of the block in this tile
            wave_lds_order();                                     // (the previous tile's reads of lds.in are done)
            lds.in[lane] = t0 != 0 ? lds.in[(TILE / 4) + lane] : 0u;     // last HALO bytes of the previous tile / tile 0: zero halo (never matched: d <= p)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (read back before the request may overwrite it)
            request_tile(src + t0, nfull);
            // zeros behind the block's end, and the chunk the block ends in: through registers, bytes at or beyond N read as zero.  (Two
            // loops: a store whose value MAY come from a load makes hipcc wait vmcnt(0) in front of it -- for the flush stores too.)
            {
                const uint32_t part = (nfull < NCHUNK && ((n - t0) & 15u) != 0u) ? 1u : 0u;
                for (uint32_t c = nfull + part + lane; c < NCHUNK; c += 64) *reinterpret_cast<uint4*>(lin8 + HALO + c * 16u) = make_uint4(0, 0, 0, 0);
                if (part) {
                    if (lane == 0u) *reinterpret_cast<uint4*>(lin8 + HALO + nfull * 16u) = load_chunk16(src, t0 + nfull * 16u, n, aligned16, mis);
                }
            }
            // zero the bit buffer, seed the carry
            if constexpr (!HASH) zero_bit_buffer(lout, lane, carry_word);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tile has landed (and the last tile's output stores have left)
            wave_lds_order();

            // -------------------------------------------------------------- 2..6: the shared tile phases (hdlz_compress_common.h)
            const uint32_t run_dw = (HALO / 4) + lane * (RUN / 4);   // dword index of the run in lds.in
            const uint32_t p_run = t0 + lane * RUN;                   // first position of this run
            const uint32_t nrem = n - min(p_run, n);                  // positions of the block from p_run on
            uint32_t best[RUN], tok[RUN], code[RUN];
            TT(1);                                                // stage (HBM latency, LDS writes, bit buffer zeroing)
            // Wave priorities (round 5): the search is the one phase that is pure VALU work; every other phase is a chain of LDS round
            // trips, scalar work or memory waits with little to issue.  With the search at the LOWEST priority a wave in any other phase
            // issues the moment it can -- it leaves those phases sooner, and the search of the other four waves fills every slot it does
            // not need: 4.79 -> 4.57 ms on configs[1] (profiles/r05_compress_ab.txt; the opposite assignment: 4.85).  The hash finder of
            // the wide windows is LDS-bound itself: no gain there, left alone.
            if constexpr (!HASH) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(HDLZ_HP_SEARCH);
            HDLZ_MARK("search");
            if constexpr (HASH) {
                match_search_hash<NCH>(lds.in, hl, lane, (uint32_t)a.cwindow, best);               // 2. R3/R4, wide windows
                zero_bit_buffer(lout, lane, carry_word);                                           // (ordered before the scatter by the fences below)
            } else match_search<NCH, ONE_TILE && NCH == 1>(lds.in, run_dw, best);                  // 2. R3/R4 (a one-tile block: candidate keys by DPP)
            {
                TT(2);
                HDLZ_MARK("adler");
                uint32_t ow[12];                                      // own 32 bytes + 16 look-ahead (reloaded: see match_search)
                load_own(lds.in, run_dw, ow);
                {                                                                                  // 6. Adler partials
                    uint32_t sa, sc;
                    adler_run(ow, sa, sc);
                    // sum (N - p) x_p over the run = (N - p_run) * sa - sc ; bytes at p >= N are zero (then the weight does not matter);
                    // wm * sa <= 65520 * 8160, sc <= 252960 < 4 * 65521: the sum stays below 2^31
                    ad_a += sa;
                    ad_w = (ad_w + __umul24(wm, sa) + (4u * ADLER_MOD - sc)) % ADLER_MOD;
                    wm = wm >= (uint32_t)(TILE % ADLER_MOD) ? wm - (uint32_t)(TILE % ADLER_MOD) : wm + (ADLER_MOD - (uint32_t)(TILE % ADLER_MOD));
                    if (((t0 >> 11) & 0x3FFFFu) == 0x3FFFFu) ad_a %= ADLER_MOD;      // (every 2^18 tiles: ad_a grows by <= 8160 per tile)
                    asm volatile("" : "+v"(ad_a), "+v"(ad_w), "+v"(wm));              // computed HERE, while the bytes are in registers
                }
                TT(3);
                if constexpr (!HASH) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(HDLZ_HP_EXTEND);
                HDLZ_MARK("extend");
                make_tokens<NCH, FULLWIN, true>(lds.in, HALO + lane * RUN, ow, best, cw4, kmax, 4u * min(p_run, 32u * NCH), nrem, tok, (int32_t)(n - t0));   // 3. R5
            }
            pin(tok);
            PHASE_FENCE();
            TT(4);
            if constexpr (HASH) __builtin_amdgcn_s_setprio(HDLZ_HP_REST);
            HDLZ_MARK("parse");
            const uint64_t P = run_transfer(tok);                                                  // 4. greedy parse
            TT(5);
            HDLZ_MARK("chain");
            uint32_t myskip = chain_skips(P, lane, skip_in);          // (skip_in: carried into the next tile)
            pin(tok); asm volatile("" : "+v"(myskip));
            PHASE_FENCE();
            TT(6);
            HDLZ_MARK("codes");
            uint32_t lane_bits = token_codes<NCH, false>(lut8, tok, myskip, 0u, code);             // 5. R6/R7
            pin(code);
            PHASE_FENCE();
            TT(7);
            HDLZ_MARK("scan");
            uint32_t incl = wave_scan_incl(lane_bits, lane);
            const uint32_t tile_bits_all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            pin(code); asm volatile("" : "+v"(incl), "+v"(lane_bits));
            PHASE_FENCE();
            TT(8);
            HDLZ_MARK("scatter");
            scatter_codes(out8, code, base_bits + incl - lane_bits);                               // bit writer
            TT(9);
            HDLZ_MARK("flush");
            wave_lds_order();
            TT(10);                                               // (kept for the table's layout)

            // -------------------------------------------------------------- 7. flush
            const bool last = ONE_TILE || (t0 + TILE >= n);
            if (!last) {
                const uint32_t end_bits = base_bits + tile_bits_all;
                const uint32_t full = end_bits >> 5;
                store_words(outw + gw, lout, full, lane);
                carry_word = lout[full];
                gw += full;
                base_bits = end_bits & 31u;
            } else {
                // every position >= N of this tile was emitted as one 8-bit literal (zero byte, never a match,
                // and the last two real bytes are always literals so the parse lands exactly on N)
                const uint32_t ninv = t0 + TILE - n;
                const uint32_t end_bits = base_bits + tile_bits_all - 8u * ninv;
                // wipe everything behind the real end: partial word masked, later words zeroed
                // (only the words that go out: the partial word, the EOB / pad bits and the trailer end within four words of `ew`, and
                //  ew + 3 <= (19 + 9 * 2048) / 32 + 3 < OUT_WORDS; the rest of the buffer is zeroed when the next tile is staged)
                {
                    const uint32_t ew = end_bits >> 5, rb = end_bits & 31u;
                    if (lane < 4u) lout[ew + lane] = (lane == 0u) ? (lout[ew] & ((1u << rb) - 1u)) : 0u;
                }
                // R8: EOB = 7 zero bits, zero pad to a byte, Adler-32 big-endian (s2 then s1)
                uint32_t s1 = wave_sum(ad_a % ADLER_MOD), s2 = wave_sum(ad_w);
                s1 = (s1 + 1u) % ADLER_MOD;
                s2 = (s2 + n % ADLER_MOD) % ADLER_MOD;
                const uint32_t nbytes = (end_bits + 7u + 7u) >> 3;
                wave_lds_order();
                if (lane == 0) {
                    out8[nbytes] = (uint8_t)(s2 >> 8);
                    out8[nbytes + 1] = (uint8_t)s2;
                    out8[nbytes + 2] = (uint8_t)(s1 >> 8);
                    out8[nbytes + 3] = (uint8_t)s1;
                }
                wave_lds_order();
                const uint32_t total = nbytes + 4u;
                const uint32_t words = (total + 3u) >> 2;
                store_words(outw + gw, lout, words, lane);
                if (lane == 0) {
                    a.out_len[blk] = gw * 4u + total;     // R9
                    a.status[blk] = HDLZ_OK;
                }
            }
            TT(11);                                               // flush
            TT_TILE();
        }
    }
#ifdef HDLZ_TILE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        uint64_t blk = blockIdx.x;
#pragma unroll
        for (int k = 0; k < 13; k++, blk += gridDim.x)
            if (blk < a.nblocks) { a.out_len[blk] = tacc[k]; a.status[blk] = 0u; }
        // the wave's lifetime in s_memtime ticks and in s_memrealtime ticks (100 MHz): their ratio is the clock the first one counts
        if (blk < a.nblocks) { a.out_len[blk] = (uint32_t)__builtin_readcyclecounter() - tt_c0; a.status[blk] = 0u; }
        blk += gridDim.x;
        if (blk < a.nblocks) { a.out_len[blk] = (uint32_t)__builtin_amdgcn_s_memrealtime() - tt_r0; a.status[blk] = 0u; }
    }
#endif
}

template __global__ void k_compress<1, true, true>(CompressArgs);
template __global__ void k_compress<1, false, true>(CompressArgs);
template __global__ void k_compress<1, true, false>(CompressArgs);
template __global__ void k_compress<1, false, false>(CompressArgs);
template __global__ void k_compress<2, true, false>(CompressArgs);
template __global__ void k_compress<2, false, false>(CompressArgs);
template __global__ void k_compress<8, true, false>(CompressArgs);
template __global__ void k_compress<8, false, false>(CompressArgs);

hipError_t launch_compress(const CompressArgs& a, hipStream_t stream) {
    if (a.nblocks == 0) return hipSuccess;
    // persistent single-wave workgroups: 256 per CU queued (20 resident at 5 waves/SIMD) so that the
    // hardware dispatcher balances uneven blocks; each wave strides over the batch
    // (the CU count is cached per DEVICE: a process may drive several GPUs)
    static int ncu_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    int ncu = (dev >= 0 && dev < 64) ? ncu_of[dev] : 0;
    if (ncu == 0) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return hipGetLastError();
        if (ncu <= 0) ncu = 256;
        if (dev >= 0 && dev < 64) ncu_of[dev] = ncu;
    }
    // small blocks (the reference's own input scale): several blocks per wave-tile -- uniform 16-byte aligned batches,
    // or ragged ones whose caller states an upper bound on the block lengths in in_len
    if (a.cwindow <= 256 && a.in_len >= 5u && a.in_len <= 1024u && a.out_pitch >= (uint64_t)out_bound(a.in_len) &&
        (a.in_off || ((a.in_pitch & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15u) == 0)))
        return launch_compress_small(a, stream, ncu);
    // (round 5: 256 waves queued per CU instead of 64 -- a wave's blocks all belong to one family when the families alternate with a
    //  period that divides the grid, and waves of different families differ 3x in their run time (profiles/r05_tile_timing.txt); the
    //  shorter a wave lives, the shorter the tail in which the GPU drains: 4.585 -> 4.475 ms on configs[1], 128: 4.515, 512: 4.475, 1024: 4.51)
    uint64_t g = (uint64_t)ncu * 256u;
    if (g > a.nblocks) g = a.nblocks;
    const dim3 grid((unsigned)g), block(64);
    // every block within one wave-tile (fixed size, or a ragged batch whose caller states such a bound in in_len)
    const bool one_tile = a.in_len >= 5u && a.in_len <= (uint32_t)TILE;
    if (a.cwindow == 32 && one_tile) hipLaunchKernelGGL((k_compress<1, true, true>), grid, block, 0, stream, a);
    else if (a.cwindow < 32 && one_tile) hipLaunchKernelGGL((k_compress<1, false, true>), grid, block, 0, stream, a);
    else if (a.cwindow == 32) hipLaunchKernelGGL((k_compress<1, true, false>), grid, block, 0, stream, a);
    else if (a.cwindow < 32) hipLaunchKernelGGL((k_compress<1, false, false>), grid, block, 0, stream, a);
    else if (a.cwindow == 64) hipLaunchKernelGGL((k_compress<2, true, false>), grid, block, 0, stream, a);
    else if (a.cwindow < 64) hipLaunchKernelGGL((k_compress<2, false, false>), grid, block, 0, stream, a);
    else if (a.cwindow == 256) hipLaunchKernelGGL((k_compress<8, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_compress<8, false, false>), grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
