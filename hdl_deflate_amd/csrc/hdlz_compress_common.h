// hdlz_compress_common.h -- constants, device helpers and THE TILE PHASES shared by all compress kernels
// (hdlz_compress.hip: one block per wave, any size; hdlz_compress_small.hip: several small blocks per wave-tile;
//  hdlz_compress_stream.hip: one large stream / a few large blocks spread over the whole GPU)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"

namespace hdlz {

// waves per SIMD the CWINDOW <= 32 kernels are register-capped for (96 VGPRs at 5)
#ifndef HDLZ_W1
#define HDLZ_W1 5
#endif

#ifndef HDLZ_EXT_GROUP
#define HDLZ_EXT_GROUP 4
#endif
#ifndef HDLZ_CODE_GROUP
#define HDLZ_CODE_GROUP 4
#endif
constexpr int RUN = 32;             // positions per lane
constexpr int TILE = 64 * RUN;      // 2048 positions per wave-tile
constexpr int HALO = 256;           // bytes kept in front of the tile (max CWINDOW)
constexpr int LOOKAHEAD = 16;       // bytes staged behind the tile (need p+9 and p+2)
constexpr int IN_BYTES = HALO + TILE + LOOKAHEAD;   // 2320
constexpr int OUT_WORDS = 592;      // 9 bits * 2048 = 576 words + carry word + slack
constexpr int LUT_LIT = 256;        // [byte]                                  at LUT word 0
constexpr int LUT_MATCH = 256;      // [len-3][dist-1] (CWINDOW <= 32) or [dist-1]  at LUT word 256
constexpr uint32_t LUT_MATCH_BYTE = 4u * LUT_LIT;
constexpr int LUT_LEN = 16;         // [len-1] -> the length code (wide windows: the match LUT is [dist-1] only)   at LUT word 512
constexpr uint32_t LUT_LEN_BYTE = 4u * (LUT_LIT + LUT_MATCH);
constexpr uint32_t ADLER_MOD = 65521u;
constexpr uint32_t NB_SHIFT = 27;   // LUT entry = code (27 bits) | nbits << 27
constexpr uint32_t CODE_MASK = (1u << NB_SHIFT) - 1u;

struct __attribute__((aligned(16))) WaveLds {
    uint32_t in[IN_BYTES / 4];      // byte index = position - tile_start + HALO
    uint32_t out[OUT_WORDS];        // bit buffer of the current tile
    uint32_t lut[LUT_LIT + LUT_MATCH + LUT_LEN];
};

constexpr uint32_t GATHER_SPAN = 4u * (0x3FFu + 3u);      // bytes behind `in` that make_tokens' masked gather may touch (GMASK = 0x3FF)
static_assert(sizeof(WaveLds) >= GATHER_SPAN, "make_tokens: masked gather inside the LDS block");

struct __attribute__((aligned(16))) WaveLdsNoOut {       // kernels with the hash finder: the bit buffer lives in HashLds::D's memory
    uint32_t in[IN_BYTES / 4];
    uint32_t lut[LUT_LIT + LUT_MATCH + LUT_LEN];
};

// fence for the instruction scheduler + value fences: keep independent phases from being overlapped
// (that blew the VGPR budget to 239 and spilled ~200 SGPR lane masks in the first version)
#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
// a comment in the ISA at a phase boundary (tools/phase_count.py counts the instructions between two marks)
#define HDLZ_MARK(name) do { PHASE_FENCE(); asm volatile("; @@PHASE " name ::: "memory"); PHASE_FENCE(); } while (0)
template <int N>
__device__ __forceinline__ void pin(uint32_t (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(a[i]));
}
template <int B, int E, int N>
__device__ __forceinline__ void pin_range(uint32_t (&a)[N]) {
#pragma unroll
    for (int i = B; i < E; i++) asm volatile("" : "+v"(a[i]));
}

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);      // bytes [sh, sh+4) of hi:lo
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t t = a < b ? a : b;
    return t < c ? t : c;
}
// index of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32)
__device__ __forceinline__ uint32_t ffbl(uint32_t x) { return (uint32_t)(__builtin_ffs((int)x) - 1); }

// key of the 3-byte string at byte J of d[]: bytes 1..3 = the string, byte 0 = tag (4 * window index)
template <int J>
__device__ __forceinline__ uint32_t key3(const uint32_t* d, uint32_t tag) {
    constexpr int w = J >> 2, sh = J & 3;
    if constexpr (sh == 0) return (d[w] << 8) | tag;
    else if constexpr (sh == 1) return (d[w] & 0xFFFFFF00u) | tag;
    else return (alignbyte(d[w + 1], d[w], sh - 1) & 0xFFFFFF00u) | tag;
}

// ---- fixed Huffman token bits (used to fill the LUTs) ---------------------------------------
// literal (R7, deflate.py:1005-1016 + out_codes :112-149): sym<144 -> 8 bits rev8(0x30+sym),
// else 9 bits rev9(0x100+sym)
__device__ __forceinline__ uint32_t literal_entry(uint32_t b) {
    const bool big = b >= 144u;
    const uint32_t v = big ? (0x100u + b) : (0x30u + b);
    const uint32_t nb = big ? 9u : 8u;
    return (__builtin_bitreverse32(v) >> (32u - nb)) | (nb << NB_SHIFT);
}
// distance part of a match token (R6, deflate.py:836-882): rev5(dist code) | extra<<5 in 5+eb bits,
// placed behind the 7-bit length code; nbits = 12 + eb
__device__ __forceinline__ uint32_t dist_entry(uint32_t d) {
    const uint32_t dd = d - 1u;
    uint32_t c, eb, extra;
    if (dd < 4u) {
        c = dd; eb = 0; extra = 0;
    } else {
        const uint32_t hb = 31u - (uint32_t)__builtin_clz(dd);
        eb = hb - 1u;
        c = 2u * hb + ((dd >> eb) & 1u);
        extra = dd & ((1u << eb) - 1u);
    }
    const uint32_t dcode = __builtin_bitreverse32(c) >> 27;
    return ((dcode | (extra << 5)) << 7) | ((12u + eb) << NB_SHIFT);
}
// 7-bit code of length symbol 254+m (no extra bits for m <= 10)
__device__ __forceinline__ uint32_t length_code(uint32_t m) { return __builtin_bitreverse32(m - 2u) >> 25; }

// =====================================================================================================================
// The tile phases.  ONE source for all compress kernels: k_compress (one block per wave, hdlz_compress.hip),
// k_compress_small (several small blocks per wave-tile, hdlz_compress_small.hip) and the multi-wave stream passes
// k_stream_tails / _xfer / _tile (hdlz_compress_stream.hip).  A wave-tile is 2048 positions, lane l owns the RUN of 32
// consecutive positions [32l, 32l+32); `in` is the staged input (byte index = position - tile start + halo).
// Everything is forced inline: the phases are straight-line code over register arrays, and the value fences (pin) and
// scheduling barriers (PHASE_FENCE) inside them are what keeps hipcc from overlapping the phases.
// =====================================================================================================================

// ---- the lane's own 32 bytes + 16 bytes of look-ahead
__device__ __forceinline__ void load_own(const uint32_t* in, uint32_t run_dw, uint32_t (&ow)[12]) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(&in[run_dw]);
    const uint4 v1 = *reinterpret_cast<const uint4*>(&in[run_dw + 4]);
    const uint4 v2 = *reinterpret_cast<const uint4*>(&in[run_dw + 8]);
    ow[0] = v0.x; ow[1] = v0.y; ow[2] = v0.z; ow[3] = v0.w;
    ow[4] = v1.x; ow[5] = v1.y; ow[6] = v1.z; ow[7] = v1.w;
    ow[8] = v2.x; ow[9] = v2.y; ow[10] = v2.z; ow[11] = v2.w;
}

// ---- phase 2, match search (R3/R4; the reference's matcher3 x CWINDOW + first-set-bit pick, deflate.py:407-421, :975-994):
// best[i] = 4 * (nearest distance d in [1, 32*NCH] with x[p-d .. p-d+2] == x[p .. p+2]) for own position i; a value > 128 * NCH
// (NCH > 1: NO_MATCH4) = none.
// Keys K = (3-byte string) << 8 | 4 * window index: Ko - Kc (u32 wrap) is 4*distance (<= 128) iff the strings are equal and
// > 256 otherwise, so the MIN over a chunk of 32 candidates is four times the nearest matching distance: one v_sub and
// half a v_min3 per compare.  Windows > 32 iterate chunks of 32 distances far -> near.
// (The lane's own bytes are loaded here and die with the keys: the 32 own keys + 32 running minima + the candidate
// bytes are the register peak of the whole kernel; later phases reload the 48 bytes from LDS, three ds_read_b128.)
// wide windows (NCH > 1), "no candidate": a value every eligibility test rejects (> 4 * 256 + 124) whose distance field -- bits
// 2..10, what make_tokens gathers with -- reads 1: one v_bfe instead of compare + select + shift per position
constexpr uint32_t NO_MATCH4 = 0x8004u;
// PREV_DPP (NCH == 1, a tile that starts its block): the 32 candidates in front of the run are the own positions of the PREVIOUS lane, so
// their keys are that lane's own keys minus 128 (tag 4 (j + 32) -> 4 j): one DPP add each instead of two 16-byte LDS loads and ~1.5
// shift / align / mask instructions per key.  The DPP is a ROTATE (wave_ror:1): lane 0 has no previous lane and takes lane 63's keys --
// real keys with candidate tags, so whatever they match lies i + 32 - j > i = p positions back: in front of the block's first byte.  A
// nearer in-block candidate always wins the minimum, and a result from out there fails the position test of make_tokens like "none" does.
// (wave_shr with bound_ctrl would feed lane 0 the constant -128, which an own key FF FF FF | tag turns into a distance <= p: wrong.)
template <int NCH, bool PREV_DPP = false>
__device__ __forceinline__ void match_search(const uint32_t* in, uint32_t run_dw, uint32_t (&best)[RUN]) {
    uint32_t ko[RUN];
    uint32_t ow0;
    {
        uint32_t ow[12];
        load_own(in, run_dw, ow);
        static_for<0, RUN>([&](auto I) { constexpr int i = decltype(I)::value; ko[i] = key3<i>(ow, (uint32_t)(4 * (i + 32))); });
        ow0 = ow[0];
    }
    pin(ko); asm volatile("" : "+v"(ow0));
    PHASE_FENCE();
#pragma unroll
    for (int i = 0; i < RUN; i++) best[i] = NCH == 1 ? 0xFFFFFFFFu : NO_MATCH4;

#pragma unroll 1
    for (int k = NCH - 1; k >= 0; k--) {                      // far chunks first, nearer ones overwrite
        uint32_t cd[17];                                      // 64 candidate positions + 2 bytes
        const uint32_t cdw = run_dw - 8u * (uint32_t)(k + 1);
        [[maybe_unused]] uint32_t m128 = 0xFFFFFF80u;         // (-128 in a register: the DPP form takes no literal)
        if constexpr (PREV_DPP) {
            static_assert(NCH == 1, "own keys double as the next lane's candidates only for a 32-wide window");
            asm volatile("" : "+v"(m128));
#pragma unroll
            for (int j = 0; j < 17; j++) cd[j] = 0;
        } else {
            const uint4 c0 = *reinterpret_cast<const uint4*>(&in[cdw]);
            const uint4 c1 = *reinterpret_cast<const uint4*>(&in[cdw + 4]);
            cd[0] = c0.x; cd[1] = c0.y; cd[2] = c0.z; cd[3] = c0.w;
            cd[4] = c1.x; cd[5] = c1.y; cd[6] = c1.z; cd[7] = c1.w;
            if (NCH == 1) {
                cd[8] = ow0;                                  // candidate 31 needs the first own bytes
#pragma unroll
                for (int j = 9; j < 17; j++) cd[j] = 0;      // unused: own keys double as candidates
            } else {
                const uint4 c2 = *reinterpret_cast<const uint4*>(&in[cdw + 8]);
                const uint4 c3 = *reinterpret_cast<const uint4*>(&in[cdw + 12]);
                cd[8] = c2.x; cd[9] = c2.y; cd[10] = c2.z; cd[11] = c2.w;
                cd[12] = c3.x; cd[13] = c3.y; cd[14] = c3.z; cd[15] = c3.w;
                cd[16] = in[cdw + 16];
            }
        }
        uint32_t m[RUN];
#pragma unroll
        for (int i = 0; i < RUN; i++) m[i] = 0xFFFFFFFFu;
        // candidate-major order: two candidate keys live at a time, 32 running minima.  (Round 4: pairing the odd rows as (j+1, j+2) puts
        // their 32 single v_min_u32 into v_min3 as well, -12 instructions -- but the third live key spills 8 bytes per lane: not adopted)
        static_for<0, 63>([&](auto J) {
            constexpr int j = decltype(J)::value;             // handles candidates j and j+1 (j even)
            if constexpr ((j & 1) == 0) {
                uint32_t kc0, kc1;
                if constexpr (NCH == 1 && j >= 32) {          // own position j-32 IS candidate j (same tag 4j)
                    // pin in place: without it the scheduler precomputes all ~500 own-vs-own differences
                    asm volatile("" : "+v"(ko[j - 32]), "+v"(ko[j - 31]));
                    kc0 = ko[j - 32];
                    kc1 = ko[j - 31];
                } else if constexpr (PREV_DPP) {
                    // (the builtin, not inline asm: hipcc folds the move into v_add_u32_dpp and keeps the DPP read-after-write wait states)
                    kc0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ko[j], 0x13C, 0xF, 0xF, false) + m128;          // wave_ror:1
                    kc1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ko[j + 1], 0x13C, 0xF, 0xF, false) + m128;
                } else {
                    kc0 = key3<j>(cd, (uint32_t)(4 * j));
                    kc1 = key3<j + 1>(cd, (uint32_t)(4 * (j + 1)));
                }
                // own index i pairs with candidates j in [i, i+31]
                static_for<0, RUN>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    constexpr bool use0 = (j >= i) && (j <= i + 31);
                    constexpr bool use1 = (j + 1 >= i) && (j + 1 <= i + 31);
                    if constexpr (use0 && use1) m[i] = umin3(m[i], ko[i] - kc0, ko[i] - kc1);
                    else if constexpr (use0) m[i] = min(m[i], ko[i] - kc0);
                    else if constexpr (use1) m[i] = min(m[i], ko[i] - kc1);
                });
                if constexpr ((j & 7) == 6) { pin(m); PHASE_FENCE(); }
            }
        });
        // NCH == 1: the raw minimum IS the result (>= 256 = "none"; make_tokens' position test rejects those values)
#pragma unroll
        for (int i = 0; i < RUN; i++) {
            if (NCH == 1) best[i] = m[i];
            else if (m[i] < 256u) best[i] = m[i] + 128u * (uint32_t)k;
        }
    }
}

// ---- phase 2 for WIDE windows: a window-independent match finder ------------------------------------------
// "Nearest previous occurrence of the same 3-byte string" (matcher3 x CWINDOW + first-set-bit pick, deflate.py:407-421, :966-1016)
// is a predecessor query; its cost need not grow with the window.  match_search pays 1.5 VALU instructions per (position,
// distance) pair -- 48 per position at CWINDOW = 32, 384 at 256.  Here the tile is walked in ROUNDS of 64 consecutive positions,
// one lane per position, through three LDS structures:
//   T[h]    last position with hash h so far.  ONE returning atomic per position -- old = ds_max_rtn_u32(T[h], own position) -- both
//           inserts the position and answers "previous position of my hash class", INCLUDING the up to 63 nearer members of the
//           same round, if the LDS applies the lanes of a wave instruction to one address in lane order: the positions of a round
//           rise with the lane.  That order is not architected, so it is CHECKED instead of assumed: if a later lane b is applied
//           before an earlier lane a of the same class, a reads back a value >= position(b) > its own position -- every inversion
//           shows up as a negative distance in some lane.  A group of rounds that saw one is redone by a 64-step readlane loop (the
//           minimum of the values a class got back is the table entry before the round; the nearest lower lane of the class the
//           candidate inside the round): exact whatever the hardware does, and T's final content (a maximum) never depends on the
//           order.  (Round 3 read T, ds_max-ed, read the class's LEADER back and kept a 64-bit lane map per leader slot -- six LDS
//           operations and ~15 VALU instructions per round instead of one and two: the LDS was 65 % busy, VERDICT r3 #2.)
//   E[p]    (distance from p to the previous position with the same HASH) - 1 as one byte, 255 = none or >= 256: every hash class
//           is a chain in descending position order.
// h = the top HB bits of (K * odd) mod 2^24 (K = the three bytes): a bijection of the key, well mixed in its high bits.  A lane
// follows its chain, nearest first, comparing the exact three bytes, until they are equal (the answer), or the window is left;
// positions with the same hash and another key are the only wasted steps (about CWINDOW / 2^HB per query).
// Result: best[i] = 4 * distance for own position i of the lane's RUN (the layout of all later phases; transposed through LDS),
// NO_MATCH4 = none.  The zero halo in front of a block's first tile only yields distances > p (rejected by
// make_tokens' position test; a valid candidate is always nearer and therefore found first).
template <int NCH> struct HashCfg {
#ifndef HDLZ_HASH_BITS
#define HDLZ_HASH_BITS 10
#endif
    static constexpr int HB = HDLZ_HASH_BITS;
    static constexpr int NT = 1 << HB;
    static constexpr int PRE = 64 * ((32 * NCH + 63) / 64);     // positions in front of the tile that are inserted first
    static_assert(PRE <= HALO && HB >= 8 && HB <= 16, "halo / tag position");
    // E[] is only written from byte index HALO - PRE on and an empty T[] entry reads as index 0.  A query of a tile position
    // (index >= HALO) walks at most cwindow <= 32 * NCH entries back, i.e. never below HALO - PRE -- as long as this holds:
    static_assert(32 * NCH <= PRE, "cwindow <= PRE: chain walks must stay inside the entries this tile has written (ADVICE r3)");
};
template <int NCH> struct __attribute__((aligned(16))) HashLds {
    union {
        uint32_t T[HashCfg<NCH>::NT];
        uint16_t D[TILE];                   // the results on their way from the round layout to the run layout
    };
    uint16_t E[HALO + TILE];                // per position: eight tag bits of the key << 8 | link
    static_assert(sizeof(uint32_t) * HashCfg<NCH>::NT >= sizeof(uint16_t) * TILE, "D overlays T");
};

__device__ __forceinline__ uint32_t* lds_out(WaveLds& l, uint32_t&) { return l.out; }
template <int NCH> __device__ __forceinline__ uint32_t* lds_out(WaveLdsNoOut&, HashLds<NCH>& h) {
    static_assert(sizeof(h.T) >= sizeof(uint32_t) * OUT_WORDS, "the bit buffer overlays T / D");
    return h.T;
}

template <int NCH>
__device__ __forceinline__ void match_search_hash(const uint32_t* in, HashLds<NCH>& hl, uint32_t lane, uint32_t cw, uint32_t (&best)[RUN]) {
    constexpr int HB = HashCfg<NCH>::HB, NT = HashCfg<NCH>::NT, PRE = HashCfg<NCH>::PRE;
    constexpr uint32_t base = HALO - PRE;
    // empty table (it was the transposition buffer and the bit buffer of the previous tile)
    for (uint32_t k = lane * 4u; k < (uint32_t)NT; k += 256u) *reinterpret_cast<uint4*>(&hl.T[k]) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // Rounds are batched in groups of G: the LDS executes a wave's instructions in order, so the table operations of G rounds
    // are ISSUED back to back (round r+1's T read behind round r's ds_max) and their results are consumed afterwards -- one
    // LDS round trip per group instead of four dependent ones per round (a first version that finished a round before it
    // started the next was latency-bound at 5 waves per CU: 63 k cycles per tile).
    // Costs in CU cycles per wave instruction (tools/ubench/lds_ops.hip): random ds_read_b32 7, ds_max_u32 7, ds_or_b64 11,
    // ds_read_b64 7..10, ds_write_b64 11 -- and 65 for an UNALIGNED ds_read_b32 (one lane per cycle), 11..22 for ds_write_b8: the key
    // therefore comes from two aligned dwords + v_alignbyte, and the chain entries are dwords that carry the key themselves
    constexpr int G = 8;
    static_assert(RUN % G == 0 && PRE / 64 <= G, "groups");
    uint32_t res2[RUN / 2];                  // round layout, two rounds per register: 4 * distance of position 64 r + lane, 16 bits each
    // A group's work comes in three pieces: issue (key, hash, the table operations), consume (first candidate, chain entry) and
    // walk (the chain steps of its queries).  (Issuing group g+1 before walking group g changed nothing measurable.)
    struct Grp { uint32_t kw[G], ow[G], c0[G], dd[G]; };
    auto hash_of = [&](uint32_t kw, uint32_t& tag) -> uint32_t {
        uint32_t mix;
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(mix) : "v"(kw), "v"(0x9E3779u));     // (ignores the top byte; hipcc picks the quarter-rate v_mul_lo_u32)
        tag = (mix << (HB - 8)) & 0xFF00u;                    // the eight bits BELOW the hash bits of (K * odd) mod 2^24, as the entry's tag
        return __builtin_amdgcn_ubfe(mix, 24 - HB, HB);
    };
    auto issue = [&](auto G0, auto CNT, Grp& q) {             // rounds [g0, g0 + cnt), counted from position `base`
        constexpr int g0 = decltype(G0)::value, cnt = decltype(CNT)::value;
        uint32_t w0_[cnt], w1_[cnt];
        static_for<0, cnt>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const uint32_t idx = base + 64u * (uint32_t)(g0 + j) + lane;      // byte index of the position in `in`
            w0_[j] = in[idx >> 2]; w1_[j] = in[(idx >> 2) + 1u];
        });
        static_for<0, cnt>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr uint32_t rbase = base + 64u * (uint32_t)(g0 + j);
            q.kw[j] = alignbyte(w1_[j], w0_[j], lane);                        // (rbase is a multiple of 4) the key is its low three bytes
            const uint32_t h = hash_of(q.kw[j], q.ow[j]);
            // insert the position and get the previous one of the hash class back (the rounds of a group back to back: the LDS
            // executes a wave's instructions in order)
            q.c0[j] = __hip_atomic_fetch_max(&hl.T[h], rbase + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        });
    };
    // (never taken on this hardware so far: see the head of this section) the previous position of the hash class of every lane of
    // one round, from the values the atomics returned, whatever order they were applied in
    auto reorder = [&](uint32_t rbase, uint32_t kw, uint32_t got) -> uint32_t {
        uint32_t tag;
        const uint32_t h = hash_of(kw, tag);
        uint32_t before = got, nearer = 0u;
        bool has = false;
#pragma unroll 1
        for (uint32_t s_ = 0; s_ < 64u; s_++) {
            const uint32_t hs = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)s_), gs = (uint32_t)__builtin_amdgcn_readlane((int)got, (int)s_);
            if (hs == h) {
                before = min(before, gs);
                if (s_ < lane) { nearer = rbase + s_; has = true; }
            }
        }
        return has ? nearer : before;
    };
    auto consume = [&](auto G0, auto CNT, Grp& q) {
        constexpr int g0 = decltype(G0)::value, cnt = decltype(CNT)::value;
        uint32_t inv = 0;                                     // sign bit: some lane got a position behind its own back
        static_for<0, cnt>([&](auto J) {
            constexpr int j = decltype(J)::value;
            inv |= (base + 64u * (uint32_t)(g0 + j) + lane) - q.c0[j];
        });
#ifdef HDLZ_HASH_FORCE_REORDER                     // test build (tools/r4_exp13.sh): every group takes the path the hardware never asks for
        inv = 0x80000000u;
#endif
        if (ballot64((int32_t)inv < 0) != 0ull) {
            static_for<0, cnt>([&](auto J) {
                constexpr int j = decltype(J)::value;
                q.c0[j] = reorder(base + 64u * (uint32_t)(g0 + j), q.kw[j], q.c0[j]);
            });
        }
        static_for<0, cnt>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr uint32_t rbase = base + 64u * (uint32_t)(g0 + j);
            const uint32_t idx = rbase + lane;
            const uint32_t d = idx - q.c0[j];                 // (an empty T entry reads as position 0: a distance that is either outside
                                                              //  the window or leads to a real position whose key is then compared)
            hl.E[idx] = (uint16_t)(q.ow[j] | min(d - 1u, 255u));
            q.dd[j] = (d - 1u) < cw ? d : 0u;                 // 0 = this lane has nothing (more) to look at
        });
    };
    // follow the chains, nearest first: the nearest position with the same three bytes inside the window.  The first step of
    // the group's rounds is issued together; the (rare) further steps run per round, branch-free in the body (the scalar unit
    // is shared by the CU's four SIMDs: a compare-and-branch ladder per step would make IT the limit).  An entry carries eight
    // tag bits of its key (a dword per entry with the whole key cost 9 KB of LDS = a wave per SIMD): a candidate whose tag agrees
    // is CONFIRMED against its three bytes, and the one in 256 wrong candidates that gets that far sends its lane back to the walk.
    auto walk = [&](auto G0, Grp& q) {
        constexpr int g0 = decltype(G0)::value;
        uint32_t ev_[G], found_[G];                           // found: 0 = none
        auto step = [&](uint32_t ev, uint32_t ow, uint32_t& dd, uint32_t& found) {
            const uint32_t nd = dd + (ev & 255u) + 1u;
            const bool live = dd != 0u;
            const bool same = (ev ^ ow) < 256u;
            found = (live & same) ? dd : found;
            dd = (live & !same & (nd <= cw)) ? nd : 0u;
        };
        // two steps for all rounds of the group, their reads in flight together (every dependent LDS round trip that is paid per
        // ROUND -- a step, the confirmation -- is paid 32 times per tile: at two to three waves per SIMD that latency is the kernel)
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            static_for<0, G>([&](auto J) {
                constexpr int j = decltype(J)::value;
                ev_[j] = hl.E[base + 64u * (uint32_t)(g0 + j) + lane - q.dd[j]];     // (a finished lane reads its own entry: in range, unused)
            });
            static_for<0, G>([&](auto J) {
                constexpr int j = decltype(J)::value;
                if (pass == 0) found_[j] = 0u;
                step(ev_[j], q.ow[j], q.dd[j], found_[j]);
            });
        }
        // the chains that are longer, round by round
        static_for<0, G>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const uint32_t idx = base + 64u * (uint32_t)(g0 + j) + lane;
            while (ballot64(q.dd[j] != 0u) != 0ull) step(hl.E[idx - q.dd[j]], q.ow[j], q.dd[j], found_[j]);
        });
        // confirm the candidates: the three bytes at idx - found against the own ones (none: the own position, trivially equal)
        uint32_t c0_[G], c1_[G];
        static_for<0, G>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const uint32_t a = base + 64u * (uint32_t)(g0 + j) + lane - found_[j];
            c0_[j] = in[a >> 2]; c1_[j] = in[(a >> 2) + 1u];
        });
        uint32_t anybad = 0;
        static_for<0, G>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const uint32_t a = base + 64u * (uint32_t)(g0 + j) + lane - found_[j];
            ev_[j] = (alignbyte(c1_[j], c0_[j], a) ^ q.kw[j]) << 8;          // != 0: a wrong candidate with the right tag
            anybad |= ev_[j];
        });
        if (ballot64(anybad != 0u) != 0ull) {
            // (rare: one candidate in 256 wrong ones) the lane goes on behind it -- serial walk and confirmation
            static_for<0, G>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const uint32_t idx = base + 64u * (uint32_t)(g0 + j) + lane;
                bool bad = ev_[j] != 0u;
                while (ballot64(bad) != 0ull) {
                    uint32_t dd = 0u;
                    if (bad) {
                        const uint32_t nd = found_[j] + ((uint32_t)hl.E[idx - found_[j]] & 255u) + 1u;
                        dd = nd <= cw ? nd : 0u;
                        found_[j] = 0u;
                    }
                    while (ballot64(dd != 0u) != 0ull) step(hl.E[idx - dd], q.ow[j], dd, found_[j]);
                    const uint32_t a = idx - found_[j];
                    bad = bad && ((alignbyte(in[(a >> 2) + 1u], in[a >> 2], a) ^ q.kw[j]) << 8) != 0u;
                }
            });
        }
        static_for<0, G>([&](auto J) {
            constexpr int j = decltype(J)::value;
            constexpr int r = g0 + j - PRE / 64;
            const uint32_t f = found_[j] ? found_[j] : (NO_MATCH4 >> 2);
            if constexpr ((r & 1) == 0) res2[r >> 1] = f << 2;
            else res2[r >> 1] |= f << 18;
        });
    };
    {
        Grp q;
        issue(std::integral_constant<int, 0>{}, std::integral_constant<int, PRE / 64>{}, q);      // the positions in front of the tile
        consume(std::integral_constant<int, 0>{}, std::integral_constant<int, PRE / 64>{}, q);
        static_for<0, RUN / G>([&](auto GI) {
            constexpr int g0 = PRE / 64 + decltype(GI)::value * G;
            issue(std::integral_constant<int, g0>{}, std::integral_constant<int, G>{}, q);
            consume(std::integral_constant<int, g0>{}, std::integral_constant<int, G>{}, q);
            walk(std::integral_constant<int, g0>{}, q);
            pin(res2);
            PHASE_FENCE();
        });
    }
    // round layout -> run layout.  res2[k] holds rounds 2k (low half) and 2k+1 of this lane: written as ONE dword at [k][lane]
    // (16 conflict-free ds_write_b32 instead of 32 ds_write_b16); position p = 64 r + j sits in dword (r >> 1) * 64 + j, half r & 1,
    // so the run of lane l -- positions 32 l .. 32 l + 31, all of round l >> 1 -- is 32 consecutive dwords and one half of each
    __syncthreads();
    uint32_t* D32 = reinterpret_cast<uint32_t*>(hl.D);
#pragma unroll
    for (int k = 0; k < RUN / 2; k++) D32[64 * k + lane] = res2[k];
    __syncthreads();
    const uint32_t dbase = (lane >> 2) * 64u + 32u * (lane & 1u), hsh = 16u * ((lane >> 1) & 1u);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint4 v = *reinterpret_cast<const uint4*>(&D32[dbase + 4u * q]);
        best[4 * q + 0] = __builtin_amdgcn_ubfe(v.x, hsh, 16u); best[4 * q + 1] = __builtin_amdgcn_ubfe(v.y, hsh, 16u);
        best[4 * q + 2] = __builtin_amdgcn_ubfe(v.z, hsh, 16u); best[4 * q + 3] = __builtin_amdgcn_ubfe(v.w, hsh, 16u);
    }
    __syncthreads();
}

// which kernels use the hash finder, and the waves per SIMD they are register-capped for
#ifndef HDLZ_WH
#define HDLZ_WH 3                               // waves per SIMD of the kernels with the hash finder (13.6 KB of LDS per wave: 12 per CU)
#endif
template <int NCH> constexpr bool wide_hash() { return NCH > 1; }      // every window > 32 (round 4: CWINDOW = 64 too, 279 against 246 GB/s for the one-pass brute force, which is gone)
template <int NCH> constexpr int waves_eu() { return NCH == 1 ? HDLZ_W1 : HDLZ_WH; }


// ---- phase 3, eligibility + extension (R3/R5; SEARCHF / SEARCH10, deflate.py:899-964, :1018-1062):
// tok[i] = 4 * (len-1) << 16 | LUT byte offset of the token  (len-1 = 0 for a literal).
//   lds_run  byte offset of the run in `in`;  nrem = positions of the block from the run's first position on (0 = none)
//   p4_run   4 * min(block-relative position of the run, 32 * NCH): with that clamp `d4 <= p4_run + 4i` is at once R4's
//            d <= p, the "a candidate exists" test AND -- for NCH == 1, where best[] holds raw minima -- the rejection of
//            the >= 256 values that stand for "no match" (p4_run + 4i <= 128 + 124)
//   FULLWIN  cwindow == 32 * NCH: every distance the search can return is inside the window (no cw4 compare)
// Measured and dropped (profiles/r02_tokens_ab.txt): the candidate's 8 bytes from ONE unaligned ds_read_b64 -- 8 VALU
// instructions fewer per position, but gfx950's LDS splits unaligned 64-bit reads: LDS busy 57 -> 82 %, 1 % slower.
//   SCALAR_TAIL: the lane runs are consecutive pieces of ONE block and tile_rem = positions of the block from the tile's first
//            position on: R3's "p <= N-5" then holds for a PREFIX of the lanes, so it is evaluated on the scalar unit --
//            at most two different lane masks per tile -- and and-ed into the compare's lane mask (else: per lane, from nrem)
//   GMASK    dword-index mask of the extension's gather: the gather of a position WITHOUT a candidate (NCH == 1: a raw minimum up to
//            2^32 - 1 stands for "none") would address LDS far outside `in`; masked, it stays inside the first 4 (GMASK + 3) bytes behind
//            `in` -- which every caller's LDS block covers (static_assert there) -- and its result is discarded as before.  The mask
//            replaces the ~3 of the dword address: no instruction more.  (an out-of-allocation LDS read.)
template <int NCH, bool FULLWIN, bool SCALAR_TAIL = false, uint32_t GMASK = 0x3FFu>
__device__ __forceinline__ void make_tokens(const uint32_t* in, uint32_t lds_run, uint32_t (&ow)[12], uint32_t (&best)[RUN],
                                            uint32_t cw4, uint32_t kmax, uint32_t p4_run, uint32_t nrem, uint32_t (&tok)[RUN],
                                            int32_t tile_rem = 0) {
    pin(best); pin(ow);
    PHASE_FENCE();
    const uint32_t nrem_m5 = nrem - 5u;                       // (wraps when nrem < 5: then nothing is eligible)
    const uint32_t kmax_m3 = kmax - 3u;
    // lanes l with tile_rem - 32 l >= i + 5  <=>  l < cnt_i,  cnt_i = ((tile_rem - 5 - i) >> 5) + 1: two values over i = 0..31
    const int32_t rem5 = tile_rem - 5;
    const uint32_t thr = rem5 >= 0 ? ((uint32_t)rem5 & 31u) : 0u;                        // i <= thr: the larger count
    const int32_t cnt_hi = rem5 >= 0 ? (rem5 >> 5) + 1 : 0, cnt_lo = rem5 >= 0 ? (rem5 >> 5) : 0;
    const uint64_t m_hi = cnt_hi >= 64 ? ~0ull : ((1ull << cnt_hi) - 1ull), m_lo = cnt_lo >= 64 ? ~0ull : ((1ull << cnt_lo) - 1ull);
    // token word of a match of length len = l3 + 3 at distance d:  4 * (len-1) << 16 | LUT offset
    //   CWINDOW <= 32: LUT [len-3][d-1] -> l3 * (4 * 65536 + 128) + 4d + K;   wider: LUT [d-1] -> l3 * 4 * 65536 + 4d + K
    // (the upper half holds 4 * (len-1): the parse shifts its nibble table by exactly that, the LUT_LEN offset is exactly that, and a plain
    //  v_lshrrev is a full-rate instruction where the shift-and-scale forms are not)
    const uint32_t tok_k = (2u << 18) + LUT_MATCH_BYTE - 4u;
    const uint32_t tok_mul = NCH == 1 ? (4u * 65536u + 128u) : 4u * 65536u;
    static_for<0, RUN>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t d4 = best[i];
        // R3: 1 <= p <= N-5;  R4: d <= min(CWINDOW, p)   ('&': no short-circuit branches)
        uint64_t okm = __builtin_amdgcn_ballot_w64(d4 <= p4_run + (uint32_t)(4 * i));      // (the compare's own lane mask)
        if constexpr (!FULLWIN) okm &= __builtin_amdgcn_ballot_w64(d4 <= cw4);
        if constexpr (SCALAR_TAIL) okm &= ((uint32_t)i <= thr ? m_hi : m_lo);
        else okm &= __builtin_amdgcn_ballot_w64(nrem >= (uint32_t)(i + 5));
        // distance for the gather; for "no match" any in-range value will do (the result is discarded)
        // (NCH == 1: a plain shift -- v_lshrrev is a full-rate instruction, v_bfe / v_and + shift are not, profiles/r04_ubench/ubench_valu_cycles2.txt;
        //  a raw minimum that means "none" then gathers at some masked address inside the caller's LDS block, see GMASK; the result is discarded)
        const uint32_t d = (NCH == 1) ? (d4 >> 2) : __builtin_amdgcn_ubfe(d4, 2u, 9u);      // (NO_MATCH4 reads as 1)
        // R5: common prefix of x[p+3..p+10] and x[p-d+3..p-d+10]
        const uint32_t q = lds_run + (uint32_t)(i + 3) - d;   // byte offset of the candidate's 4th byte
        const uint32_t qd = (q >> 2) & GMASK;
        const uint32_t a0 = in[qd], a1 = in[qd + 1], a2 = in[qd + 2];
        const uint32_t clo = alignbyte(a1, a0, q), chi = alignbyte(a2, a1, q);      // (v_alignbyte takes the shift from q[1:0])
        constexpr int o = i + 3;
        uint32_t olo, ohi;
        if constexpr ((o & 3) == 0) { olo = ow[o >> 2]; ohi = ow[(o >> 2) + 1]; }
        else { olo = alignbyte(ow[(o >> 2) + 1], ow[o >> 2], o & 3); ohi = alignbyte(ow[(o >> 2) + 2], ow[(o >> 2) + 1], o & 3); }
        // equal low BITS of the two 8-byte windows.  Only 7 bytes can matter (len <= 10): a sentinel in the top bit of
        // the eighth bounds the count at 63 (-> 7 bytes) and keeps ffbl away from its "no bit set" value
        const uint32_t zhi = ffbl((chi ^ ohi) | 0x80000000u) | 32u;
        uint32_t zlo;                                            // v_ffbl_b32(0) = 0xFFFFFFFF: "no difference in the low half"
        asm("v_ffbl_b32 %0, %1" : "=v"(zlo) : "v"(clo ^ olo));   // (asm: hipcc turns ffs()-1 + min into a compare and a select)
        uint32_t zb;                                             // min(zlo, zhi): zhi <= 63 and zlo is a bit index or 0xFFFFFFFF, so the 16-bit
        asm("v_min_u16 %0, %1, %2" : "=v"(zb) : "v"(zlo), "v"(zhi));   // minimum is the minimum (full rate; v_min_u32 is not), upper half zero
        // len - 3 = min(equal bytes, Kmax - 3, N-2-p - 3): a match never covers the last two bytes
        const uint32_t l3 = umin3(zb >> 3, kmax_m3, nrem_m5 - (uint32_t)i);
        // literal byte -> LUT offset 4*byte
        constexpr int bsh = 8 * (i & 3);
        uint32_t lit;
        if constexpr (bsh == 0) lit = (ow[i >> 2] << 2) & 0x3FCu;
        else lit = (ow[i >> 2] >> (bsh - 2)) & 0x3FCu;
        uint32_t mt;     // l3 * tok_mul + d4 + tok_k  (hipcc folds the C form into a quarter-rate v_mul_lo_u32)
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(mt) : "v"(l3), "s"(tok_mul), "v"(d4 + tok_k));
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(tok[i]) : "v"(lit), "v"(mt), "s"(okm));      // tok = ok ? mt : lit
        // (positions per scheduling group: the gathers of a group are in flight together; profiles/r05_tile_timing.txt)
        constexpr int EG = HDLZ_EXT_GROUP;
        if constexpr ((i & (EG - 1)) == EG - 1) { pin_range<(i & ~(EG - 1)), (i & ~(EG - 1)) + EG>(tok); PHASE_FENCE(); }
    });
}

// ---- phase 4a, greedy parse of one run ("di += m / di += 1", deflate.py:960,1008 -- a serial chain in the reference) folded
// into a TRANSFER FUNCTION "entry skip (0..9) -> exit skip": nibble s of the result = exit skip for entry skip s.
// Backward pass: E[i] = exit skip if a token starts at local index i; nibbles of P hold E[i+1..i+10].
__device__ __forceinline__ uint64_t run_transfer(const uint32_t (&tok)[RUN]) {
    uint64_t P = 0x9876543210ull;
#pragma unroll
    for (int i = RUN - 1; i >= 0; i--) {
        const uint32_t sh = tok[i] >> 16;                     // 4 * (len-1)
        const uint32_t e = (uint32_t)(P >> sh) & 15u;
        P = (P << 4) | e;
    }
    return P;
}

// ---- phase 4b: compose the 64 run functions across the wave.  s: in = entry skip of lane 0, out = exit skip of lane 63; returns this
// lane's entry skip.
// Round 5: a run function is almost always CONSTANT -- whatever the entry skip, the parse of 32 positions falls into step at the first
// literal or cut-short match (0 / 0 / 0 / 4 % of the runs of the four bench families are not constant; data whose matches follow each
// other at full length through the whole run -- zeros, short periods -- is what keeps a function a rotation).  A lane behind a constant
// function knows its entry skip at once, a lane behind a chain of k others after k steps of "take the previous lane's exit through my own
// function" (one DPP move + a nibble pick per step, all lanes at once).  Up to CHAIN_STEPS such steps; a wave that still has an open lane
// then takes the serial composition on the scalar unit, which is exact for every input (64 steps of ~6 dependent scalar instructions:
// 7.7 k of the 52.6 k cycles a wave spent per tile, profiles/r05_tile_timing.txt -- a wave in that loop issues nothing else).
constexpr int CHAIN_STEPS = 4;
__device__ __forceinline__ uint32_t chain_skips_serial(uint64_t P, uint32_t lane, uint32_t& s);
__device__ __forceinline__ uint32_t chain_skips(uint64_t P, uint32_t lane, uint32_t& s) {
    const uint32_t plo = (uint32_t)P, phi = (uint32_t)(P >> 32) & 0xFFu;
#ifndef HDLZ_CHAIN_SERIAL_ONLY
    // all ten nibbles equal  <=>  P ^ (P >> 4) has no bit in its low 36
    const uint32_t dl = plo ^ __builtin_amdgcn_alignbit(phi, plo, 4u), dh = (phi ^ (phi >> 4)) & 15u;
    uint32_t x = (dl | dh) == 0u ? (plo & 15u) : 16u;           // this lane's EXIT skip, 16 = not known yet
#ifdef HDLZ_CHAIN_FORCE_SERIAL                                   // test build: every wave takes the scalar composition
    x = 16u;
#endif
    bool open = ballot64(x >= 16u) != 0ull;
#pragma unroll 1
    for (int it = 0; it < CHAIN_STEPS && open; it++) {
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)s, (int)x, 0x138, 0xF, 0xF, false);      // wave_shr:1; lane 0 keeps s
        const uint32_t via = (uint32_t)(P >> (4u * (prev & 15u))) & 15u;
        x = (x >= 16u && prev < 16u) ? via : x;
        open = ballot64(x >= 16u) != 0ull;
    }
    if (!open) {
        const uint32_t e = (uint32_t)__builtin_amdgcn_update_dpp((int)s, (int)x, 0x138, 0xF, 0xF, false);
        s = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        return e;
    }
#endif
    return chain_skips_serial(P, lane, s);
}
__device__ __forceinline__ uint32_t chain_skips_serial(uint64_t P, uint32_t lane, uint32_t& s) {
    const uint32_t plo = (uint32_t)P;
    // the upper part of a function is 8 bits (entry skips 8 and 9): the four of a quad of lanes are packed into ONE register first (two
    // quad-permute moves), so the scalar chain reads 64 + 16 lanes instead of 2 x 64 (v_readlane is a 3.25-cycle instruction)
    uint32_t ph4 = (uint32_t)(P >> 32) & 0xFFu;
    ph4 |= (uint32_t)__builtin_amdgcn_mov_dpp((int)ph4, 0xB1, 0xF, 0xF, true) << 8;      // quad_perm [1,0,3,2]: lanes 0 / 2 of a quad get their neighbour
    ph4 |= (uint32_t)__builtin_amdgcn_mov_dpp((int)ph4, 0x4E, 0xF, 0xF, true) << 16;     // quad_perm [2,3,0,1]: lane 0 gets the pair of lanes 2, 3
    uint64_t sv[4] = {0, 0, 0, 0};     // entry skips of all 64 lanes, one nibble each (scalar regs)
    // 4 segments of 16 lanes; the scheduling barriers keep the compiler from hoisting all the
    // readlanes to the top (that needed ~260 SGPR spills = v_writelane/v_readlane traffic)
    static_for<0, 4>([&](auto G) {
        constexpr int g = decltype(G)::value;
        uint64_t acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            // NB: readlane returns a signed int -- cast before widening or bit 31 smears into the high half
            const uint32_t hi4 = (uint32_t)__builtin_amdgcn_readlane((int)ph4, g * 16 + q * 4);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int l = q * 4 + k;
                acc |= (uint64_t)s << (4 * l);
                const uint64_t f = ((uint64_t)((hi4 >> (8 * k)) & 0xFFu) << 32) |
                                   (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)plo, g * 16 + l);
                s = (uint32_t)(f >> (4u * s)) & 15u;
            }
        }
        sv[g] = acc;
        __builtin_amdgcn_sched_barrier(0);
    });
    const uint32_t g = lane >> 4;
    const uint64_t mine = g == 0 ? sv[0] : g == 1 ? sv[1] : g == 2 ? sv[2] : sv[3];
    return (uint32_t)(mine >> (4u * (lane & 15u))) & 15u;
}

// ---- phase 5a, token bits (R6/R7; DISTANCE deflate.py:836-882, literal emit :1005-1016) from the per-wave LDS LUTs:
// code[i] = LUT entry (code | nbits << 27) of every token START, 0 elsewhere; returns the lane's bit count.
//   c0      this lane's entry skip (a value >= 32 = the lane starts no token at all)
//   LIMIT   only positions i < nlimit may start a token (padding positions of a packed small block)
template <int NCH, bool LIMIT>
__device__ __forceinline__ uint32_t token_codes(const uint8_t* lut8, uint32_t (&tok)[RUN], uint32_t c0, uint32_t nlimit,
                                                uint32_t (&code)[RUN]) {
    uint32_t lane_bits = 0;
    uint32_t c = 4u * c0;                                     // positions still covered by the last token, times four (the token word's unit)
    static_for<0, RUN>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const uint32_t e = *reinterpret_cast<const uint32_t*>(lut8 + (tok[i] & 0xFFFFu));
        bool start = (c == 0u);
        if constexpr (LIMIT) start = start & ((uint32_t)i < nlimit);
        const uint32_t lenm1x4 = tok[i] >> 16;
        c = start ? lenm1x4 : (c - 4u);
        uint32_t ee = e;
        if constexpr (NCH != 1)                               // wide windows: [dist] LUT + the length code from its own small LUT
            ee |= *reinterpret_cast<const uint32_t*>(lut8 + LUT_LEN_BYTE + lenm1x4);        // (computed: six VALU instructions per position)
        code[i] = start ? ee : 0u;
        lane_bits += code[i] >> NB_SHIFT;
        constexpr int CG = HDLZ_CODE_GROUP;
        if constexpr ((i & (CG - 1)) == CG - 1) { pin_range<(i & ~(CG - 1)), (i & ~(CG - 1)) + CG>(code); asm volatile("" : "+v"(c), "+v"(lane_bits)); PHASE_FENCE(); }
    });
    return lane_bits;
}

// inclusive wave scan (all 64 lanes must call it): Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8; a lane without a source adds 0),
// then the row totals across (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3) -- six DPP adds; the shuffle form
// (__shfl_up = ds_bpermute + compare + select per step) was 38 VALU + 6 LDS instructions
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, uint32_t) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);      // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);     // row_bcast15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);     // row_bcast31 -> rows 2, 3
    return v;
}

// ---- phase 5b, the bit writer (put / do_flush, deflate.py:535-567): OR every token into the LDS bit buffer at its own
// bit offset (ds_or_b32), the lane's first token at bit `bp`.
// The codes of two consecutive positions are joined first and go out as ONE value: 32 ds_or pairs per run instead of 64, 11 instead
// of 14 VALU instructions per pair.  A pair never holds two match tokens -- a match covers at least three positions, so the position
// behind a match start starts nothing --: at most a 9-bit literal and a match of 7 + 5 + 6 bits (CWINDOW 256), 27 bits in a 32-bit
// word.  (Round 4, first half: CWINDOW <= 32 only, reasoned from two 15-bit tokens; the bound above holds for every window.)
__device__ __forceinline__ void scatter_codes(uint8_t* out8, const uint32_t (&code)[RUN], uint32_t bp) {
#pragma unroll
    for (int i = 0; i < RUN; i += 2) {
        const uint32_t n0 = code[i] >> NB_SHIFT;
        const uint32_t c = ((code[i + 1] & CODE_MASK) << n0) | (code[i] & CODE_MASK);
        const uint64_t v = (uint64_t)c << (bp & 31u);
        uint32_t* w = reinterpret_cast<uint32_t*>(out8 + ((bp >> 3) & ~3u));
        __hip_atomic_fetch_or(w, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or(w + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bp += n0 + (code[i + 1] >> NB_SHIFT);
        if ((i & 6) == 6) { asm volatile("" : "+v"(bp)); PHASE_FENCE(); }
    }
}

// ---- phase 6, Adler-32 partials of the run (deflate.py:826-831, :888-897): sa = sum x_i, sc = sum i * x_i (i = 0..31)
__device__ __forceinline__ void adler_run(const uint32_t (&ow)[12], uint32_t& sa, uint32_t& sc) {
    sa = 0; sc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        sa = __builtin_amdgcn_sad_u8(ow[k], 0u, sa);
        const uint32_t wts = (uint32_t)(4 * k) | ((uint32_t)(4 * k + 1) << 8) | ((uint32_t)(4 * k + 2) << 16) | ((uint32_t)(4 * k + 3) << 24);
        sc = __builtin_amdgcn_udot4(ow[k], wts, sc, false);
    }
}

// sum of v over the 64 lanes, as a wave-uniform value: four DPP adds inside the rows of 16 (quad permutes, half mirror, mirror: every lane
// then holds its row's sum) and four v_readlane -- the xor butterfly through ds_bpermute was 36 VALU + 12 LDS instructions for two sums
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);       // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);      // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true);      // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// zero the bit buffer of a tile and seed its first word (16-byte stores: 3 LDS instructions per lane instead of 10)
__device__ __forceinline__ void zero_bit_buffer(uint32_t* lout, uint32_t lane, uint32_t carry_word) {
    static_assert(OUT_WORDS % 4 == 0, "16-byte stores");
    for (uint32_t q = lane; q < (uint32_t)OUT_WORDS / 4u; q += 64u)
        *reinterpret_cast<uint4*>(lout + 4u * q) = make_uint4(q == 0u ? carry_word : 0u, 0u, 0u, 0u);
}

// fill the per-wave LUTs (literal: [byte] -> code|nbits; match: [len-3][dist-1] (CWINDOW <= 32) or [dist-1])
template <int NCH>
__device__ __forceinline__ void fill_luts(uint32_t* lut, uint32_t lane) {
    for (uint32_t e = lane; e < (uint32_t)LUT_LIT; e += 64) lut[e] = literal_entry(e);
    for (uint32_t e = lane; e < (uint32_t)LUT_MATCH; e += 64) {
        if (NCH == 1) lut[LUT_LIT + e] = dist_entry((e & 31u) + 1u) | length_code((e >> 5) + 3u);
        else lut[LUT_LIT + e] = dist_entry(e + 1u);
    }
    if (lane < (uint32_t)LUT_LEN) lut[LUT_LIT + LUT_MATCH + lane] = lane ? length_code(lane + 1u) : 0u;      // index len-1; 0: a literal
}

// one 16-byte chunk of a block at position p (p < n), from a source of any alignment; bytes at or beyond n read as zero
__device__ __forceinline__ uint4 load_chunk16(const uint8_t* __restrict__ src, uint32_t p, uint32_t n, bool aligned16, uint32_t mis) {
    uint4 v;
    if (aligned16) {
        v = *reinterpret_cast<const uint4*>(src + p);
    } else {
        // realign with aligned dword loads + v_alignbyte
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + p - mis);
        const uint32_t nd = (n - p + mis + 3u) >> 2;      // dwords that hold valid bytes
        uint32_t d0 = q[0];
        uint32_t d1 = nd > 1 ? q[1] : 0, d2 = nd > 2 ? q[2] : 0, d3 = nd > 3 ? q[3] : 0, d4 = nd > 4 ? q[4] : 0;
        v.x = alignbyte(d1, d0, mis);
        v.y = alignbyte(d2, d1, mis);
        v.z = alignbyte(d3, d2, mis);
        v.w = alignbyte(d4, d3, mis);
    }
    const uint32_t valid = n - p;                // bytes of this chunk inside the block
    if (valid < 16u) {                           // positions >= N must read as zero bytes
        uint32_t* vv = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lo = 4u * k;
            const uint32_t m = valid <= lo ? 0u : (valid >= lo + 4u ? 0xFFFFFFFFu : ((1u << (8u * (valid - lo))) - 1u));
            vv[k] &= m;
        }
    }
    return v;
}


// ---- stage the tile that starts at position t0: [t0 - 256, t0 + 2048 + 16) straight from HBM (no carried halo: tiles are independent)
__device__ __forceinline__ void stage_tile(uint8_t* lin8, const uint8_t* __restrict__ src, uint32_t t0, uint32_t n, bool aligned16,
                                           uint32_t mis, uint32_t lane) {
    __syncthreads();
    const uint32_t nchunk = (HALO + TILE + LOOKAHEAD) / 16;          // 145 16-byte chunks
    for (uint32_t c = lane; c < nchunk; c += 64) {
        const int64_t p = (int64_t)t0 - HALO + (int64_t)c * 16;       // first position of the chunk
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p >= 0 && p < (int64_t)n) v = load_chunk16(src, (uint32_t)p, n, aligned16, mis);
        *reinterpret_cast<uint4*>(lin8 + c * 16u) = v;
    }
}


}  // namespace hdlz
