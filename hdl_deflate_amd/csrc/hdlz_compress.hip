// hdlz_compress.hip -- STARTC for a batch of independent blocks on gfx950 (CDNA4, wave64).
//
// Replaces the reference's compress FSM (/root/reference/deflate.py:734-1082 + :407-515 +
// :535-567) with a data-parallel formulation.  Rule names R0..R9 are SURVEY.md 8(a)'s.
//
// Mapping: ONE WAVE PER BLOCK, the block is walked in tiles of 2048 positions; lane l owns
// the RUN of 32 consecutive positions [32l, 32l+32) of the tile.
//   1. tile + 256-byte look-back halo + 16-byte look-ahead staged in LDS (coalesced 16-B loads)
//   2. match search (R3/R4): each lane builds keys K = (3-byte-string << 6) | window_index for
//      its own 32 positions and the 32..256 positions before them, all in VGPRs.  For an own
//      key Ko and a candidate key Kc,  Ko - Kc  equals the distance d in [1,32] iff the three
//      bytes are equal and is >= 64 (as u32) otherwise, so  min over the 32 candidates  IS the
//      nearest matching distance: one v_sub + half a v_min3 per compare, no branches.
//   3. extension (R5): 8-byte LDS gather at p-d+3, xor with the own bytes, count-trailing-zeros.
//   4. greedy parse (R8a "di += m / di += 1"): every lane folds its run into a transfer
//      function "entry skip (0..9) -> exit skip", 10 nibbles packed in 40 bits, computed by a
//      backward pass; a 64-step readlane chain composes them across the wave; a forward pass
//      then marks the token starts.
//   5. fixed-Huffman token bits (R6/R7), in-lane prefix sums + wave scan -> bit offsets,
//      per-lane 64-bit accumulator packing with ds_or_b32 into an LDS bit buffer,
//      coalesced dword flush to HBM; partial word carried to the next tile.
//   6. Adler-32 (R8) by per-lane byte sums / index-weighted sums (v_sad_u8 / v_dot4_u32_u8).
// There is no MFMA here: nothing is a dense contraction (HBM/VALU-bound byte work).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"

namespace hdlz {

constexpr int RUN = 32;             // positions per lane
constexpr int TILE = 64 * RUN;      // 2048 positions per wave-tile
constexpr int HALO = 256;           // bytes kept in front of the tile (max CWINDOW)
constexpr int LOOKAHEAD = 16;       // bytes staged behind the tile (need p+9 and p+2)
constexpr int IN_BYTES = HALO + TILE + LOOKAHEAD;   // 2320
constexpr int OUT_WORDS = 592;      // 9 bits * 2048 = 576 words + carry word + slack
constexpr uint32_t ADLER_MOD = 65521u;
#ifndef WAVES_PER_EU
#define WAVES_PER_EU 1
#endif
// fence for the instruction scheduler: keeps independent phases from being overlapped (which blew the
// VGPR budget to 239 and spilled ~200 SGPR lane masks in the first version)
#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
// value fence: an empty asm that "redefines" each element pins producers before / consumers after this
// point in program order (no instruction is emitted)
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

struct __attribute__((aligned(16))) WaveLds {
    uint32_t in[IN_BYTES / 4];      // byte index = position - tile_start + HALO
    uint32_t out[OUT_WORDS];        // bit buffer of the current tile
};

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    // bytes [sh, sh+4) of the 8-byte value hi:lo   (v_alignbyte_b32)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t t = a < b ? a : b;
    return t < c ? t : c;
}

// key of the 3-byte string starting at byte `j` of the dword array d[], tagged with `tag` (<64)
template <int J>
__device__ __forceinline__ uint32_t key3(const uint32_t* d, uint32_t tag) {
    constexpr int w = J >> 2, sh = J & 3;
    uint32_t t;
    if (sh == 0) t = d[w] & 0xFFFFFFu;
    else if (sh == 1) t = d[w] >> 8;
    else t = alignbyte(d[w + 1], d[w], sh) & 0xFFFFFFu;
    return (t << 6) | tag;
}

// ---- fixed Huffman token bits -------------------------------------------------------------
// literal (R7, deflate.py:1005-1016 + out_codes :112-149): sym<144 -> 8 bits rev8(0x30+sym),
// else 9 bits rev9(0x100+sym)
__device__ __forceinline__ void literal_bits(uint32_t b, uint32_t& code, uint32_t& nb) {
    const bool big = b >= 144u;
    const uint32_t v = big ? (0x100u + b) : (0x30u + b);
    nb = big ? 9u : 8u;
    code = __builtin_bitreverse32(v) >> (32u - nb);
}
// match (R6, deflate.py:836-882): 7-bit length code for symbol 254+m (no extra bits for
// m<=10), then rev5(dist code) | extra<<5 in 5+eb bits
__device__ __forceinline__ void match_bits(uint32_t m, uint32_t d, uint32_t& code, uint32_t& nb) {
    const uint32_t lcode = __builtin_bitreverse32(m - 2u) >> 25;   // 7 bits
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
    const uint32_t dcode = __builtin_bitreverse32(c) >> 27;        // 5 bits
    code = lcode | (dcode << 7) | (extra << 12);
    nb = 12u + eb;
}

template <int NCH>   // NCH = ceil(cwindow / 32): 1, 2 or 8 chunks of 32 candidate distances
__global__ __launch_bounds__(64, WAVES_PER_EU) void k_compress(CompressArgs a) {
    __shared__ WaveLds lds;
    const uint32_t lane = threadIdx.x;
    const uint64_t blk = blockIdx.x;
    if (blk >= a.nblocks) return;

    uint64_t off;
    uint32_t n;
    if (a.in_off) {
        off = a.in_off[blk];
        n = (uint32_t)(a.in_off[blk + 1] - off);
    } else {
        off = blk * a.in_pitch;
        n = a.in_len;
    }
    const uint8_t* __restrict__ src = a.in + off;
    uint32_t* __restrict__ outw = reinterpret_cast<uint32_t*>(a.out + blk * a.out_pitch);

    if (n < 5u) {                               // R0: the reference never starts
        if (lane == 0) { a.out_len[blk] = 0; a.status[blk] = HDLZ_E_SHORT_INPUT; }
        return;
    }
    if ((uint64_t)out_bound(n) > a.out_pitch) {
        if (lane == 0) { a.out_len[blk] = 0; a.status[blk] = HDLZ_E_OUT_CAPACITY; }
        return;
    }
    const uint32_t cw = (uint32_t)a.cwindow;
    const uint32_t kmax = (uint32_t)a.maxmatch;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
    const bool aligned16 = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;

    uint8_t* lin8 = reinterpret_cast<uint8_t*>(lds.in);
    uint32_t gw = 0;            // output words already flushed to HBM
    uint32_t base_bits = 19;    // R1: 78 9C + bits 1,1,0
    uint32_t carry_word = 0x78u | (0x9Cu << 8) | (0x3u << 16);
    uint32_t skip_in = 0;       // positions at the tile start covered by the previous tile's last match
    uint32_t ad_a = 0, ad_w = 0;   // per-lane Adler partials (sum x, sum (N-p) x mod 65521)

    for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
        // ------------------------------------------------------------------ 1. stage the tile
        uint32_t keep = 0;
        if (t0 != 0) keep = lds.in[(TILE / 4) + lane];      // last HALO bytes of the previous tile
        __syncthreads();
        lds.in[lane] = keep;                                 // tile 0: zero halo (never matched: d <= p)
        {
            const uint32_t nchunk = (TILE + LOOKAHEAD) / 16;     // 129 16-byte chunks
            for (uint32_t c = lane; c < nchunk; c += 64) {
                const uint32_t p = t0 + c * 16u;                 // first position of the chunk
                uint4 v = make_uint4(0, 0, 0, 0);
                if (p < n) {
                    if (aligned16) {
                        v = *reinterpret_cast<const uint4*>(src + p);
                    } else {
                        // realign with aligned dword loads + v_alignbyte
                        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + p - mis);
                        const uint32_t nd = (n - p + mis + 3u) >> 2;      // dwords that hold valid bytes
                        uint32_t d0 = q[0];
                        uint32_t d1 = nd > 1 ? q[1] : 0, d2 = nd > 2 ? q[2] : 0, d3 = nd > 3 ? q[3] : 0,
                                 d4 = nd > 4 ? q[4] : 0;
                        v.x = alignbyte(d1, d0, mis);
                        v.y = alignbyte(d2, d1, mis);
                        v.z = alignbyte(d3, d2, mis);
                        v.w = alignbyte(d4, d3, mis);
                    }
                    const uint32_t valid = n - p;                // bytes of this chunk inside the block
                    if (valid < 16u) {                           // zero the tail (R5 clamp relies on N, not on data)
                        uint32_t* vv = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t lo = 4u * k;
                            uint32_t m = valid <= lo ? 0u : (valid >= lo + 4u ? 0xFFFFFFFFu : ((1u << (8u * (valid - lo))) - 1u));
                            vv[k] &= m;
                        }
                    }
                }
                *reinterpret_cast<uint4*>(lin8 + HALO + c * 16u) = v;
            }
        }
        // zero the bit buffer, seed the carry
        for (uint32_t w = lane; w < OUT_WORDS; w += 64) lds.out[w] = (w == 0) ? carry_word : 0u;
        __syncthreads();

        // ------------------------------------------------------------------ 2. match search
        const uint32_t run_dw = (HALO / 4) + lane * (RUN / 4);   // dword index of the run in lds.in
        uint32_t ow[12];                                          // own 32 bytes + 16 look-ahead
        {
            const uint4 v0 = *reinterpret_cast<const uint4*>(&lds.in[run_dw]);
            const uint4 v1 = *reinterpret_cast<const uint4*>(&lds.in[run_dw + 4]);
            const uint4 v2 = *reinterpret_cast<const uint4*>(&lds.in[run_dw + 8]);
            ow[0] = v0.x; ow[1] = v0.y; ow[2] = v0.z; ow[3] = v0.w;
            ow[4] = v1.x; ow[5] = v1.y; ow[6] = v1.z; ow[7] = v1.w;
            ow[8] = v2.x; ow[9] = v2.y; ow[10] = v2.z; ow[11] = v2.w;
        }
        uint32_t ko[RUN];
        static_for<0, RUN>([&](auto I) { constexpr int i = decltype(I)::value; ko[i] = key3<i>(ow, (uint32_t)(i + 32)); });

        pin(ko); pin(ow);
        PHASE_FENCE();
        uint32_t best[RUN];                                       // nearest distance, >= 0x10000 = none
#pragma unroll
        for (int i = 0; i < RUN; i++) best[i] = 0xFFFFFFFFu;

#pragma unroll 1
        for (int k = NCH - 1; k >= 0; k--) {                      // far chunks first, nearer ones overwrite
            uint32_t cd[17];                                      // 64 candidate positions + 2 bytes
            const uint32_t cdw = run_dw - 8u * (uint32_t)(k + 1);
            {
                const uint4 c0 = *reinterpret_cast<const uint4*>(&lds.in[cdw]);
                const uint4 c1 = *reinterpret_cast<const uint4*>(&lds.in[cdw + 4]);
                cd[0] = c0.x; cd[1] = c0.y; cd[2] = c0.z; cd[3] = c0.w;
                cd[4] = c1.x; cd[5] = c1.y; cd[6] = c1.z; cd[7] = c1.w;
                if (NCH == 1) {
#pragma unroll
                    for (int j = 0; j < 9; j++) cd[8 + j] = ow[j];
                } else {
                    const uint4 c2 = *reinterpret_cast<const uint4*>(&lds.in[cdw + 8]);
                    const uint4 c3 = *reinterpret_cast<const uint4*>(&lds.in[cdw + 12]);
                    cd[8] = c2.x; cd[9] = c2.y; cd[10] = c2.z; cd[11] = c2.w;
                    cd[12] = c3.x; cd[13] = c3.y; cd[14] = c3.z; cd[15] = c3.w;
                    cd[16] = lds.in[cdw + 16];
                }
            }
            uint32_t m[RUN];
#pragma unroll
            for (int i = 0; i < RUN; i++) m[i] = 0xFFFFFFFFu;
            // candidate-major order: two candidate keys live at a time, 32 running minima
            static_for<0, 63>([&](auto J) {
                constexpr int j = decltype(J)::value;             // handles candidates j and j+1 (j even)
                if constexpr ((j & 1) == 0) {
                    const uint32_t kc0 = key3<j>(cd, (uint32_t)j);
                    const uint32_t kc1 = key3<j + 1>(cd, (uint32_t)(j + 1));
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
#pragma unroll
            for (int i = 0; i < RUN; i++)
                if (m[i] < 64u) best[i] = m[i] + 32u * (uint32_t)k;
        }

        pin(best); pin(ow);
        PHASE_FENCE();
        // ------------------------------------------------------------------ 3. eligibility + extension
        // afterwards tok[i] = (len << 16) | dist  with len = 1 (literal) or 3..10
        const uint32_t p_run = t0 + lane * RUN;                   // first position of this run
        const uint32_t lds_run = HALO + lane * RUN;               // its byte offset in lds.in
        uint32_t tok[RUN];
        static_for<0, RUN>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const uint32_t p = p_run + i;
            uint32_t d = best[i];
            // R3: 1 <= p <= N-5;  R4: d <= min(CWINDOW, p)
            const bool ok = (d <= cw) && (d <= p) && (p + 5u <= n);
            d = ok ? d : 1u;
            // R5: common prefix of x[p+3..p+9] and x[p-d+3..p-d+9]
            const uint32_t q = lds_run + i + 3u - d;              // byte offset of the candidate's 4th byte
            const uint32_t qd = q >> 2, qs = q & 3u;
            const uint32_t a0 = lds.in[qd], a1 = lds.in[qd + 1], a2 = lds.in[qd + 2];
            const uint32_t clo = alignbyte(a1, a0, qs), chi = alignbyte(a2, a1, qs);
            constexpr int o = i + 3;
            uint32_t olo, ohi;
            if constexpr ((o & 3) == 0) { olo = ow[o >> 2]; ohi = ow[(o >> 2) + 1]; }
            else { olo = alignbyte(ow[(o >> 2) + 1], ow[o >> 2], o & 3); ohi = alignbyte(ow[(o >> 2) + 2], ow[(o >> 2) + 1], o & 3); }
            const uint64_t x = ((uint64_t)((chi ^ ohi) & 0x00FFFFFFu) << 32) | (uint64_t)(clo ^ olo) | (1ull << 56);
            const uint32_t cpl = (uint32_t)__builtin_ctzll(x) >> 3;   // 0..7 equal bytes beyond the first three
            uint32_t mlen = 3u + cpl;
            mlen = min(mlen, kmax);
            mlen = min(mlen, n - 2u - p);                         // never covers the last two bytes
            tok[i] = ok ? ((mlen << 16) | d) : (1u << 16);
            if constexpr ((i & 3) == 3) { pin_range<(i & ~3), (i & ~3) + 4>(tok); pin(ow); PHASE_FENCE(); }
        });

        pin(tok); pin(ow);
        PHASE_FENCE();
        // ------------------------------------------------------------------ 4. greedy parse
        // backward pass: E[i] = exit skip if a token starts at local index i; nibbles of P hold E[i+1..i+10]
        uint64_t P = 0x9876543210ull;
#pragma unroll
        for (int i = RUN - 1; i >= 0; i--) {
            const uint32_t sh = ((tok[i] >> 16) - 1u) * 4u;
            const uint32_t e = (uint32_t)(P >> sh) & 15u;
            P = (P << 4) | e;
        }
        // now nibble s of P = exit skip for entry skip s.  Compose across the wave (serial, scalar).
        uint32_t myskip;
        {
            const uint32_t plo = (uint32_t)P, phi = (uint32_t)(P >> 32);
            uint32_t s = skip_in;
            uint64_t sv[4] = {0, 0, 0, 0};     // entry skips of all 64 lanes, one nibble each (scalar regs)
            // 4 segments of 16 lanes; the scheduling barriers keep the compiler from hoisting all 128
            // readlanes to the top (that needed ~260 SGPR spills = v_writelane/v_readlane traffic)
            static_for<0, 4>([&](auto G) {
                constexpr int g = decltype(G)::value;
                uint64_t acc = 0;
#pragma unroll
                for (int l = 0; l < 16; l++) {
                    acc |= (uint64_t)s << (4 * l);
                    // NB: readlane returns a signed int -- cast before widening or bit 31 smears into the high half
                    const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)phi, g * 16 + l) << 32) |
                                       (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)plo, g * 16 + l);
                    s = (uint32_t)(f >> (4u * s)) & 15u;
                }
                sv[g] = acc;
                __builtin_amdgcn_sched_barrier(0);
            });
            skip_in = s;
            const uint32_t g = lane >> 4;
            const uint64_t mine = g == 0 ? sv[0] : g == 1 ? sv[1] : g == 2 ? sv[2] : sv[3];
            myskip = (uint32_t)(mine >> (4u * (lane & 15u))) & 15u;
        }

        pin(tok); pin(ow); asm volatile("" : "+v"(myskip));
        PHASE_FENCE();
        // ------------------------------------------------------------------ 5. token bits
        uint32_t code[RUN];     // (nb << 24) | bits   (bits <= 18)
        uint32_t lane_bits = 0;
        {
            uint32_t c = myskip;
            static_for<0, RUN>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const uint32_t p = p_run + i;
                const bool start = (c == 0u) && (p < n);
                const uint32_t len = tok[i] >> 16;
                c = (c == 0u) ? (len - 1u) : (c - 1u);
                const uint32_t byte = (ow[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                uint32_t lc, ln, mc, mn;
                literal_bits(byte, lc, ln);
                match_bits(len, tok[i] & 0xFFFFu, mc, mn);
                const bool is_match = len > 1u;
                uint32_t bits = is_match ? mc : lc;
                uint32_t nb = is_match ? mn : ln;
                bits = start ? bits : 0u;
                nb = start ? nb : 0u;
                code[i] = bits | (nb << 24);
                lane_bits += nb;
                if constexpr ((i & 3) == 3) { pin_range<(i & ~3), (i & ~3) + 4>(code); asm volatile("" : "+v"(c), "+v"(lane_bits)); PHASE_FENCE(); }
            });
        }
        pin(code); pin(ow);
        PHASE_FENCE();
        // wave exclusive scan of lane_bits
        uint32_t incl = lane_bits;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) {
            const uint32_t o = __shfl_up(incl, ofs, 64);
            if (lane >= (uint32_t)ofs) incl += o;
        }
        const uint32_t tile_bits = __builtin_amdgcn_readlane(incl, 63);
        uint32_t bitpos = base_bits + incl - lane_bits;

        // per-lane packing through a 64-bit accumulator
        {
            uint32_t widx = bitpos >> 5;
            uint32_t fill = bitpos & 31u;
            uint64_t acc = 0;
#pragma unroll
            for (int i = 0; i < RUN; i++) {
                acc |= (uint64_t)(code[i] & 0xFFFFFFu) << fill;
                fill += code[i] >> 24;
                if (fill >= 32u) {
                    atomicOr(&lds.out[widx], (uint32_t)acc);
                    widx++;
                    acc >>= 32;
                    fill -= 32u;
                }
            }
            if (fill) atomicOr(&lds.out[widx], (uint32_t)acc);
        }

        PHASE_FENCE();
        // ------------------------------------------------------------------ 6. Adler partials
        {
            uint32_t sa = 0, sc = 0;    // sum x_i, sum i*x_i over the run
#pragma unroll
            for (int k = 0; k < 8; k++) {
                sa = __builtin_amdgcn_sad_u8(ow[k], 0u, sa);
                const uint32_t wts = (uint32_t)(4 * k) | ((uint32_t)(4 * k + 1) << 8) | ((uint32_t)(4 * k + 2) << 16) | ((uint32_t)(4 * k + 3) << 24);
                sc = __builtin_amdgcn_udot4(ow[k], wts, sc, false);
            }
            // sum (N - p) x_p over the run = (N - p_run) * sa - sc ; bytes at p >= N are zero
            const uint32_t wgt = (p_run < n) ? ((n - p_run) % ADLER_MOD) : 0u;
            ad_a = (ad_a + sa) % ADLER_MOD;
            ad_w = (ad_w + (wgt * sa) % ADLER_MOD + ADLER_MOD * 8u - (sc % ADLER_MOD)) % ADLER_MOD;
        }
        __syncthreads();

        // ------------------------------------------------------------------ 7. flush
        const uint32_t end_bits = base_bits + tile_bits;
        const bool last = (t0 + TILE >= n);
        if (!last) {
            const uint32_t full = end_bits >> 5;
            for (uint32_t w = lane; w < full; w += 64) outw[gw + w] = lds.out[w];
            carry_word = lds.out[full];
            gw += full;
            base_bits = end_bits & 31u;
        } else {
            // R8: EOB = 7 zero bits, zero pad to a byte, Adler-32 big-endian (s2 then s1)
            uint32_t s1 = ad_a, s2 = ad_w;
#pragma unroll
            for (int ofs = 32; ofs > 0; ofs >>= 1) {
                s1 += __shfl_xor(s1, ofs, 64);
                s2 += __shfl_xor(s2, ofs, 64);
            }
            s1 = (s1 + 1u) % ADLER_MOD;
            s2 = (s2 + n % ADLER_MOD) % ADLER_MOD;
            const uint32_t nbytes = (end_bits + 7u + 7u) >> 3;
            if (lane == 0) {
                uint8_t* ob = reinterpret_cast<uint8_t*>(lds.out);
                ob[nbytes] = (uint8_t)(s2 >> 8);
                ob[nbytes + 1] = (uint8_t)s2;
                ob[nbytes + 2] = (uint8_t)(s1 >> 8);
                ob[nbytes + 3] = (uint8_t)s1;
            }
            __syncthreads();
            const uint32_t total = nbytes + 4u;
            const uint32_t words = (total + 3u) >> 2;
            for (uint32_t w = lane; w < words; w += 64) outw[gw + w] = lds.out[w];
            if (lane == 0) {
                a.out_len[blk] = gw * 4u + total;     // R9
                a.status[blk] = HDLZ_OK;
            }
        }
    }
}

template __global__ void k_compress<1>(CompressArgs);
template __global__ void k_compress<2>(CompressArgs);
template __global__ void k_compress<8>(CompressArgs);

hipError_t launch_compress(const CompressArgs& a, hipStream_t stream) {
    if (a.nblocks == 0) return hipSuccess;
    const dim3 grid((unsigned)a.nblocks), block(64);
    if (a.cwindow <= 32) hipLaunchKernelGGL(k_compress<1>, grid, block, 0, stream, a);
    else if (a.cwindow <= 64) hipLaunchKernelGGL(k_compress<2>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(k_compress<8>, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
