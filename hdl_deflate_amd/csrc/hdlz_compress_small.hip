// hdlz_compress_small.hip -- STARTC for batches of SMALL blocks (N <= 1024, CWINDOW <= 32), uniform or ragged.
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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"
#include "hdlz_compress_common.h"

namespace hdlz {

constexpr int SMALL_OUT_WORDS = 704;          // G * ceil(out_bound(N)/4) is largest for N = 32: 64 * 11

struct __attribute__((aligned(16))) SmallLds {
    uint32_t in[IN_BYTES / 4];
    uint32_t out[SMALL_OUT_WORDS];
    uint32_t lut[LUT_LIT + LUT_MATCH];
    uint32_t ad[2][64];                       // per-lane Adler partials, summed per block by its first lane
};

template <bool RAGGED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_compress_small(CompressArgs a) {
    constexpr int NCH = 1;
    __shared__ SmallLds lds;
    const uint32_t lane = threadIdx.x;
    for (uint32_t e = lane; e < (uint32_t)LUT_LIT; e += 64) lds.lut[e] = literal_entry(e);
    for (uint32_t e = lane; e < (uint32_t)LUT_MATCH; e += 64)
        lds.lut[LUT_LIT + e] = dist_entry((e & 31u) + 1u) | length_code((e >> 5) + 3u);
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
    uint8_t* out8 = reinterpret_cast<uint8_t*>(lds.out);
    const uint64_t ngroups = (a.nblocks + G - 1u) / G;

    for (uint64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const uint64_t blk = grp * G + g;
        const bool has_blk = (g < G) && (blk < a.nblocks);
        uint64_t boff = 0;
        uint32_t nb = n;                                          // this lane's block: offset and length
        if (has_blk) {
            if (ragged) { boff = a.in_off[blk]; nb = (uint32_t)(a.in_off[blk + 1] - boff); }
            else boff = blk * a.in_pitch;
        }
        const bool lane_ok = has_blk && nb >= 5u && nb <= n;      // R0: shorter blocks never start (status below)
        if (has_blk && !lane_ok && r == 0u) {                     // (nb > n: the caller's bound on the lengths was wrong)
            a.out_len[blk] = 0;
            a.status[blk] = nb < 5u ? HDLZ_E_SHORT_INPUT : HDLZ_E_BAD_PARAM;
        }
        const uint32_t nrem = lane_ok ? nb - min(p_run, nb) : 0u; // positions of the block from this run on
        const uint32_t nrem_m2 = nrem - 2u;
        // -------------------------------------------------------------- 1. stage: every lane loads its own run
        __syncthreads();
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
        for (uint32_t w = lane; w < (uint32_t)SMALL_OUT_WORDS; w += 64) lds.out[w] = 0u;
        __syncthreads();
        if (lane_ok && r == 0u) lds.out[g * Wb] = 0x78u | (0x9Cu << 8) | (0x3u << 16);   // R1 per block
        __syncthreads();

            // -------------------------------------------------------------- 2. match search
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
            static_for<0, RUN>([&](auto I) { constexpr int i = decltype(I)::value; ko[i] = key3<i>(ow, (uint32_t)(4 * (i + 32))); });

            pin(ko); pin(ow);
            PHASE_FENCE();
            uint32_t best[RUN];                                       // 4 * nearest distance, huge = none
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
                        cd[8] = ow[0];                                // candidate 31 needs the first own bytes
#pragma unroll
                        for (int j = 9; j < 17; j++) cd[j] = 0;      // unused: own keys double as candidates
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
                        uint32_t kc0, kc1;
                        if constexpr (NCH == 1 && j >= 32) {          // own position j-32 IS candidate j (same tag 4j)
                            // pin in place: without it the scheduler precomputes all ~500 own-vs-own differences
                            asm volatile("" : "+v"(ko[j - 32]), "+v"(ko[j - 31]));
                            kc0 = ko[j - 32];
                            kc1 = ko[j - 31];
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
#pragma unroll
                for (int i = 0; i < RUN; i++)
                    if (m[i] < 256u) best[i] = m[i] + 128u * (uint32_t)k;
            }

            // -------------------------------------------------------------- 3. eligibility + extension
            // afterwards tok[i] = (len-1) << 16 | LUT byte offset of the token  (len-1 = 0 for a literal)
            pin(best); pin(ow);
            PHASE_FENCE();
            const uint32_t lds_run = HALO + lane * RUN;               // byte offset of this run in lds.in
            const uint32_t p4_run = 4u * p_run;                       // block-relative: a block's first run has no history
            uint32_t tok[RUN];
            static_for<0, RUN>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const uint32_t d4 = best[i];
                // R3: 1 <= p <= N-5;  R4: d <= min(CWINDOW, p)
                const bool ok = (d4 <= cw4) & (d4 <= p4_run + (uint32_t)(4 * i)) & (nrem >= (uint32_t)(i + 5));   // '&': no short-circuit branches
                // distance for the gather; for "no match" any in-range value will do (the result is discarded)
                const uint32_t d = (NCH == 1) ? ((d4 & 0xFCu) >> 2) : (ok ? (d4 >> 2) : 1u);
                // R5: common prefix of x[p+3..p+10] and x[p-d+3..p-d+10]
                const uint32_t q = lds_run + (uint32_t)(i + 3) - d;   // byte offset of the candidate's 4th byte
                const uint32_t qd = q >> 2, qs = q & 3u;
                const uint32_t a0 = lds.in[qd], a1 = lds.in[qd + 1], a2 = lds.in[qd + 2];
                const uint32_t clo = alignbyte(a1, a0, qs), chi = alignbyte(a2, a1, qs);
                constexpr int o = i + 3;
                uint32_t olo, ohi;
                if constexpr ((o & 3) == 0) { olo = ow[o >> 2]; ohi = ow[(o >> 2) + 1]; }
                else { olo = alignbyte(ow[(o >> 2) + 1], ow[o >> 2], o & 3); ohi = alignbyte(ow[(o >> 2) + 2], ow[(o >> 2) + 1], o & 3); }
                // equal low BITS of the two 8-byte windows (ffbl(0) = 0xFFFFFFFF = "no difference in this half")
                const uint32_t zhi = min(ffbl(chi ^ ohi), 32u) + 32u;
                const uint32_t zb = min(ffbl(clo ^ olo), zhi);
                // m = min(3 + equal bytes, Kmax, N-2-p): a match never covers the last two bytes
                const uint32_t mlen = umin3(3u + (zb >> 3), kmax, nrem_m2 - (uint32_t)i);
                // literal byte -> LUT offset 4*byte
                constexpr int bsh = 8 * (i & 3);
                uint32_t lit;
                if constexpr (bsh == 0) lit = (ow[i >> 2] << 2) & 0x3FCu;
                else lit = (ow[i >> 2] >> (bsh - 2)) & 0x3FCu;
                uint32_t mt;
                if (NCH == 1) {
                    // (len-1)<<16 | base + ((len-3)*32 + d-1)*4 = mlen*65664 + d4 + const, as two shift-adds:
                    // hipcc folds the C form into a quarter-rate v_mul_lo_u32
                    uint32_t t1;
                    asm("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(t1) : "v"(mlen), "v"(d4 + (LUT_MATCH_BYTE - 65924u)));
                    asm("v_lshl_add_u32 %0, %1, 16, %2" : "=v"(mt) : "v"(mlen), "v"(t1));
                }
                else mt = (mlen << 16) + d4 + (LUT_MATCH_BYTE - 65540u);                     // (len-1)<<16 | base + (d-1)*4
                tok[i] = ok ? mt : lit;
                if constexpr ((i & 3) == 3) { pin_range<(i & ~3), (i & ~3) + 4>(tok); pin(ow); PHASE_FENCE(); }
            });

            pin(tok); pin(ow);
            PHASE_FENCE();
            // -------------------------------------------------------------- 4. greedy parse
            // backward pass: E[i] = exit skip if a token starts at local index i; nibbles of P hold E[i+1..i+10]
            uint64_t P = 0x9876543210ull;
#pragma unroll
            for (int i = RUN - 1; i >= 0; i--) {
                const uint32_t sh = (tok[i] >> 16) * 4u;              // 4 * (len-1)
                const uint32_t e = (uint32_t)(P >> sh) & 15u;
                P = (P << 4) | e;
            }
            // now nibble s of P = exit skip for entry skip s.  Compose across the wave (serial, scalar).
            uint32_t myskip;
            {
                const uint32_t plo = (uint32_t)P, phi = (uint32_t)(P >> 32);
                uint32_t s = 0;                      // no token ever crosses a block end (R5), so skips reset by themselves
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
                (void)s;
                const uint32_t g = lane >> 4;
                const uint64_t mine = g == 0 ? sv[0] : g == 1 ? sv[1] : g == 2 ? sv[2] : sv[3];
                myskip = (uint32_t)(mine >> (4u * (lane & 15u))) & 15u;
            }

            pin(tok); asm volatile("" : "+v"(myskip));
            PHASE_FENCE();
            // -------------------------------------------------------------- 5. token bits
            // pass A: LUT entry (code | nbits << 27) of every token start, 0 elsewhere
            uint32_t code[RUN];
            uint32_t lane_bits = 0;
            {
                uint32_t c = lane_ok ? myskip : 0xFFFFu;        // lanes without a block never start a token
                static_for<0, RUN>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    const uint32_t e = *reinterpret_cast<const uint32_t*>(lut8 + (tok[i] & 0xFFFFu));
                    const bool start = (c == 0u) & ((uint32_t)i < nrem);   // padding positions of a block emit nothing
                    const uint32_t lenm1 = tok[i] >> 16;
                    c = start ? lenm1 : (c - 1u);
                    uint32_t ee = e;
                    if constexpr (NCH != 1)                               // wide windows: [dist] LUT + computed length code
                        ee |= lenm1 ? (__builtin_bitreverse32(lenm1 - 1u) >> 25) : 0u;
                    code[i] = start ? ee : 0u;
                    lane_bits += code[i] >> NB_SHIFT;
                    if constexpr ((i & 3) == 3) { pin_range<(i & ~3), (i & ~3) + 4>(code); asm volatile("" : "+v"(c), "+v"(lane_bits)); PHASE_FENCE(); }
                });
            }
            pin(code);
            PHASE_FENCE();
            // wave scan of lane_bits; bit offsets are per block (segmented by the block's first lane)
            uint32_t incl = lane_bits;
#pragma unroll
            for (int ofs = 1; ofs < 64; ofs <<= 1) {
                const uint32_t o = __shfl_up(incl, ofs, 64);
                if (lane >= (uint32_t)ofs) incl += o;
            }
            const uint32_t first_lane = g * Rb;
            // (shuffles must be executed by ALL lanes: a source lane that skipped it reads as garbage)
            const uint32_t prev_incl = (uint32_t)__shfl((int)incl, (int)((first_lane - 1u) & 63u), 64);
            const uint32_t before_blk = first_lane == 0u ? 0u : prev_incl;
            pin(code); asm volatile("" : "+v"(incl), "+v"(lane_bits));
            PHASE_FENCE();
            // pass B: OR every token into the block's region of the LDS bit buffer
            {
                uint32_t bp = 32u * g * Wb + 19u + (incl - lane_bits - before_blk);
                if (!lane_ok) bp = 0;                             // (emits nothing: all codes are zero)
#pragma unroll
                for (int i = 0; i < RUN; i++) {
                    const uint64_t v = (uint64_t)(code[i] & CODE_MASK) << (bp & 31u);
                    uint32_t* w = reinterpret_cast<uint32_t*>(out8 + ((bp >> 3) & ~3u));
                    __hip_atomic_fetch_or(w, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or(w + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    bp += code[i] >> NB_SHIFT;
                    if ((i & 3) == 3) { asm volatile("" : "+v"(bp)); PHASE_FENCE(); }
                }
            }
            pin(ow);
            PHASE_FENCE();
            // -------------------------------------------------------------- 6. Adler partials per lane -> LDS
            {
                uint32_t sa = 0, sc = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    sa = __builtin_amdgcn_sad_u8(ow[k], 0u, sa);
                    const uint32_t wts = (uint32_t)(4 * k) | ((uint32_t)(4 * k + 1) << 8) | ((uint32_t)(4 * k + 2) << 16) | ((uint32_t)(4 * k + 3) << 24);
                    sc = __builtin_amdgcn_udot4(ow[k], wts, sc, false);
                }
                lds.ad[0][lane] = sa;                              // <= 8160
                lds.ad[1][lane] = nrem * sa - sc;                  // sum (N - p) x_p over the run, < 2^24 for N <= 1024
            }
            __syncthreads();
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
            __syncthreads();
            for (uint32_t gg = 0; gg < G; gg++) {                       // flush block by block, coalesced dwords
                const uint64_t b2 = grp * G + gg;
                if (b2 >= a.nblocks) break;
                const uint32_t words = ((uint32_t)__builtin_amdgcn_readlane((int)total, (int)(gg * Rb)) + 3u) >> 2;
                uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(a.out + b2 * a.out_pitch);
                for (uint32_t w = lane; w < words; w += 64) dst[w] = lds.out[gg * Wb + w];
            }
    }
}

template __global__ void k_compress_small<false>(CompressArgs);
template __global__ void k_compress_small<true>(CompressArgs);

hipError_t launch_compress_small(const CompressArgs& a, hipStream_t stream, int ncu) {
    const uint32_t Rb = (a.in_len + 31u) >> 5;
    const uint64_t G = 64u / Rb;
    uint64_t groups = (a.nblocks + G - 1u) / G;
    uint64_t grid = (uint64_t)ncu * 64u;
    if (grid > groups) grid = groups;
    if (a.in_off) hipLaunchKernelGGL(k_compress_small<true>, dim3((unsigned)grid), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(k_compress_small<false>, dim3((unsigned)grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
