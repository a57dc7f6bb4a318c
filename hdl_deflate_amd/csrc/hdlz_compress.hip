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

template <int NCH>   // NCH = ceil(cwindow / 32): 1, 2 or 8 chunks of 32 candidate distances
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NCH == 1 ? 5 : 4, NCH == 1 ? 5 : 4))) void k_compress(CompressArgs a) {
    __shared__ WaveLds lds;
    const uint32_t lane = threadIdx.x;

    // ---- per-wave look-up tables (once per wave lifetime)
    for (uint32_t e = lane; e < (uint32_t)LUT_LIT; e += 64) lds.lut[e] = literal_entry(e);
    for (uint32_t e = lane; e < (uint32_t)LUT_MATCH; e += 64) {
        if (NCH == 1) lds.lut[LUT_LIT + e] = dist_entry((e & 31u) + 1u) | length_code((e >> 5) + 3u);
        else lds.lut[LUT_LIT + e] = dist_entry(e + 1u);
    }
    __syncthreads();

    const uint32_t cw4 = 4u * (uint32_t)a.cwindow;
    const uint32_t kmax = (uint32_t)a.maxmatch;
    uint8_t* lin8 = reinterpret_cast<uint8_t*>(lds.in);
    const uint8_t* lut8 = reinterpret_cast<const uint8_t*>(lds.lut);
    uint8_t* out8 = reinterpret_cast<uint8_t*>(lds.out);

    for (uint64_t blk = blockIdx.x; blk < a.nblocks; blk += gridDim.x) {
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
            continue;
        }
        if ((uint64_t)out_bound(n) > a.out_pitch) {
            if (lane == 0) { a.out_len[blk] = 0; a.status[blk] = HDLZ_E_OUT_CAPACITY; }
            continue;
        }
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
        const bool aligned16 = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;

        uint32_t gw = 0;            // output words already flushed to HBM
        uint32_t base_bits = 19;    // R1: 78 9C + bits 1,1,0
        uint32_t carry_word = 0x78u | (0x9Cu << 8) | (0x3u << 16);
        uint32_t skip_in = 0;       // positions at the tile start covered by the previous tile's last match
        uint32_t ad_a = 0, ad_w = 0;   // per-lane Adler partials (sum x, sum (N-p) x mod 65521)

        for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
            // -------------------------------------------------------------- 1. stage the tile
            uint32_t keep = 0;
            if (t0 != 0) keep = lds.in[(TILE / 4) + lane];      // last HALO bytes of the previous tile
            __syncthreads();                                     // (also orders the previous flush reads)
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
                        if (valid < 16u) {                           // positions >= N must read as zero bytes
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
            const uint32_t p_run = t0 + lane * RUN;                   // first position of this run
            const uint32_t lds_run = HALO + lane * RUN;               // its byte offset in lds.in
            const uint32_t nrem = n - min(p_run, n);                  // positions of the block from p_run on
            const uint32_t nrem_m2 = nrem - 2u;                       // (wraps when nrem < 2: then nothing is eligible)
            const uint32_t p4_run = 4u * min(p_run, 1024u);           // 4*p saturated: only p < CWINDOW <= 256 matters
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

            pin(tok); asm volatile("" : "+v"(myskip));
            PHASE_FENCE();
            // -------------------------------------------------------------- 5. token bits
            // pass A: LUT entry (code | nbits << 27) of every token start, 0 elsewhere
            uint32_t code[RUN];
            uint32_t lane_bits = 0;
            {
                uint32_t c = myskip;
                static_for<0, RUN>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    const uint32_t e = *reinterpret_cast<const uint32_t*>(lut8 + (tok[i] & 0xFFFFu));
                    const bool start = (c == 0u);
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
            // wave exclusive scan of lane_bits
            uint32_t incl = lane_bits;
#pragma unroll
            for (int ofs = 1; ofs < 64; ofs <<= 1) {
                const uint32_t o = __shfl_up(incl, ofs, 64);
                if (lane >= (uint32_t)ofs) incl += o;
            }
            const uint32_t tile_bits_all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            pin(code); asm volatile("" : "+v"(incl), "+v"(lane_bits));
            PHASE_FENCE();
            // pass B: OR every token into the LDS bit buffer at its own bit offset
            {
                uint32_t bp = base_bits + incl - lane_bits;
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
            // -------------------------------------------------------------- 6. Adler partials
            {
                uint32_t sa = 0, sc = 0;    // sum x_i, sum i*x_i over the run
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    sa = __builtin_amdgcn_sad_u8(ow[k], 0u, sa);
                    const uint32_t wts = (uint32_t)(4 * k) | ((uint32_t)(4 * k + 1) << 8) | ((uint32_t)(4 * k + 2) << 16) | ((uint32_t)(4 * k + 3) << 24);
                    sc = __builtin_amdgcn_udot4(ow[k], wts, sc, false);
                }
                // sum (N - p) x_p over the run = (N - p_run) * sa - sc ; bytes at p >= N are zero
                const uint32_t wgt = nrem % ADLER_MOD;
                ad_a = (ad_a + sa) % ADLER_MOD;
                ad_w = (ad_w + (wgt * sa) % ADLER_MOD + ADLER_MOD * 8u - (sc % ADLER_MOD)) % ADLER_MOD;
            }
            __syncthreads();

            // -------------------------------------------------------------- 7. flush
            const bool last = (t0 + TILE >= n);
            if (!last) {
                const uint32_t end_bits = base_bits + tile_bits_all;
                const uint32_t full = end_bits >> 5;
                for (uint32_t w = lane; w < full; w += 64) outw[gw + w] = lds.out[w];
                carry_word = lds.out[full];
                gw += full;
                base_bits = end_bits & 31u;
            } else {
                // every position >= N of this tile was emitted as one 8-bit literal (zero byte, never a match,
                // and the last two real bytes are always literals so the parse lands exactly on N)
                const uint32_t ninv = t0 + TILE - n;
                const uint32_t end_bits = base_bits + tile_bits_all - 8u * ninv;
                // wipe everything behind the real end: partial word masked, later words zeroed
                {
                    const uint32_t ew = end_bits >> 5, rb = end_bits & 31u;
                    for (uint32_t w = ew + lane; w < OUT_WORDS; w += 64)
                        lds.out[w] = (w == ew) ? (lds.out[w] & ((1u << rb) - 1u)) : 0u;
                }
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
                    a.out_len[blk] = gw * 4u + total;     // R9
                    a.status[blk] = HDLZ_OK;
                }
            }
        }
    }
}

template __global__ void k_compress<1>(CompressArgs);
template __global__ void k_compress<2>(CompressArgs);
template __global__ void k_compress<8>(CompressArgs);

hipError_t launch_compress(const CompressArgs& a, hipStream_t stream) {
    if (a.nblocks == 0) return hipSuccess;
    // persistent single-wave workgroups: 64 per CU queued (16 resident at 4 waves/SIMD) so that the
    // hardware dispatcher balances uneven blocks; each wave strides over the batch
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipGetLastError();
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // small blocks (the reference's own input scale): several blocks per wave-tile -- uniform 16-byte aligned batches,
    // or ragged ones whose caller states an upper bound on the block lengths in in_len
    if (a.cwindow <= 32 && a.in_len >= 5u && a.in_len <= 1024u && a.out_pitch >= (uint64_t)out_bound(a.in_len) &&
        (a.in_off || ((a.in_pitch & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15u) == 0)))
        return launch_compress_small(a, stream, ncu);
    uint64_t g = (uint64_t)ncu * 64u;
    if (g > a.nblocks) g = a.nblocks;
    const dim3 grid((unsigned)g), block(64);
    if (a.cwindow <= 32) hipLaunchKernelGGL(k_compress<1>, grid, block, 0, stream, a);
    else if (a.cwindow <= 64) hipLaunchKernelGGL(k_compress<2>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(k_compress<8>, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
