// hdlz_inflate_grp.hip -- STARTD for a batch of independent zlib streams, SIXTEEN LANES per stream (round 5).
//
// Same contract, same reference lines and the same order of the reference's checks as hdlz_inflate_tok.hip
// (/root/reference/deflate.py:635-732 IDLE/HEADER, :1402-1445 NEXT, :1519-1591 INFLATE, :1593-1659 COPY, :517-533 get4/adv; rule
// names D0..D8 are SURVEY.md 8(a)'s); fixed-Huffman and stored blocks -- a stream that meets a dynamic-tree block comes back flagged
// HDLZ_E_DYNAMIC_UNSUPPORTED for the caller's second pass, exactly like pass 1 of the lane kernel.
//
// Why a third mapping.  One LANE per stream (k_inflate_tok) is the cheapest decode per byte -- 64 serial decoders per wave
// instruction -- but (a) every memory instruction of the wave touches 64 different rows: the input slots and the far history of the
// copies cost one 64-byte sector each (TA 94 % busy, 8.7x the algorithmic traffic on BASELINE configs[3], profiles/r04_inflate_tok_pmc_summary.txt),
// and (b) one stream takes ~0.55 ms however few there are: the GPU is full only from ~10^5 streams on.  One WAVE per stream
// (k_inflate_dyn) fills the GPU with 8 k streams but spends 64 lanes on a serial chain.  Here a group of 16 lanes owns a stream:
//   * the stream's HISTORY stays on chip: a 2 KiB ring per stream in LDS (the whole output of a configs[3] stream; a copy that reaches
//     further back -- up to 32 KiB, D8 -- reads the stream's own flushed output), so no far-history sectors at all;
//   * input and output move in full lines: the group fetches 128 input bytes at a time (16 lanes x 8 bytes, one request ahead of the
//     reader, through a 256-byte LDS FIFO) and flushes 1 KiB of output with 16-byte stores -- HBM traffic ~1.0x the algorithmic bytes;
//   * the decode itself is replicated: every lane of the group holds the same bit buffer and decodes the same token (no cross-lane
//     traffic on the serial chain); the lanes differ only where the bytes move -- lane l copies byte l of up to 16 per step
//     (out[o + l] = out[o - dist + l mod dist]: overlap-correct by construction, deflate.py:1627-1659);
//   * four streams per wave, 16 waves per CU: 4096 streams already put a wave on every SIMD.
// A step takes up to three literals and the match behind them per stream and moves up to 16 bytes (a longer copy goes on in the next
// steps): ~180 steps for a 2 KiB stream of BASELINE configs[3].  That is still several times the wave instructions per byte of the lane
// kernel, so huge batches stay there; this mapping is for the batches in between (hdlz_inflate_batch: HDLZ_INFLATE_GROUP_MIN .. _MAX streams) and whenever the caller asks for it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"
#include "hdlz_inflate_tables.h"

namespace hdlz {
namespace grp {
using namespace tok;

constexpr uint32_t G = 16;                   // lanes per stream
constexpr uint32_t NS = 64 / G;              // streams per wave
constexpr uint32_t WAVES = 4;                // waves per workgroup (they share the 2 KiB literal/length table, nothing else)
constexpr uint32_t RB = 2048, RMASK = RB - 1u;       // history ring per stream
constexpr uint32_t RSTRIDE = RB + 64;        // (the four rings of a wave start 16 banks apart)
constexpr uint32_t FB = 256, FMASK = FB - 1u, FH = FB / 2;   // input FIFO per stream: two halves of 16 lanes x 8 bytes
constexpr uint32_t FLUSH = 1024;             // output bytes per flush
constexpr uint32_t NEAR = RB - 32;           // distances up to this are served from the ring (a step writes at most 16 bytes ahead of o)
static_assert(FLUSH + 64 < NEAR && FLUSH % (16 * G) == 0, "a far copy reads flushed bytes only: unflushed < FLUSH + 16 when it is decoded");

struct __attribute__((aligned(16))) Lds {
    uint8_t ring[WAVES][NS][RSTRIDE];
    uint32_t fifo[WAVES][NS][FB / 4];
    uint32_t lit[512];
};

// 8 stream bytes at position p of z (zn bytes long); bytes at or beyond zn read as zero
__device__ __forceinline__ uint64_t load8(const uint8_t* __restrict__ z, uint32_t p, uint32_t zn) {
    if (p + 8u <= zn) return *reinterpret_cast<const u64_unaligned*>(z + p);
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8u; k++)
        if (p + k < zn) v |= (uint64_t)z[p + k] << (8u * k);
    return v;
}

__global__ __launch_bounds__(64 * WAVES) void k_inflate_grp(InflateArgs a) {
    __shared__ Lds lds;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t l = lane & (G - 1u), g = lane / G;
    for (uint32_t c = threadIdx.x; c < 512u; c += 64u * WAVES) lds.lit[c] = lit_entry(c, (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0u);
    __syncthreads();                 // the only workgroup barrier: the waves are independent from here on

    const uint64_t sid = ((uint64_t)blockIdx.x * WAVES + wave) * NS + g;
    const bool exists = sid < a.nstreams;
    uint64_t off = 0;
    uint32_t zn = 0;
    if (exists) {
        if (a.in_off) { off = a.in_off[sid]; zn = (uint32_t)(a.in_off[sid + 1] - off); }
        else { off = sid * a.in_pitch; zn = a.in_len; }
    }
    // D0: the two zlib header bytes are skipped unvalidated -- the FIFO is filled from byte 2 on, which keeps the reader's dwords aligned
    const uint8_t* __restrict__ z2 = a.in + off + 2u;
    const uint32_t zn2 = zn >= 2u ? zn - 2u : 0u;
    uint8_t* out = a.out + (exists ? sid : 0ull) * a.out_pitch;
    uint8_t* ring = lds.ring[wave][g];
    uint32_t* fifo = lds.fifo[wave][g];
    const uint32_t cap = a.out_pitch > 0xFFFFFE00ull ? 0xFFFFFE00u : (uint32_t)a.out_pitch;   // o + 258 never wraps
    const uint32_t obsize = a.obsize ? a.obsize : 32768u;
    const uint32_t len_mask = a.obsize ? ((1u << (31u - (uint32_t)__builtin_clz(a.obsize))) - 1u) : 0xFFFFu;   // deflate.py:329,:714
    const bool assume_fixed = (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0;
    const uint32_t oneblock = (a.flags & HDLZ_INFLATE_ONEBLOCK) ? 1u : 0u;
    const int32_t isize = (int32_t)zn - 1;            // deflate.py:605

    uint32_t status = HDLZ_OK, out_len = 0;
    bool active = exists;
    if (exists && zn < 5u) { status = HDLZ_E_SHORT_INPUT; active = false; }

    // ---- everything below is the same in the 16 lanes of a group, except `nxt` and what the move / flush steps address
    uint64_t bb = 0;            // bit buffer (LSB first)
    uint32_t bc = 0;            // valid bits in bb
    uint32_t ipq = 0;           // next FIFO byte to go into bb (stream byte ipq + 2): a multiple of 4
    uint32_t loaded = FB;       // the FIFO holds the stream bytes [loaded - FB, loaded) (counted from byte 2)
    uint32_t o = 0, flushed = 0;
    uint32_t rem = 0, dist = 1; // pending LZ copy
    uint32_t litv = 0, litn = 0;// pending literal / stored byte
    uint32_t srem = 0, final_ = 0;
    bool need_header = true;
    uint32_t qw = 0;            // the FIFO dword at ipq, read when its predecessor was taken: the refill itself waits for nothing
    uint64_t nxt = 0;           // this lane's 8 bytes of the NEXT half, [loaded + 8 l, + 8): requested a whole half ahead of the reader
    // (the FIFO is written and read as uint32_t -- one type for the compiler's alias analysis -- and wave_lds_order() keeps the
    //  reader's loads behind the stores: the LDS executes a wave's instructions in order)
#define GRP_FIFO_PUT(dw, v64) do { fifo[(dw)] = (uint32_t)(v64); fifo[(dw) + 1u] = (uint32_t)((v64) >> 32); } while (0)
    if (active) {
        const uint64_t h0 = load8(z2, 8u * l, zn2), h1 = load8(z2, FH + 8u * l, zn2);
        GRP_FIFO_PUT(2u * l, h0);
        GRP_FIFO_PUT((FH / 4u) + 2u * l, h1);
        nxt = load8(z2, FB + 8u * l, zn2);
        wave_lds_order();
        qw = fifo[0];
    }
#define GRP_IP() (ipq + 2u)
#define GRP_BITPOS() (8u * GRP_IP() - bc)
#define GRP_REFILL() do { if (bc <= 32u) { bb |= (uint64_t)qw << bc; bc += 32u; ipq += 4u; qw = fifo[(ipq & FMASK) >> 2]; } } while (0)
#define GRP_FAIL(code) do { status = (code); out_len = 0; active = false; } while (0)

#ifdef HDLZ_GRP_TIMING        // diagnostic build (tools/exp_grp_timing.py): s_memtime per part of the step; group g of a wave reports part g INSTEAD of its result
    uint32_t tacc[5] = {0, 0, 0, 0, 0}, tsteps = 0, tlast = (uint32_t)__builtin_readcyclecounter();
#define GT(k) do { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define GT(k) do {} while (0)
#endif
    for (;;) {
        GT(4);
        // ------------------------------------------------------------ 1. input: the half the reader has left is replaced by the one that
        // was requested when the reader entered the half before it, and the one after that is requested
        {
            const bool adv = active && ipq + FH >= loaded;
            if (ballot64(adv) != 0ull) {
                if (adv) {
                    GRP_FIFO_PUT(((loaded & FMASK) >> 2) + 2u * l, nxt);
                    loaded += FH;
                    nxt = load8(z2, loaded + 8u * l, zn2);
                }
                wave_lds_order();
            }
        }
        if (active) GRP_REFILL();
        GT(0);
        // ------------------------------------------------------------ 2. fast path, inside a fixed block: up to three literals and the match
        // behind them in ONE step.  A literal is 8 or 9 bits, so the second symbol starts at bit 8 or 9 and the third at 16, 17 or 18: the
        // six look-ups are issued together -- one LDS round trip on the serial chain of the stream instead of three (a wave is alone on
        // its SIMD when the batch is small: nothing else hides them).  Straight-line code, every lane computes every field and selects
        // (the nested-if form compiled to ~350 instructions per step: exec-mask regions).
        bool slow;
        {
            const bool idle = active & (srem == 0u) & (rem == 0u);          // this stream takes new tokens in this step
            // input guard as in k_inflate_tok: bc >= 33 after the refill, a token is only taken when a buffered bit is left behind it, so
            // its bit position lies below byte ip; with ip + 3 <= zn both reference checks (deflate.py:1535-1539, :1600) pass
            const bool can = idle & !need_header & (GRP_IP() + 3u <= zn);
            const uint32_t w = (uint32_t)bb;
            const uint32_t e0 = lds.lit[w & 511u], e1a = lds.lit[(w >> 8) & 511u], e1b = lds.lit[(w >> 9) & 511u];
            const uint32_t e2a = lds.lit[(w >> 16) & 511u], e2b = lds.lit[(w >> 17) & 511u], e2c = lds.lit[(w >> 18) & 511u];
            constexpr uint32_t TMASK = 3u << 13;                      // entry: nbits[3:0] | sym[12:4] | type[14:13] | lbase[24:16] | leb[27:25]
            const uint32_t n0 = e0 & 15u;
            const uint32_t e1 = n0 == 8u ? e1a : e1b;                 // (used behind a literal only: n0 is 8 or 9 then)
            const uint32_t off2 = n0 + (e1 & 15u);
            const uint32_t e2 = off2 == 16u ? e2a : off2 == 17u ? e2b : e2c;
            const uint32_t l0 = (e0 & TMASK) == 0u ? 1u : 0u;         // T_LIT = 0
            const uint32_t l1 = (e1 & TMASK) == 0u ? l0 : 0u;
            const uint32_t l2 = (e2 & TMASK) == 0u ? l1 : 0u;
            const uint32_t nraw = l0 + l1 + l2;                       // leading literals (3: the third symbol is one too)
            const uint32_t nl = can ? min(nraw, cap - o) : 0u;        // ... that fit (a literal at o == cap is the slow path's OUT_CAPACITY)
            // the token behind them (nraw == 3: a literal -- its type fails the match test below)
            const uint32_t em = nraw == 0u ? e0 : nraw == 1u ? e1 : e2;
            const uint32_t moff = nraw == 0u ? 0u : nraw == 1u ? n0 : off2;
            const uint32_t nbm = moff + (em & 15u);
            const uint32_t leb = (em >> 25) & 7u, lbase = (em >> 16) & 0x1FFu;
            const uint32_t x = (uint32_t)(bb >> nbm);                 // (5 + 5 + 13 + 5 bits at most lie behind the length symbol)
            const uint32_t tlength = lbase + (x & ((1u << leb) - 1u));
            const uint32_t y = x >> leb;
            // the distance code in closed form (no second dependent table look-up): five reversed bits, RFC1951 3.2.5
            const uint32_t dc = __builtin_bitreverse32(y) >> 27;
            const uint32_t deb = max(dc >> 1, 1u) - 1u;
            const uint32_t dbase = 1u + (dc < 4u ? dc : ((2u + (dc & 1u)) << deb));
            const uint32_t distance = dbase + ((y >> 5) & ((1u << deb) - 1u));
            const uint32_t mbits = nbm + leb + 5u + deb;
            const uint32_t om = o + nl;                               // where the copy will start
            const bool len_ok = can & (nl == nraw) & ((em & TMASK) == ((uint32_t)T_LEN << 13)) & (mbits < bc) & (dc < 30u) &
                                (distance <= om) & (distance <= obsize) & (om + tlength <= cap);
            const uint32_t lbits = nl == 0u ? 0u : nl == 1u ? n0 : nl == 2u ? off2 : off2 + (e2 & 15u);
            const uint32_t take = len_ok ? mbits : lbits;
            bb >>= take; bc -= take;
            litv = can ? (((e0 >> 4) & 0xFFu) | (((e1 >> 4) & 0xFFu) << 8) | (((e2 >> 4) & 0xFFu) << 16)) : litv;
            litn = can ? nl : litn;
            rem = len_ok ? tlength : rem;
            dist = len_ok ? distance : dist;
            slow = idle & (nl == 0u) & !len_ok;                        // header, EOB, end of the input, invalid data, any failing check
        }
        GT(1);
        // ------------------------------------------------------------ 3. slow path (wave-uniform branch, rare): k_inflate_tok's, verbatim in
        // its checks and their order.  ONE block header or end-of-block per step: the FIFO advances once per step (part 1, one half of
        // 128 bytes), and a step that walked through any number of empty blocks -- 5 bytes per empty stored block, 10 bits per empty
        // fixed one -- would read past what is loaded (~26 empty stored blocks in a row did)
        if (ballot64(slow || (active && srem != 0u)) != 0ull) {
            while (slow && active && rem == 0u && srem == 0u && litn == 0u) {
                GRP_REFILL();
                if (need_header) {
                    // HEADER (deflate.py:677-732)
                    final_ = ((uint32_t)bb & 1u) | oneblock;
                    const uint32_t hm = assume_fixed ? 1u : ((uint32_t)(bb >> 1) & 3u);
                    if (hm == 3u) { GRP_FAIL(HDLZ_E_BAD_BTYPE); break; }
                    if (hm == 2u) { GRP_FAIL(HDLZ_E_DYNAMIC_UNSUPPORTED); break; }
                    need_header = false;
                    if (hm == 0u) {
                        // stored (deflate.py:709-717): LEN sits `skip` bits after the header start
                        const uint32_t dio = GRP_BITPOS() & 7u;
                        uint32_t skip = 8u - dio;
                        if (skip <= 2u) skip = 16u - dio;
                        const uint32_t length = (uint32_t)(bb >> skip) & 0xFFFFu & len_mask;
                        bb >>= (skip + 16u); bc -= (skip + 16u);          // now at NLEN = the reference's di
                        GRP_REFILL();
                        bb >>= 16; bc -= 16u;                             // NLEN unchecked (D2); data follows
                        srem = length;
                        if (length == 0u) {
                            // COPY with nothing to copy (deflate.py:1617-1626)
                            if ((int32_t)(GRP_BITPOS() >> 3) >= isize) { GRP_FAIL(HDLZ_E_NO_EOF); break; }
                            if (final_) { out_len = o; active = false; break; }
                            need_header = true;
                        }
                    } else {
                        bb >>= 3; bc -= 3u;
                    }
                    break;                                               // (the block's first token: next step)
                }
                // NEXT (deflate.py:1409-1445)
                const uint32_t e = lds.lit[(uint32_t)bb & 511u];
                const uint32_t nb = e & 15u, code = (e >> 4) & 0x1FFu;
                if (nb < 1u) { GRP_FAIL(HDLZ_E_BAD_SYMBOL); break; }
                bb >>= nb; bc -= nb;
                // INFLATE (deflate.py:1519-1591)
                if ((int32_t)(GRP_BITPOS() >> 3) > isize - 3) { GRP_FAIL(HDLZ_E_NO_EOF); break; }   // :1535-1539
                if (code == 256u) {
                    if (final_) { out_len = o; active = false; break; }   // D6
                    need_header = true;
                    break;                                               // (the next block's header: next step)
                }
                if (code < 256u) {
                    if (o >= cap) { GRP_FAIL(HDLZ_E_OUT_CAPACITY); break; }
                    litv = code; litn = 1;
                    break;
                }
                const uint32_t token = code - 257u;
                if (token >= 29u) { GRP_FAIL(HDLZ_E_BAD_SYMBOL); break; }
                uint32_t lbase, leb;
                length_info(token, lbase, leb);
                const uint32_t tlength = lbase + ((uint32_t)bb & ((1u << leb) - 1u));
                bb >>= leb; bc -= leb;
                const uint32_t dc = __builtin_bitreverse32((uint32_t)bb) >> 27;
                if (dc >= 30u) { GRP_FAIL(HDLZ_E_BAD_DISTANCE); break; }
                bb >>= 5;
                uint32_t dbase, deb;
                dist_info(dc, dbase, deb);
                const uint32_t distance = dbase + ((uint32_t)bb & ((1u << deb) - 1u));
                bb >>= deb;
                bc -= 5u + deb;
                if (distance > o || distance > obsize) { GRP_FAIL(HDLZ_E_BAD_DISTANCE); break; }        // D8
                if ((int32_t)(GRP_BITPOS() >> 3) >= isize - 2) { GRP_FAIL(HDLZ_E_NO_EOF); break; }      // COPY hold, :1600
                if ((uint64_t)o + tlength > cap) { GRP_FAIL(HDLZ_E_OUT_CAPACITY); break; }
                rem = tlength;
                dist = distance;
            }
            // stored COPY (deflate.py:1603-1616): one byte per step (rare: level-0 streams, incompressible blocks)
            if (active && srem != 0u && litn == 0u && rem == 0u) {
                GRP_REFILL();
                if ((int32_t)(GRP_BITPOS() >> 3) >= isize) { GRP_FAIL(HDLZ_E_NO_EOF); }
                else if (o >= cap) { GRP_FAIL(HDLZ_E_OUT_CAPACITY); }
                else {
                    litv = (uint32_t)bb & 0xFFu; litn = 1;
                    bb >>= 8; bc -= 8u;
                    srem--;
                    if (srem == 0u) {                              // the block ends with this byte (deflate.py:1617-1626)
                        if ((int32_t)(GRP_BITPOS() >> 3) >= isize) { GRP_FAIL(HDLZ_E_NO_EOF); litn = 0; }
                        else if (final_) { out_len = o + 1u; active = false; }      // (the byte is still emitted below)
                        else need_header = true;
                    }
                }
            }
        }
        GT(2);
        // ------------------------------------------------------------ 4. move, up to 16 bytes per step: lanes 0 .. litn - 1 write the literals,
        // the lanes behind them the next bytes of the copy (the literals first: a copy may start right behind them and read them back --
        // the LDS executes a wave's instructions in order)
        {
            const uint32_t kc = min(rem, G - litn);
            if (l < litn) ring[(o + l) & RMASK] = (uint8_t)(litv >> (8u * l));
            const uint32_t lc = l - litn;                             // (wraps for a literal lane: then it is not below kc)
            // byte lc of the copy repeats the pattern of `dist` bytes in front of it: source offset lc mod dist (lc < 16; exact in floats)
            uint32_t r = lc;
            if (ballot64((rem != 0u) & (dist < G)) != 0ull) {
                const uint32_t q = (uint32_t)(((float)(lc & 15u) + 0.5f) * __builtin_amdgcn_rcpf((float)dist));
                r = dist < G ? (lc & 15u) - q * dist : lc;
            }
            const uint32_t src = o + litn - dist + r;
            uint32_t b = 0;
            if (lc < kc) b = ring[src & RMASK];
            if (ballot64((rem != 0u) & (dist > NEAR)) != 0ull) {
                // far history: the stream's own output, flushed by other lanes of this group a while ago (src + 16 <= flushed, see
                // NEAR / FLUSH) -- the stores have to be complete and the load must not be served from this CU's L1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((lc < kc) & (dist > NEAR)) b = __hip_atomic_load(out + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lc < kc) ring[(o + l) & RMASK] = (uint8_t)b;
            o += litn + kc;
            rem -= kc;
            litn = 0u;
        }
        GT(3);
        // ------------------------------------------------------------ 5. flush a kilobyte of finished output: 16 lanes x 16 bytes per store
        {
            const bool fl = (o - flushed) >= FLUSH;
            if (ballot64(fl) != 0ull) {
                if (fl) {
#pragma unroll
                    for (uint32_t q = 0; q < FLUSH / (16u * G); q++) {
                        const uint32_t p = flushed + q * 16u * G + 16u * l;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(ring + (p & RMASK));
                        // (4-byte alignment is all gfx950 asks of a 16-byte store; s_nop: the store-data hazard is not seen inside inline asm)
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(out + p), "v"(v) : "memory");
                    }
                    flushed += FLUSH;
                }
            }
        }
#ifdef HDLZ_GRP_TIMING
        tsteps += 1u;
#endif
        if (ballot64(active || rem != 0u) == 0ull) break;
    }
#undef GRP_REFILL
#undef GRP_FIFO_PUT
#undef GRP_FAIL
#undef GRP_BITPOS
#undef GRP_IP
    // ---- tail: what the ring still holds of [flushed, out_len)
    if (exists && status == HDLZ_OK) {
        uint32_t p = flushed + 16u * l;
        for (; p + 16u <= out_len; p += 16u * G) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(ring + (p & RMASK));
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(out + p), "v"(v) : "memory");
        }
        // (the last partial 16 bytes: p of exactly one lane lies in [out_len - 15, out_len))
        if (p < out_len) for (uint32_t q = p; q < out_len; q++) out[q] = ring[q & RMASK];
    }
#ifdef HDLZ_GRP_TIMING
    if (exists && l == 0u) { a.out_len[sid] = g == 0u ? tacc[0] : g == 1u ? tacc[1] : g == 2u ? tacc[2] : tacc[3]; a.status[sid] = g == 0u ? tsteps : tacc[4]; }
#else
    if (exists && l == 0u) { a.out_len[sid] = out_len; a.status[sid] = status; }
#endif
}

}  // namespace grp

hipError_t launch_inflate_grp(const InflateArgs& a, hipStream_t stream) {
    if (a.nstreams == 0) return hipSuccess;
    const uint64_t per_wg = (uint64_t)grp::NS * grp::WAVES;
    hipLaunchKernelGGL(grp::k_inflate_grp, dim3((unsigned)((a.nstreams + per_wg - 1u) / per_wg)), dim3(64 * grp::WAVES), 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
