// hdlz_inflate.hip -- STARTD for a batch of independent zlib streams on gfx950.
//
// Replaces the reference's inflate FSM for stored + fixed-Huffman blocks
// (/root/reference/deflate.py:635-732 IDLE/HEADER, :1402-1445 NEXT, :1519-1591 INFLATE,
// :1593-1659 COPY, :517-533 get4/adv).  Rule names D0..D8 are SURVEY.md 8(a)'s.
//
// Inflate is serial per stream, so the parallelism is ACROSS streams: one lane per stream, 64
// streams per wave, run in LOCKSTEP: every iteration each active lane produces exactly ONE output
// byte (a literal, a stored byte, or the next byte of a pending LZ copy -- the reference's COPY state
// also moves one byte per clock, deflate.py:1627-1659).  Consequences:
//   * the output offset `o` is wave-uniform, so output is staged in a lane-interleaved LDS ring
//     (dword w of lane l at dword index w*64 + l: every access of the wave hits 64 different banks
//     whatever the per-lane history offset is) and flushed as 64 full 64-byte lines per 64 iterations
//   * LZ copies with distance <= 64 read the ring (ds_read_u8); longer distances (up to OBSIZE)
//     read the stream's own, already flushed, output in HBM/L2
//   * the decode path and the copy path are both short, so divergence between "lane decodes a
//     symbol" and "lane continues a copy" costs the sum of two short paths, not a loop of one.
// The 512-entry fixed-tree leaf table (the reference's stat_leaves, deflate.py:151-216:
// leaf = (sym << 4) | nbits, indexed by the next 9 bits) lives in LDS, generated from RFC1951 3.2.6.
// Input is pulled per lane through a 64-bit bit buffer refilled 4 bytes at a time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"

namespace hdlz {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr uint32_t RING_BYTES = 64;           // history kept in LDS per stream = one flush chunk (4 KiB per wave).
                                              // Occupancy is what matters here: measured 110 / 150 / 202 GB/s for 256 / 128 / 64
constexpr uint32_t RING_DW = RING_BYTES / 4;  // dwords per lane
constexpr uint32_t CHUNK = 64;                // bytes per stream per flush
constexpr uint32_t FAR_BUF_MIN = 79;          // 8-byte far prefetch needs src+15 < flushed end: distance > 78
constexpr uint32_t WAVES = 4;                 // independent waves per workgroup, sharing the decode tables

struct __attribute__((aligned(16))) InflateLds {
    uint32_t ring[WAVES][RING_DW * 64];   // per wave: [dword][lane]
    uint32_t inq[WAVES][64];              // per lane: the stream dword at `ip`, written by LDS-DMA loads (see refill_async)
    uint32_t lit[512];             // literal/length table, see lit_entry()
    uint32_t dst[32];              // distance table indexed by the RAW 5 stream bits
};

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

__device__ __forceinline__ uint32_t rev(uint32_t v, uint32_t nbits) { return __builtin_bitreverse32(v) >> (32u - nbits); }

// 4 stream bytes at byte `ip` (little endian); bytes at or beyond zn read as zero (the reference's
// input memory holds nothing there)
__device__ __forceinline__ uint32_t load32(const uint8_t* __restrict__ z, uint32_t ip, uint32_t zn) {
    if (ip + 4u <= zn) return *reinterpret_cast<const u32_unaligned*>(z + ip);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4u; k++)
        if (ip + k < zn) v |= (uint32_t)z[ip + k] << (8u * k);
    return v;
}

// ---- asynchronous input refill ---------------------------------------------------------------------------
// Every lane reads its own stream.  With the next dword prefetched into a VGPR, the wave had to wait for ITS
// LATEST load before any lane could consume an OLDER one (s_waitcnt counts instructions, not lanes), i.e. one
// full memory latency per lockstep iteration.  The dword now travels HBM -> LDS directly (global_load_lds_dword:
// no destination VGPR, so the compiler adds no wait) and the consumer waits by hand: loads complete in order, so
// a request with `after` LDS-DMA instructions issued behind it has landed once at most `after` loads are
// outstanding.  `issued` counts the LDS-DMA instructions the wave has executed (wave-uniform), every lane
// remembers the count right after its own request, and the wait is vmcnt(min(after over the consuming lanes, 2)).
__device__ __forceinline__ void lds_dma_load32(const uint8_t* gptr, uint32_t lds_base) {
    // LDS address = M0 + lane * 4; only the lanes active here load / write.  M0 is saved and restored: the compiler
    // does not expect inline asm to change it.
    uint32_t save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(gptr), "s"(lds_base) : "memory");
}

// RFC1951 tables in closed form (deflate.py:100-110)
__device__ __forceinline__ void length_info(uint32_t token, uint32_t& base, uint32_t& eb) {
    if (token < 8u) { base = 3u + token; eb = 0; }
    else if (token == 28u) { base = 258u; eb = 0; }
    else { eb = (token >> 2) - 1u; base = 3u + ((4u + (token & 3u)) << eb); }
}
__device__ __forceinline__ void dist_info(uint32_t dc, uint32_t& base, uint32_t& eb) {
    if (dc < 4u) { base = 1u + dc; eb = 0; }
    else { eb = (dc >> 1) - 1u; base = 1u + ((2u + (dc & 1u)) << eb); }
}

// literal/length table entry for the next 9 stream bits (the reference's stat_leaves, deflate.py:151-216,
// widened): nbits[3:0] | sym[12:4] | type[14:13] | lbase[24:16] | leb[27:25]
//   type 0 literal, 1 length symbol 257..285, 2 end of block, 3 invalid (286: nbits 8; 287: nbits 0 =
//   the reference's zero leaf at index 483 -> "< 1 bits")
enum { T_LIT = 0, T_LEN = 1, T_EOB = 2, T_BAD = 3 };
__device__ __forceinline__ uint32_t lit_entry(uint32_t c, bool zero_leaf) {
    uint32_t sym, nb;
    const uint32_t r7 = rev(c & 127u, 7), r8 = rev(c & 255u, 8), r9 = rev(c, 9);
    if (r7 < 24u) { sym = 256u + r7; nb = 7; }                        // 0000000..0010111
    else if (r8 >= 0x30u && r8 < 0xC0u) { sym = r8 - 0x30u; nb = 8; }  // 00110000..10111111
    else if (r8 >= 0xC0u && r8 < 0xC8u) { sym = 280u + (r8 - 0xC0u); nb = 8; }
    else { sym = r9 - 256u; nb = 9; }                                 // 110010000..111111111 -> 144..255
    uint32_t type = sym < 256u ? T_LIT : sym == 256u ? T_EOB : sym <= 285u ? T_LEN : T_BAD;
    uint32_t lbase = 0, leb = 0;
    if (type == T_LEN) length_info(sym - 257u, lbase, leb);
    if (c == 483u && zero_leaf) nb = 0;   // (DYNAMIC=False build only, see hdlz_inflate_tables.h; index 227 is an ordinary leaf)
    return nb | (sym << 4) | (type << 13) | (lbase << 16) | (leb << 25);
}
// distance table entry for the raw 5 bits: dbase[15:0] | deb[19:16], 0xFFFFFFFF for codes 30/31
__device__ __forceinline__ uint32_t dst_entry(uint32_t raw5) {
    const uint32_t dc = rev(raw5, 5);
    if (dc >= 30u) return 0xFFFFFFFFu;
    uint32_t dbase, deb;
    dist_info(dc, dbase, deb);
    return dbase | (deb << 16);
}

__device__ __forceinline__ uint32_t ring_addr(uint32_t pos, uint32_t lane) {
    const uint32_t b = pos & (RING_BYTES - 1u);
    return ((b >> 2) << 8) | (lane << 2) | (b & 3u);     // byte address inside InflateLds::ring
}

__global__ __launch_bounds__(64 * WAVES) void k_inflate(InflateArgs a) {
    __shared__ InflateLds lds;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t c = threadIdx.x; c < 512u; c += 64u * WAVES) lds.lit[c] = lit_entry(c, (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0u);
    if (threadIdx.x < 32u) lds.dst[threadIdx.x] = dst_entry(threadIdx.x);
    __syncthreads();                 // the only workgroup barrier: the waves are independent from here on

    const uint64_t sid0 = ((uint64_t)blockIdx.x * WAVES + wave) * 64u;
    const uint64_t sid = sid0 + lane;
    const bool exists = sid < a.nstreams;
    uint64_t off = 0;
    uint32_t zn = 0;
    if (exists) {
        if (a.in_off) {
            off = a.in_off[sid];
            zn = (uint32_t)(a.in_off[sid + 1] - off);
        } else {
            off = sid * a.in_pitch;
            zn = a.in_len;
        }
    }
    const uint8_t* __restrict__ z = a.in + off;
    uint8_t* out = a.out + sid * a.out_pitch;     // (not __restrict__: the chunk flush stores the same bytes through a.out)
    uint8_t* ring8 = reinterpret_cast<uint8_t*>(lds.ring[wave]);
    const uint32_t cap = a.out_pitch > 0xFFFFFE00ull ? 0xFFFFFE00u : (uint32_t)a.out_pitch;   // o + 258 never wraps
    // obsize != 0: reference-exact OBSIZE build -- the stored LEN register is LOBSIZE bits wide
    // (deflate.py:329,:714), so LEN is taken mod 2^floor(log2(obsize)); obsize == 0: RFC behaviour.
    const uint32_t obsize = a.obsize ? a.obsize : 32768u;
    const uint32_t len_mask = a.obsize ? ((1u << (31u - (uint32_t)__builtin_clz(a.obsize))) - 1u) : 0xFFFFu;
    const bool assume_fixed = (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0;
    const uint32_t oneblock = (a.flags & HDLZ_INFLATE_ONEBLOCK) ? 1u : 0u;       // deflate.py:678,:1542,:1617
    const int32_t isize = (int32_t)zn - 1;            // deflate.py:605

    uint32_t status = HDLZ_OK;
    uint32_t out_len = 0;
    bool active = exists;
    if (exists && zn < 5u) { status = HDLZ_E_SHORT_INPUT; active = false; }

    uint64_t bb = 0;            // bit buffer (LSB first)
    uint32_t bc = 0;            // valid bits in bb
    uint32_t ip = 2;            // D0: next byte to load; the 2 zlib header bytes are skipped unvalidated
    uint32_t rem = 0, dist = 0; // pending LZ copy
    uint32_t srem = 0;          // pending stored bytes
    uint32_t lit = 0;
    uint32_t final_ = 0;
    bool need_header = true;
    uint64_t fb = 0;            // far-copy buffer: up to 8 source bytes fetched from the flushed output
    uint32_t fbn = 0;
    uint64_t fpre = 0;          // ... and the NEXT 8, requested one refill period ahead (latency hiding)
    // (an LDS-typed pointer: through a generic `volatile uint32_t*` the accesses became flat loads, which wait on vmcnt)
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
    lds_vu32* inq = (lds_vu32*)&lds.inq[wave][0];
    // low half of the flat address = LDS offset; wave-uniform, but derived from threadIdx: pin it to an SGPR
    const uint32_t inq_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(&lds.inq[wave][0]));
    uint32_t issued = 0;        // wave-uniform: LDS-DMA load instructions issued so far
    uint32_t myissue = 0;       // value of `issued` when this lane's pending dword was requested
    // request the dword at byte `ip` into inq[lane]: LDS-DMA when it lies fully inside the stream, else assembled
    // from byte loads (zero beyond zn) and written synchronously -- that happens a few times per stream
#define HDLZ_REQUEST(asyncv) do {                                                                        \
        if (ip + 4u <= zn) lds_dma_load32(z + ip, inq_base);                                               \
        else inq[lane] = load32(z, ip, zn);                                                                \
        myissue = (asyncv);                                                                                \
    } while (0)
    // synchronous refill for the rare mid-iteration needs of the slow path
#define HDLZ_REFILL() do { if (bc <= 32u) {                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
        bb |= (uint64_t)inq[lane] << bc; bc += 32u; ip += 4u;                                               \
        inq[lane] = load32(z, ip, zn);                                                                      \
        myissue = issued - 1000u;                                                                           \
    } } while (0)
    if (active) { HDLZ_REQUEST(issued + 1u); }
    if (ballot64(active && ip + 4u <= zn) != 0ull) issued += 1u;      // counts LDS-DMA instructions actually executed

#define HDLZ_FAIL(code) do { status = (code); out_len = 0; active = false; } while (0)
#define HDLZ_BITPOS() (8u * ip - bc)

    bool any_stored = false;    // wave-uniform: some lane is inside a stored block (set by the slow path, which is where they start)
    for (uint32_t o = 0;; ++o) {
        // ------------------------------------------------------------ 0. input refill (uniform control flow)
        {
            const bool need = active && rem == 0u && bc <= 32u;
            if (ballot64(need) != 0ull) {
                // loads complete in order: a request with `after` LDS-DMA instructions issued behind it has landed once at
                // most `after` loads are outstanding; wait for the youngest request that is consumed now, no further
                const uint32_t after = issued - myissue;
                if (ballot64(need && after < 2u) == 0ull) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // the usual case first
                else if (ballot64(need && after < 1u) == 0ull) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bool dma = false;
                if (need) {
                    bb |= (uint64_t)inq[lane] << bc; bc += 32u; ip += 4u;
                    dma = ip + 4u <= zn;
                    HDLZ_REQUEST(issued + 1u);
                }
                if (ballot64(dma) != 0ull) issued += 1u;          // counts LDS-DMA instructions actually executed
            }
        }
        // ------------------------------------------------------------ 1a. fast path: literal / match inside a fixed block
        bool have = false;
        bool slow = false;
        if (active && rem == 0u && srem == 0u) {
            slow = need_header;                                    // (bc >= 33 here: refilled in step 0)
            const uint32_t e = lds.lit[(uint32_t)bb & 511u];
            const uint32_t nb = e & 15u, type = (e >> 13) & 3u;
            const uint32_t leb = (e >> 25) & 7u, lbase = (e >> 16) & 0x1FFu;
            uint64_t x = bb >> nb;
            const uint32_t tlength = lbase + ((uint32_t)x & ((1u << leb) - 1u));
            x >>= leb;
            const uint32_t de = lds.dst[(uint32_t)x & 31u];
            const uint32_t deb = (de >> 16) & 15u;
            const uint32_t distance = (de & 0xFFFFu) + ((uint32_t)(x >> 5) & ((1u << deb) - 1u));
            const uint32_t mbits = nb + leb + 5u + deb;
            // input guard: after the refill bc >= 33, so every bit position touched by this token is below
            // byte ip; with ip + 3 <= zn both reference checks (deflate.py:1535-1539 after the symbol, :1600
            // before the copy) are guaranteed to pass -- anything closer to the end goes to the slow path
            const bool in_ok = ip + 3u <= zn;
            const bool lit_ok = (type == (uint32_t)T_LIT) & in_ok & (o < cap);
            const bool len_ok = (type == (uint32_t)T_LEN) & in_ok & (de != 0xFFFFFFFFu) & (distance <= o) &
                                (distance <= obsize) & (o + tlength <= cap);
            if (!slow && (lit_ok | len_ok)) {
                const uint32_t used = lit_ok ? nb : mbits;
                bb >>= used; bc -= used;
                if (lit_ok) { lit = (e >> 4) & 0xFFu; have = true; }
                else {
                    rem = tlength; dist = distance; fbn = 0;
                    if (distance >= FAR_BUF_MIN) fpre = *reinterpret_cast<const u64_unaligned*>(out + (o - distance));
                }
            } else {
                slow = true;                                        // EOB, header, invalid data, any failing check
            }
        }
        // ------------------------------------------------------------ 1b. slow path (wave-uniform branch, rare)
        if (ballot64(slow) != 0ull) {
            while (slow && active && rem == 0u && srem == 0u && !have) {
                HDLZ_REFILL();
                if (need_header) {
                    // HEADER (deflate.py:677-732)
                    final_ = ((uint32_t)bb & 1u) | oneblock;
                    const uint32_t hm = assume_fixed ? 1u : ((uint32_t)(bb >> 1) & 3u);
                    if (hm == 3u) { HDLZ_FAIL(HDLZ_E_BAD_BTYPE); break; }
                    if (hm == 2u) { HDLZ_FAIL(HDLZ_E_DYNAMIC_UNSUPPORTED); break; }
                    need_header = false;
                    if (hm == 0u) {
                        // stored (deflate.py:709-717): LEN sits `skip` bits after the header start
                        const uint32_t dio = HDLZ_BITPOS() & 7u;
                        uint32_t skip = 8u - dio;
                        if (skip <= 2u) skip = 16u - dio;
                        const uint32_t length = (uint32_t)(bb >> skip) & 0xFFFFu & len_mask;
                        bb >>= (skip + 16u); bc -= (skip + 16u);          // now at NLEN = the reference's di
                        HDLZ_REFILL();
                        bb >>= 16; bc -= 16u;                             // NLEN unchecked (D2); data follows
                        srem = length;
                        if (length == 0u) {
                            // COPY with nothing to copy (deflate.py:1617-1626)
                            if ((int32_t)(HDLZ_BITPOS() >> 3) >= isize) { HDLZ_FAIL(HDLZ_E_NO_EOF); break; }
                            if (final_) { out_len = o; active = false; break; }
                            need_header = true;
                        }
                    } else {
                        bb >>= 3; bc -= 3u;
                    }
                    continue;
                }
                // NEXT (deflate.py:1409-1445)
                const uint32_t e = lds.lit[(uint32_t)bb & 511u];
                const uint32_t nb = e & 15u, code = (e >> 4) & 0x1FFu;
                if (nb < 1u) { HDLZ_FAIL(HDLZ_E_BAD_SYMBOL); break; }
                bb >>= nb; bc -= nb;
                // INFLATE (deflate.py:1519-1591)
                if ((int32_t)(HDLZ_BITPOS() >> 3) > isize - 3) { HDLZ_FAIL(HDLZ_E_NO_EOF); break; }   // :1535-1539
                if (code == 256u) {
                    if (final_) { out_len = o; active = false; break; }   // D6
                    need_header = true;
                    continue;
                }
                if (code < 256u) {
                    if (o >= cap) { HDLZ_FAIL(HDLZ_E_OUT_CAPACITY); break; }
                    lit = code;
                    have = true;
                    break;
                }
                const uint32_t token = code - 257u;
                if (token >= 29u) { HDLZ_FAIL(HDLZ_E_BAD_SYMBOL); break; }
                uint32_t lbase, leb, dbase, deb;
                length_info(token, lbase, leb);
                const uint32_t tlength = lbase + ((uint32_t)bb & ((1u << leb) - 1u));
                bb >>= leb;
                const uint32_t dc = rev((uint32_t)bb & 31u, 5);
                bb >>= 5;
                if (dc >= 30u) { HDLZ_FAIL(HDLZ_E_BAD_DISTANCE); break; }
                dist_info(dc, dbase, deb);
                const uint32_t distance = dbase + ((uint32_t)bb & ((1u << deb) - 1u));
                bb >>= deb;
                bc -= leb + 5u + deb;
                if (distance > o || distance > obsize) { HDLZ_FAIL(HDLZ_E_BAD_DISTANCE); break; }        // D8
                if ((int32_t)(HDLZ_BITPOS() >> 3) >= isize - 2) { HDLZ_FAIL(HDLZ_E_NO_EOF); break; }      // COPY hold, :1600
                if ((uint64_t)o + tlength > cap) { HDLZ_FAIL(HDLZ_E_OUT_CAPACITY); break; }
                rem = tlength;
                dist = distance;
                fbn = 0;
                if (distance >= FAR_BUF_MIN) fpre = *reinterpret_cast<const u64_unaligned*>(out + (o - distance));
            }
            any_stored = any_stored || (ballot64(active && srem != 0u) != 0ull);
        }
        if (ballot64(active) == 0ull) break;

        // ------------------------------------------------------------ 2. one output byte per active lane
        bool wrote = false;        // this lane put a byte into the ring in this iteration
        uint32_t byte = lit;
        bool stored_done = false;
        if (any_stored) {                                          // stored COPY (deflate.py:1603-1616): rare, uniform branch
            if (active && srem != 0u) {
                HDLZ_REFILL();
                if ((int32_t)(HDLZ_BITPOS() >> 3) >= isize) { HDLZ_FAIL(HDLZ_E_NO_EOF); }
                else if (o >= cap) { HDLZ_FAIL(HDLZ_E_OUT_CAPACITY); }
                byte = (uint32_t)bb & 0xFFu;
                bb >>= 8; bc -= 8u;
                srem--;
                stored_done = (srem == 0u);
            }
        }
        if (active) {
            if (rem != 0u) {                                       // COPY (deflate.py:1627-1659)
                const uint32_t rb = ring8[ring_addr(o - dist, lane)];     // LDS history (valid for dist <= RING_BYTES)
                byte = rb;
                if (dist >= FAR_BUF_MIN) {                         // far history: the stream's own flushed output
                    if (fbn == 0u) {          // take the prefetched 8 bytes, request the following 8 (all already flushed)
                        fb = fpre; fbn = 8u;
                        if (rem > 8u) fpre = *reinterpret_cast<const u64_unaligned*>(out + (o - dist) + 8u);
                    }
                    byte = (uint32_t)fb & 0xFFu;
                    fb >>= 8; fbn--;
                } else if (dist > RING_BYTES) {                    // 65..78: flushed, but too close for the 8-byte prefetch
                    byte = out[o - dist];
                }
                rem--;
            }
            ring8[ring_addr(o, lane)] = (uint8_t)byte;
            wrote = true;
        }
        if (any_stored) {                                          // a stored block may just have ended (deflate.py:1617-1626)
            if (active && stored_done) {
                if ((int32_t)(HDLZ_BITPOS() >> 3) >= isize) { HDLZ_FAIL(HDLZ_E_NO_EOF); }
                else if (final_) { out_len = o + 1u; active = false; }
                else need_header = true;
            }
            any_stored = ballot64(active && srem != 0u) != 0ull;    // leave stored mode when the last such block ended
        }

        // ------------------------------------------------------------ 3. flush 64 bytes per stream every 64 iterations
        if ((o & (CHUNK - 1u)) == CHUNK - 1u) {
            const uint64_t live = ballot64(wrote);      // lanes that filled this whole chunk (incl. one finishing on it)
            const uint32_t c0 = o - (CHUNK - 1u);                       // first byte of the chunk
            const uint32_t w0 = (c0 & (RING_BYTES - 1u)) >> 2;          // its ring dword
            const uint32_t q = lane & 3u;
#pragma unroll
            for (uint32_t r = 0; r < 4u; r++) {
                const uint32_t s = (lane >> 2) + 16u * r;               // stream (lane index) this lane copies for
                if ((live >> s) & 1ull) {
                    const uint32_t* src = &lds.ring[wave][(w0 + 4u * q) * 64u + s];
                    u32x4_t v;
                    v.x = src[0]; v.y = src[64]; v.z = src[128]; v.w = src[192];
                    uint8_t* dst = a.out + (sid0 + s) * a.out_pitch + c0 + 16u * q;
                    // one 16-byte store whatever the alignment (4 bytes is all gfx950 asks for): written as `aligned ? x4 : 4 x dword`
                    // hipcc merged the two branches into a dwordx3 + dword pair (see hdlz_inflate_tok.hip; s_nop: store-data hazard)
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
                }
            }
            // later far copies (distance > 256) read these bytes back through L1/L2 (the stores above are asm: the compiler does
            // not count them, the release fence would not wait for them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
#undef HDLZ_FAIL
#undef HDLZ_BITPOS
#undef HDLZ_REFILL
#undef HDLZ_REQUEST

    // no LDS-DMA load may still be in flight when this wave's LDS is handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- tail: the bytes of the last, partial chunk are still only in the ring
    if (exists && status == HDLZ_OK) {
        const uint32_t c0 = out_len & ~(CHUNK - 1u);
        for (uint32_t p = c0; p < out_len; p++) out[p] = ring8[ring_addr(p, lane)];
    }
    if (exists) {
        a.out_len[sid] = out_len;
        a.status[sid] = status;
    }
}

hipError_t launch_inflate(const InflateArgs& a, hipStream_t stream) {
    if (a.nstreams == 0) return hipSuccess;
    const uint64_t per_wg = 64u * WAVES;
    const dim3 grid((unsigned)((a.nstreams + per_wg - 1u) / per_wg)), block(64 * WAVES);
    hipLaunchKernelGGL(k_inflate, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
