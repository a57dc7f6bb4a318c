// hdlz_inflate_tok.hip -- STARTD for a batch of independent zlib streams, one LANE per stream, TOKENS (not bytes) per round.
//
// Replaces the reference's inflate FSM (/root/reference/deflate.py:635-732 IDLE/HEADER, :1402-1445 NEXT, :1519-1591 INFLATE,
// :1593-1659 COPY, :517-533 get4/adv); rule names D0..D8 are SURVEY.md 8(a)'s.  Inflate is serial per stream, so the parallelism is
// ACROSS streams: 64 per wave.  (Round 1's kernel ran them in lockstep ONE OUTPUT BYTE per iteration -- the reference's COPY state
// moves one byte per clock too, deflate.py:1627-1659 -- and paid the whole token decode, ~50 VALU + ~40 SALU instructions, every
// iteration although only the lanes at a token boundary need it; it was slower everywhere measured and left with round 6.)  Here a ROUND decodes, in every lane, up to three literals and the match behind
// them, and a short loop moves the bytes: up to four per lane and iteration (the literals, or the next bytes of a copy out of the
// ring) and, behind the loop, up to sixteen bytes of far history in one step, so the decode is paid once per token.  What changes with it:
//   * every lane has its own output position.  The ring stays lane-interleaved (dword w of lane l at dword index w*64 + l: each
//     lane owns a bank, whatever the positions are) and is 128 bytes per lane; two dwords (far step: three to five) are written at
//     once, unmasked -- the bytes behind the new end are not-yet-produced positions whose old content (history > 112 back) is never read again;
//   * a lane's 64-byte line is flushed when complete, by the lane itself (16 conflict-free ds_read_b32 + 4 x global_store_dwordx4: a
//     full 64-byte sector per lane); flushes are batched: they run when a quarter of the wave is ready or one lane is about to overrun;
//   * near history (distance <= 112) is read from the ring, far history from the stream's own flushed output (one 16-byte load
//     when the token is decoded, taken a round later; a longer far copy asks for its next 16 bytes when it takes the last ones);
//   * input arrives through 16-byte LDS-DMA slots per lane.  The ordering rule: with the next bytes prefetched into a VGPR the wave
//     would wait for ITS LATEST load before any lane could consume an OLDER one (s_waitcnt counts instructions, not lanes); an LDS-DMA
//     load has no destination VGPR, so the compiler adds no wait, and the consumer waits by hand -- loads complete in order, so a
//     request with `after` LDS-DMA instructions issued behind it has landed once at most `after` loads are outstanding.
// Status codes follow the ORDER of the reference's checks: the wave-uniform slow path (headers, EOB, stored blocks, failing checks).
//
// DYN = true is the same kernel for streams with DYNAMIC-TREE blocks (BTYPE = 2; deflate.py:1084-1202 BL/READBL/REPEAT, :1204-1400
// HF1..HF4/SPREAD, :1447-1517 D_NEXT), the second pass over the streams the first one flagged HDLZ_E_DYNAMIC_UNSUPPORTED.  A lane
// cannot hold a 512-entry look-up table per stream, but a canonical code needs none: with the codes left-aligned to 15 bits the
// codes of length <= l end at hi[l], and the word X[l] = hi[l] << 16 | l << 9 | index of the last symbol of length <= l, minus the
// reversed stream bits R = r << 16 | 0xFFFF, is -- for the right length -- the smallest of the fifteen differences that did not
// wrap: 15 v_sub + 7 v_min3 per symbol, all registers (the key-difference trick of the compress search, hdlz_compress_common.h).
// What is left per lane in LDS is the symbol list sorted by (length, value): 286 bytes + 286 ninth bits + 30 distance symbols,
// 440 bytes with the counters of the build -- 37 KB per wave with the ring, four waves per CU.  The code lengths themselves are
// never stored: the header is decoded twice, once to count the lengths and once to place the symbols.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"
#include "hdlz_inflate_tables.h"

namespace hdlz {
namespace tok {

#ifndef HDLZ_TOK_RING
#define HDLZ_TOK_RING 128
#endif
constexpr uint32_t RINGB = HDLZ_TOK_RING;     // ring bytes per stream (128 or 64)
constexpr uint32_t RING_DW = RINGB / 4;
constexpr uint32_t NEAR = RINGB - 16;         // distances up to this are served from the ring: NEAR + 3 (dword alignment of the
                                              // source) + 12 (the unmasked write ahead of the end) < RINGB
constexpr uint32_t CHUNK = RINGB >= 128u ? 64u : RINGB / 2u;   // bytes per flush
constexpr uint32_t URGENT = RINGB - 52;       // a lane with this many unflushed bytes forces a flush.  ONE flush check per round, behind the far
                                              // step: a round advances o by at most 19 (three literals + 16 far bytes; 12 out of the ring), so
                                              // o - flushed < URGENT + 19 when a far load is issued, and the 16 bytes it reads end at
                                              // o - dist + 16 <= o - NEAR + 15 = o - 97: below `flushed`.  (The ring itself only needs
                                              // URGENT + 19 written bytes + 11 written ahead < RINGB.)
constexpr uint32_t SLOT_DW = 4;               // input dwords fetched per lane and refill
constexpr uint32_t BATCH = 16;             // lanes with a complete line that start a flush
#ifndef HDLZ_TOK_STEP2_MIN
#define HDLZ_TOK_STEP2_MIN 16
#endif
constexpr uint32_t STEP2_MIN = HDLZ_TOK_STEP2_MIN;   // lanes of a wave that can take a SECOND group of tokens in a round (0 = never; fixed blocks)
constexpr uint32_t MOVES = 3;              // move iterations (up to 4 bytes per lane each) per round; round 3, with the far copies in the loop: 2 / 3 / 4 / 6: 4.21 / 3.95 / 4.10 / 3.99 ms
#ifdef HDLZ_TOK_MARKS                         // tools/phase_count.py --src hdlz_inflate_tok.hip -DHDLZ_TOK_MARKS: static counts per part
#define TOK_MARK(name) asm volatile("; @@PHASE " name ::: "memory")
#else
#define TOK_MARK(name) do {} while (0)
#endif
#ifdef HDLZ_TOK_TIMING                        // diagnostic build (tools/exp_tok_timing.py): s_memtime per part of the round; lanes 0..7 of every
                                              // wave report the wave's totals in out_len / status INSTEAD of the results
#define TOK_TIME_DECL() uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter()
#define TOK_TIME(k) do { const uint64_t t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } while (0)
#define TOK_TIME_ROUND() tacc[7] += 1u
#define TOK_STORE_RESULT() do { TOK_TIME(4);       /* -DHDLZ_TOK_TIMING=1: the DYN = false kernel reports, =2: the DYN = true kernels */ \
        if (DYN != (HDLZ_TOK_TIMING == 2)) { if (exists) { a.out_len[sid] = out_len; a.status[sid] = status; } }  \
        else if (exists) { uint64_t v_ = 0;          /* by GRID lane; lanes >= 8 report 0 */                          \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; k_++) v_ = lane == (uint32_t)k_ ? tacc[k_] : v_;             \
        a.out_len[gid] = (uint32_t)v_; a.status[gid] = (uint32_t)(v_ >> 32); } } while (0)
#else
#define TOK_TIME_DECL() do {} while (0)
#define TOK_TIME(k) do {} while (0)
#define TOK_TIME_ROUND() do {} while (0)
#define TOK_STORE_RESULT() do { if (exists) { a.out_len[sid] = out_len; a.status[sid] = status; } } while (0)
#endif

// DYN: per-lane tables in LDS, rows of 64 dwords (row j of lane l = dword j * 64 + l: every lane stays in its own bank): only the
// literal/length symbols, sorted by (code length, value) -- CAP low bytes and CAP ninth bits.  Everything else a lane needs of its
// codes is in registers: the X words, the sorted distance symbols (30 x 5 bits), and during the build the code-length code and the
// per-length counters.  CAP = 144 (41 rows = 10.25 KB, 19.3 KB per wave with ring and input slot: EIGHT waves per CU) holds the codes
// of small blocks; a stream whose block codes more symbols stays flagged and goes through the CAP = 288 build (81 rows, 5 per CU).
#ifndef HDLZ_TOK_CAP_SMALL
#define HDLZ_TOK_CAP_SMALL 144
#endif
constexpr uint32_t CAP_SMALL = HDLZ_TOK_CAP_SMALL, CAP_FULL = 288;
template <uint32_t CAP> struct Tab {
    static constexpr uint32_t LS8 = 0;                         // CAP bytes: the low 8 bits of the symbols
    static constexpr uint32_t LBIT = CAP / 4;                  // CAP bits: their ninth bit
    static constexpr uint32_t ROWS = CAP / 4 + (CAP + 31) / 32;
};

template <bool DYN, uint32_t CAP> struct Lds;
template <uint32_t CAP>
struct __attribute__((aligned(16))) Lds<false, CAP> {
    static constexpr uint32_t WAVES = 4u;
    uint32_t ring[WAVES][RING_DW * 64];   // per wave: [dword][lane]
    uint32_t inq[WAVES][64 * SLOT_DW];    // per lane: SLOT_DW stream dwords from byte `sbase` on (LDS-DMA)
    uint32_t lit[512];
    uint32_t dst[32];
};
template <uint32_t CAP>
struct __attribute__((aligned(16))) Lds<true, CAP> {
    static constexpr uint32_t WAVES = 1u;
    uint32_t ring[WAVES][RING_DW * 64];
    uint32_t inq[WAVES][64 * SLOT_DW];
    uint32_t tab[Tab<CAP>::ROWS * 64];
};

// sixteen counters in registers, addressed per lane: 16-bit fields in four 64-bit words
struct F16 {
    uint64_t w[4];
    __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = w[3] = 0ull; }
    __device__ __forceinline__ void add(uint32_t i, uint32_t v) {
        const uint64_t a = (uint64_t)v << ((i & 3u) * 16u);
        const uint32_t j = i >> 2;
        w[0] += j == 0u ? a : 0ull; w[1] += j == 1u ? a : 0ull; w[2] += j == 2u ? a : 0ull; w[3] += j == 3u ? a : 0ull;
    }
    __device__ __forceinline__ uint32_t get(uint32_t i) const {
        const uint32_t j = i >> 2;
        const uint64_t x = j == 0u ? w[0] : j == 1u ? w[1] : j == 2u ? w[2] : w[3];
        return (uint32_t)(x >> ((i & 3u) * 16u)) & 0xFFFFu;
    }
    // (compile-time index)
    __device__ __forceinline__ uint32_t at(int i) const { return (uint32_t)(w[i >> 2] >> ((i & 3) * 16)) & 0xFFFFu; }
    __device__ __forceinline__ void put(int i, uint32_t v) { w[i >> 2] |= (uint64_t)v << ((i & 3) * 16); }
};
// up to 36 five-bit symbols in three 64-bit words (twelve each), addressed per lane
struct S5 {
    uint64_t w[3];
    __device__ __forceinline__ void clear() { w[0] = w[1] = w[2] = 0ull; }
    __device__ __forceinline__ void put(uint32_t q, uint32_t sym) {
        const uint32_t j = (q * 43u) >> 9;                      // q / 12 for q < 36
        const uint64_t a = (uint64_t)sym << ((q - 12u * j) * 5u);
        w[0] |= j == 0u ? a : 0ull; w[1] |= j == 1u ? a : 0ull; w[2] |= j == 2u ? a : 0ull;
    }
    __device__ __forceinline__ uint32_t get(uint32_t q) const {
        const uint32_t j = (q * 43u) >> 9;
        const uint64_t x = j == 0u ? w[0] : j == 1u ? w[1] : w[2];
        return (uint32_t)(x >> ((q - 12u * j) * 5u)) & 31u;
    }
};

// (16 bytes per lane straight into LDS: lds_dma16, hdlz_device.h -- LDS address = M0 + lane * 16, see tools/ubench/lds_dma.hip)
// byte address of stream position `pos` of lane `lane` inside a wave's ring
__device__ __forceinline__ uint32_t ring_addr(uint32_t pos, uint32_t lane4) { return ((pos & (RINGB - 4u)) << 6) | lane4 | (pos & 3u); }

// the X words of a canonical code (see the head of the file), from the per-length symbol counts c(1..NL):
//   X[l-1] = (hi_l << 16 | l << 9 | last_l) - 1,  hi_l = (number of codes of length <= l, as l-bit values) << (15 - l)
// `left` is zlib's / puff's completeness count (0 = complete, < 0 = over-subscribed), `cum` the number of coded symbols
template <int NL, class CountOf>
__device__ __forceinline__ void x_build(uint32_t (&X)[NL], CountOf count_of, int32_t& left, uint32_t& cum) {
    uint32_t first = 0;
    left = 1; cum = 0;
#pragma unroll
    for (int l = 1; l <= NL; l++) {
        const uint32_t c = count_of(l);
        left = (left << 1) - (int32_t)c;
        cum += c;
        first += c;
        X[l - 1] = (((first << (15 - l)) << 16) | ((uint32_t)l << 9) | (max(cum, 1u) - 1u)) - 1u;
        first <<= 1;
    }
}
// the code in front of `bits` (LSB first): its length and the index of its symbol in the sorted list.  A difference that wrapped
// belongs to a length whose codes all lie below the stream bits; equal hi (lengths without codes) tie on the shorter length.
template <int NL>
__device__ __forceinline__ void x_decode(const uint32_t (&X)[NL], uint32_t bits, uint32_t& len, uint32_t& idx, bool& valid) {
    const uint32_t R = (__builtin_bitreverse32(bits) >> 1) | 0xFFFFu;
    uint32_t d[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) d[l] = X[l] - R;
    uint32_t m = d[0];
    static_assert(NL % 2 == 1, "min3 tree");
#pragma unroll
    for (int l = 1; l < NL; l += 2) m = min(m, min(d[l], d[l + 1]));
    valid = (int32_t)m >= 0;
    len = (m >> 9) & 15u;
    idx = (m & 511u) - ((m >> 16) >> (15u - len));
}

constexpr uint32_t ORDER_BAD = 0xFFFFFFFFu;      // list length word of a counting sort whose counts did not add up (see bin_flush_and_finish)
template <bool DYN, uint32_t CAP>
__global__ __launch_bounds__(DYN ? 64 : 256) void k_inflate_tok(InflateArgs a, const uint32_t* __restrict__ list,
                                                                const uint32_t* __restrict__ list_n, uint32_t lane_min,
                                                                const uint32_t* __restrict__ order_word = nullptr,
                                                                const uint32_t* __restrict__ list_alt = nullptr
                                                                ) {
    constexpr uint32_t WAVES = Lds<DYN, CAP>::WAVES;
    __shared__ Lds<DYN, CAP> lds;
    if (list != nullptr && *list_n != ORDER_BAD && (uint64_t)blockIdx.x * (64u * WAVES) >= (uint64_t)*list_n) return;      // (the grid covers the longest possible list)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    if constexpr (!DYN) {
        for (uint32_t c = threadIdx.x; c < 512u; c += 64u * WAVES) lds.lit[c] = lit_entry(c, (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0u);
        if (threadIdx.x < 32u) lds.dst[threadIdx.x] = dst_entry(threadIdx.x);
        __syncthreads();             // the only workgroup barrier: the waves are independent from here on
    }
    // DYN: lane g of the grid takes the g-th stream of the list k_collect_dyn made of the streams pass 1 could not finish
    const uint64_t gid = ((uint64_t)blockIdx.x * WAVES + wave) * 64u + lane;
    uint64_t sid = gid;
    bool exists = gid < a.nstreams;
    // the length-ordered list is only used when the counting sort vouched for it (bin_flush_and_finish: ORDER_BAD = its counts did not
    // add up): else the unordered list it was made from (dynamic-tree pass) or plain stream order (pass 1) -- slower lanes, same results
    if (list && order_word && *order_word == ORDER_BAD) list = list_alt;
    if (list) {
        if (*list_n < lane_min) return;                            // few such streams: k_inflate_dyn takes them, one wave each
        exists = gid < (uint64_t)*list_n;
        sid = exists ? (uint64_t)list[gid] : 0ull;
    }
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
    uint8_t* out = a.out + sid * a.out_pitch;
    uint8_t* ring8 = reinterpret_cast<uint8_t*>(lds.ring[wave]);
    const uint32_t lane4 = lane << 2;
    const uint32_t cap = a.out_pitch > 0xFFFFFE00ull ? 0xFFFFFE00u : (uint32_t)a.out_pitch;   // o + 258 never wraps
    const uint32_t obsize = a.obsize ? a.obsize : 32768u;
    const uint32_t len_mask = a.obsize ? ((1u << (31u - (uint32_t)__builtin_clz(a.obsize))) - 1u) : 0xFFFFu;   // deflate.py:329,:714
    const bool assume_fixed = (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0;
    const uint32_t oneblock = (a.flags & HDLZ_INFLATE_ONEBLOCK) ? 1u : 0u;
    const int32_t isize = (int32_t)zn - 1;            // deflate.py:605

    uint32_t status = HDLZ_OK;
    uint32_t out_len = 0;
    bool active = exists;
    if (exists && zn < 5u) { status = HDLZ_E_SHORT_INPUT; active = false; }

    uint64_t bb = 0;            // bit buffer (LSB first)
    uint32_t bc = 0;            // valid bits in bb
    uint32_t ip = 2;            // D0: next byte to load; the 2 zlib header bytes are skipped unvalidated
    uint32_t sbase = 2;         // first stream byte of this lane's input slot in LDS (the 16 bytes that are in flight or have landed)
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 0;   // the qn stream dwords in front of the slot, [ip, sbase): a slot is DRAINED into these
                                // registers as soon as a lane opens it, and the next 16 bytes are requested at once -- four refills (two
                                // in a literal run) before they are needed instead of one (round 4: the wave waited ~400 cycles per
                                // round for a request made a round earlier, profiles/r04_tok_refill.txt); one ds_read_b128 per slot
    uint32_t o = 0;             // bytes produced by THIS lane
    uint32_t flushed = 0;       // ... of which in HBM (a multiple of 64)
    uint32_t pend = 0;          // the valid low bytes of the ring dword that holds position o
    uint32_t rem = 0, dist = 0; // pending LZ copy
    u32x4 far4 = {0u, 0u, 0u, 0u};                  // far copy: the next 16 source bytes, requested a round ahead
    uint32_t litv = 0, litn = 0;// pending literals (up to three) / a stored byte
    // a SECOND pending group (fixed blocks): up to three literals and a NEAR match behind the first group.  A round decodes one group per lane
    // -- at ~400 instructions per round a stream of short matches or of literals advanced 3..4 bytes per round --; when enough lanes of the
    // wave have their first group pending and room for a second one, a second decode step fills it.  The move loop takes the groups in order.
    [[maybe_unused]] uint32_t litv2 = 0, litn2 = 0, rem2 = 0, dist2 = 0;
    // wave-uniform (SGPRs): some lane has a second group pending; rounds for which step 2 stays off, and the next such pause.  Streams whose
    // matches are mostly FAR (stock zlib, 32 KiB window) rarely get a whole second group: a wave whose step 2 fills less than half of its
    // lanes pauses it for 32, 64, ... 1024 rounds, and while nothing is pending the move loop's hand-over costs one scalar branch
    [[maybe_unused]] bool s2_live = false;
    [[maybe_unused]] uint32_t s2_hold = 0, s2_back = 32;
    uint32_t srem = 0;          // pending stored bytes
    uint32_t final_ = 0;
    bool need_header = true;
    typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
    lds_vu32* inq = (lds_vu32*)&lds.inq[wave][0];
    const uint32_t inq_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(&lds.inq[wave][0]));
    uint32_t issued = 0;        // wave-uniform: LDS-DMA instructions issued so far
    uint32_t myissue = 0;       // value of `issued` right after this lane's pending request

#define TOK_REQUEST(asyncv) do {                                                                        \
        if (sbase + 16u <= zn) lds_dma16(z + sbase, inq_base);                                       \
        else for (uint32_t k_ = 0; k_ < SLOT_DW; k_++) inq[lane * SLOT_DW + k_] = load32(z, sbase + 4u * k_, zn); \
        myissue = (asyncv);                                                                               \
    } while (0)
    typedef __attribute__((address_space(3))) volatile u32x4 lds_v4;
    // the next queued dword goes into the bit buffer
#define TOK_POP() do { bb |= (uint64_t)q0 << bc; bc += 32u; ip += 4u; q0 = q1; q1 = q2; q2 = q3; qn -= 1u; } while (0)
    // the lane's slot -> the queue (the slot must have landed); the slot then stands for the NEXT 16 bytes
#define TOK_DRAIN() do { const u32x4 s4_ = *(lds_v4*)(inq + lane * SLOT_DW);                              \
        q0 = s4_.x; q1 = s4_.y; q2 = s4_.z; q3 = s4_.w; qn = 4u; sbase += 16u; } while (0)
    // synchronous refill for the slow path (divergent code: no LDS-DMA, `issued` stays wave-uniform)
#define TOK_REFILL() do { if (bc <= 32u) {                                                              \
        if (qn == 0u) {                                                                                    \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                               \
            TOK_DRAIN();                                                                                   \
            for (uint32_t k_ = 0; k_ < SLOT_DW; k_++) inq[lane * SLOT_DW + k_] = load32(z, sbase + 4u * k_, zn); \
            myissue = issued - 1000u;                                                                      \
        }                                                                                                  \
        TOK_POP();                                                                                         \
    } } while (0)
#define TOK_FAIL(code) do { status = (code); out_len = 0; active = false; } while (0)
#define TOK_BITPOS() (8u * ip - bc)
    // flush the completed 64-byte lines: batched -- when a quarter of the wave is ready or a lane is about to overrun its ring
#define TOK_FLUSH() do {                                                                                \
        const bool ready_ = (o - flushed) >= CHUNK;        /* (a lane without a stream keeps o = flushed = 0) */ \
        const uint64_t rm_ = ballot64(ready_);                                                             \
        if (rm_ != 0ull && (__popcll(rm_) >= (int)BATCH || ballot64((o - flushed) >= URGENT) != 0ull)) {      \
            if (ready_) {                                                                                  \
                const uint32_t* rp_ = &lds.ring[wave][((flushed & (RINGB - 1u)) >> 2) * 64u + lane];       \
                uint8_t* dp_ = out + flushed;                                                              \
                _Pragma("unroll") for (uint32_t q_ = 0; q_ < CHUNK / 16u; q_++) {                          \
                    u32x4 v_;                                                                              \
                    v_.x = rp_[(4u * q_) * 64u]; v_.y = rp_[(4u * q_ + 1u) * 64u];                         \
                    v_.z = rp_[(4u * q_ + 2u) * 64u]; v_.w = rp_[(4u * q_ + 3u) * 64u];                    \
                    /* one 16-byte store whatever the alignment (d_out / out_pitch are 4-byte aligned, gfx950 needs no more): */ \
                    /* written as `aligned ? x4 : 4 x dword` hipcc merged the two branches into TEN stores per line (TA 80 % busy). */ \
                    /* s_nop: the data registers of a store wider than 8 bytes must not be written in the next two wait states, and  */ \
                    /* the hazard recognizer does not look into inline asm                                                            */ \
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dp_ + 16u * q_), "v"(v_) : "memory");     \
                }                                                                                          \
                flushed += CHUNK;                                                                          \
            }                                                                                              \
            issued += CHUNK / 16u;                         /* (VMEM instructions of the wave: see the refill) */ \
            /* (no fence: a far copy reads bytes that THIS lane stored -- program order of one thread; k_inflate's lines are */ \
            /* stored by other lanes and need the workgroup-scope release / acquire, a full store round trip per flush)     */ \
        }                                                                                                  \
    } while (0)

    // ---- the symbol in front of the stream bits, as a stat_leaves-style entry (lit_entry's layout), and the distance code
    // behind a length: dbase | deb << 16 and the bits of the code; NO_DCODE / BAD_DCODE = no such code / symbols 30, 31
    constexpr uint32_t NO_DCODE = 0xFFFFFFFEu, BAD_DCODE = 0xFFFFFFFFu;
    [[maybe_unused]] uint32_t XL[15], XD[15];       // DYN: the X words of this lane's literal/length and distance code
    [[maybe_unused]] S5 DS;                         // DYN: its distance symbols, sorted by (code length, value)
    typedef Tab<CAP> T;
#define TROW(j) lds.tab[(j) * 64u + lane]
#define TBYTE(base, i) reinterpret_cast<uint8_t*>(lds.tab)[(((base) + ((i) >> 2)) << 8) | lane4 | ((i) & 3u)]
    // the idx-th literal/length symbol of the sorted list
#define TOK_LSYM(idx) ((uint32_t)TBYTE(T::LS8, (idx)) | (((TROW(T::LBIT + ((idx) >> 5)) >> ((idx) & 31u)) & 1u) << 8))
    auto lit_at = [&](const uint32_t bits) -> uint32_t {
        if constexpr (!DYN) return lds.lit[bits & 511u];
        else {
            uint32_t len, idx; bool valid;
            x_decode<15>(XL, bits, len, idx, valid);
            idx = min(idx, CAP - 1u);
            const uint32_t sym = TOK_LSYM(idx);
            const uint32_t type = sym < 256u ? T_LIT : sym == 256u ? T_EOB : sym <= 285u ? T_LEN : T_BAD;
            uint32_t lbase, leb;
            length_info(sym - 257u, lbase, leb);
            // (no zero leaf here: this kernel never runs the DYNAMIC=False build, and a DYNAMIC=True build decodes a fixed block
            // through leaves built from the fixed lengths, where symbol 287 is an ordinary 8-bit leaf -- deflate.py:1066-1073)
            const uint32_t nb = !valid ? 0u : len;
            return nb | (sym << 4) | (type << 13) | ((lbase & 0x1FFu) << 16) | ((leb & 7u) << 25);
        }
    };
    auto dst_at = [&](const uint32_t bits, uint32_t& nbits) -> uint32_t {
        if constexpr (!DYN) { nbits = 5u; return lds.dst[bits & 31u]; }
        else {
            uint32_t idx; bool valid;
            x_decode<15>(XD, bits, nbits, idx, valid);
            const uint32_t ds = DS.get(min(idx, 31u));
            uint32_t dbase, deb;
            dist_info(ds, dbase, deb);
            return !valid ? NO_DCODE : ds >= 30u ? BAD_DCODE : (dbase | (deb << 16));
        }
    };
    // ---- fixed blocks, the fast path's GROUP: up to three literals and the match behind them, decoded at output position AT into
    // (LV, LN, RM, DS).  ONE source for both decode steps of a round -- a macro, not a lambda: called through a lambda the same text
    // compiled to 3 VALU + 13 SALU instructions more per round (configs[4] round trip 401 -> 392 GB/s).  SECOND = false is step 1 (the lane
    // has nothing pending; bc >= 33 after the refill, so the literals need no bit count -- the third look-up still has 33 - 9 - 9 = 15
    // valid bits --; a far match is taken and the caller requests its history), SECOND = true is step 2 (behind a pending group: every
    // symbol only while a buffered bit is left behind it, near matches only).  A match is only taken with every check of
    // deflate.py:1576-1585, :1597-1602 passing at OM = AT + the literals; OK says whether one was taken.
#define TOK_FIXED_GROUP(SECOND, AT, LV, LN, RM, DS, OM, OK) do {                                        \
        uint32_t nl_ = 0;                                                                                  \
        _Pragma("unroll") for (uint32_t extra = 0; extra < 3u; extra++) {                                  \
            const uint32_t e2 = lit_at((uint32_t)bb);                                                      \
            if (nl_ == extra && ((e2 >> 13) & 3u) == (uint32_t)T_LIT && (AT) + extra < cap && (!(SECOND) || (e2 & 15u) < bc)) { \
                LV = extra == 0u ? ((e2 >> 4) & 0xFFu) : (LV | (((e2 >> 4) & 0xFFu) << (8u * extra)));     \
                nl_ = extra + 1u;                                                                          \
                bb >>= (e2 & 15u); bc -= (e2 & 15u);                                                       \
            }                                                                                              \
        }                                                                                                  \
        LN = nl_;                                                                                          \
        const uint32_t e = lit_at((uint32_t)bb);                                                           \
        const uint32_t nb = e & 15u, type = (e >> 13) & 3u;                                                \
        const uint32_t leb = (e >> 25) & 7u, lbase = (e >> 16) & 0x1FFu;                                   \
        uint64_t x = bb >> nb;                                                                             \
        const uint32_t tlength = lbase + ((uint32_t)x & ((1u << leb) - 1u));                               \
        x >>= leb;                                                                                         \
        uint32_t dnb;                                                                                      \
        const uint32_t de = dst_at((uint32_t)x, dnb);                                                      \
        const uint32_t deb = (de >> 16) & 15u;                                                             \
        const uint32_t distance = (de & 0xFFFFu) + ((uint32_t)(x >> dnb) & ((1u << deb) - 1u));            \
        const uint32_t mbits = nb + leb + dnb + deb;                                                       \
        OM = (AT) + nl_;                                        /* where the copy will start */            \
        OK = (type == (uint32_t)T_LEN) & (mbits < bc) & (de < NO_DCODE) & (distance <= OM) & (distance <= obsize) & \
             (OM + tlength <= cap) & (!(SECOND) || distance <= NEAR);                                      \
        if (OK) {                                                                                          \
            bb >>= mbits; bc -= mbits;                                                                     \
            RM = tlength; DS = distance;                                                                   \
        }                                                                                                  \
    } while (0)
    // ---- DYN: restart the bit reader at an absolute bit position (the second pass over a block header)
#define TOK_RESYNC(bitp) do {                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* no LDS-DMA may land in the slot after this */ \
        ip = (bitp) >> 3; sbase = ip; qn = 0u;                                                             \
        for (uint32_t k_ = 0; k_ < SLOT_DW; k_++) inq[lane * SLOT_DW + k_] = load32(z, sbase + 4u * k_, zn); \
        myissue = issued - 1000u;                                                                          \
        bb = 0; bc = 0;                                                                                    \
        TOK_REFILL();                                                                                      \
        bb >>= ((bitp) & 7u); bc -= ((bitp) & 7u);                                                         \
    } while (0)
    // ---- DYN: the tables of a block, fixed (hm = 1: deflate.py:1066-1073) or dynamic (hm = 2: BL/READBL/REPEAT deflate.py:1084-1202,
    // the canonical codes :1204-1400).  Returns the status; the acceptance rules are k_inflate_dyn's (zlib's).  A block whose
    // literal/length code has more than CAP symbols is HDLZ_E_DYNAMIC_UNSUPPORTED here: the stream stays flagged for the next pass.
    auto build_tables = [&](const uint32_t hm) -> uint32_t {
        if constexpr (!DYN) return HDLZ_E_DYNAMIC_UNSUPPORTED;
        else {
            for (uint32_t k = 0; k < (CAP + 31u) / 32u; k++) TROW(T::LBIT + k) = 0u;
            uint32_t nlen = 288u, ndist = 32u, p0 = 0u;
            uint32_t XC[7];
            F16 nL, nD;                       // symbols per code length (pass 1), then the next free slot per length (pass 2)
            S5 CS;                            // the symbols of the code-length code, sorted
            nL.clear(); nD.clear(); CS.clear(); DS.clear();
            // pass 1 (place = false) counts the symbols per code length, pass 2 puts every symbol into its slot of the sorted lists
            auto lens_pass = [&](auto place) -> uint32_t {
                const uint32_t total = nlen + ndist;
                uint32_t idx = 0, prev = 0;
                bool has256 = false;
                while (idx < total) {
                    TOK_REFILL();
                    uint32_t len, ci; bool valid;
                    x_decode<7>(XC, (uint32_t)bb, len, ci, valid);
                    if (!valid) return HDLZ_E_BAD_TREE;
                    const uint32_t sym = CS.get(min(ci, 18u));
                    const uint32_t eb = sym < 16u ? 0u : sym == 16u ? 2u : sym == 17u ? 3u : 7u;
                    const uint32_t ev = (uint32_t)(bb >> len) & ((1u << eb) - 1u);
                    const uint32_t rep = sym < 16u ? 1u : sym == 18u ? 11u + ev : 3u + ev;
                    bb >>= (len + eb); bc -= len + eb;
                    if (sym == 16u && idx == 0u) return HDLZ_E_BAD_TREE;
                    if (idx + rep > total) return HDLZ_E_BAD_TREE;
                    const uint32_t val = sym < 16u ? sym : sym == 16u ? prev : 0u;
                    // (a repeat may run from the literal/length lengths into the distance lengths)
                    const uint32_t nl = idx < nlen ? min(rep, nlen - idx) : 0u, nd = rep - nl;
                    if constexpr (!decltype(place)::value) {
                        nL.add(val, nl);
                        nD.add(val, nd);
                        has256 = has256 || (val != 0u && idx <= 256u && idx + rep > 256u);
                    } else if (val != 0u) {
                        if (nl) {
                            const uint32_t pos = nL.get(val);
                            nL.add(val, nl);
                            for (uint32_t k = 0; k < nl; k++) {
                                const uint32_t s_ = idx + k, q = pos + k;
                                TBYTE(T::LS8, q) = (uint8_t)s_;
                                if (s_ >= 256u) atomicOr(&TROW(T::LBIT + (q >> 5)), 1u << (q & 31u));
                            }
                        }
                        if (nd) {
                            const uint32_t pos = nD.get(val);
                            nD.add(val, nd);
                            for (uint32_t k = 0; k < nd; k++) DS.put(pos + k, idx + nl + k - nlen);
                        }
                    }
                    prev = val;
                    idx += rep;
                }
                if constexpr (!decltype(place)::value) { if (!has256) return HDLZ_E_BAD_TREE; }      // no end-of-block code
                return HDLZ_OK;
            };
            if (hm == 1u) {
                if constexpr (CAP < 288u) return HDLZ_E_DYNAMIC_UNSUPPORTED;
                // 24 codes of 7 bits (256..279), 152 of 8 (0..143, 280..287), 112 of 9 (144..255); 32 distance codes of 5 bits
                nL.put(7, 24u); nL.put(8, 152u); nL.put(9, 112u); nD.put(5, 32u);
                for (uint32_t q = 0; q < 288u; q++) {
                    const uint32_t s_ = q < 24u ? 256u + q : q < 168u ? q - 24u : q < 176u ? q + 112u : q - 32u;
                    TBYTE(T::LS8, q) = (uint8_t)s_;
                    if (s_ >= 256u) atomicOr(&TROW(T::LBIT + (q >> 5)), 1u << (q & 31u));
                }
#pragma unroll
                for (uint32_t q = 0; q < 32u; q++) DS.w[q / 12u] |= (uint64_t)q << ((q % 12u) * 5u);
            } else {
                // BL (deflate.py:1090-1114)
                TOK_REFILL();
                nlen = ((uint32_t)bb & 31u) + 257u;
                ndist = ((uint32_t)(bb >> 5) & 31u) + 1u;
                const uint32_t ncode = ((uint32_t)(bb >> 10) & 15u) + 4u;
                bb >>= 14; bc -= 14u;
                if (nlen > 286u || ndist > 30u) return HDLZ_E_BAD_TREE;
                // the code-length code: 3-bit lengths in the order of RFC1951 3.2.7, kept as 3-bit fields by symbol
                constexpr uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint64_t cl = 0;
#pragma unroll
                for (int i = 0; i < 19; i++) {
                    if ((i & 7) == 0) TOK_REFILL();
                    if ((uint32_t)i < ncode) { cl |= (uint64_t)((uint32_t)bb & 7u) << (3 * order[i]); bb >>= 3; bc -= 3u; }
                }
                uint64_t cw = 0;                       // symbols per length, 5-bit fields
#pragma unroll
                for (int s_ = 0; s_ < 19; s_++) cw += 1ull << (5u * ((uint32_t)(cl >> (3 * s_)) & 7u));
                int32_t left; uint32_t cum;
                x_build<7>(XC, [&](int l) { return (uint32_t)(cw >> (5 * l)) & 31u; }, left, cum);
                if (left != 0) return HDLZ_E_BAD_TREE;
                uint64_t ow = 0;                       // first slot per length in the sorted list, then the next free one
                {
                    uint32_t off = 0;
#pragma unroll
                    for (int l = 1; l < 8; l++) { ow |= (uint64_t)off << (5 * l); off += (uint32_t)(cw >> (5 * l)) & 31u; }
                }
#pragma unroll
                for (int s_ = 0; s_ < 19; s_++) {
                    const uint32_t l5 = 5u * ((uint32_t)(cl >> (3 * s_)) & 7u);
                    if (l5) { CS.put((uint32_t)(ow >> l5) & 31u, (uint32_t)s_); ow += 1ull << l5; }
                }
                p0 = TOK_BITPOS();
                const uint32_t st1 = lens_pass(std::false_type{});
                if (st1 != HDLZ_OK) return st1;
            }
            int32_t l1, l2; uint32_t c1, c2;
            x_build<15>(XL, [&](int l) { return nL.at(l); }, l1, c1);
            x_build<15>(XD, [&](int l) { return nD.at(l); }, l2, c2);
            if (hm == 2u) {
                // incomplete sets only with a single code -- or, for the distance code, with none at all (RFC1951 3.2.7)
                if (l1 < 0 || (l1 > 0 && c1 != 1u)) return HDLZ_E_BAD_TREE;
                if (l2 < 0 || (l2 > 0 && c2 > 1u)) return HDLZ_E_BAD_TREE;
                if (c1 > CAP) return HDLZ_E_DYNAMIC_UNSUPPORTED;       // (only CAP_SMALL: at most 286 symbols are coded)
                {                                      // counts -> first slot per length
                    F16 fL, fD;
                    fL.clear(); fD.clear();
                    uint32_t oL = 0, oD = 0;
#pragma unroll
                    for (int l = 1; l < 16; l++) { fL.put(l, oL); fD.put(l, oD); oL += nL.at(l); oD += nD.at(l); }
                    nL = fL; nD = fD;
                }
                TOK_RESYNC(p0);
                lens_pass(std::true_type{});
                if ((int32_t)(TOK_BITPOS() >> 3) > isize - 3) return HDLZ_E_NO_EOF;
            }
            return HDLZ_OK;
        }
    };

    if (active) { TOK_REQUEST(issued + 1u); }
    if (ballot64(active && sbase + 16u <= zn) != 0ull) issued += 1u;
    TOK_TIME_DECL();

    for (;;) {
        TOK_TIME(5);
        TOK_MARK("move");
        // ------------------------------------------------------------ 0. move the bytes of the tokens decoded in the PREVIOUS round: up to
        // four per lane and iteration.  (Moving first, decoding after: the far history a token needs was requested when it was
        // decoded, a whole round ago -- waiting for it right after the decode left 67 % of the wave cycles in s_waitcnt.)
        // At most MOVES iterations per round: a long copy goes on in the next rounds while the other lanes decode on -- waiting
        // for the longest copy of the wave in every round left the short tokens idle (measured: 91 VALU per byte instead of ~25)
        // Far copies (distance > NEAR) take no part in this loop: they move in ONE step of up to 16 bytes behind it (below), so the loop
        // holds no load at all -- round 3's form (8 far bytes per iteration, the next chunk requested inside the loop) made hipcc wait
        // vmcnt(0) in EVERY iteration, near branch included: for the flush stores and the input DMA too (profiles/r04_tok_round_timing.txt)
        // (the lane predicates of this loop are single integer compares: a compound bool goes through an SGPR lane mask, and a
        // ballot of such a mask costs hipcc a v_cndmask + v_cmp_ne to put it back under exec; a lane without a stream never decodes)
#define TOK_PROMOTE() do { if constexpr (!DYN && STEP2_MIN != 0u) { if (s2_live) {                     \
            const bool pr_ = (litn | rem) == 0u;          /* the first group is done: the second one moves up (it may be empty) */ \
            litn = pr_ ? litn2 : litn; litv = pr_ ? litv2 : litv; rem = pr_ ? rem2 : rem; dist = pr_ ? dist2 : dist;        \
            litn2 = pr_ ? 0u : litn2; rem2 = pr_ ? 0u : rem2;                                              \
            s2_live = ballot64((litn2 | rem2) != 0u) != 0ull;                                              \
        } } } while (0)
        for (uint32_t mvi = 0; mvi < MOVES; mvi++) {
            if (mvi != 0u) TOK_PROMOTE();                  // (at the top of a round the hand-over behind the far step has been made)
            const bool mv = (litn | (dist <= NEAR ? rem : 0u)) != 0u;
            if (ballot64(mv) == 0ull) break;
            if (mv) {
                uint32_t v = litv, k = litn;                       // 1..3 literals, or
                if (litn == 0u) {                                  // near history: the ring (unflushed bytes live only here), 4 bytes
                    const uint32_t src = o - dist;
                    k = min(rem, 4u);
                    const uint32_t a0 = ((src & (RINGB - 4u)) << 6) | lane4, a1 = (((src + 4u) & (RINGB - 4u)) << 6) | lane4;
                    const uint32_t w0 = *reinterpret_cast<const uint32_t*>(ring8 + a0), w1 = *reinterpret_cast<const uint32_t*>(ring8 + a1);
                    v = __builtin_amdgcn_alignbyte(w1, w0, src);
                    // an overlapping copy repeats a pattern of `dist` bytes: only its first period is there yet
                    if (dist < 4u) v = __builtin_amdgcn_perm(v, v, dist == 1u ? 0x00000000u : dist == 2u ? 0x01000100u : 0x00020100u);
                    rem -= k;
                }
                litn = 0;
                // four bytes at position o, unmasked; `pend` holds the valid low bytes of the dword at o
                const uint32_t s8 = (o & 3u) * 8u;
                const uint64_t ta = (uint64_t)v << s8;
                const uint32_t d0 = (uint32_t)ta | pend, d1 = (uint32_t)(ta >> 32);
                const uint32_t b0 = ((o & (RINGB - 4u)) << 6) | lane4, b1 = (((o + 4u) & (RINGB - 4u)) << 6) | lane4;
                *reinterpret_cast<uint32_t*>(ring8 + b0) = d0;
                *reinterpret_cast<uint32_t*>(ring8 + b1) = d1;
                const uint32_t q1 = (o & 3u) + k;                  // 1..7
                o += k;
                const uint32_t nd = q1 >= 4u ? d1 : d0;
                pend = nd & ((1u << ((q1 & 3u) * 8u)) - 1u);
            }
            TOK_TIME(6);
        }
        // ------------------------------------------------------------ 0a. the far step: up to 16 bytes of far history -- the stream's own output,
        // flushed long ago (URGENT) -- requested when the token was decoded (or by this step a round ago) and taken once the token's
        // literals are out.  The only place that waits for a history load, and only in rounds in which a lane takes one.
        {
            const bool fc = (litn == 0u ? (dist > NEAR ? rem : 0u) : 0u) != 0u;
            if (ballot64(fc) != 0ull) {
                if (fc) {
                    const uint32_t k = min(rem, 16u);
                    const uint32_t s8 = (o & 3u) * 8u;
                    const uint64_t t0 = (uint64_t)far4.x << s8, t1 = (uint64_t)far4.y << s8, t2 = (uint64_t)far4.z << s8, t3 = (uint64_t)far4.w << s8;
                    const uint32_t d0 = (uint32_t)t0 | pend, d1 = (uint32_t)(t0 >> 32) | (uint32_t)t1, d2 = (uint32_t)(t1 >> 32) | (uint32_t)t2,
                                   d3 = (uint32_t)(t2 >> 32) | (uint32_t)t3, d4 = (uint32_t)(t3 >> 32);
                    const uint32_t q1 = (o & 3u) + k;              // 1..19
                    // three dwords unmasked as in the loop; the fourth and fifth only when valid bytes reach them (what is written
                    // ahead of the new end stays below 12 bytes: NEAR + 3 + 12 < RINGB holds as before)
                    *reinterpret_cast<uint32_t*>(ring8 + (((o & (RINGB - 4u)) << 6) | lane4)) = d0;
                    *reinterpret_cast<uint32_t*>(ring8 + ((((o + 4u) & (RINGB - 4u)) << 6) | lane4)) = d1;
                    *reinterpret_cast<uint32_t*>(ring8 + ((((o + 8u) & (RINGB - 4u)) << 6) | lane4)) = d2;
                    if (q1 > 12u) *reinterpret_cast<uint32_t*>(ring8 + ((((o + 12u) & (RINGB - 4u)) << 6) | lane4)) = d3;
                    if (q1 > 16u) *reinterpret_cast<uint32_t*>(ring8 + ((((o + 16u) & (RINGB - 4u)) << 6) | lane4)) = d4;
                    o += k;
                    rem -= k;
                    const uint32_t qd = q1 >> 2;
                    const uint32_t nd = qd == 0u ? d0 : qd == 1u ? d1 : qd == 2u ? d2 : qd == 3u ? d3 : d4;
                    pend = nd & ((1u << ((q1 & 3u) * 8u)) - 1u);
                    if (rem != 0u) far4 = *reinterpret_cast<const u32x4_unaligned*>(out + (o - dist));
                }
            }
        }
        TOK_PROMOTE();                                   // (behind the far step, which may have finished the first group: from here on a
                                                         //  pending second group implies a pending first one -- the decode relies on it)
        TOK_FLUSH();
        TOK_TIME(0);
        TOK_MARK("refill");
        // ------------------------------------------------------------ 0b. input refill (a lane waits only when it opens a new slot)
        {
            const bool need = active && bc <= 32u;
            if (ballot64(need) != 0ull) {
                const bool fresh = need && qn == 0u;
                if (ballot64(fresh) != 0ull) {
                    // `issued` counts the wave's LDS-DMA instructions and flush stores, `myissue` is its value right after this lane's
                    // request: at least `after` VMEM instructions are younger than the request (far loads are not counted: the wait
                    // is then longer than it has to be, never shorter), and vmcnt retires in order
                    const uint32_t after = issued - myissue;
                    if (ballot64(fresh && after < 8u) == 0ull) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (ballot64(fresh && after < 4u) == 0ull) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (ballot64(fresh && after < 2u) == 0ull) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if (ballot64(fresh && after < 1u) == 0ull) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    bool dma = false;
                    if (fresh) {
                        TOK_DRAIN();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slot is read before the next request may overwrite it
                        dma = sbase + 16u <= zn;
                        TOK_REQUEST(issued + 1u);
                    }
                    if (ballot64(dma) != 0ull) issued += 1u;
                }
                if (need) TOK_POP();
            }
        }
        TOK_TIME(1);
        TOK_MARK("decode");
        // ------------------------------------------------------------ 1a. fast path: up to three literals and then a match, inside a
        // fixed block.  (One token per round made the literal-heavy streams the lanes the whole wave waits for: 362 -> 390 GB/s with
        // literal triples, more with the match behind them.)
        bool slow = false;
        if (active && srem == 0u && (rem | litn) == 0u) {     // (a lane with a pending group takes its next one in step 2, if at all)
            slow = need_header;
            // input guard: after the refill bc >= 33, and a token is only taken when at least one buffered bit is left behind it, so
            // its bit position lies below byte ip; with ip + 3 <= zn both reference checks (deflate.py:1535-1539 after the symbol,
            // :1600 before the copy) are guaranteed to pass -- anything closer to the end goes the slow way
            const bool in_ok = ip + 3u <= zn;
            if (!slow && in_ok) {
              if constexpr (DYN) {
                // four symbols at once: the LENGTH of a code needs no table (x_decode), so the bit offsets of the next three
                // symbols are known before a look-up returns -- the four look-ups are in flight together (one LDS round trip
                // instead of four: this kernel runs one wave per SIMD, nothing else hides them).  Symbols behind a non-literal
                // are simply not used.
                uint32_t ln[4], ix[4], of[4], sy[4];
                bool vd[4];
                uint32_t off = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    of[k] = off;
                    x_decode<15>(XL, (uint32_t)(bb >> off), ln[k], ix[k], vd[k]);
                    off += ln[k];
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i_ = min(ix[k], CAP - 1u);
                    sy[k] = TOK_LSYM(i_);
                }
                bool lt[3];                                        // a literal is taken while a buffered bit is left behind it
#pragma unroll
                for (int k = 0; k < 3; k++) lt[k] = vd[k] && sy[k] < 256u && of[k] + ln[k] < bc && o + (uint32_t)k < cap;
                const uint32_t nl = !lt[0] ? 0u : !lt[1] ? 1u : !lt[2] ? 2u : 3u;
                litv = (sy[0] & 255u) | ((sy[1] & 255u) << 8) | ((sy[2] & 255u) << 16);     // (bytes behind the nl-th are never used)
                litn = nl;
                // the token behind the literals
                const uint32_t tsy = nl == 0u ? sy[0] : nl == 1u ? sy[1] : nl == 2u ? sy[2] : sy[3];
                const uint32_t tof = nl == 0u ? 0u : nl == 1u ? of[1] : nl == 2u ? of[2] : of[3];
                const uint32_t tln = nl == 0u ? ln[0] : nl == 1u ? ln[1] : nl == 2u ? ln[2] : ln[3];
                const bool tvd = nl == 0u ? vd[0] : nl == 1u ? vd[1] : nl == 2u ? vd[2] : vd[3];
                const uint32_t token = tsy - 257u;
                uint32_t lbase, leb;
                length_info(token, lbase, leb);
                leb &= 7u;
                uint64_t x = bb >> (tof + tln);
                const uint32_t tlength = lbase + ((uint32_t)x & ((1u << leb) - 1u));
                x >>= leb;
                uint32_t dnb, dix; bool dvd;
                x_decode<15>(XD, (uint32_t)x, dnb, dix, dvd);
                const uint32_t ds = DS.get(min(dix, 31u));
                uint32_t dbase, deb;
                dist_info(ds, dbase, deb);
                deb &= 15u;
                const uint32_t distance = dbase + ((uint32_t)(x >> dnb) & ((1u << deb) - 1u));
                const uint32_t mbits = tof + tln + leb + dnb + deb;
                const uint32_t om = o + nl;                 // where the copy will start
                const bool len_ok = tvd & (token < 29u) & dvd & (ds < 30u) & (mbits < bc) & (distance <= om) & (distance <= obsize) &
                                    (om + tlength <= cap);
                const uint32_t take = len_ok ? mbits : tof;
                bb >>= take; bc -= take;
                if (len_ok) {
                    rem = tlength; dist = distance;
                    if (distance > NEAR) {
                        // (one 16-byte request: the TA is the busiest unit of this kernel; src + 16 <= om - 97 < flushed)
                        far4 = *reinterpret_cast<const u32x4_unaligned*>(out + (om - distance));
                    }
                } else if (nl == 0u) {
                    slow = true;                                    // EOB, invalid data, any failing check
                }
              } else {
                // (after the refill bc >= 33: the third literal look-up still has 33 - 9 - 9 = 15 valid bits, no bit count to check)
                uint32_t om; bool len_ok;
                TOK_FIXED_GROUP(false, o, litv, litn, rem, dist, om, len_ok);
                if (len_ok) {
                    if (dist > NEAR) {
                        // (one 16-byte request: the TA is the busiest unit of this kernel; src + 16 <= om - 97 < flushed)
                        far4 = *reinterpret_cast<const u32x4_unaligned*>(out + (om - dist));
                    }
                } else if (litn == 0u) {
                    slow = true;                                    // EOB, invalid data, any failing check
                }
              }
            } else {
                slow = true;                                        // header, end of the input
            }
        }
        // ------------------------------------------------------------ 1a'. a second group for the lanes whose first one is pending (fixed blocks):
        // up to three literals and a NEAR match (a far one stays in the stream: its history load belongs to step 1), only what the fast path
        // can take -- anything else (end of block, a header, the end of the input, a failing check) is left for step 1 of a later round,
        // which sees it with nothing pending, as before.  The bit buffer may be short here (step 1 has eaten from it): one more dword is
        // popped when the queue has one, and every symbol is taken only while a buffered bit is left behind it.  The group starts at
        // ob = o + (bytes still pending), which is where its distance and capacity checks apply (deflate.py:1576-1585 at that position).
        // Bounds: the move loop still moves at most MOVES * 4 bytes and the far step only the FIRST group's bytes, so a round advances o
        // by at most 19 as before (flush / far-load safety, see URGENT).
        if constexpr (!DYN && STEP2_MIN != 0u) {
          if (s2_hold != 0u) s2_hold -= 1u;
          else {
            const bool s2 = active && !slow && !need_header && srem == 0u && (litn | rem) != 0u && (litn2 | rem2) == 0u;
            const uint32_t ns2 = (uint32_t)__popcll(ballot64(s2));
            if (ns2 >= STEP2_MIN) {
                if (s2 && bc <= 32u && qn != 0u) TOK_POP();
                if (s2 && ip + 3u <= zn) { uint32_t om2; bool ok2; TOK_FIXED_GROUP(true, o + litn + rem, litv2, litn2, rem2, dist2, om2, ok2); }
                // worth it when most of those lanes got a match or three literals; else pause, twice as long every time
                const uint32_t ngood = (uint32_t)__popcll(ballot64(s2 && (rem2 != 0u || litn2 == 3u)));
                s2_live = s2_live || ballot64((litn2 | rem2) != 0u) != 0ull;
                if (2u * ngood < ns2) { s2_hold = s2_back; s2_back = min(2u * s2_back, 1024u); }
                else s2_back = 32u;
            }
          }
        }
        TOK_TIME(2);
        TOK_MARK("slow");
        // ------------------------------------------------------------ 1b. slow path (wave-uniform branch, rare)
        if (ballot64(slow || (active && srem != 0u)) != 0ull) {
            while (slow && active && rem == 0u && srem == 0u && litn == 0u) {
                TOK_REFILL();
                if (need_header) {
                    // HEADER (deflate.py:677-732)
                    final_ = ((uint32_t)bb & 1u) | oneblock;
                    const uint32_t hm = assume_fixed ? 1u : ((uint32_t)(bb >> 1) & 3u);
                    if (hm == 3u) { TOK_FAIL(HDLZ_E_BAD_BTYPE); break; }
                    if (!DYN && hm == 2u) { TOK_FAIL(HDLZ_E_DYNAMIC_UNSUPPORTED); break; }
                    need_header = false;
                    if (hm == 0u) {
                        // stored (deflate.py:709-717): LEN sits `skip` bits after the header start
                        const uint32_t dio = TOK_BITPOS() & 7u;
                        uint32_t skip = 8u - dio;
                        if (skip <= 2u) skip = 16u - dio;
                        const uint32_t length = (uint32_t)(bb >> skip) & 0xFFFFu & len_mask;
                        bb >>= (skip + 16u); bc -= (skip + 16u);          // now at NLEN = the reference's di
                        TOK_REFILL();
                        bb >>= 16; bc -= 16u;                             // NLEN unchecked (D2); data follows
                        srem = length;
                        if (length == 0u) {
                            // COPY with nothing to copy (deflate.py:1617-1626)
                            if ((int32_t)(TOK_BITPOS() >> 3) >= isize) { TOK_FAIL(HDLZ_E_NO_EOF); break; }
                            if (final_) { out_len = o; active = false; break; }
                            need_header = true;
                        }
                    } else {
                        bb >>= 3; bc -= 3u;
                        if constexpr (DYN) {
                            const uint32_t tst = build_tables(hm);
                            if (tst != HDLZ_OK) { TOK_FAIL(tst); break; }
                        }
                    }
                    continue;
                }
                // NEXT (deflate.py:1409-1445; DYN: D_NEXT :1447-1517 for the distance)
                const uint32_t e = lit_at((uint32_t)bb);
                const uint32_t nb = e & 15u, code = (e >> 4) & 0x1FFu;
                if (nb < 1u) { TOK_FAIL(HDLZ_E_BAD_SYMBOL); break; }
                bb >>= nb; bc -= nb;
                // INFLATE (deflate.py:1519-1591)
                if ((int32_t)(TOK_BITPOS() >> 3) > isize - 3) { TOK_FAIL(HDLZ_E_NO_EOF); break; }   // :1535-1539
                if (code == 256u) {
                    if (final_) { out_len = o; active = false; break; }   // D6
                    need_header = true;
                    continue;
                }
                if (code < 256u) {
                    if (o >= cap) { TOK_FAIL(HDLZ_E_OUT_CAPACITY); break; }
                    litv = code; litn = 1;
                    break;
                }
                const uint32_t token = code - 257u;
                if (token >= 29u) { TOK_FAIL(HDLZ_E_BAD_SYMBOL); break; }
                uint32_t lbase, leb, dnb;
                length_info(token, lbase, leb);
                const uint32_t tlength = lbase + ((uint32_t)bb & ((1u << leb) - 1u));
                bb >>= leb; bc -= leb;
                if constexpr (DYN) TOK_REFILL();                   // 15 + 5 bits are gone, 15 + 13 may follow
                const uint32_t de = dst_at((uint32_t)bb, dnb);
                if (de == NO_DCODE) { TOK_FAIL(HDLZ_E_BAD_SYMBOL); break; }
                if (de == BAD_DCODE) { TOK_FAIL(HDLZ_E_BAD_DISTANCE); break; }
                bb >>= dnb;
                const uint32_t deb = (de >> 16) & 15u;
                const uint32_t distance = (de & 0xFFFFu) + ((uint32_t)bb & ((1u << deb) - 1u));
                bb >>= deb;
                bc -= dnb + deb;
                if (distance > o || distance > obsize) { TOK_FAIL(HDLZ_E_BAD_DISTANCE); break; }        // D8
                if ((int32_t)(TOK_BITPOS() >> 3) >= isize - 2) { TOK_FAIL(HDLZ_E_NO_EOF); break; }      // COPY hold, :1600
                if ((uint64_t)o + tlength > cap) { TOK_FAIL(HDLZ_E_OUT_CAPACITY); break; }
                rem = tlength;
                dist = distance;
                if (distance > NEAR) {
                    far4 = *reinterpret_cast<const u32x4_unaligned*>(out + (o - distance));
                }
            }
            // stored COPY (deflate.py:1603-1616): one byte per round (rare: level-0 streams, incompressible blocks)
            if (active && srem != 0u && litn == 0u && rem == 0u) {
                TOK_REFILL();
                if ((int32_t)(TOK_BITPOS() >> 3) >= isize) { TOK_FAIL(HDLZ_E_NO_EOF); }
                else if (o >= cap) { TOK_FAIL(HDLZ_E_OUT_CAPACITY); }
                else {
                    litv = (uint32_t)bb & 0xFFu; litn = 1;
                    bb >>= 8; bc -= 8u;
                    srem--;
                    if (srem == 0u) {                              // the block ends with this byte (deflate.py:1617-1626)
                        if ((int32_t)(TOK_BITPOS() >> 3) >= isize) { TOK_FAIL(HDLZ_E_NO_EOF); litn = 0; }
                        else if (final_) { out_len = o + 1u; active = false; }      // (the byte is still emitted below)
                        else need_header = true;
                    }
                }
            }
        }
        TOK_TIME(3);
        TOK_TIME_ROUND();
        TOK_MARK("loopend");
        if (ballot64(active || (rem | litn | rem2 | litn2) != 0u) == 0ull) break;
    }
    TOK_MARK("epilogue");
    // the lines completed since the last batch
    if (exists && (o - flushed) >= CHUNK) {
        const uint32_t* rp = &lds.ring[wave][((flushed & (RINGB - 1u)) >> 2) * 64u + lane];
        uint32_t* d32 = reinterpret_cast<uint32_t*>(out + flushed);
#pragma unroll
        for (uint32_t q = 0; q < CHUNK / 4u; q++) d32[q] = rp[q * 64u];
        flushed += CHUNK;
    }
#undef TROW
#undef TOK_LSYM
#undef TBYTE
#undef TOK_RESYNC
#undef TOK_FAIL
#undef TOK_BITPOS
#undef TOK_REFILL
#undef TOK_REQUEST
#undef TOK_FLUSH
#undef TOK_FIXED_GROUP

    // no LDS-DMA load may still be in flight when this wave's LDS is handed to another workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- tail: the bytes behind the last flushed line are still only in the ring
    if (exists && status == HDLZ_OK) {
        for (uint32_t p = flushed; p < out_len; p++) out[p] = ring8[ring_addr(p, lane4)];
    }
    TOK_STORE_RESULT();
}

// ---- length-binned lane assignment (round 4).  The 64 streams of a wave run in lockstep until the LAST of them is done, and a
// round costs the same whatever its lanes have to do: a wave of mixed streams pays the rounds of its longest stream with the
// instruction mix of its most demanding one (BASELINE configs[4] read back, blocks of four families side by side: 85.7 M cycles per
// wave against 57 M / 40 M for waves of one family, profiles/r04_tok_round_timing.txt).  With a ragged archive the compressed
// lengths are known before the launch, so the streams are handed to the lanes in the order of their length class -- quarter octaves,
// longest first (the long streams start first, the short ones fill the tail; the order is worth 13 % on the dynamic-tree entry) --: a
// counting sort in two launches (counts + first slots by the last block to arrive, then the scatter), the permutation in
// stream-ordered scratch.  The order inside a class is whatever the atomics give: streams are independent, results do not depend on it.
constexpr uint32_t NBIN = 128;
constexpr uint32_t BIN_WORDS = NBIN + 2u;   // bins[0 .. NBIN): counts, then cursors; bins[NBIN]: the list length; bins[NBIN + 1]: arrival ticket
__device__ __forceinline__ uint32_t len_bin(uint64_t len64) {
    const uint32_t l = (uint32_t)(len64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : len64) | 4u;
    const uint32_t msb = 31u - (uint32_t)__builtin_clz(l);
    return (NBIN - 1u) - (4u * msb + ((l >> (msb - 2u)) & 3u));        // bin 0 = the longest streams
}
// the block's counts go to the global ones; the LAST of `nact` blocks to arrive turns them into first slots (exclusive scan, in place) and
// stores the list length -- what was a launch of its own (k_bin_scan).  Called by all 256 threads of every participating block.
__device__ __forceinline__ void bin_flush_and_finish(const uint32_t* lh, uint32_t* __restrict__ bins, uint32_t nact, const uint32_t* nlist_p,
                                                     uint32_t nlist_max) {
    __shared__ uint32_t last;
    // (no __threadfence: an agent-scope fence writes the XCD's L2 back, 200 us over 4096 blocks.  The counts are only ever touched by
    //  agent-scope atomics, which are performed at the coherence point; what is needed is their ORDER -- this block's adds performed (their
    //  old values returned) before its ticket -- and the last block reading them with atomic loads)
    if (threadIdx.x < NBIN && lh[threadIdx.x]) {
        const uint32_t old = atomicAdd(&bins[threadIdx.x], lh[threadIdx.x]);     // RETURNING: the value is back = the add has been performed
        asm volatile("" :: "v"(old) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0u) last = atomicAdd(&bins[NBIN + 1u], 1u) == nact - 1u ? 1u : 0u;
    __syncthreads();
    if (last == 0u || threadIdx.x >= 64u) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t nl = nlist_p ? __hip_atomic_load(nlist_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : nlist_max;
    const uint32_t expect = min(nl, nlist_max);
    // the ordering above is argued, not fenced -- so it is CHECKED: the counts must add up to the number of listed streams.
    // A count that is still on its way is read again (it is an atomic at the coherence point: it arrives); if the sum never matches, the
    // list length word says so and the decode kernel does not use the ordered list (a wrong permutation would decode streams twice
    // and leave rows unwritten, silently).
    uint32_t c0 = 0, c1 = 0, v = 0, total = 0;
    for (uint32_t tries = 0; tries < 1024u; tries++) {
        c0 = __hip_atomic_load(&bins[2u * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c1 = __hip_atomic_load(&bins[2u * lane + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = c0 + c1;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) {
            const uint32_t o = __shfl_up(v, ofs, 64);
            if (lane >= (uint32_t)ofs) v += o;
        }
        total = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
        if (total == expect) break;
        __builtin_amdgcn_s_sleep(8);
    }
    const uint32_t excl = v - (c0 + c1);
    bins[2u * lane] = excl; bins[2u * lane + 1u] = excl + c0;
    if (lane == 0u) bins[NBIN] = total == expect ? expect : ORDER_BAD;
}

// the streams pass 1 flagged HDLZ_E_DYNAMIC_UNSUPPORTED, as a dense list: the lanes of k_inflate_tok<true> are then all busy whatever
// the share of such streams is (list[0 .. *n) in no particular order: the streams are independent).  With `in_off` / `bins` the kernel
// also counts the length classes of the listed streams (and its last block makes the first slots of them): the counting sort of
// the dynamic-tree pass then needs only the scatter.
__global__ __launch_bounds__(256) void k_collect_dyn(const uint32_t* __restrict__ status, uint64_t nstreams, uint32_t* __restrict__ list,
                                                      uint32_t* __restrict__ n, const uint64_t* __restrict__ in_off = nullptr,
                                                      uint32_t* __restrict__ bins = nullptr, uint32_t code = HDLZ_E_DYNAMIC_UNSUPPORTED) {
    // ONE atomic per workgroup (round 4: one per wave was 4096 atomics on one counter for 262144 streams, 49 us of a 1.4 ms job)
    __shared__ uint32_t wcnt[4], wbase, lh[NBIN];
    const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const bool mine = gid < nstreams && status[gid] == code;
    const uint64_t m = ballot64(mine);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) wcnt[wave] = (uint32_t)__popcll(m);
    if (bins && threadIdx.x < NBIN) lh[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t c0 = wcnt[0], c1 = wcnt[1], c2 = wcnt[2], c3 = wcnt[3];
    if (c0 + c1 + c2 + c3 != 0u) {
        if (threadIdx.x == 0u) wbase = atomicAdd(n, c0 + c1 + c2 + c3);
        if (bins && mine) atomicAdd(&lh[len_bin(in_off[gid + 1] - in_off[gid])], 1u);
        __syncthreads();
        const uint32_t base = wbase + (wave > 0u ? c0 : 0u) + (wave > 1u ? c1 : 0u) + (wave > 2u ? c2 : 0u);
        if (mine) list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint32_t)gid;
    }
    if (bins) bin_flush_and_finish(lh, bins, gridDim.x, n, (uint32_t)min(nstreams, (uint64_t)0xFFFFFFFFull));
}

// (src / src_n: the streams to order are src[0 .. *src_n) -- the list of the dynamic-tree pass -- instead of 0 .. n - 1)
__global__ __launch_bounds__(256) void k_bin_hist(const uint64_t* __restrict__ in_off, uint32_t n, uint32_t* __restrict__ bins) {
    __shared__ uint32_t lh[NBIN];
    if (threadIdx.x < NBIN) lh[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) atomicAdd(&lh[len_bin(in_off[i + 1] - in_off[i])], 1u);
    __syncthreads();
    bin_flush_and_finish(lh, bins, gridDim.x, nullptr, n);
}
__global__ __launch_bounds__(256) void k_bin_scatter(const uint64_t* __restrict__ in_off, uint32_t n, uint32_t* __restrict__ cursor,
                                                      uint32_t* __restrict__ list, const uint32_t* __restrict__ src,
                                                      const uint32_t* __restrict__ src_n) {
    __shared__ uint32_t lh[NBIN], lbase[NBIN];
    if (cursor[NBIN] == ORDER_BAD) return;                  // (the counts did not add up: nobody reads the ordered list)
    if (src_n) n = min(n, *src_n);
    if (blockIdx.x * 256u >= n) return;
    if (threadIdx.x < NBIN) lh[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t bin = 0, rank = 0, sid = 0;
    if (i < n) { sid = src ? src[i] : i; bin = len_bin(in_off[sid + 1] - in_off[sid]); rank = atomicAdd(&lh[bin], 1u); }
    __syncthreads();
    if (threadIdx.x < NBIN && lh[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], lh[threadIdx.x]);
    __syncthreads();
    if (i < n) list[lbase[bin] + rank] = sid;
}

// pass 1's order: bins[0 .. BIN_WORDS) (zeroed here) and `list` (n words) in caller-provided scratch; the decode kernel takes `list`
// and the count at bins + NBIN
static hipError_t bin_streams(const uint64_t* in_off, uint32_t n, uint32_t* bins, uint32_t* list, hipStream_t stream) {
    hipError_t e = zero_words(bins, BIN_WORDS, stream);
    if (e != hipSuccess) return e;
    const dim3 bgrid((n + 255u) / 256u), bblock(256);
    hipLaunchKernelGGL(k_bin_hist, bgrid, bblock, 0, stream, in_off, n, bins);
    hipLaunchKernelGGL(k_bin_scatter, bgrid, bblock, 0, stream, in_off, n, bins, list, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    return hipGetLastError();
}

}  // namespace tok

size_t inflate_tok_work_bytes(uint64_t nstreams, bool ragged) {
    if (nstreams == 0 || nstreams > 0xFFFFFFFFull) return 0;
    const size_t pass1 = ragged && nstreams > HDLZ_INFLATE_BIN_MIN ? sizeof(uint32_t) * ((size_t)nstreams + tok::BIN_WORDS) : 0u;
    const size_t pass2 = sizeof(uint32_t) * ((size_t)2u + tok::BIN_WORDS + nstreams + (ragged && nstreams > HDLZ_INFLATE_BIN_MIN ? (size_t)nstreams : 0u));
    return ((pass1 > pass2 ? pass1 : pass2) + 255u) & ~(size_t)255u;
}

hipError_t launch_inflate_tok(const InflateArgs& a, hipStream_t stream, const Work& w) {
    if (a.nstreams == 0) return hipSuccess;
    typedef tok::Lds<false, tok::CAP_FULL> L;
    const uint64_t per_wg = 64u * L::WAVES;
    const dim3 grid((unsigned)((a.nstreams + per_wg - 1u) / per_wg)), block(64 * L::WAVES);
    // ragged input of more than one wave: the lanes take the streams in the order of their length class (see k_bin_*)
    if (a.in_off && a.nstreams > HDLZ_INFLATE_BIN_MIN && a.nstreams <= 0xFFFFFFFFull) {
        uint32_t* ws = nullptr;                  // ws[0 .. BIN_WORDS): the bins (see tok::BIN_WORDS); the list behind them
        const uint32_t n = (uint32_t)a.nstreams;
        hipError_t e = w.get(sizeof(uint32_t) * ((size_t)n + tok::BIN_WORDS), stream, reinterpret_cast<uint8_t**>(&ws));
        if (e == hipSuccess) {
            e = tok::bin_streams(a.in_off, n, ws, ws + tok::BIN_WORDS, stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL((tok::k_inflate_tok<false, tok::CAP_FULL>), grid, block, 0, stream, a,
                                   (const uint32_t*)(ws + tok::BIN_WORDS), (const uint32_t*)(ws + tok::NBIN), 0u,
                                   (const uint32_t*)(ws + tok::NBIN), (const uint32_t*)nullptr);
                e = hipGetLastError();
            }
            const hipError_t e2 = w.put(reinterpret_cast<uint8_t*>(ws), stream);
            return e != hipSuccess ? e : e2;
        }
        (void)hipGetLastError();                 // no scratch: stream order
    }
    hipLaunchKernelGGL((tok::k_inflate_tok<false, tok::CAP_FULL>), grid, block, 0, stream, a, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, 0u, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    return hipGetLastError();
}

// second pass of the lane-per-stream mapping: the streams with dynamic-tree blocks, one lane each again.  `all`: every stream
// is decoded here (no first pass).  Two stages: the CAP_SMALL build of the kernel (eight waves per CU) takes every such stream;
// the ones it leaves flagged -- a literal/length code of more than CAP_SMALL symbols, or a fixed block between dynamic ones --
// are collected again and go through the CAP_FULL build.  Either stage hands its streams to k_inflate_dyn (one wave each) when
// they are fewer than `lane_min`; the counts stay on the device.  The lists live in stream-ordered scratch (4 bytes per stream).
hipError_t launch_inflate_tok_dyn(const InflateArgs& a, hipStream_t stream, bool all, const Work& w) {
    if (a.nstreams == 0 || (a.flags & HDLZ_INFLATE_ASSUME_FIXED)) return hipSuccess;
    const dim3 grid((unsigned)((a.nstreams + 63u) / 64u)), block(64);
    const dim3 cgrid((unsigned)((a.nstreams + 255u) / 256u)), cblock(256);
    // ws[0], ws[1]: the two counts; ws[2 .. 2 + BIN_WORDS): the bins of stage 1's counting sort (zeroed with the counts in one launch);
    // the collected list behind them (both stages: the launches are ordered; 4 bytes per stream), the length-ordered list of stage 1
    // behind that (4 more per stream when binned)
    uint32_t* ws = nullptr;
    // (the explicit lane hint keeps every such stream in the lane kernels)
    const uint32_t lane_min = (a.flags & HDLZ_INFLATE_LANE_PER_STREAM) ? 0u : HDLZ_INFLATE_DYN_LANE_MIN;
    // ragged input: stage 1 takes its streams in the order of their length class, like pass 1: k_collect_dyn counts the classes of the
    // streams it lists and its last block makes the first slots, so the sort adds ONE launch (the scatter)
    const bool binned = a.in_off != nullptr && a.nstreams > HDLZ_INFLATE_BIN_MIN && a.nstreams <= 0xFFFFFFFFull && !all;   // (32-bit stream ids in the lists)
    const uint32_t head = 2u + tok::BIN_WORDS;
    const size_t nws = (size_t)head + a.nstreams + (binned ? (size_t)a.nstreams : 0u);
    hipError_t e = w.get(sizeof(uint32_t) * nws, stream, reinterpret_cast<uint8_t**>(&ws));
    if (e != hipSuccess) {                      // no scratch: the wave-per-stream pass needs none and finishes the job
        (void)hipGetLastError();
        return launch_inflate_dyn(a, stream, all);
    }
    uint32_t* const bins = ws + 2;
    uint32_t* const list0 = ws + head;
    e = zero_words(ws, head, stream);
    if (e == hipSuccess) {
        if (all) {
            hipLaunchKernelGGL((tok::k_inflate_tok<true, tok::CAP_SMALL>), grid, block, 0, stream, a, (const uint32_t*)nullptr,
                               (const uint32_t*)nullptr, 0u, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
            e = hipGetLastError();
        } else {
            const uint32_t* list1 = list0;
            if (binned) {
                hipLaunchKernelGGL(tok::k_collect_dyn, cgrid, cblock, 0, stream, a.status, a.nstreams, list0, ws, a.in_off, bins,
                                   (uint32_t)HDLZ_E_DYNAMIC_UNSUPPORTED);
                uint32_t* const ordered = list0 + a.nstreams;
                hipLaunchKernelGGL(tok::k_bin_scatter, cgrid, cblock, 0, stream, a.in_off, (uint32_t)a.nstreams, bins, ordered,
                                   (const uint32_t*)list0, (const uint32_t*)ws);
                list1 = ordered;
            } else {
                hipLaunchKernelGGL(tok::k_collect_dyn, cgrid, cblock, 0, stream, a.status, a.nstreams, list0, ws, (const uint64_t*)nullptr,
                                   (uint32_t*)nullptr, (uint32_t)HDLZ_E_DYNAMIC_UNSUPPORTED);
            }
            e = hipGetLastError();
            if (e == hipSuccess) {
                hipLaunchKernelGGL((tok::k_inflate_tok<true, tok::CAP_SMALL>), grid, block, 0, stream, a, list1, (const uint32_t*)ws, lane_min,
                                   binned ? (const uint32_t*)(bins + tok::NBIN) : (const uint32_t*)nullptr, (const uint32_t*)list0);
                e = hipGetLastError();
            }
            if (e == hipSuccess && lane_min != 0u) e = launch_inflate_dyn(a, stream, false, ws, lane_min);   // (one of the two returns at once)
        }
    }
    // stage 2: what is still flagged
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tok::k_collect_dyn, cgrid, cblock, 0, stream, a.status, a.nstreams, list0, ws + 1, (const uint64_t*)nullptr,
                           (uint32_t*)nullptr, (uint32_t)HDLZ_E_DYNAMIC_UNSUPPORTED);
        hipLaunchKernelGGL((tok::k_inflate_tok<true, tok::CAP_FULL>), grid, block, 0, stream, a, (const uint32_t*)list0,
                           (const uint32_t*)(ws + 1), lane_min, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        e = hipGetLastError();
        if (e == hipSuccess && lane_min != 0u) e = launch_inflate_dyn(a, stream, false, ws + 1, lane_min);
    }
    const hipError_t e2 = w.put(reinterpret_cast<uint8_t*>(ws), stream);
    return e != hipSuccess ? e : e2;
}

}  // namespace hdlz
