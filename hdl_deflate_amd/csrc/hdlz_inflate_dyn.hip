// hdlz_inflate_dyn.hip -- second inflate pass for streams that contain dynamic-tree (BTYPE=2) blocks.
//
// SURVEY.md 8(f) rank 1.  Replaces the reference's dynamic-tree machinery: BL/READBL/REPEAT/INIT3/
// DISTTREE (/root/reference/deflate.py:1084-1202), canonical code construction HF1..HF4/SPREAD
// (:1204-1400) and the distance decode D_NEXT/D_NEXT_2 (:1447-1517); stored and fixed blocks met in
// the same stream are handled here too (the stream is restarted from its first block).
//
// Per-stream Huffman tables cannot live per LANE in LDS, so the mapping differs from k_inflate:
// ONE WAVE PER STREAM, and the 64 lanes decode speculatively at the 64 next bit offsets of the stream ("window
// decode", see below); an LZ copy of length L, distance D is lane-parallel too, in ceil(L/64) steps with
//     out[o+i] = out[o - D + (i mod D)]
// which has no dependency on bytes produced by the same copy, whatever the overlap.  Output goes through
// a 2 KiB LDS history ring and is flushed to HBM as full 64-byte lines.
// The reference's table layout (10-bit instant table + incremental-mask retry) is FPGA-specific; a
// canonical count/offset decoder returns the same symbols for every valid code.  Invalid code
// descriptions are HDLZ_E_BAD_TREE (zlib's acceptance rules: over-subscribed sets rejected, incomplete
// sets only with a single code -- or, for the distance code, with none at all).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"

namespace hdlz {

constexpr uint32_t DRING = 2048;    // history ring bytes (power of two)
constexpr uint32_t DCHUNK = 64;
constexpr uint32_t IWIN = 512;      // compressed-input window staged in LDS (bytes)
constexpr uint32_t DYN_ALL = 0x80000000u;   // internal flag bit: this kernel is the only pass, every stream is its
constexpr uint32_t DYN_FINAL = 0x40000000u; // internal flag bit (STREAM): in_len is the final stream length
constexpr uint32_t DYN_HDR_BYTES = 700;     // a dynamic block header is at most 14 + 19*3 + 316*(7+7) bits = 562 bytes
constexpr uint32_t LUTL = 9, LUTD = 7;  // bits of the two look-up tables in front of the X-word decode
constexpr uint32_t WCAP = 448;       // output bytes committed per decode window (< 512: see the ring argument at the commit)

struct __attribute__((aligned(16))) DynLds {
    uint8_t ring[DRING];
    uint32_t iwin[IWIN / 4];         // input window: stream bytes [iwbase, iwbase + IWIN)
    uint8_t lengths[320];
    uint16_t lsym[288];
    uint16_t dsym[32];
    uint16_t csym[20];
    uint16_t cnt[3][16];             // code counts per length: [0] code-length code, [1] lit/len, [2] distance
    int32_t left[3];
    uint32_t cnt32[16];              // scratch of canon_build
    uint32_t next[16];
    uint16_t lutl[1u << 9];          // window decode: literal/length codes of up to LUTL bits, symbol | length << 9 by the next LUTL stream bits
    uint16_t lutd[1u << 7];          // ... distance codes of up to LUTD bits, symbol | length << 5 (0 = no such short code: the X words decide)
    uint64_t mdesc[64];              // commit: the window's matches whose source lies before the window (position, length, distance)
};

typedef uint32_t __attribute__((aligned(1))) u32u;
#ifdef HDLZ_DYN_MARKS                         // tools/phase_count.py --src hdlz_inflate_dyn.hip -DHDLZ_DYN_MARKS --kernel k_inflate_dynILb0E
#define DYN_MARK(name) asm volatile("; @@PHASE " name ::: "memory")
#else
#define DYN_MARK(name) do {} while (0)
#endif

__device__ __forceinline__ uint32_t dload32(const uint8_t* __restrict__ z, uint32_t ip, uint32_t zn) {
    if (ip + 4u <= zn) return *reinterpret_cast<const u32u*>(z + ip);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4u; k++)
        if (ip + k < zn) v |= (uint32_t)z[ip + k] << (8u * k);
    return v;
}

// RFC1951 3.2.5 in closed form, branch-free (these run per lane inside the window decode)
__device__ __forceinline__ void d_length_info(uint32_t token, uint32_t& base, uint32_t& eb) {
    const uint32_t e = max(token >> 2, 1u) - 1u;                     // 0 for token < 8
    const uint32_t b = token < 8u ? token : ((4u + (token & 3u)) << e);
    eb = token == 28u ? 0u : e;
    base = token == 28u ? 258u : 3u + b;
}
__device__ __forceinline__ void d_dist_info(uint32_t dc, uint32_t& base, uint32_t& eb) {
    eb = max(dc >> 1, 1u) - 1u;                                      // 0 for dc < 4
    base = 1u + (dc < 4u ? dc : ((2u + (dc & 1u)) << eb));
}

// canonical code construction, wave-parallel (all 64 lanes call it; n <= 320):
//   counts by LDS atomics, offsets by a 15-step serial prefix, symbol placement ordered by (length, value)
//   through one ballot per code length and round: rank = bits of the ballot below this lane (v_mbcnt).
// (A single-lane version is a chain of ~600 dependent LDS round trips per block and dominated the kernel.)
__device__ void canon_build(DynLds& L, int which, const uint8_t* len, uint16_t* symbol, int n, uint32_t lane) {
    uint32_t* cnt32 = L.cnt32;
    uint32_t* nxt = L.next;
    __syncthreads();
    if (lane < 16u) cnt32[lane] = 0;
    __syncthreads();
    for (int s = (int)lane; s < n; s += 64) atomicAdd(&cnt32[len[s]], 1u);
    __syncthreads();
    if (lane == 0) {
        int left = 1;
        uint32_t o = 0;
        for (int l = 1; l < 16; l++) {
            const uint32_t c = cnt32[l];
            nxt[l] = o;
            o += c;
            left = (left << 1) - (int)c;
            if (left < 0) break;
        }
        nxt[0] = 0;
        L.left[which] = left;
    }
    if (lane < 16u) L.cnt[which][lane] = (uint16_t)cnt32[lane];
    __syncthreads();
    if (L.left[which] < 0) return;
    for (int r = 0; r < n; r += 64) {
        const int s = r + (int)lane;
        const uint32_t mylen = s < n ? len[s] : 0u;
        uint32_t rank = 0, addv = 0;
#pragma unroll
        for (uint32_t l = 1; l < 16u; l++) {
            const bool is = mylen == l;
            const uint64_t m = ballot64(is);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            rank = is ? below : rank;
            addv = lane == l ? (uint32_t)__popcll(m) : addv;
        }
        if (mylen != 0u) symbol[nxt[mylen] + rank] = (uint16_t)s;
        __syncthreads();
        if (lane < 16u) nxt[lane] += addv;
        __syncthreads();
    }
}

// ---- window decode -----------------------------------------------------------------------------------------
// The token decode of ONE stream is a serial chain, and hipcc runs a wave-uniform chain on the CU's single scalar
// unit (0.97 instructions per cycle per CU, tools/ubench/salu_rate.hip): ~90 scalar instructions per token was the
// bound of the first version.  Now all 64 lanes decode speculatively, lane k at bit offset k of the next 64 stream
// bits (a canonical code needs no table for that: with the codes left-aligned to 15 bits, the codes of length <= l
// end at hi[l]; the length of the code in front of a lane is found by 15 compare+select steps against per-length
// words X[l] = hi[l] | first-symbol-index << 16 | (l+1) << 25 held in registers).  A short scalar loop then follows
// the real chain through the per-lane bit counts, a wave scan gives every chained token its output position, the
// reference's checks are evaluated per lane in its order, literals are written in parallel and only the LZ copies
// remain serial (lane-parallel inside each copy).
struct XCode { uint32_t x[16]; };
__device__ __forceinline__ void build_x(XCode& X, const uint16_t* count) {
    uint32_t first = 0, index = 0;
    X.x[0] = 1u << 25;                                               // shorter than hi[1]: length 1, first symbol 0, base 0
#pragma unroll
    for (int l = 1; l < 16; l++) {
        const uint32_t c = count[l];
        index += c;
        X.x[l] = ((first + c) << (15 - l)) | (index << 16) | ((uint32_t)(l + 1) << 25);
        first = (first + c) << 1;
    }
}
// -> code length (16 = no code starts with these bits) and index into the sorted symbol list
__device__ __forceinline__ void xwalk(const XCode& X, uint32_t bits15, uint32_t& len, uint32_t& symi) {
    const uint32_t V = __builtin_bitreverse32(bits15 & 0x7FFFu) >> 17;
    uint32_t sel = X.x[0];
#pragma unroll
    for (int l = 1; l < 16; l++) {
        // sel = (V >= hi[l]) ? X[l] : sel  -- the 16-bit compare reads hi[l] straight out of the packed word (two VALU
        // instructions per step; the C form needs a third one for the mask)
        uint64_t m;
        asm("v_cmp_le_u16_e64 %1, %2, %3\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(sel), "=&s"(m) : "v"(X.x[l]), "v"(V));
    }
    len = sel >> 25;
    const uint32_t base = sel & 0xFFFFu, idx = (sel >> 16) & 0x1FFu;
    symi = idx + ((V - base) >> (15u - min(len, 15u)));
}
// STREAM = the resumable form for ONE stream that arrives in pieces (hdlz_inflate_chunk): the decoder
// state lives in a device-resident hdlz_istate between the calls; a call decodes until the stream ends, the input known so
// far runs out (the reference's `di >= isize - 4 and not i_mode == IDLE` stall, deflate.py:1529-1530) or the output limit
// is reached (its `do >= i_raddr + OBSIZE` hold, deflate.py:1531-1534, :1597-1599), always stopping BETWEEN two tokens.
template <bool STREAM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_inflate_dyn(InflateArgs a, hdlz_istate* ist, uint32_t out_limit, const uint32_t* few_n, uint32_t lane_min) {
    __shared__ DynLds L;
    const uint32_t lane = threadIdx.x;
    // second pass of the lane mapping: this kernel takes the flagged streams only when they are few (*few_n of them, counted by
    // k_collect_dyn) -- from `lane_min` (HDLZ_INFLATE_DYN_LANE_MIN; 0 with the explicit lane hint) on one lane per stream is
    // faster (k_inflate_tok<true>)
    // (a second pass with nothing left to do -- the stage behind k_inflate_tok<true, CAP_SMALL> -- does not scan the statuses at all)
    if (!STREAM && few_n && (*few_n >= lane_min || (*few_n == 0u && !(a.flags & DYN_ALL)))) return;
    for (uint64_t sid = blockIdx.x; sid < a.nstreams; sid += gridDim.x) {
        if (!STREAM && !(a.flags & DYN_ALL) && a.status[sid] != HDLZ_E_DYNAMIC_UNSUPPORTED) continue;   // pass 1 finished this stream
        uint64_t off;
        uint32_t zn;
        if (a.in_off) {
            off = a.in_off[sid];
            zn = (uint32_t)(a.in_off[sid + 1] - off);
        } else {
            off = sid * a.in_pitch;
            zn = a.in_len;
        }
        const bool sfinal = !STREAM || (a.flags & DYN_FINAL) != 0;      // the whole stream is here: the reference's end-of-input checks apply
        uint32_t need = 0;                                              // STREAM: why this call stops (1 = input, 2 = output room)
        if (STREAM) {
            if (ist->done || ist->status != HDLZ_OK) return;            // a finished or failed session stays as it is
            if (!sfinal && zn < 16u) {                                  // nothing decodable yet
                if (lane == 0) { ist->need = 1; }
                return;
            }
        }
        if (zn < 5u) {                              // R0/D0: the reference never starts (only reachable when this kernel is the first pass)
            if (STREAM) { if (lane == 0) { ist->status = HDLZ_E_SHORT_INPUT; ist->out_pos = 0; } return; }
            if (lane == 0) { a.out_len[sid] = 0; a.status[sid] = HDLZ_E_SHORT_INPUT; }
            continue;
        }
        const uint8_t* __restrict__ z = a.in + off;
        uint8_t* __restrict__ out = a.out + sid * a.out_pitch;
        const uint32_t cap0 = a.out_pitch > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a.out_pitch;
        const uint32_t cap = STREAM ? (out_limit < cap0 ? out_limit : cap0) : cap0;      // STREAM: reaching it is a stop, not an error
        const uint32_t inbits = 8u * zn;                                                 // (zn < 2^28)
        const uint32_t obsize = a.obsize ? a.obsize : 32768u;
        const uint32_t len_mask = a.obsize ? ((1u << (31u - (uint32_t)__builtin_clz(a.obsize))) - 1u) : 0xFFFFu;
        const int32_t isize = (int32_t)zn - 1;

        uint32_t status = HDLZ_OK;
        uint32_t o = 0;                     // bytes produced
        uint64_t bb = 0;
        uint32_t bc = 0, ip = 2;            // D0: zlib header skipped unvalidated
        uint32_t save_bit = 16, save_phase = 0, save_srem = 0;          // STREAM: where and how the next call resumes
        // a serial decoder must not pay a global-load latency per 4 input bytes: the compressed stream is
        // staged through an LDS window, filled cooperatively (coalesced) whenever the reader runs off its end
        uint32_t iwbase = ip - IWIN;        // forces a fill at the first refill
#define REFILL() do {                                                                                   \
        if (bc <= 32u) {                                                                                 \
            if (ip - iwbase >= IWIN) {                                                                   \
                __syncthreads();                                                                         \
                iwbase = ip;                                                                             \
                for (uint32_t k = lane; k < IWIN / 4u; k += 64u) L.iwin[k] = dload32(z, iwbase + 4u * k, zn); \
                __syncthreads();                                                                         \
            }                                                                                            \
            bb |= (uint64_t)L.iwin[(ip - iwbase) >> 2] << bc; bc += 32u; ip += 4u;                       \
        }                                                                                                \
    } while (0)
#define BITPOS() (8u * ip - bc)
#define TAKE(n) do { bb >>= (n); bc -= (n); } while (0)
#define FAIL(code) do { status = (code); goto done; } while (0)
        // flush every completed 64-byte line in [from, to)
        auto flush_lines = [&](uint32_t from, uint32_t to) {
            for (uint32_t c0 = from & ~(DCHUNK - 1u); c0 + DCHUNK <= to; c0 += DCHUNK)
                out[c0 + lane] = L.ring[(c0 + lane) & (DRING - 1u)];
        };

        // the LDS window must hold the 64 + 48 bits from `bitp` on and the two dwords a lane's funnel shift reads beyond
        auto ensure_window = [&](uint32_t bitp) {
            if ((bitp >> 3) - iwbase >= IWIN - 32u) {               // (also true for the "force a fill" value of iwbase)
                __syncthreads();
                iwbase = (bitp >> 3) & ~3u;
                for (uint32_t k = lane; k < IWIN / 4u; k += 64u) L.iwin[k] = dload32(z, iwbase + 4u * k, zn);
                __syncthreads();
            }
        };
        // 64 stream bits from bit `bitpos` on (per lane)
        auto bits64 = [&](uint32_t bitpos) -> uint64_t {
            const uint32_t rel = bitpos - 8u * iwbase;
            const uint32_t w = rel >> 5, sh = rel & 31u;
            const uint32_t d0 = L.iwin[w], d1 = L.iwin[w + 1u], d2 = L.iwin[w + 2u];
            return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
        };
        // hand an absolute bit position back to the scalar bit reader
#define RESYNC(bitp) do { ip = (bitp) >> 3; bb = 0; bc = 0; iwbase = ip - IWIN; REFILL(); TAKE((bitp) & 7u); } while (0)

        // ---- STREAM: pick the session up where the previous call parked it
        uint32_t r_phase = 0, r_final = 0, r_hm = 0, r_srem = 0, r_nlen = 0, r_ndist = 0;
        if (STREAM && ist->started) {
            o = ist->out_pos; r_phase = ist->phase; r_final = ist->final_; r_hm = ist->hm; r_srem = ist->srem;
            r_nlen = ist->nlen; r_ndist = ist->ndist;
            save_bit = ist->bitpos; save_phase = r_phase; save_srem = r_srem;
            // the history ring is re-primed from the output produced so far
            const uint32_t h0 = o > DRING ? o - DRING : 0u;
            for (uint32_t k = h0 + lane; k < o; k += 64u) L.ring[k & (DRING - 1u)] = out[k];
            if (r_phase == 1u && r_hm == 2u)
                for (uint32_t k = lane; k < 320u; k += 64u) L.lengths[k] = ist->lengths[k];
            __syncthreads();
            RESYNC(save_bit);
        }
        for (;;) {
            uint32_t final_, hm;
            uint32_t blk_bit = 0;
            bool resumed = false;
            if (STREAM && r_phase != 0u) {          // inside a block: no header to read
                final_ = r_final; hm = r_hm; resumed = true;
            } else {
            REFILL();
            blk_bit = BITPOS();
            if (STREAM && !sfinal && blk_bit + 64u > inbits) { save_bit = blk_bit; save_phase = 0; need = 1; goto done; }
            // HEADER (deflate.py:677-732)
            final_ = ((uint32_t)bb & 1u) | ((a.flags & HDLZ_INFLATE_ONEBLOCK) ? 1u : 0u);   // ONEBLOCK: deflate.py:678,:1542,:1617
            hm = (a.flags & HDLZ_INFLATE_ASSUME_FIXED) ? 1u : ((uint32_t)(bb >> 1) & 3u);   // DYNAMIC=False build: deflate.py:724-732
            }
            if (hm == 3u) FAIL(HDLZ_E_BAD_BTYPE);
            if (hm == 0u) {
                // stored (deflate.py:709-717, COPY :1603-1626)
                uint32_t length, p0;
                if (STREAM && resumed) {                      // the rest of a stored block the previous call could not finish
                    length = r_srem; p0 = save_bit >> 3;
                } else {
                    const uint32_t dio = BITPOS() & 7u;
                    uint32_t skip = 8u - dio;
                    if (skip <= 2u) skip = 16u - dio;
                    length = (uint32_t)(bb >> skip) & 0xFFFFu & len_mask;
                    TAKE(skip + 16u);
                    REFILL();
                    TAKE(16u);                                // NLEN unchecked (D2)
                    p0 = BITPOS() >> 3;                       // first data byte (bit reader is byte aligned here)
                }
                // the reference checks, byte by byte and in this order: input left (deflate.py:1600), then room
                const uint32_t i_noeof = (int32_t)p0 >= isize ? 0u : (uint32_t)isize - p0;
                const uint32_t i_cap = cap - o;
                const uint32_t whole = length;
                if (STREAM) {
                    // input known so far: the reference's COPY holds while di >= isize - 2 (deflate.py:1600)
                    const uint32_t i_in = sfinal ? i_noeof : (zn >= p0 + 2u ? zn - 2u - p0 : 0u);
                    const uint32_t lim = i_in < i_cap ? i_in : i_cap;
                    if (length > lim) {
                        if (i_in <= i_cap) { if (sfinal) FAIL(HDLZ_E_NO_EOF); need = 1; }
                        else need = 2;
                        length = lim;                         // copy what can be copied, park the rest
                    }
                } else if (length > (i_noeof < i_cap ? i_noeof : i_cap)) FAIL(i_noeof <= i_cap ? HDLZ_E_NO_EOF : HDLZ_E_OUT_CAPACITY);
                {
                    // bytes of the current, still unflushed line live only in the ring: push them out first
                    const uint32_t c0 = o & ~(DCHUNK - 1u);
                    if (c0 + lane < o) out[c0 + lane] = L.ring[(c0 + lane) & (DRING - 1u)];
                    // the block itself goes straight to the output; the ring keeps its last DRING bytes
                    for (uint32_t i = lane; i < length; i += 64u) out[o + i] = z[p0 + i];
                    const uint32_t t0 = length > DRING ? length - DRING : 0u;
                    for (uint32_t i = t0 + lane; i < length; i += 64u) L.ring[(o + i) & (DRING - 1u)] = z[p0 + i];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                o += length;
                if (STREAM) {
                    if (need == 0u && !sfinal && p0 + length + 2u > zn) need = 1;           // is the block properly followed?  not known yet
                    if (need != 0u) {
                        save_bit = 8u * (p0 + length); save_phase = 2; save_srem = whole - length;
                        r_final = final_; r_hm = 0;
                        goto done;
                    }
                    r_phase = 0;
                }
                if ((int32_t)(p0 + length) >= isize) FAIL(HDLZ_E_NO_EOF);      // deflate.py:1617-1626 with the COPY hold
                ip = p0 + length; bb = 0; bc = 0;                              // resynchronise the bit reader
                iwbase = ip - IWIN;                                            // ... and refill the window there
                if (final_) break;
                continue;
            }
            if (!(STREAM && resumed)) {
                // STREAM: a dynamic header is only parsed when it is here as a whole
                if (STREAM && !sfinal && hm == 2u && blk_bit + 8u * DYN_HDR_BYTES > inbits) { save_bit = blk_bit; save_phase = 0; need = 1; goto done; }
                TAKE(3u);
            }
            uint32_t nlen = r_nlen, ndist = r_ndist;
            if (hm == 1u) {
                // fixed code lengths (deflate.py:1066-1073)
                for (uint32_t s = lane; s < 288u; s += 64u) L.lengths[s] = (uint8_t)(s < 144u ? 8 : s < 256u ? 9 : s < 280u ? 7 : 8);
                if (lane < 32u) L.lengths[288u + lane] = 5;
                __syncthreads();
                canon_build(L, 1, L.lengths, L.lsym, 288, lane);
                canon_build(L, 2, L.lengths + 288, L.dsym, 32, lane);
                __syncthreads();
            } else if (STREAM && resumed) {
                // the code lengths of the block came back with the state: rebuild the two codes
                canon_build(L, 1, L.lengths, L.lsym, (int)nlen, lane);
                canon_build(L, 2, L.lengths + nlen, L.dsym, (int)ndist, lane);
                __syncthreads();
            } else {
                // BL (deflate.py:1090-1114)
                REFILL();
                nlen = ((uint32_t)bb & 31u) + 257u;
                ndist = ((uint32_t)(bb >> 5) & 31u) + 1u;
                const uint32_t ncode = ((uint32_t)(bb >> 10) & 15u) + 4u;
                TAKE(14u);
                if (nlen > 286u || ndist > 30u) FAIL(HDLZ_E_BAD_TREE);
                for (uint32_t s = lane; s < 320u; s += 64u) L.lengths[s] = 0;
                __syncthreads();
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint32_t hbp = BITPOS();                            // from here on the header is read through the window
                {
                    // the ncode 3-bit lengths of the code-length code: lane i takes its own field (ncode * 3 <= 57 bits)
                    ensure_window(hbp);
                    const uint64_t x = bits64(hbp);
                    if (lane < ncode) L.lengths[order[lane]] = (uint8_t)((uint32_t)(x >> (3u * lane)) & 7u);
                    hbp += 3u * ncode;
                }
                __syncthreads();
                canon_build(L, 0, L.lengths, L.csym, 19, lane);
                __syncthreads();
                if (L.left[0] != 0) FAIL(HDLZ_E_BAD_TREE);
                XCode XC;
                build_x(XC, L.cnt[0]);
                __syncthreads();
                for (uint32_t s = lane; s < 320u; s += 64u) L.lengths[s] = 0;     // reuse as the real length list
                __syncthreads();
                // READBL / REPEAT (deflate.py:1116-1164, :1190-1202), window decode: every lane decodes the code-length
                // symbol (+ its repeat field) that would start at its bit offset; the scalar chain applies them in order
                uint32_t idx = 0, prev = 0;
                while (idx < nlen + ndist) {
                    ensure_window(hbp);
                    uint32_t packed;                               // bits | sym << 8 | repeat << 16, 0 = no code here
                    {
                        const uint64_t x = bits64(hbp + lane);
                        uint32_t len, symi;
                        xwalk(XC, (uint32_t)x, len, symi);
                        const uint32_t sym = L.csym[min(symi, 18u)];
                        const uint32_t eb = sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u;
                        const uint32_t ev = (uint32_t)(x >> min(len, 15u)) & ((1u << eb) - 1u);
                        const uint32_t rep = sym < 16u ? 1u : sym == 18u ? 11u + ev : 3u + ev;
                        packed = len <= 15u ? ((len + eb) | (sym << 8) | (rep << 16)) : 0u;
                    }
                    uint32_t cur = 0;
                    while (cur < 64u && idx < nlen + ndist) {
                        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)packed, (int)cur);
                        if (t == 0u) FAIL(HDLZ_E_BAD_TREE);
                        const uint32_t sym = (t >> 8) & 255u, rep = t >> 16;
                        uint32_t val = 0;
                        if (sym < 16u) val = sym;
                        else if (sym == 16u) { if (idx == 0u) FAIL(HDLZ_E_BAD_TREE); val = prev; }
                        if (idx + rep > nlen + ndist) FAIL(HDLZ_E_BAD_TREE);
                        for (uint32_t k = lane; k < rep; k += 64u) L.lengths[idx + k] = (uint8_t)val;
                        prev = val;
                        idx += rep;
                        cur += t & 255u;
                    }
                    hbp += cur;
                }
                RESYNC(hbp);
                __syncthreads();
                if (L.lengths[256] == 0) FAIL(HDLZ_E_BAD_TREE);          // no end-of-block code
                canon_build(L, 1, L.lengths, L.lsym, (int)nlen, lane);
                canon_build(L, 2, L.lengths + nlen, L.dsym, (int)ndist, lane);
                __syncthreads();
                {
                    const int l1 = L.left[1], l2 = L.left[2];
                    if (l1 < 0 || (l1 > 0 && (int)nlen - (int)L.cnt[1][0] != 1)) FAIL(HDLZ_E_BAD_TREE);
                    // an EMPTY distance set (a block of literals only) is legal, RFC1951 3.2.7 -- zlib's inflate_table (max == 0)
                    // and puff accept it; a distance symbol met later then finds no code (dlen > 15 -> BAD_SYMBOL)
                    if (l2 < 0 || (l2 > 0 && (int)ndist - (int)L.cnt[2][0] > 1)) FAIL(HDLZ_E_BAD_TREE);
                }
                if (sfinal && (int32_t)(BITPOS() >> 3) > isize - 3) FAIL(HDLZ_E_NO_EOF);
                if (STREAM) {                                        // the block's code lengths travel with the session
                    for (uint32_t k = lane; k < 320u; k += 64u) ist->lengths[k] = L.lengths[k];
                    if (lane == 0) { ist->nlen = nlen; ist->ndist = ndist; }
                }
            }
            if (STREAM) { r_final = final_; r_hm = hm; r_nlen = nlen; r_ndist = ndist; r_phase = 0; }
            {
                XCode XL, XD;
                build_x(XL, L.cnt[1]);
                build_x(XD, L.cnt[2]);
                // the two tables: entry e = the code that the stream bits e (LSB first) start with, if it is short enough
                for (uint32_t e = lane; e < (1u << LUTL); e += 64u) {
                    uint32_t l_, si_;
                    xwalk(XL, e, l_, si_);
                    L.lutl[e] = (uint16_t)(l_ <= LUTL ? (L.lsym[min(si_, 287u)] | (l_ << 9)) : 0u);
                }
                for (uint32_t e = lane; e < (1u << LUTD); e += 64u) {
                    uint32_t l_, si_;
                    xwalk(XD, e, l_, si_);
                    L.lutd[e] = (uint16_t)(l_ <= LUTD ? (L.dsym[min(si_, 31u)] | (l_ << 5)) : 0u);
                }
                __syncthreads();
                uint32_t bp = BITPOS();                             // absolute bit position of the next token
#ifdef HDLZ_DYN_X_HDRONLY
                for (bool eob = true; !eob;) {
#else
                for (bool eob = false; !eob;) {
#endif
                    DYN_MARK("window");
                    ensure_window(bp);
                    DYN_MARK("decode");
                    // ---- every lane: the token that would start at bit bp + lane
                    const uint32_t bitpos = bp + lane;
                    uint32_t len, sym, tlen, distance, total, token, dlen, ds;
                    bool lit, eobt, ismatch, valid, mvalid, zleaf;
                    {
                        // branch-free: every lane computes a match's fields, the flags say what they are worth.  Codes of up to
                        // LUTL / LUTD bits -- all of them in most blocks -- come out of a table by the next stream bits; when a lane
                        // that matters sees a longer one the whole wave takes the X-word decode (xwalk, ~38 VALU instructions)
                        const uint64_t x = bits64(bitpos);
                        const uint32_t el = L.lutl[(uint32_t)x & ((1u << LUTL) - 1u)];
                        len = el >> 9; sym = el & 511u;
                        if (ballot64(el == 0u) != 0ull) {
                            uint32_t symi;
                            xwalk(XL, (uint32_t)x, len, symi);
                            sym = L.lsym[min(symi, 287u)];
                        }
                        valid = len <= 15u;
                        // the reference's ONE zero leaf: stat_leaves[483] of the DYNAMIC=False build = symbol 287's 8-bit code followed by a 1
                        // (deflate.py:212); its other slot (a 0 follows) is an ordinary leaf -- code 287 passes NEXT and fails in INFLATE, after
                        // the end-of-input check --, and so are both in a DYNAMIC=True build (leaves built from the fixed lengths)
                        zleaf = (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0u && sym == 287u && (((uint32_t)x >> 8) & 1u) != 0u;
                        lit = valid && sym < 256u;
                        eobt = valid && sym == 256u;
                        ismatch = valid && sym > 256u;
                        token = sym - 257u;
                        uint32_t lbase, leb, dbase, deb;
                        d_length_info(min(token, 28u), lbase, leb);
                        const uint64_t x1 = x >> len;
                        tlen = lbase + ((uint32_t)x1 & ((1u << leb) - 1u));
                        const uint64_t x2 = x1 >> leb;
                        const uint32_t ed = L.lutd[(uint32_t)x2 & ((1u << LUTD) - 1u)];
                        dlen = ed >> 5; ds = ed & 31u;
                        if (ballot64(ismatch && token < 29u && ed == 0u) != 0ull) {
                            uint32_t dsymi;
                            xwalk(XD, (uint32_t)x2, dlen, dsymi);
                            ds = L.dsym[min(dsymi, 31u)];
                        }
                        d_dist_info(min(ds, 29u), dbase, deb);
                        distance = dbase + ((uint32_t)(x2 >> min(dlen, 15u)) & ((1u << deb) - 1u));
                        mvalid = ismatch && token < 29u && dlen - 1u <= 14u && ds < 30u;
                        total = mvalid ? len + leb + dlen + deb : len;
                    }
                    // ---- the real chain: token at bit 0, then at the end of each chained token, up to bit 63; it stops
                    // at anything that is not a plain literal / complete match (their bit count is not trusted)
                    DYN_MARK("chain");
                    const bool plain = lit || (ismatch && mvalid);
                    uint64_t chain = 0;
                    uint32_t cur = 0;
                    uint32_t excl = 0;                              // output bytes of the chained tokens before this one
                    uint32_t acc = 0;
                    {
                        // bits | output length << 8; 0 = stop.  The running output offset is written into the chained lanes
                        // on the way: a wave scan would cost six LDS-crossbar round trips per window.
                        const uint32_t step = plain ? (total | ((lit ? 1u : tlen) << 8)) : 0u;
                        while (cur < 64u) {
                            chain |= 1ull << cur;
                            excl = (lane == cur) ? acc : excl;       // (v_writelane with two SGPR operands breaks the constant-bus rule)
                            const uint32_t st_ = (uint32_t)__builtin_amdgcn_readlane((int)step, (int)cur);
                            if (st_ == 0u) break;
                            acc += st_ >> 8;
                            cur += st_ & 255u;
                        }
                    }
                    DYN_MARK("checks");
                    const bool inchain = (chain >> lane) & 1ull;
                    const uint32_t outlen = (inchain && plain) ? (lit ? 1u : tlen) : 0u;
                    const uint32_t pos = o + excl;
                    // ---- the reference's checks, per token, in its order (deflate.py:1409-1445, :1519-1591, :1447-1517, :1600)
                    constexpr uint32_t ST_EOB = 100, ST_CUT = 101, ST_NEEDIN = 102, ST_NEEDOUT = 103;
                    // (evaluated as a chain of selects from the LAST check of the reference's order to the first -- an earlier check
                    // overrides a later one; the same conditions as `if ... else if` cost ~90 scalar mask instructions per window)
                    uint32_t st;
                    {
                        const uint32_t outcode = STREAM ? ST_NEEDOUT : (uint32_t)HDLZ_E_OUT_CAPACITY;
                        uint32_t m = ((uint64_t)pos + tlen > cap) ? outcode : 0u;
                        m = (sfinal && (int32_t)((bitpos + total) >> 3) >= isize - 2) ? (uint32_t)HDLZ_E_NO_EOF : m;      // COPY hold, :1600
                        m = (distance > pos || distance > obsize) ? (uint32_t)HDLZ_E_BAD_DISTANCE : m;                    // deflate.py:1506-1508, D8
                        m = ds >= 30u ? (uint32_t)HDLZ_E_BAD_DISTANCE : m;
                        m = (dlen > 15u || token >= 29u) ? (uint32_t)HDLZ_E_BAD_SYMBOL : m;
                        st = lit ? (pos >= cap ? outcode : 0u) : m;
                        st = eobt ? ST_EOB : st;
                        st = (sfinal && (int32_t)((bitpos + len) >> 3) > isize - 3) ? (uint32_t)HDLZ_E_NO_EOF : st;       // deflate.py:1535-1539
                        st = (!valid || zleaf) ? (uint32_t)HDLZ_E_BAD_SYMBOL : st;                    // zero leaf, deflate.py:212,:1437-1439
                        // STREAM, more input to come: a token is only decoded when its 64 bits are here; the end-of-input checks
                        // (deflate.py:1535-1539, :1600) are the final call's business -- and hold for every token decoded earlier
                        if (STREAM) st = (!sfinal && bitpos + 64u > inbits) ? ST_NEEDIN : st;
                    }
#ifdef HDLZ_DYN_X_NOCHECK
                    st = eobt ? ST_EOB : 0u;
#endif
                    if (st == 0u && excl + outlen > WCAP) st = ST_CUT;                               // rest of the chain: next window
                    const uint64_t special = ballot64(inchain && st != 0u);
                    uint64_t commit = chain;
                    uint32_t consumed = cur, o_new;
                    if (special != 0ull) {
                        const uint32_t f = (uint32_t)__builtin_ctzll(special);
                        const uint32_t stf = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)f);
                        if (stf < ST_EOB) FAIL(stf);
                        commit = chain & ((1ull << f) - 1ull);
                        consumed = f;
                        if (stf == ST_EOB) { consumed = f + (uint32_t)__builtin_amdgcn_readlane((int)len, (int)f); eob = true; }
                        if (STREAM && (stf == ST_NEEDIN || stf == ST_NEEDOUT)) need = stf == ST_NEEDIN ? 1u : 2u;
                        o_new = o + (uint32_t)__builtin_amdgcn_readlane((int)excl, (int)f);
                    } else {
                        o_new = o + acc;                            // (no stop inside the window: acc covers the whole chain)
                    }
                    // ---- commit.  All literals first, in parallel; then the copies in stream order.  Ring argument: a window
                    // adds at most WCAP < 512 bytes, a copy reads the ring only up to DRING - 512 back, so a literal written
                    // ahead of a copy can never land in a ring slot that copy still reads.
                    DYN_MARK("commit");
                    const bool mine = (commit >> lane) & 1ull;
                    if (mine && lit) L.ring[pos & (DRING - 1u)] = (uint8_t)sym;
                    uint64_t mm = ballot64(mine && ismatch);
#ifdef HDLZ_DYN_X_NOCOPY
                    mm = 0ull;
#endif
                    {
                        // matches whose source lies wholly BEFORE this window (distance >= bytes of the window in front of the
                        // match + its length) and inside the ring depend on nothing the window produces: four at a time, sixteen
                        // lanes each -- the serial loop below (~30 VALU + ~30 scalar instructions per match) keeps the others
                        const bool par = mine && ismatch && distance >= excl + tlen && distance + tlen <= DRING - 512u;
                        const uint64_t pm = ballot64(par);
#ifdef HDLZ_DYN_X_NOCOPY
                        const uint32_t np = 0u;
#else
                        const uint32_t np = (uint32_t)__popcll(pm);
#endif
                        if (np != 0u) {
                            mm &= ~pm;
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                            if (par) L.mdesc[rank] = (uint64_t)pos | ((uint64_t)(tlen | (distance << 16)) << 32);
                            const uint32_t g = lane >> 4, sub = lane & 15u;
                            for (uint32_t base = 0; base < np; base += 4u) {
                                const uint32_t idx = base + g;
                                const uint64_t dsc = L.mdesc[min(idx, 63u)];
                                const uint32_t P = (uint32_t)dsc, tl = idx < np ? (uint32_t)(dsc >> 32) & 0xFFFFu : 0u, D = (uint32_t)(dsc >> 48);
                                for (uint32_t i = sub; i < tl; i += 16u) L.ring[(P + i) & (DRING - 1u)] = L.ring[(P - D + i) & (DRING - 1u)];
                            }
                        }
                    }
                    while (mm != 0ull) {
                        const uint32_t k = (uint32_t)__builtin_ctzll(mm);
                        mm &= mm - 1ull;
                        const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)k);
                        const uint32_t tlength = (uint32_t)__builtin_amdgcn_readlane((int)tlen, (int)k);
                        const uint32_t D = (uint32_t)__builtin_amdgcn_readlane((int)distance, (int)k);
                        // COPY (deflate.py:1627-1659), lane-parallel: out[P+i] = out[P - D + (i mod D)]
                        // (P, D, tlength are wave-uniform: the common cases get their own lean loops)
                        if (D + tlength <= DRING - 512u) {                      // source and destination inside the ring
                            if (D >= tlength) {
                                for (uint32_t i = lane; i < tlength; i += 64u) L.ring[(P + i) & (DRING - 1u)] = L.ring[(P - D + i) & (DRING - 1u)];
                            } else {
                                for (uint32_t i = lane; i < tlength; i += 64u) L.ring[(P + i) & (DRING - 1u)] = L.ring[(P - D + i % D) & (DRING - 1u)];
                            }
                            continue;
                        }
                        // far history is read back from HBM: only then must the earlier line flushes have landed
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        for (uint32_t i = lane; i < tlength; i += 64u) {
                            const uint32_t src = P - D + (D >= tlength ? i : i % D);
                            uint32_t byte;
                            if ((P + i) - src <= DRING - 512u) byte = L.ring[src & (DRING - 1u)];   // still in the ring for sure
                            else byte = out[src];                               // far history: flushed long ago
                            L.ring[(P + i) & (DRING - 1u)] = (uint8_t)byte;
                        }
                    }
                    DYN_MARK("flush");
                    flush_lines(o, o_new);
                    o = o_new;
                    bp += consumed;
                    DYN_MARK("loopend");
                    if (STREAM && need != 0u) { save_bit = bp; save_phase = 1; goto done; }   // parked between two tokens
                }
                DYN_MARK("after");
                RESYNC(bp);                                         // back to the scalar bit reader (block headers)
            }
            if (final_) break;                                                   // D6
        }
    done:
        __syncthreads();
        // STREAM: a stop for output room that no larger out_limit can ever give (the limit is not below the capacity) is the
        // batch call's HDLZ_E_OUT_CAPACITY, not need = 2 forever
        if (STREAM && status == HDLZ_OK && need == 2u && cap0 <= out_limit) status = HDLZ_E_OUT_CAPACITY;
        if (status == HDLZ_OK) {
            const uint32_t c0 = o & ~(DCHUNK - 1u);
            if (c0 + lane < o) out[c0 + lane] = L.ring[(c0 + lane) & (DRING - 1u)];
        }
        if (STREAM) {
            if (lane == 0) {
                ist->started = 1; ist->status = status; ist->need = status == HDLZ_OK ? need : 0u;
                ist->done = (status == HDLZ_OK && need == 0u) ? 1u : 0u;
                ist->out_pos = status == HDLZ_OK ? o : 0u;
                ist->bitpos = save_bit; ist->phase = save_phase; ist->srem = save_srem;
                ist->final_ = r_final; ist->hm = r_hm; ist->nlen = r_nlen; ist->ndist = r_ndist;
            }
            return;
        }
        if (lane == 0) {
            a.out_len[sid] = status == HDLZ_OK ? o : 0u;
            a.status[sid] = status;
        }
        __syncthreads();
#undef REFILL
#undef RESYNC
#undef BITPOS
#undef TAKE
#undef FAIL
    }
}

// second pass after k_inflate (streams it flagged with HDLZ_E_DYNAMIC_UNSUPPORTED), or -- `all` -- the only pass
// the streams whose status is HDLZ_E_DYNAMIC_UNSUPPORTED, whatever the flags (the several-streams form of hdlz_inflate_par.hip flags what
// it gives up on -- under HDLZ_INFLATE_ASSUME_FIXED too, where the batch kernels' second pass has nothing to do)
hipError_t launch_inflate_dyn_flagged(const InflateArgs& a, hipStream_t stream) {
    if (a.nstreams == 0) return hipSuccess;
    const uint64_t g = a.nstreams < 65536u ? a.nstreams : 65536u;
    hipLaunchKernelGGL(k_inflate_dyn<false>, dim3((unsigned)g), dim3(64), 0, stream, a, (hdlz_istate*)nullptr, 0u, (const uint32_t*)nullptr, 0u);
    return hipGetLastError();
}

hipError_t launch_inflate_dyn(const InflateArgs& a0, hipStream_t stream, bool all, const uint32_t* few_n, uint32_t lane_min) {
    if (a0.nstreams == 0 || (!all && (a0.flags & HDLZ_INFLATE_ASSUME_FIXED))) return hipSuccess;
    InflateArgs a = a0;
    if (all) a.flags |= DYN_ALL;
    uint64_t g = a.nstreams < 65536u ? a.nstreams : 65536u;
    if (few_n && lane_min != 0u && g > lane_min) g = lane_min;       // (list mode runs for fewer than lane_min streams: no more blocks than that)
    hipLaunchKernelGGL(k_inflate_dyn<false>, dim3((unsigned)g), dim3(64), 0, stream, a, (hdlz_istate*)nullptr, 0u, few_n, lane_min);
    return hipGetLastError();
}

// one call of a resumable session (hdlz_inflate_chunk): the stream bytes known so far, the session state, the output limit
hipError_t launch_inflate_chunk(const uint8_t* in, uint32_t in_len, int final_, uint32_t flags, uint32_t obsize, uint8_t* out,
                                uint64_t out_cap, uint32_t out_limit, void* state, hipStream_t stream) {
    InflateArgs a{in, nullptr, 0, in_len, 1, (flags & (HDLZ_INFLATE_ASSUME_FIXED | HDLZ_INFLATE_ONEBLOCK)) | DYN_ALL | (final_ ? DYN_FINAL : 0u),
                  obsize, out, out_cap, nullptr, nullptr};
    hipLaunchKernelGGL(k_inflate_dyn<true>, dim3(1), dim3(64), 0, stream, a, static_cast<hdlz_istate*>(state), out_limit,
                       (const uint32_t*)nullptr, 0u);
    return hipGetLastError();
}

}  // namespace hdlz
