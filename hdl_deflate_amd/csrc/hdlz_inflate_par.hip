// hdlz_inflate_par.hip -- STARTD for ONE large stream -- or a batch of up to HDLZ_INFLATE_PAR_BATCH_MAX of them, blockIdx.y = the stream -- on
// the whole GPU (the port adapter's case: the reference inflates one stream at a time, /root/reference/deflate.py:635-732 HEADER, :1402-1445 NEXT, :1519-1591 INFLATE, :1593-1659 COPY).
//
// One wave decoding one stream is a serial chain: ~9 MB/s (k_inflate_dyn), a fifth of what the FPGA does at 100 MHz.  A stream of ONE
// fixed-Huffman block -- what STARTC writes (deflate.py:429-466: 78 9C, BFINAL = 1, BTYPE = 1) and all the reference's DYNAMIC=False
// build reads -- can be cut anywhere, because a token is at most 32 bits long (9 + 5 + 5 + 13):
//   1. k_par_head    every piece of the stream (1 KiB; 512 / 256 bytes for smaller streams) is decoded from all 32 bit offsets a token can start
//                    at behind its first bit -- for its first 256 bits only: chains from different offsets fall into step at the first token
//                    boundary they share, so the lanes that stand at the same bit afterwards are ONE chain (listed once);
//      k_par_tail    one lane per listed chain decodes the rest of its piece: where the chain leaves the piece (offset into the next one,
//                    or EOB, or an undecodable symbol), how many bytes it produces, and the same at three sub-boundaries of the piece;
//      k_par_resolve every (piece, offset) takes its chain's results -- a 32-entry map per piece (k_par_spec: the 32-fold decode of whole
//                    pieces these three replace; kept for A/B with -DHDLZ_PAR_SPEC32);
//   2. k_par_scan_*  the 32-entry maps are walked from the stream's first token on (per group of 64 pieces for all 32 offsets, one
//                    wave over the groups, the pieces of every group again): the true entry offset and the output position of every
//                    piece, the total length;
//   3. k_par_tokens  one LANE per SUB-piece (a quarter of a piece; entry offsets from the sub-boundary maps) decodes it for real, with the reference's checks in the reference's order, into a token list;
//      k_par_emit    one wave per piece writes the bytes, 64 tokens at a time -- except that the history before the piece's own output
//                    is not there yet.  A byte copied from there becomes a MARKER: src[p] = the absolute position it comes from
//                    (markers are copied like bytes);
//   4. k_par_jump    pointer jumping over the markers IN PLACE: src[p] <- up to HOPS steps along its chain, until the source is a byte
//                    (one launch per pass, log_HOPS(pieces) + 1 passes at most; a pass with nothing left returns at once).
//   3'. k_par_ends   (round 6) streams of SEVERAL fixed blocks: the chains pass "end-of-block + next fixed header" like a token; the
//                    true end is the first listed end-of-block code of a block whose BFINAL was set (see starts_fixed below);
// Anything else -- a block of another type (a stream that does not even start with a fixed block: hdlz_inflate_any.hip), a failed check
// (NO EOF, bad symbol, bad distance, capacity) -- sets
// a fallback flag on the device and k_inflate_dyn redoes the stream from its first byte (it is launched behind the chain and
// returns at once otherwise): status words and bytes are those of the serial decoder by construction, the parallel path
// only ever reports HDLZ_OK.  Scratch (stream-ordered, from the library's own pool): ~1.6 KB per piece (the maps of the 32 offsets, of the sub-boundaries and of the
// listed chains, the token lists) and 4 bytes per possible output byte (markers; 8 up to round 4: two buffers).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "hdlz_device.h"
#include "hdlz_inflate_tables.h"
#include "hdlz_inflate_par.h"

namespace hdlz {
namespace par {

using tok::T_BAD;
using tok::T_EOB;
using tok::T_LIT;
using tok::T_LEN;

// (pointer arithmetic, NOT a round trip through an integer: that makes the pointer generic and every access through it a flat_load /
//  flat_store, which waits on the LDS counter as well -- k_par_emit ran 7.7 ms instead of 0.15 that way)
// XCD-aware item order (MI355X_MICROARCH.md: block b runs on XCD b % 8, each XCD has its own L2): workgroup b of n takes item
// xcd_item(b, n) -- XCD x works through ONE contiguous eighth of the items, so the history its copies and marker chains read (the bytes
// and marker words of the items just in front) is in ITS L2 instead of being fetched by all eight (measured on a 16 MiB zlib stream:
// profiles/r06_xcd_order_ab.txt).  A bijection of [0, n); speed only, no correctness depends on it.
__device__ __forceinline__ uint32_t xcd_item(uint32_t b, uint32_t n) {
#ifdef HDLZ_NO_XCD_ORDER                       // (A/B build)
    return b;
#endif
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u;
    return x * q + min(x, r) + (b >> 3);
}
template <typename T> __device__ __forceinline__ void shift_ptr(T*& p, size_t bytes) {
    p = reinterpret_cast<T*>(reinterpret_cast<uint8_t*>(p) + bytes);
}
// the arguments of stream blockIdx.y: every kernel below works on ONE stream and never looks at another one's arrays
__device__ __forceinline__ ParArgs of_stream(ParArgs a) {
    const uint32_t s = blockIdx.y;
    if (a.in_off) {
        const uint64_t o0 = a.in_off[s], n64 = a.in_off[s + 1u] - o0;
        a.z += o0;
        a.zn = n64 > (uint64_t)a.zn ? 0xFFFFFFFFu : (uint32_t)n64;      // longer than the stated bound: starts_fixed() sends it to the serial pass
    } else a.z += (uint64_t)s * a.in_pitch;
    if (s == 0u) return a;
    a.out += (uint64_t)s * a.out_pitch;
    a.out_len += s;
    a.status += s;
    const size_t d = (size_t)s * a.ws_stride;
    shift_ptr(a.ctl, d); shift_ptr(a.exit8, d); shift_ptr(a.nb32, d); shift_ptr(a.entry8, d); shift_ptr(a.opos, d);
    shift_ptr(a.gexit8, d); shift_ptr(a.gstop8, d); shift_ptr(a.gnb32, d); shift_ptr(a.gentry8, d); shift_ptr(a.gopos, d);
    shift_ptr(a.tokens, d); shift_ptr(a.ntok, d); shift_ptr(a.srcA, d); shift_ptr(a.mexit8, d); shift_ptr(a.mnb32, d); shift_ptr(a.mext, d);
    shift_ptr(a.cross, d); shift_ptr(a.nfail, d);
    return a;
}

__device__ __forceinline__ void fill_tables(uint32_t* lit, uint32_t* dst, uint32_t tid, uint32_t nthreads) {
    for (uint32_t c = tid; c < 512u; c += nthreads) lit[c] = tok::lit_entry(c, true);      // (symbols 286 / 287 send the stream to the serial decoder whatever their leaf says)
    if (tid < 32u) dst[tid] = tok::dst_entry(tid);
}
// the window of piece c: stream dwords from byte B0 = (first bit of the piece / 8) & ~3 on
__device__ __forceinline__ void stage_window(uint32_t* win, const uint8_t* z, uint32_t zn, uint32_t b_c, uint32_t chbits, uint32_t tid,
                                             uint32_t nthreads) {
    const uint32_t B0 = (b_c >> 3) & ~3u;
    for (uint32_t k = tid; k < chbits / 32u + 8u; k += nthreads) win[k] = tok::load32(z, B0 + 4u * k, zn);
}
// 64 stream bits from absolute bit position `pos` on
__device__ __forceinline__ uint64_t bits_at(const uint32_t* win, uint32_t b_c, uint32_t pos) {
    const uint32_t rel = pos - 8u * ((b_c >> 3) & ~3u);
    const uint32_t w = rel >> 5, sh = rel & 31u;
    const uint32_t d0 = win[w], d1 = win[w + 1u], d2 = win[w + 2u];
    return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
}

// The stream must START with a fixed block (or be read as fixed blocks: the DYNAMIC=False build) for this chain of kernels; round 6: it
// may go on with MORE fixed blocks -- what zlib's Z_FIXED strategy writes for anything beyond ~16 K symbols, what the DYNAMIC=False build
// makes of every stream (deflate.py:1540-1548 EOB -> HEADER, :677-685, :724-732).  An end-of-block code followed by the header of another
// fixed block is one more token of the chains -- 7 + 3 bits, no bytes -- WHATEVER the block's BFINAL was: a speculative chain does not know
// it.  So the chain of pieces runs through the true end of the stream into the trailer bytes behind it (and ends there, "bad" or at an
// end-of-block code that nothing fixed follows); the real decode (k_par_tokens) lists every end-of-block code it passes, and k_par_ends
// finds the first one that ends a block whose BFINAL was set: the true end, the total length; what lies behind it is cut off.
// k_par_scan_top gives the first verdict (not this chain's stream at all); the kernels in front of it only SKIP such a stream -- a batch
// of small dynamic-tree streams must not pay a speculative fixed-Huffman decode of every stream before another path takes them.
__device__ __forceinline__ bool starts_fixed(const ParArgs& a) {
    const uint32_t hdr = (a.zn >= 5u && a.zn != 0xFFFFFFFFu) ? (uint32_t)a.z[2] : 0u;
    const bool fixed = (a.flags & HDLZ_INFLATE_ASSUME_FIXED) || ((hdr >> 1) & 3u) == 1u;
    return a.zn >= 5u && a.zn != 0xFFFFFFFFu && fixed;
}
// what follows an end-of-block code: 0 = the chain ends there (the ONEBLOCK build: the stream ends with its first block), 1 = it goes on if
// the next header says BTYPE = 01, 2 = it goes on (the DYNAMIC=False build reads every block as fixed)
__device__ __forceinline__ uint32_t eob_mode(const ParArgs& a) {
    return (a.flags & HDLZ_INFLATE_ONEBLOCK) ? 0u : (a.flags & HDLZ_INFLATE_ASSUME_FIXED) ? 2u : 1u;
}
__device__ __forceinline__ bool eob_goes_on(uint32_t mode, uint32_t hdr3) { return mode == 2u || (mode == 1u && ((hdr3 >> 1) & 3u) == 1u); }

// ---- 1. speculative decode: lane (piece, offset)
template <bool SUBMAPS>
__global__ __launch_bounds__(64) void k_par_spec(ParArgs a_) {
    const ParArgs a = of_stream(a_);
    __shared__ uint32_t lit[512], dst[32], win[2][WIN_DW];
    const uint32_t lane = threadIdx.x, half = lane >> 5, e = lane & 31u;
    fill_tables(lit, dst, lane, 64u);
    const uint32_t c = blockIdx.x * 2u + half;
    const bool have = c < a.nchunks;
    const uint32_t b_c = FIRST_BIT + c * a.chbits, end = b_c + a.chbits;
    if (have) stage_window(win[half], a.z, a.zn, b_c, a.chbits, e, 32u);
    __syncthreads();
    uint32_t pos = b_c + e, nbytes = 0, exitc = 0;
    bool run = have;
    // one token of the chain (lengths and byte counts only), branch-free: the 32 chains of a piece stand at different kinds of tokens
    // at any moment, so a ladder of branches runs every path anyway and pays the exec-mask bookkeeping on top (the kernel is
    // issue-bound: 9 waves per SIMD of ~450 token steps).  A token needs at most 9 + 5 + 5 bits here: a 32-bit window (two LDS
    // dwords, one funnel shift) instead of bits_at's 64.
    const uint32_t bit0 = 8u * ((b_c >> 3) & ~3u);
    const uint32_t* w_ = win[half];
    auto token = [&]() {
        const uint32_t rel = pos - bit0;
        const uint32_t x = __builtin_amdgcn_alignbit(w_[(rel >> 5) + 1u], w_[rel >> 5], rel);       // (the shift is taken modulo 32)
        const uint32_t e0 = lit[x & 511u];
        const uint32_t nb = e0 & 15u, type = (e0 >> 13) & 3u, leb = (e0 >> 25) & 7u, lbase = (e0 >> 16) & 0x1FFu;
        const uint32_t y = x >> nb;
        const uint32_t tl = lbase + __builtin_amdgcn_ubfe(y, 0u, leb);
        const uint32_t de = dst[__builtin_amdgcn_ubfe(y, leb, 5u)];
        const bool islit = type == (uint32_t)T_LIT;
        const bool bad = (nb == 0u) | (type == (uint32_t)T_BAD) | ((type == (uint32_t)T_LEN) & (de == 0xFFFFFFFFu));
        const bool eob = type == (uint32_t)T_EOB;
        const uint32_t used = islit ? nb : nb + leb + 5u + ((de >> 16) & 15u);
        const uint32_t made = islit ? 1u : tl;
        const bool adv = !(bad | eob);
        exitc = bad ? X_BAD : eob ? X_EOB : exitc;
        pos += adv ? used : 0u;
        nbytes += adv ? made : 0u;
        run = adv;
    };
    if constexpr (SUBMAPS) {
        // The real decode (k_par_tokens) is ONE lane's serial chain per piece, ~1000 cycles per token at one wave per SIMD: for streams
        // that do not fill the GPU it runs on pieces SUB times shorter than these -- the entry offsets and output positions at the
        // sub-boundaries are read off here, where every chain passes them anyway (the decode work of this kernel does not depend on
        // the piece size; its maps do not get finer).  One loop per sub-piece: the chains of the 32 offsets pass a boundary within
        // one token of each other.  It still costs 25 % of this kernel (315 -> 393 us at 16 MiB, against -237 us in k_par_tokens);
        // a per-token test with the stores under it cost 14 %, a shifting 64-bit bit buffer in place of bits_at 27 %.
        const uint32_t nsub = a.sub, fb = a.chbits / nsub;
        for (uint32_t sb = 1u; sb <= nsub; sb++) {
            const uint32_t bound = b_c + sb * fb;            // (the last one: the end of the piece)
            while (ballot64(run && pos < bound) != 0ull) {
                if (run && pos < bound) token();
            }
            if (have && sb < nsub) {
                const uint32_t m = (c * (nsub - 1u) + (sb - 1u)) * 32u + e;
                a.mexit8[m] = run ? (uint8_t)(pos - bound) : (uint8_t)X_EOB;      // (X_EOB: the chain ended in front of this boundary)
                a.mnb32[m] = nbytes;
            }
        }
        if (run) exitc = pos - end;
    } else {
        while (ballot64(run) != 0ull) {
            if (run) {
                token();
                if (run && pos >= end) { exitc = pos - end; run = false; }
            }
        }
    }
    if (have) { a.exit8[c * 32u + e] = (uint8_t)exitc; a.nb32[c * 32u + e] = nbytes; }
}

// ---- 1'. the same maps with the 32-fold work only where it is needed (round 3).  Chains that start at different offsets of a piece fall
// into step at the first token boundary they share -- about one chance in nine per token -- and are ONE chain from there on.  So:
//   k_par_head     all 32 offsets of a piece decode its first HEAD_BITS bits only (6 % of k_par_spec's work at 512-byte pieces); the
//                  lanes that stand at the same bit afterwards are one chain: its first lane appends it to a global chain list;
//   k_par_tail     one LANE per listed chain decodes the rest of its piece (exit offset, byte counts, the sub-boundary maps);
//   k_par_resolve  every (piece, offset) takes its chain's results (+ its own bytes of the head) -- the arrays k_par_scan_* read.
// Nothing is assumed about the data: a stream whose chains never merge (a period of a few tokens) lists 32 chains per piece and costs
// what k_par_spec cost, plus the head.  Ordinary data lists two or three.
#ifndef HDLZ_HEAD_MAX
#define HDLZ_HEAD_MAX 256
#endif
constexpr uint32_t HEAD_MAX = HDLZ_HEAD_MAX;     // bits of a piece all 32 offsets decode (never beyond the piece's first sub-boundary: the tail records those)
// (512 / 1024 bits: fewer chains are left for the tail, but 32 lanes wide costs more than it saves -- 16 MiB 0.524 -> 0.538 / 0.564 ms,
//  256 MiB 4.07 -> 4.14 / 4.40, 256 x 1 MiB 4.52 -> 4.57 / 4.79)
struct Chains {
    uint32_t* rep;              // [nchunks][32]  the chain of (piece, offset); NONE: ended inside the head (its maps are final)
    uint32_t* cpos;             // [chains]  bit position at which the chain stands behind the head
    uint8_t* cexit;             // [chains]  exit offset of the piece / X_EOB / X_BAD
    uint32_t* cnb;              // [chains]  bytes from cpos to the end of the piece
    uint8_t* cmx;               // [chains][SUB-1]  the sub-boundary maps, as in ParArgs::mexit8 / mnb32 (bytes from cpos on)
    uint32_t* cmn;
};
__device__ __forceinline__ Chains of_stream(Chains ch, size_t ws_stride) {
    const size_t d = (size_t)blockIdx.y * ws_stride;
    shift_ptr(ch.rep, d); shift_ptr(ch.cpos, d); shift_ptr(ch.cexit, d); shift_ptr(ch.cnb, d); shift_ptr(ch.cmx, d); shift_ptr(ch.cmn, d);
    return ch;
}

// one token of a chain, lengths and byte counts only, branch-free (see k_par_spec); x = the next 32 stream bits
__device__ __forceinline__ void spec_token(uint32_t x, const uint32_t* lit, const uint32_t* dst, uint32_t& pos, uint32_t& nbytes,
                                           uint32_t& exitc, bool& run, uint32_t& used_out, uint32_t emode) {
    const uint32_t e0 = lit[x & 511u];
    const uint32_t nb = e0 & 15u, type = (e0 >> 13) & 3u, leb = (e0 >> 25) & 7u, lbase = (e0 >> 16) & 0x1FFu;
    const uint32_t y = x >> nb;
    const uint32_t tl = lbase + __builtin_amdgcn_ubfe(y, 0u, leb);
    // the distance code's extra bits in closed form (RFC1951: codes 0..3 none, then (code >> 1) - 1; 30 and 31 do not exist) instead of
    // the dst[] look-up: a second DEPENDENT LDS round trip on every token of a chain that is one lane's serial latency
    const uint32_t dc = __builtin_bitreverse32(__builtin_amdgcn_ubfe(y, leb, 5u)) >> 27;
    const uint32_t deb = max(dc >> 1, 1u) - 1u;
    const bool islit = type == (uint32_t)T_LIT;
    const bool bad = (nb == 0u) | (type == (uint32_t)T_BAD) | ((type == (uint32_t)T_LEN) & (dc >= 30u));
    const bool iseob = type == (uint32_t)T_EOB;
    const bool on = iseob & eob_goes_on(emode, (x >> 7) & 7u);        // the end-of-block code + the next fixed block's header: 10 bits, no bytes
    const bool eob = iseob & !on;
    const uint32_t used = on ? 10u : islit ? nb : nb + leb + 5u + deb;
    const uint32_t made = on ? 0u : islit ? 1u : tl;
    const bool adv = !(bad | eob);
    exitc = bad ? X_BAD : eob ? X_EOB : exitc;
    used_out = adv ? used : 0u;
    pos += used_out;
    nbytes += adv ? made : 0u;
    run = adv;
}

constexpr uint32_t HEAD_WAVES = 8;            // waves per workgroup of k_par_head: ONE atomic on the chain counter per workgroup (one per
                                              // wave -- 9.4 k same-address atomics at 16 MiB -- serialised in L2: 121 us for 25 us of work)
__global__ __launch_bounds__(64 * HEAD_WAVES) void k_par_head(ParArgs a_, Chains ch_) {
    const ParArgs a = of_stream(a_);
    const Chains ch = of_stream(ch_, a_.ws_stride);
    if (!starts_fixed(a)) return;
    const uint32_t emode = eob_mode(a);
    __shared__ uint32_t lit[512], dst[32], win[HEAD_WAVES][2][HEAD_MAX / 32 + 8], first[HEAD_WAVES][2][32], slotof[HEAD_WAVES][2][32];
    __shared__ uint32_t wcount[HEAD_WAVES], wbase[HEAD_WAVES];
    const uint32_t tid = threadIdx.x, wv = tid >> 6, lane = tid & 63u, half = lane >> 5, e = lane & 31u;
    fill_tables(lit, dst, tid, 64u * HEAD_WAVES);
    first[wv][half][e] = 0xFFFFFFFFu;
    const uint32_t c = (blockIdx.x * HEAD_WAVES + wv) * 2u + half;
    const bool have = c < a.nchunks;
    const uint32_t HEAD_BITS = min(HEAD_MAX, a.chbits / a.sub);
    const uint32_t b_c = FIRST_BIT + c * a.chbits, hb = b_c + HEAD_BITS;
    if (have) stage_window(win[wv][half], a.z, a.zn, b_c, HEAD_BITS, e, 32u);
    __syncthreads();
    uint32_t pos = b_c + e, nbytes = 0, exitc = 0, used;
    bool run = have;
    const uint32_t bit0 = 8u * ((b_c >> 3) & ~3u);
    const uint32_t* w_ = win[wv][half];
    while (ballot64(run && pos < hb) != 0ull) {
        if (run && pos < hb) {
            const uint32_t rel = pos - bit0;
            spec_token(__builtin_amdgcn_alignbit(w_[(rel >> 5) + 1u], w_[rel >> 5], rel), lit, dst, pos, nbytes, exitc, run, used, emode);
        }
    }
    const uint32_t key = (pos - hb) & 31u;                 // (a token is at most 32 bits long)
    if (run) atomicMin(&first[wv][half][key], e);
    __syncthreads();
    const bool leader = run && first[wv][half][key] == e;
    const uint64_t lm = ballot64(leader);
    if (lane == 0u) wcount[wv] = (uint32_t)__popcll(lm);
    __syncthreads();
    if (tid == 0u) {
        uint32_t tot = 0;
        for (uint32_t k = 0; k < HEAD_WAVES; k++) { wbase[k] = tot; tot += wcount[k]; }
        const uint32_t base = tot ? atomicAdd(&a.ctl[C_NCHAIN], tot) : 0u;
        for (uint32_t k = 0; k < HEAD_WAVES; k++) wbase[k] += base;
    }
    __syncthreads();
    if (leader) {
        const uint32_t slot = wbase[wv] + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
        slotof[wv][half][key] = slot;
        ch.cpos[slot] = pos;
    }
    __syncthreads();
    if (have) {
        const uint32_t t = c * 32u + e;
        a.nb32[t] = nbytes;                                // the bytes of the head; k_par_resolve adds the chain's
        if (run) ch.rep[t] = slotof[wv][half][key];
        else {
            ch.rep[t] = NONE;
            a.exit8[t] = (uint8_t)exitc;
            for (uint32_t sb = 1u; sb < a.sub; sb++) a.mexit8[(c * (a.sub - 1u) + (sb - 1u)) * 32u + e] = (uint8_t)X_EOB;
        }
    }
}

// (The lane's bits copied into an LDS row of its own first -- no global load, hence no `s_waitcnt vmcnt` in the loop -- changed nothing
// here: 138 -> 143 us at 16 MiB.  The ~800 cycles per token of a wave that is alone on its SIMD are the DEPENDENT issue of ~35
// instructions and two LDS round trips, not memory; k_par_tokens, which also stores, gained 10 % from the same rows and keeps them.)
__global__ __launch_bounds__(64) void k_par_tail(ParArgs a_, Chains ch_) {
    const ParArgs a = of_stream(a_);
    const Chains ch = of_stream(ch_, a_.ws_stride);
    if (!starts_fixed(a)) return;
    const uint32_t emode = eob_mode(a);
    __shared__ uint32_t lit[512], dst[32];
    const uint32_t lane = threadIdx.x, i = blockIdx.x * 64u + lane;
    const uint32_t nchain = a.ctl[C_NCHAIN];
    if (blockIdx.x * 64u >= nchain) return;
    fill_tables(lit, dst, lane, 64u);
    __syncthreads();
    const bool have = i < nchain;
    uint32_t pos = have ? ch.cpos[i] : FIRST_BIT;
    const uint32_t c = (pos - FIRST_BIT) / a.chbits;
    const uint32_t b_c = FIRST_BIT + c * a.chbits, end = b_c + a.chbits;
    const uint32_t nsub = a.sub, fb = a.chbits / nsub;
    // bit reader straight from the stream, as in k_par_tokens<false> (the lanes of a wave stand in different pieces)
    uint32_t ip = (pos >> 3) & ~3u, bc = 64u - (pos - 8u * ip);
    uint64_t bb = have ? (((uint64_t)tok::load32(a.z, ip + 4u, a.zn) << 32) | tok::load32(a.z, ip, a.zn)) >> (pos - 8u * ip) : 0ull;
    ip += 8u;
    uint32_t nxt = have ? tok::load32(a.z, ip, a.zn) : 0u;
    uint32_t nbytes = 0, exitc = 0, used;
    bool run = have;
    for (uint32_t sb = 1u; sb <= nsub; sb++) {
        const uint32_t bound = b_c + sb * fb;
        while (ballot64(run && pos < bound) != 0ull) {
            if (run && pos < bound) {
                if (bc <= 32u) { bb |= (uint64_t)nxt << bc; bc += 32u; ip += 4u; nxt = tok::load32(a.z, ip, a.zn); }
                spec_token((uint32_t)bb, lit, dst, pos, nbytes, exitc, run, used, emode);
                bb >>= used; bc -= used;
            }
        }
        if (have && sb < nsub) {
            ch.cmx[i * (nsub - 1u) + (sb - 1u)] = run ? (uint8_t)(pos - bound) : (uint8_t)X_EOB;
            ch.cmn[i * (nsub - 1u) + (sb - 1u)] = nbytes;
        }
    }
    if (have) { ch.cexit[i] = run ? (uint8_t)(pos - end) : (uint8_t)exitc; ch.cnb[i] = nbytes; }
}

__global__ __launch_bounds__(256) void k_par_resolve(ParArgs a_, Chains ch_) {
    const ParArgs a = of_stream(a_);
    const Chains ch = of_stream(ch_, a_.ws_stride);
    if (!starts_fixed(a)) return;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= a.nchunks * 32u) return;
    const uint32_t r = ch.rep[t];
    if (r == NONE) return;
    const uint32_t c = t >> 5, e = t & 31u, nsub = a.sub, hnb = a.nb32[t];
    a.exit8[t] = ch.cexit[r];
    a.nb32[t] = hnb + ch.cnb[r];
    for (uint32_t sb = 1u; sb < nsub; sb++) {
        const uint32_t m = (c * (nsub - 1u) + (sb - 1u)) * 32u + e;
        a.mexit8[m] = ch.cmx[r * (nsub - 1u) + (sb - 1u)];
        a.mnb32[m] = hnb + ch.cmn[r * (nsub - 1u) + (sb - 1u)];
    }
}

// ---- 2. the true chain through the pieces.  The maps compose: (a) every group of 64 pieces is walked from all 32 entry offsets at
// once, (b) one wave walks the groups from the stream's first token on, (c) every group walks its pieces again from its true
// entry offset and writes their entry offsets and output positions.  (One wave over all the pieces: 115 ns per piece, more than
// half of the whole path.)
constexpr uint32_t GROUP = 64;
struct GroupLds {
    uint8_t ex[GROUP * 32];
    uint32_t nb[GROUP * 32];
};
__device__ __forceinline__ uint32_t stage_group(GroupLds& L, const ParArgs& a, uint32_t g, uint32_t lane) {
    const uint32_t base = g * GROUP, cnt = min(GROUP, a.nchunks - base);
    const uint32_t* ex32 = reinterpret_cast<const uint32_t*>(a.exit8 + (size_t)base * 32u);      // (256-byte aligned scratch)
    for (uint32_t k = lane; k < cnt * 8u; k += 64u) reinterpret_cast<uint32_t*>(L.ex)[k] = ex32[k];
    for (uint32_t k = lane; k < cnt * 32u; k += 64u) L.nb[k] = a.nb32[(size_t)base * 32u + k];
    __syncthreads();
    return cnt;
}
__global__ __launch_bounds__(64) void k_par_scan_groups(ParArgs a_) {
    const ParArgs a = of_stream(a_);
    if (!starts_fixed(a)) return;
    __shared__ GroupLds L;
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t cnt = stage_group(L, a, g, lane);
    if (lane < 32u) {
        uint32_t e = lane, acc = 0, x = 0, j = 0;
        for (; j < cnt; j++) {
            x = L.ex[j * 32u + e];
            acc += L.nb[j * 32u + e];
            if (x & (X_EOB | X_BAD)) break;
            e = x;
        }
        a.gexit8[g * 32u + lane] = (uint8_t)x;             // exit offset of the group, or where and why the chain ends in it
        a.gstop8[g * 32u + lane] = (uint8_t)j;
        a.gnb32[g * 32u + lane] = acc;                     // (a group makes < 64 * 175 KB)
    }
}
__global__ __launch_bounds__(256) void k_par_scan_top(ParArgs a_) {
    const ParArgs a = of_stream(a_);
    __shared__ GroupLds L;                                 // (the maps of 64 GROUPS at a time, staged like a group's pieces)
    __shared__ __attribute__((aligned(16))) uint8_t stp[GROUP * 32];
    __shared__ uint8_t ent[GROUP];
    __shared__ uint32_t op[GROUP];
    __shared__ uint32_t sh_stop, sh_e, sh_nused, sh_cnt;                  // (WHY the chain of pieces ended is of no consequence since round 6: k_par_ends)
    __shared__ uint64_t sh_acc;
    __shared__ uint8_t pathb[8][32][8], segmap[8][32], segstop[8][32], segent[8];
    const uint32_t lane = threadIdx.x;                     // (256 threads: staging, and 8 segments x 32 entry offsets for the walk)
    // the stream must be one fixed block (or be read as one: the DYNAMIC=False / ONEBLOCK builds)
    if (!starts_fixed(a)) { if (lane == 0u) { a.ctl[C_FALLBACK] = 1u; a.ctl[C_NOTFIXED] = 1u; } return; }
    const uint32_t ngroups = (a.nchunks + GROUP - 1u) / GROUP;
    if (lane == 0u) { sh_stop = 0; sh_e = 0; sh_nused = 0; sh_acc = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < ngroups && sh_stop == 0u; base += GROUP) {
        const uint32_t cnt = min(GROUP, ngroups - base);
        for (uint32_t k = lane; k < cnt * 8u; k += 256u) {                 // (dwords: byte loads made the staging the longest part of this kernel)
            reinterpret_cast<uint32_t*>(L.ex)[k] = reinterpret_cast<const uint32_t*>(a.gexit8 + (size_t)base * 32u)[k];
            reinterpret_cast<uint32_t*>(stp)[k] = reinterpret_cast<const uint32_t*>(a.gstop8 + (size_t)base * 32u)[k];
        }
        for (uint32_t k = lane; k < cnt * 32u; k += 256u) L.nb[k] = a.gnb32[(size_t)base * 32u + k];
        if (lane < 8u) segent[lane] = 0;
        __syncthreads();
        // the chain through the groups is serial, so ONLY the offset look-up is on it (one dependent LDS read per group; with the
        // byte counts on the same chain it was 250 ns per group: 74 of 1107 us at 16 MiB, 587 us at 256 MiB); the output positions
        // are a prefix sum of the counts the chain picked -- one group per lane, a wave scan
        // ... and it is walked in 8 segments of 8 groups from ALL 32 entry offsets at once (the maps compose), then one thread
        // goes through the 8 segment maps: 16 dependent steps per batch instead of 64 (round 3: 43 -> 20 us at 16 MiB)
        {
            const uint32_t sg = lane >> 5, e0 = lane & 31u, g0 = sg * 8u;
            uint32_t e = e0, stopk = 0xFFu;
            for (uint32_t k = 0; k < 8u && g0 + k < cnt; k++) {
                const uint32_t x = L.ex[(g0 + k) * 32u + e];
                pathb[sg][e0][k] = (uint8_t)e;
                if (x & (X_EOB | X_BAD)) { stopk = k; e = x; break; }
                e = x;
            }
            segmap[sg][e0] = (uint8_t)e; segstop[sg][e0] = (uint8_t)stopk;
        }
        __syncthreads();
        if (lane == 0u) {
            uint32_t e = sh_e, walked = 0;
            for (uint32_t sg = 0; sg < 8u && sg * 8u < cnt; sg++) {
                segent[sg] = (uint8_t)e;
                const uint32_t m = segmap[sg][e], sk = segstop[sg][e];
                if (sk != 0xFFu) {
                    const uint32_t j = sg * 8u + sk;
                    sh_stop = 1u; sh_nused = (base + j) * GROUP + stp[j * 32u + pathb[sg][e][sk]] + 1u;
                    walked = j + 1u;
                    break;
                }
                e = m;
                walked = min(cnt, sg * 8u + 8u);
            }
            sh_e = e; sh_cnt = walked;                      // groups walked in this batch (the one the chain ends in included)
        }
        __syncthreads();
        if (lane < 64u) ent[lane] = pathb[lane >> 3][segent[lane >> 3]][lane & 7u];
        __syncthreads();
        if (lane < 64u)
        {
            const uint32_t walked = sh_cnt;
            const uint64_t mine = lane < walked ? (uint64_t)L.nb[lane * 32u + ent[lane]] : 0ull;
            uint64_t incl = mine;
#pragma unroll
            for (int ofs = 1; ofs < 64; ofs <<= 1) {
                const uint64_t o = ((uint64_t)(uint32_t)__shfl_up((int)(incl >> 32), ofs, 64) << 32) | (uint32_t)__shfl_up((int)(uint32_t)incl, ofs, 64);
                if (lane >= (uint32_t)ofs) incl += o;
            }
            const uint64_t before = sh_acc + incl - mine;
            if (lane < walked) op[lane] = (uint32_t)before;
            const bool over = lane < walked && before + mine > 0xFFFFFFFFull;
            if (ballot64(over) != 0ull && lane == 0u) {            // positions beyond 2^32 (only behind the true end, or the stream is not decodable here)
                const uint32_t fo = (uint32_t)__builtin_ctzll(ballot64(over));
                if (sh_stop == 0u || (base + fo) * GROUP < sh_nused) { sh_stop = 1u; sh_nused = (base + fo) * GROUP; }
            }
            if (lane == 63u) sh_acc += incl;
        }
        __syncthreads();
        if (lane < cnt) { a.gentry8[base + lane] = ent[lane]; a.gopos[base + lane] = op[lane]; }
        __syncthreads();
    }
    if (lane == 0u) {
        // (round 6) where the chain of pieces ends is NOT where the stream ends: it runs through the true end into the trailer (see
        // starts_fixed); the pieces up to there are decoded, k_par_ends finds the true end among the end-of-block codes and gives the
        // verdict.  A chain that never stopped (garbage up to the last piece): all pieces.
        a.ctl[C_NUSED] = sh_stop != 0u ? sh_nused : a.nchunks;
        a.ctl[C_TOTAL] = 0u;
    }
}
__global__ __launch_bounds__(64) void k_par_scan_pieces(ParArgs a_, uint8_t* fentry8, uint32_t* fopos) {
    const ParArgs a = of_stream(a_);
    shift_ptr(fentry8, (size_t)blockIdx.y * a_.ws_stride);
    shift_ptr(fopos, (size_t)blockIdx.y * a_.ws_stride);
    __shared__ GroupLds L;
    __shared__ uint8_t ent[GROUP];
    __shared__ uint32_t op[GROUP];
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    if (a.ctl[C_FALLBACK] != 0u || g * GROUP >= a.ctl[C_NUSED]) return;
    const uint32_t cnt = stage_group(L, a, g, lane);
    if (lane == 0u) {
        uint32_t e = a.gentry8[g], acc = a.gopos[g];
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t x = L.ex[j * 32u + e];
            ent[j] = (uint8_t)e; op[j] = acc;
            acc += L.nb[j * 32u + e];
            if (x & (X_EOB | X_BAD)) break;                // (the pieces behind it are not used)
            e = x;
        }
    }
    __syncthreads();
    if (lane < cnt) { a.entry8[g * GROUP + lane] = ent[lane]; a.opos[g * GROUP + lane] = op[lane]; }
    // ---- 2b. entry offsets and output positions of the SUB-pieces (what k_par_tokens works on), from the maps k_par_spec took at the
    // sub-boundaries and the true entry offset of the piece (one lane per piece; a launch of its own cost 9 us)
    const uint32_t c = g * GROUP + lane;
    if (lane < cnt && c < a.ctl[C_NUSED]) {
        const uint32_t nsub = a.sub, e = ent[lane], P = op[lane], f = c * nsub;
        fentry8[f] = (uint8_t)e; fopos[f] = P;
        uint32_t reached = f;
        for (uint32_t sb = 1; sb < nsub; sb++) {
            const uint32_t m = (c * (nsub - 1u) + (sb - 1u)) * 32u + e;
            const uint32_t x = a.mexit8[m];
            if (x & (X_EOB | X_BAD)) break;               // the chain ends in front of this boundary
            fentry8[f + sb] = (uint8_t)x; fopos[f + sb] = P + a.mnb32[m];
            reached = f + sb;
        }
        atomicMax(&a.ctl[C_FNUSED], reached + 1u);
    }
}

// ---- 3a. the real decode, tokens only: one LANE per piece (64 pieces per wave), the reference's checks
// in the reference's order, the tokens into a list per piece.  (One WAVE per piece decoding and copying kept the CU's one scalar unit
// 89 % busy -- the token chain of a piece is wave-uniform, 115 scalar instructions per token: 8.2 of 13.7 ms at 256 MiB.)
template <bool ROWS>
__global__ __launch_bounds__(64) void k_par_tokens(ParArgs a_) {
    const ParArgs a = of_stream(a_);
    __shared__ uint32_t lit[512], dst[32];
    extern __shared__ uint32_t rows_[];              // ROWS: the stream bits a lane will read, copied into an LDS row of its own first (dword j of lane l
                                                     // at j * 64 + l: every lane in its own bank) -- no global load and no `s_waitcnt vmcnt` in the loop
    const uint32_t lane = threadIdx.x;
    if (a.ctl[C_FALLBACK] != 0u) return;
    const uint32_t nused = a.ctl[a.cnu], c0 = blockIdx.x * 64u;
    if (c0 >= nused) return;
    fill_tables(lit, dst, lane, 64u);
    __syncthreads();
    const uint32_t c = c0 + lane;
    const bool have = c < nused;
    const uint32_t emode = eob_mode(a);
    const int32_t isize = (int32_t)a.zn - 1;              // deflate.py:605
    const uint32_t obsize = a.obsize ? a.obsize : 32768u;
    const uint32_t b_c = FIRST_BIT + c * a.chbits, end = b_c + a.chbits;
    uint32_t pos = have ? b_c + a.entry8[c] : 0u, P = have ? a.opos[c] : 0u;
    uint32_t* tk = a.tokens + (size_t)c * a.tcap;
    // bit reader straight from the stream (64 KB of LDS windows per wave left two waves per CU and every LDS / store round trip
    // exposed: 2.9 ms at 256 MiB): bb holds bc valid bits from `pos` on, `nxt` is the dword behind them, requested a refill ahead
    uint32_t ip = (pos >> 3) & ~3u, bc = 64u - (pos - 8u * ip), n = 0, jn = 2u;
    if (ROWS) {
        const uint32_t ndw = have ? ((end + 64u - 8u * ip) >> 5) + 2u : 0u;
        for (uint32_t j = 0; j < ndw; j++) rows_[j * 64u + lane] = tok::load32(a.z, ip + 4u * j, a.zn);
        __syncthreads();
    }
    uint64_t bb;
    uint32_t nxt = 0;
    if (ROWS) bb = have ? (((uint64_t)rows_[64u + lane] << 32) | rows_[lane]) >> (pos - 8u * ip) : 0ull;
    else {
        bb = have ? (((uint64_t)tok::load32(a.z, ip + 4u, a.zn) << 32) | tok::load32(a.z, ip, a.zn)) >> (pos - 8u * ip) : 0ull;
        ip += 8u;
        nxt = have ? tok::load32(a.z, ip, a.zn) : 0u;
    }
    // One token per iteration, branch-free: the 64 pieces of a wave stand at different kinds of tokens, so a ladder of branches runs
    // every path anyway.  The reference's checks (deflate.py:1409-1445, :1519-1591, :1600) are all evaluated; WHICH of them failed does
    // not matter here -- any failure hands the stream to the serial decoder, which reports the reference's status in the reference's
    // order.  A token is at most 32 bits long: the low dword of the bit buffer is all a step looks at.
    // (round 6) an end-of-block code is LISTED -- bit, output position, sub-piece, token index, the three header bits behind it -- and, when
    // another fixed block follows, passed like a token of 10 bits; a failed check is recorded per sub-piece (nfail: the token index), not
    // as the stream's verdict: this lane may be decoding the trailer behind the true end.  k_par_ends sorts it out.
    bool run = have;
    uint32_t nfail = NONE;
    uint4 q = make_uint4(0, 0, 0, 0);                        // the last (up to) four tokens
    while (ballot64(run) != 0ull) {
        if (run) {
            if (bc <= 32u) {
                if (ROWS) { bb |= (uint64_t)rows_[jn * 64u + lane] << bc; jn++; }
                else { bb |= (uint64_t)nxt << bc; ip += 4u; nxt = tok::load32(a.z, ip, a.zn); }
                bc += 32u;
            }
            const uint32_t x = (uint32_t)bb;
            const uint32_t e0 = lit[x & 511u];
            const uint32_t nb = e0 & 15u, code = (e0 >> 4) & 0x1FFu, leb = (e0 >> 25) & 7u, lbase = (e0 >> 16) & 0x1FFu;
            const uint32_t y = x >> nb;
            const uint32_t tl = lbase + __builtin_amdgcn_ubfe(y, 0u, leb);
            // (the distance table in closed form, RFC1951 3.2.5: no second dependent LDS look-up on the lane's serial chain)
            const uint32_t dc = __builtin_bitreverse32(__builtin_amdgcn_ubfe(y, leb, 5u)) >> 27;
            const uint32_t deb = max(dc >> 1, 1u) - 1u;
            const uint32_t D = (dc < 4u ? dc + 1u : 1u + ((2u + (dc & 1u)) << deb)) + __builtin_amdgcn_ubfe(y, leb + 5u, deb);
            const bool islit = code < 256u, iseob = code == 256u, islen = code > 256u;
            const uint32_t used = islen ? nb + leb + 5u + deb : nb;
            const uint32_t made = islit ? 1u : tl;
            bool f = (nb < 1u) | ((int32_t)((pos + nb) >> 3) > isize - 3);                        // zero leaf; NO EOF (deflate.py:1535-1539)
            f |= !iseob & ((uint64_t)P + made > a.cap);                                           // capacity
            f |= islen & ((code - 257u >= 29u) | (dc >= 30u) | (D > P) | (D > obsize) |           // BAD_SYMBOL, BAD_DISTANCE, D8
                          ((int32_t)((pos + used) >> 3) >= isize - 2));                           // COPY hold (deflate.py:1600)
            const bool go = !(f | iseob);
            bool stop = f;
            // (the token list is written FOUR tokens at a time: a store per token and lane is 64 requests of 4 bytes to 64 different lines
            //  per wave instruction -- 0.80 of the kernel's 1.39 ms at 256 MiB went with the stores)
            if (go) {
                const uint32_t w = islit ? (TOK_LIT | code) : (tl | (D << 9)), k4 = n & 3u;
                q.x = k4 == 0u ? w : q.x; q.y = k4 == 1u ? w : q.y; q.z = k4 == 2u ? w : q.z; q.w = k4 == 3u ? w : q.w;
                if (k4 == 3u) *reinterpret_cast<uint4*>(tk + (n & ~3u)) = q;
                n++; P += made; pos += used; bb >>= used; bc -= used;
            }
            else if (!f) {                                                                        // an end-of-block code (D6)
                const uint32_t hdr3 = (x >> 7) & 7u;
                const bool on = eob_goes_on(emode, hdr3);
                const uint32_t k = atomicAdd(&a.ctl[C_NCROSS], 1u);
                if (k < MAXCROSS) { uint32_t* cr = a.cross + 4u * k; cr[0] = pos; cr[1] = P; cr[2] = c; cr[3] = n | (hdr3 << 24) | (on ? 1u << 27 : 0u); }
                if (on) { pos += 10u; bb >>= 10; bc -= 10u; }
                else stop = true;
            }
            if (f) { nfail = n; atomicMax(&a.ctl[C_FAILF], ~c); }                                  // (rare: the lowest failing sub-piece, as ~index: 0 = none)
            run = !stop && pos < end;
        }
    }
    if (have && (n & 3u) != 0u) *reinterpret_cast<uint4*>(tk + (n & ~3u)) = q;          // (the rest of the last four: tcap is a multiple of 4)
    if (have) { a.ntok[c] = n; a.nfail[c] = nfail; }
}

// ---- 3a'. the true end of the stream: the first end-of-block code, in stream order, that ends a block whose BFINAL was set.  One
// workgroup: the listed codes ranked by bit position, one thread walks them with the BFINAL state (the stream's first header, then the
// header behind every code that is passed); everything behind the end is cut off (pieces, sub-pieces, the end sub-piece's tokens); a
// check that failed IN FRONT of the end -- or a block of another type, or no end at all -- is the fallback
__global__ __launch_bounds__(256) void k_par_ends(ParArgs a_) {
    const ParArgs a = of_stream(a_);
    __shared__ uint32_t spos[MAXCROSS], sw3[MAXCROSS];
    __shared__ uint32_t s_end, s_bad;
    const uint32_t tid = threadIdx.x;
    if (a.ctl[C_FALLBACK] != 0u) return;
    const uint32_t n = a.ctl[C_NCROSS];
    if (n > MAXCROSS || n == 0u) { if (tid == 0u) a.ctl[C_FALLBACK] = 1u; return; }      // (more blocks than the list holds / no end-of-block code at all)
    if (tid == 0u) { s_end = NONE; s_bad = 0u; }
    for (uint32_t k = tid; k < n; k += 256u) {                      // rank by position (all distinct)
        const uint32_t p = a.cross[4u * k];
        uint32_t r = 0;
        for (uint32_t j = 0; j < n; j++) r += a.cross[4u * j] < p ? 1u : 0u;
        spos[r] = k; sw3[r] = a.cross[4u * k + 3u];
    }
    __syncthreads();
    if (tid == 0u) {
        uint32_t fin = ((uint32_t)a.z[2] & 1u) | ((a.flags & HDLZ_INFLATE_ONEBLOCK) ? 1u : 0u);
        for (uint32_t r = 0; r < n; r++) {
            const uint32_t w3 = sw3[r];
            if (fin) { s_end = spos[r]; break; }
            if (!((w3 >> 27) & 1u)) { s_bad = 1u; break; }          // the stream goes on with a block that is not fixed
            fin = (w3 >> 24) & 1u;
        }
    }
    __syncthreads();
    const uint32_t ke = s_end;
    if (ke == NONE || s_bad) { if (tid == 0u) a.ctl[C_FALLBACK] = 1u; return; }
    const uint32_t* cr = a.cross + 4u * ke;
    const uint32_t f_end = cr[2], n_end = cr[3] & 0xFFFFFFu, total = cr[1];
    if (f_end >= a.nchunks * a.sub || n_end > a.tcap) { if (tid == 0u) a.ctl[C_FALLBACK] = 1u; return; }      // (cannot happen)
    if (tid == 0u) {
        const uint32_t fm = a.ctl[C_FAILF], failmin = fm ? ~fm : NONE;          // the lowest sub-piece with a failed check
        if (failmin < f_end || a.nfail[f_end] < n_end || total > a.cap || total > a.srcn) { a.ctl[C_FALLBACK] = 1u; return; }
        a.ctl[C_TOTAL] = total;
        a.ctl[C_NUSED] = f_end / a.sub + 1u;
        a.ctl[C_FNUSED] = f_end + 1u;
        a.ntok[f_end] = n_end;
    }
}

// ---- 3b. the bytes: one wave per piece, 64 tokens at a time.  A wave scan gives every token its output position; then the batch is
// resolved BYTE-parallel: every output byte of the batch finds its token (binary search over the 64 start positions), a literal is
// a value, a copied byte is a POINTER to its source distance bytes back -- inside the batch an index into the LDS ring, in front of
// it a byte (or marker) the ring already holds -- and pointer jumping in LDS (ptr <- ptr[ptr], log2 of the longest in-batch chain
// rounds; an overlapping copy is simply a chain through its own bytes) turns every pointer into a value.  The history in front of
// the piece's own output is not there yet: a byte copied from there is a MARKER (src[p] = the absolute position it comes from), and
// markers travel like bytes.  (Round 2 walked the copies of a batch one after the other, lane-parallel inside a copy: ~500 cycles per
// copy, 257 of 879 us at 16 MiB, 3.4 of 10.7 ms at 256 MiB.)
constexpr uint32_t HRING = 1024;              // (2048 with a pointer array beside it: 14.8 KB of LDS per wave, 11 waves per CU, for a kernel that is one
                                              //  serial chain of LDS round trips per 64-byte slice)
constexpr uint32_t SPAN = 768;                // a batch ends with the token that takes its output beyond this many bytes
constexpr uint32_t HREACH = HRING - 128u;     // distances served from the ring (it is written a 64-byte slice at a time)
constexpr uint32_t P_RES = 0xFFu;             // in-slice pointer: the byte / marker is there
template <bool STRIDED>
__global__ __launch_bounds__(64) void k_par_emit(ParArgs a_) {
    const ParArgs a = of_stream(a_);
    __shared__ uint8_t hb[HRING];             // ring over the piece's output positions: byte ...
    __shared__ uint32_t hm[HRING];            // ... marker (NONE = the byte is there) ...
    __shared__ uint32_t tI[64];               // the batch's token words
    __shared__ uint64_t bmw[17];              // token starts, one bit per byte of the batch
    __shared__ uint32_t bpre[17];             // token starts in front of each 64-byte slice
    const uint32_t lane = threadIdx.x;
    if (a.ctl[C_FALLBACK] != 0u) return;
    const uint32_t nused_ = a.ctl[C_NUSED];
    // (a workgroup takes the pieces blockIdx.x, + gridDim.x, ..: batches of many streams are launched with fewer workgroups per stream than
    //  pieces -- a chain that is NOT a stream's turns every workgroup away at the door, and 1.7 million of those cost 0.7 ms beside the
    //  other chain's real work)
    for (uint32_t c0_ = blockIdx.x; c0_ < (STRIDED ? nused_ : gridDim.x); c0_ += gridDim.x) {
    const uint32_t c = STRIDED ? c0_ : xcd_item(c0_, gridDim.x);      // (one workgroup per item: the XCD-aware order)
    if (c >= nused_) { if constexpr (!STRIDED) break; else continue; }
    // a piece's tokens are the lists of its sub-pieces, one behind the other (the decode runs on sub-pieces; the emit does not:
    // the history in front of a wave's own output becomes markers, and with four times shorter pieces the marker passes doubled)
    const uint32_t nsub = a.sub, fnused = a.ctl[C_FNUSED];
    const uint32_t cstart = a.opos[c];
    uint8_t* out = a.out;
    uint32_t* src = a.srcA;
    uint32_t Pb = cstart, nmark = 0, mlast = 0;
    for (uint32_t sbi = 0; sbi < nsub; sbi++) {
        const uint32_t f = c * nsub + sbi;
        if (f >= fnused) break;
        const uint32_t n = a.ntok[f];
        const uint32_t* tk = a.tokens + (size_t)f * a.tcap;
        // (the piece's tokens staged in LDS first -- no global load in front of a batch -- cost more in occupancy than it saved: 245 -> 301 us)
        // (... and the next batch's tokens requested a batch ahead, kept in registers: nothing -- 256 MiB 3.24 -> 3.31 ms; the batches of a
        //  piece are not what the kernel waits for)
        for (uint32_t base = 0; base < n;) {
            const uint32_t k = base + lane;
            const uint32_t t = k < n ? tk[k] : 0u;
            const bool islit = (t >> 31) != 0u;
            const uint32_t len = k < n ? (islit ? 1u : (t & 511u)) : 0u;
            uint32_t incl = len;
#pragma unroll
            for (int ofs = 1; ofs < 64; ofs <<= 1) {
                const uint32_t o = __shfl_up(incl, ofs, 64);
                if (lane >= (uint32_t)ofs) incl += o;
            }
            // the batch: the leading tokens up to the first one that ends beyond SPAN bytes (that one included)
            const uint64_t over = ballot64(k < n && incl > SPAN);
            const uint32_t cnt = min(over != 0ull ? (uint32_t)__builtin_ctzll(over) + 1u : 64u, n - base);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)(cnt - 1u));      // bytes of the batch (<= SPAN + 258)
            // the token of a byte: token starts as a bitmap over the batch's bytes (one 64-bit word per 64-byte slice) + the
            // number of starts in front of each word -- two broadcast reads and a bit count per slice (a binary search over the 64
            // start positions was six DEPENDENT LDS reads per slice)
            // (one wave per workgroup: its LDS accesses execute in order, the compiler only has to keep them so -- __syncthreads() also
            //  waited, four times per batch, for the output stores of the batch before; time-neutral: 256 MiB 3.24 -> 3.23 ms)
            wave_lds_order();
            if (lane < 17u) bmw[lane] = 0ull;
            tI[lane] = t;
            wave_lds_order();
            if (lane < cnt) atomicOr(&bmw[(incl - len) >> 6], 1ull << ((incl - len) & 63u));
            wave_lds_order();
            {
                const uint32_t cw = lane < 17u ? (uint32_t)__popcll(bmw[lane]) : 0u;
                uint32_t ci = cw;
#pragma unroll
                for (int ofs = 1; ofs < 32; ofs <<= 1) {
                    const uint32_t o = __shfl_up(ci, ofs, 64);
                    if (lane >= (uint32_t)ofs) ci += o;
                }
                if (lane < 17u) bpre[lane] = ci - cw;
            }
            wave_lds_order();
            // ---- ONE ascending sweep over the slices: a copied byte is a POINTER to its source `distance` back.  Sources in earlier
            // slices (or in front of the batch) are final by now and are read as values; only chains INSIDE the slice -- distances
            // below 64 -- are followed (ptr <- ptr[ptr], log2 of the longest such chain rounds; an overlapping copy is a chain
            // through its own bytes).  (All slices per round, every round: 107 of the emit's 247 us.  The tokens of all slices looked up
            // in front of the sweep -- 17 register slots, independent reads -- made it slower: 212 -> 263 us.)
            for (uint32_t q0 = 0; q0 < total; q0 += 64u) {
                const uint32_t q = q0 + lane;
                const bool in = q < total;
                const uint64_t wbits = bmw[q0 >> 6];
                const uint32_t ti = bpre[q0 >> 6] + __builtin_amdgcn_mbcnt_hi((uint32_t)(wbits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wbits, 0u)) +
                                    (uint32_t)((wbits >> lane) & 1ull) - 1u;
                const uint32_t tw = tI[in ? ti : 0u];
                const uint32_t pabs = Pb + q, slot = pabs & (HRING - 1u);
                uint32_t v = tw & 255u, m = NONE, p = P_RES;
                bool farin = false;
                if (in && (tw >> 31) == 0u) {
                    const uint32_t D = tw >> 9, s = pabs - D;                // COPY (deflate.py:1627-1659): out[p] = out[p - D]
                    if (s >= Pb + q0) p = s - (Pb + q0);                     // inside this slice: the LANE that holds it, followed below
                    else if (s < cstart) { v = 0u; m = s; }                  // in front of the piece: a marker
                    else if (pabs - s <= HREACH) { v = hb[s & (HRING - 1u)]; m = hm[s & (HRING - 1u)]; }      // the ring has it (final)
                    else farin = true;                                       // the piece's own output beyond the ring
                }
                if (ballot64(farin) != 0ull) {
                    // (rare) those bytes were stored to out[] / src[] by this wave in earlier batches: order the stores before the
                    // loads (workgroup-scope release / acquire; wave-uniform branch)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (farin) { const uint32_t s = pabs - (tw >> 9); v = out[s]; m = src[s]; }
                }
                // chains inside the slice: registers and the LDS crossbar (ds_bpermute), no memory -- a lane takes over its source
                // lane's byte and marker once that lane is resolved, its pointer otherwise
                while (ballot64(p != P_RES) != 0ull) {
                    const uint32_t sl = p != P_RES ? p : lane;
                    const uint32_t pp = (uint32_t)__shfl((int)p, (int)sl, 64), vv = (uint32_t)__shfl((int)v, (int)sl, 64),
                                   mm = (uint32_t)__shfl((int)m, (int)sl, 64);
                    if (p != P_RES) {
                        if (pp == P_RES) { v = vv; m = mm; p = P_RES; }
                        else p = pp;
                    }
                }
                if (in) { hb[slot] = (uint8_t)v; hm[slot] = m; }
                if (in) {
                    out[pabs] = (uint8_t)v; src[pabs] = m;
                    nmark += m != NONE ? 1u : 0u;
                    mlast = m != NONE ? pabs + 1u : mlast;
                }
            }
            Pb += total;
            base += cnt;
        }
    }
    // how far into the piece its markers reach: the marker passes look at these bytes only
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) mlast = max(mlast, (uint32_t)__shfl_xor((int)mlast, ofs, 64));
    if (lane == 0u) a.mext[c] = mlast > cstart ? mlast - cstart : 0u;
    if (nmark) atomicAdd(&a.ctl[C_MARK], nmark);
    if constexpr (!STRIDED) break;                       // (one workgroup per piece: the form of single streams -- the loop cost it 60 % there)
    __syncthreads();
    }
}

// ---- 4. one pass of pointer jumping over the markers, IN PLACE (round 5: one marker word per output byte instead of two buffers -- half
// the scratch).  One workgroup per piece, over the bytes up to the piece's last marker only: a stream with short distances (what
// STARTC writes: CWINDOW bytes) has its markers in the first CWINDOW bytes of every piece.
// Why in place is exact.  src[p] is written by ONE thread (the one that owns p) and only ever replaced by a position further along
// p's own chain or by its ROOT, so whatever a reader from another piece sees -- the old word (the XCDs' L2s are not coherent inside
// a launch) or the new one -- is a valid ancestor.  The one thing a reader must not do is take the BYTE of a position that was resolved
// in the launch it runs in (that store may not be visible yet): a resolved position keeps ROOT | r, r = the byte of the emit its chain
// ends in -- final since the emit -- and a chain that arrives there takes out[r], not out[p].
// (HOPS, hdlz_inflate_par.h: 8 in round 2, 256 up to round 6 -- a pass that finds nothing left still costs a launch, 4.5 us.  But a lane
//  that follows a deep chain holds its whole wave, and the chains of zlib's level 1 (copies of copies of copies) ARE deep: the first
//  pass of a 16 MiB level-1 stream ran 1.4 .. 2.2 ms with 256 steps against 0.43 at level 6.  With 32 steps the rest is left to the next
//  pass, which finds most of it resolved by then: DNA / logs at level 1 3.58 / 2.55 -> 1.99 / 1.77 ms, a ramp with CWINDOW 256 2.66 ->
//  1.90, everything else within 2 % -- except pure runs (every byte of every piece a marker): 64 MiB of zeros 1.52 -> 1.60 ms, 256 MiB
//  3.96 -> 4.34; 16 steps: 1.90 / 1.67, but zeros 1.69 / 4.83.  profiles/r06_marker_hops_ab.txt)
constexpr uint32_t JUMP_GRID = 8192;          // workgroups of the passes behind the first one (they mostly find nothing left)
// (... of ALL streams of the launch together: 196 streams x 8192 workgroups that find nothing left took 49 us per pass)
__host__ inline uint32_t later_grid(uint32_t items, uint32_t nstr) {
    const uint32_t g = JUMP_GRID / (nstr ? nstr : 1u) < 64u ? 64u : JUMP_GRID / (nstr ? nstr : 1u);
    return items < g ? items : g;
}
constexpr uint32_t ROOT = 0x80000000u;        // src word: ROOT | r = resolved, the byte is out[r] (NONE: a byte of the emit, its own root); positions are < 2^30
__global__ __launch_bounds__(64) void k_par_jump(ParArgs a_, uint32_t pass) {
    const ParArgs a = of_stream(a_);
    if (a.ctl[C_FALLBACK] != 0u) return;
    if (a.ctl[pass == 0u ? (uint32_t)C_MARK : C_PASS0 + pass - 1u] == 0u) return;            // nothing left
    uint32_t* s = a.srcA;
    const uint32_t nused = a.ctl[C_NUSED];
    uint32_t left = 0;
    __shared__ unsigned long long memo[256];
    // (a workgroup takes the pieces blockIdx.x, + gridDim.x, ..: the passes behind the first one are launched with JUMP_GRID workgroups -- a
    //  pass that finds nothing left costs what its workgroups cost to start, 20 us for the 90 000 items of a 7 MB zlib stream; the FIRST
    //  pass keeps one workgroup per piece: capped, it ran 665 us instead of 407)
    for (uint32_t c = xcd_item(blockIdx.x, gridDim.x); c < nused; c += gridDim.x) {      // (XCD-aware order within every stride of the grid)
    const uint32_t ext = a.mext[c];
    if (ext == 0u) continue;
    const uint32_t p0 = a.opos[c];
    // What a walk from marker m found is kept for the piece's later bytes (a direct-mapped table in LDS, key and result in one 8-byte
    // entry -- written with ONE ds_write_b64 per lane: lanes of an instruction that hit the same slot are applied one after the other,
    // whole, so a slot always holds one lane's pair (a reader only takes an entry whose key is its own marker)): the markers of a piece share few targets -- the bytes in front of it --, and in a run or a short period (zeros: EVERY
    // byte of every piece is a marker of the byte in front of the piece) they all share one to `period` of them: 64 MiB of zeros
    // 10.1 -> 1.6 ms, 256 MiB 36.5 -> 4.1 ms (profiles/r05_single_stream_inflate.txt).  (One wave per piece: its LDS accesses are in program order.)
    for (uint32_t k = threadIdx.x; k < 256u; k += 64u) memo[k] = ~0ull;          // (no marker is NONE)
    wave_lds_order();
    // TWO slices of the piece per step, their loads issued side by side: a step is a chain of dependent loads (the marker, its source's
    // word, the byte), and the slices of a piece need nothing from each other (a marker points in front of the piece).  (Worth 3 % on
    // pure runs -- 256 MiB of zeros 4.52 -> 4.38 ms --, nothing on zlib streams: the pass is bound by the memory system's rate of
    // scattered words, not by a wave's latency)
    const uint32_t end = p0 + ext;
    for (uint32_t q = p0; q < end; q += 128u) {
        const uint32_t pA = q + threadIdx.x, pB = pA + 64u;
        uint32_t mA = pA < end ? s[pA] : NONE, mB = pB < end ? s[pB] : NONE;
        const bool actA = !(mA & ROOT), actB = !(mB & ROOT);          // (past the extent;) a byte: the emit's, or resolved by an earlier pass
        const uint32_t m0A = mA, m0B = mB;
        const unsigned long long eA = memo[m0A & 255u], eB = memo[m0B & 255u];
        const bool hitA = actA && (uint32_t)eA == m0A, hitB = actB && (uint32_t)eB == m0B;
        uint32_t rA = hitA ? (uint32_t)(eA >> 32) : mA, rB = hitB ? (uint32_t)(eB >> 32) : mB;
        bool runA = actA && !hitA, runB = actB && !hitB;
#pragma unroll 1
        for (uint32_t h = 0; h < HOPS && (runA || runB); h++) {
            const uint32_t m2A = runA ? s[mA] : 0u, m2B = runB ? s[mB] : 0u;
            if (runA) {
                if (m2A & ROOT) { rA = ROOT | (m2A == NONE ? mA : (m2A & ~ROOT)); runA = false; }
                else { mA = m2A; rA = m2A; }
            }
            if (runB) {
                if (m2B & ROOT) { rB = ROOT | (m2B == NONE ? mB : (m2B & ~ROOT)); runB = false; }
                else { mB = m2B; rB = m2B; }
            }
        }
        if (actA && !hitA) memo[m0A & 255u] = (unsigned long long)m0A | ((unsigned long long)rA << 32);
        if (actB && !hitB) memo[m0B & 255u] = (unsigned long long)m0B | ((unsigned long long)rB << 32);
        const bool doneA = actA && (rA & ROOT), doneB = actB && (rB & ROOT);
        const uint8_t vA = doneA ? a.out[rA & ~ROOT] : (uint8_t)0, vB = doneB ? a.out[rB & ~ROOT] : (uint8_t)0;      // (bytes of the emit: nobody writes them now)
        if (doneA) a.out[pA] = vA;
        if (doneB) a.out[pB] = vB;
        left += (actA && !doneA ? 1u : 0u) + (actB && !doneB ? 1u : 0u);
        if (actA) s[pA] = rA;
        if (actB) s[pB] = rB;
    }
    wave_lds_order();
    }
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) left += (uint32_t)__shfl_xor((int)left, ofs, 64);
    if (left && threadIdx.x == 0u) atomicAdd(&a.ctl[C_PASS0 + pass], left);
}

// ---- 5. the verdict: HDLZ_OK and the length, or the serial decoder's turn
// (`actl`: the control words of the chain for any block types, stream 0's; null when that chain was not launched)
__global__ __launch_bounds__(64) void k_par_finish(ParArgs a_, uint32_t passes, const uint32_t* actl, uint32_t apasses) {
    const ParArgs a = of_stream(a_);
    if (threadIdx.x != 0) return;
    const uint32_t left = a.ctl[C_MARK] == 0u ? 0u : a.ctl[C_PASS0 + passes - 1u];
    bool ok = a.ctl[C_FALLBACK] == 0u && left == 0u;
    uint32_t total = a.ctl[C_TOTAL];
    // (the other chain's result: the stream was its from the first header on, or it opened for a short fixed block in front of other types)
    if (!ok && actl) {
        actl = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(actl) + (size_t)blockIdx.y * a_.ws_stride);
        const uint32_t aleft = actl[C_MARK] == 0u ? 0u : actl[C_PASS0 + apasses - 1u];
        ok = actl[C_FALLBACK] == 0u && actl[C_OK] != 0u && aleft == 0u;
        total = actl[C_TOTAL];
    }
    if (ok) { a.out_len[0] = total; a.status[0] = HDLZ_OK; }
    else if (a.batch) { a.out_len[0] = 0u; a.status[0] = HDLZ_E_DYNAMIC_UNSUPPORTED; }      // several streams: the serial pass redoes the flagged ones
    a.ctl[C_OK] = ok ? 1u : 0u;
}

// the control words of every stream of the launch
__global__ __launch_bounds__(64) void k_par_zero(ParArgs a_) {
    const ParArgs a = of_stream(a_);
    if (threadIdx.x < C_WORDS) a.ctl[threadIdx.x] = 0u;
}
static_assert(C_WORDS <= 64, "k_par_zero: one wave");

// streams of a batch that the chain could not take (no scratch left for their group): flagged like the chain's own give-ups
__global__ __launch_bounds__(64) void k_par_flag_rest(uint32_t* out_len, uint32_t* status, uint32_t n) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i < n) { out_len[i] = 0u; status[i] = HDLZ_E_DYNAMIC_UNSUPPORTED; }
}

}  // namespace par

hipError_t par::par_launch_emit_jump(const ParArgs& p, uint32_t nitems, uint32_t passes, uint32_t nstr, hipStream_t stream) {
    const uint32_t gx = grid_cap(nitems, nstr);
    if (gx < nitems) hipLaunchKernelGGL(k_par_emit<true>, dim3(gx, nstr), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(k_par_emit<false>, dim3(gx, nstr), dim3(64), 0, stream, p);
    for (uint32_t j = 0; j < passes; j++) hipLaunchKernelGGL(k_par_jump, dim3(j == 0u ? gx : later_grid(gx, nstr), nstr), dim3(64), 0, stream, p, j);
    return hipGetLastError();
}

// a.nstreams streams of at least HDLZ_INFLATE_PAR_MIN bytes each (fixed pitch form): the parallel chain -- every kernel once, blockIdx.y =
// the stream --, then the serial decoder for what it gave up on: ONE stream: one wave, only if needed; several: the streams the chain
// flagged (status HDLZ_E_DYNAMIC_UNSUPPORTED, as pass 1 of the batch kernels flags them)
namespace par {
// the scratch of ONE stream of the launch: every array of the chain, each aligned to 256 bytes (so `stride` is a multiple of 256 and
// every stream's arrays are aligned like the first one's)
struct Layout {
    uint32_t chbits, nchunks, sub, ngroups;
    uint64_t cap64, srcn;
    size_t o_ctl, o_ex, o_nb, o_en, o_op, o_gx, o_gs, o_gn, o_ge, o_go, o_mx, o_mn, o_fe, o_fo, o_tk, o_nt, o_sa, o_me, o_rp, o_cp, o_cx, o_cn,
           o_cmx, o_cmn, o_cr, o_nf, o_any, any_bytes, stride;
    bool ok;
};
// (`batch`: the streams of the whole call when this launch is one GROUP of it -- the choice of chains is the call's, not the group's)
static Layout layout_of(uint32_t zn, uint32_t nstr, uint64_t out_pitch, uint32_t flags, uint32_t batch = 0) {
    Layout L;
    memset(&L, 0, sizeof(L));
    L.cap64 = out_pitch > 0xFFFFFE00ull ? 0xFFFFFE00ull : out_pitch;
    L.srcn = (uint64_t)zn * 172u + 258u;                   // a token of 13 bits makes at most 258 bytes
    if (L.srcn > L.cap64) L.srcn = L.cap64;
    if (L.srcn > (1ull << 30)) return L;                   // (4 GiB of scratch: leave it to the serial decoder)
    // (with the de-duplicated speculation: 1024-bit pieces 1.10 ms at 16 MiB -- markers, scans --, 4096 bits with 8 sub-pieces 0.67, these 0.64)
    // measured with the final kernels, 1 / 4 / 16 MiB of output: 1024-bit pieces 0.191 / 0.328 / 1.00 ms, 2048 bits 0.241 / 0.286 / 0.65, 4096 bits 0.317 / 0.361 / 0.571
    // (the piece size follows the bytes of the whole LAUNCH: 256 streams of 1 MiB are cut like one stream of 256 MiB, not like 256 small ones)
    const uint64_t ztot = (uint64_t)zn * nstr;
    const uint32_t chbits = ztot < (5u << 18) ? CH_BITS_MAX / 8u : ztot < (3u << 20) ? CH_BITS_MAX / 4u : ztot < (24u << 20) ? CH_BITS_MAX / 2u : CH_BITS_MAX;
    const uint32_t nchunks = (8u * zn - FIRST_BIT + chbits - 1u) / chbits;
    // the real decode ALWAYS runs on sub-pieces staged in LDS rows (up to round 6 only below 24 MiB: above, a lane of k_par_tokens<false>
    // decoded an 8192-bit piece -- ~900 tokens -- straight from memory: 1.30 of the 3.7 ms of 196 x 1 MiB).  With the token lists written
    // 16 bytes at a time (k_par_tokens) it wins at every size: 32 / 64 / 100 MiB of random bytes 1.35 / 1.77 / 2.40 -> 1.14 / 1.58 /
    // 2.14 ms, 64 MiB of families / text 1.70 / 1.68 -> 1.43 / 1.44, 256 MiB 4.54 -> 3.28, 256 x 1 MiB 5.16 -> 4.32, 1024 x 1 MiB
    // 19.6 -> 15.3.  (While a token was a 4-byte store the sub-boundary maps cost literal-heavy streams of 24 .. 128 MiB more than they
    // saved: 64 MiB of random bytes 1.87 -> 2.08; 4096-bit pieces instead of 8192 as well: worse throughout.)
    const uint32_t sub = SUB;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255u) & ~(size_t)255u; return o; };
    const uint32_t ngroups = (nchunks + GROUP - 1u) / GROUP;
    L.chbits = chbits; L.nchunks = nchunks; L.sub = sub; L.ngroups = ngroups;
    L.o_ctl = take(4u * C_WORDS); L.o_ex = take((size_t)nchunks * 32u); L.o_nb = take((size_t)nchunks * 128u);
    L.o_en = take(nchunks); L.o_op = take((size_t)nchunks * 4u); L.o_gx = take((size_t)ngroups * 32u); L.o_gs = take((size_t)ngroups * 32u);
    L.o_gn = take((size_t)ngroups * 128u); L.o_ge = take(ngroups); L.o_go = take((size_t)ngroups * 4u);
    L.o_mx = take((size_t)nchunks * (sub - 1u) * 32u); L.o_mn = take((size_t)nchunks * (sub - 1u) * 128u);
    L.o_fe = take((size_t)nchunks * sub); L.o_fo = take((size_t)nchunks * sub * 4u);
    L.o_tk = take((size_t)nchunks * sub * tmax_of(chbits / sub) * 4u); L.o_nt = take((size_t)nchunks * sub * 4u); L.o_sa = take((size_t)L.srcn * 4u);
    L.o_me = take((size_t)nchunks * 4u);
    L.o_rp = take((size_t)nchunks * 128u); L.o_cp = take((size_t)nchunks * 128u); L.o_cx = take((size_t)nchunks * 32u);
    L.o_cn = take((size_t)nchunks * 128u); L.o_cmx = take((size_t)nchunks * 32u * (sub - 1u)); L.o_cmn = take((size_t)nchunks * 128u * (sub - 1u));
    L.o_cr = take((size_t)MAXCROSS * 16u); L.o_nf = take((size_t)nchunks * sub * 4u);
    // the chain for any block types: its own arrays behind these (it shares the marker words: one of the two chains writes them)
    L.any_bytes = any_work_bytes(zn, out_pitch, flags, batch ? batch : nstr);
    L.o_any = take(L.any_bytes);
    L.stride = off;
    L.ok = true;
    return L;
}
}  // namespace par

// what the path asks for when all `nstreams` streams go through it at once (less: it runs them in groups, or not at all)
size_t inflate_par_work_bytes(uint32_t in_len, uint64_t nstreams, uint64_t out_pitch, uint32_t flags) {
    if (nstreams == 0 || nstreams > 65535u || in_len < HDLZ_INFLATE_PAR_MIN) return 0;
    const par::Layout L = par::layout_of(in_len, (uint32_t)nstreams, out_pitch, flags);
    if (!L.ok) return 0;
    constexpr size_t BUDGET = (size_t)8 << 30;
    const size_t all = L.stride * (size_t)nstreams;
    return all > BUDGET && nstreams > 1 ? (BUDGET / L.stride ? (BUDGET / L.stride) * L.stride : L.stride) : all;
}

// The two chains need nothing from each other (each looks at the stream's first block header itself), and for any given stream one of
// them is a dozen launches that return at once -- 40 .. 90 us of launch spacing if they queue up behind each other.  So the chain for any
// block types runs on a SIDE stream of the library's (one per host thread and device, created on first use): forked from the caller's
// stream behind everything that is queued there (the scratch may still be in use by the call before), joined in front of k_par_finish.
// Events are per call.  The pattern is capturable (the side stream joins the capture at the fork and leaves it at the join).
static hipStream_t side_stream() {
    static thread_local hipStream_t side[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
    if (!side[dev] && hipStreamCreateWithFlags(&side[dev], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); side[dev] = nullptr; }
    return side[dev];
}

static hipError_t launch_inflate_par_group(const InflateArgs& a, hipStream_t stream, bool* used, const Work& w, uint32_t batch);
hipError_t launch_inflate_par(const InflateArgs& a, hipStream_t stream, bool* used, const Work& w) {
    return launch_inflate_par_group(a, stream, used, w, a.nstreams > 65535u ? 0u : (uint32_t)a.nstreams);
}
static hipError_t launch_inflate_par_group(const InflateArgs& a, hipStream_t stream, bool* used, const Work& w, uint32_t batch) {
    using namespace par;
    *used = false;
    const uint32_t zn = a.in_len;
    const uint32_t nstr = (uint32_t)a.nstreams;
    if (a.nstreams == 0 || a.nstreams > 65535u) return hipSuccess;
    const Layout L = layout_of(zn, nstr, a.out_pitch, a.flags, batch);
    if (!L.ok) return hipSuccess;
    const uint64_t cap64 = L.cap64, srcn = L.srcn;
    const uint32_t chbits = L.chbits, nchunks = L.nchunks, sub = L.sub, ngroups = L.ngroups;
    const size_t o_ctl = L.o_ctl, o_ex = L.o_ex, o_nb = L.o_nb, o_en = L.o_en, o_op = L.o_op, o_gx = L.o_gx, o_gs = L.o_gs, o_gn = L.o_gn,
                 o_ge = L.o_ge, o_go = L.o_go, o_mx = L.o_mx, o_mn = L.o_mn, o_fe = L.o_fe, o_fo = L.o_fo, o_tk = L.o_tk, o_nt = L.o_nt,
                 o_sa = L.o_sa, o_me = L.o_me, o_rp = L.o_rp, o_cp = L.o_cp, o_cx = L.o_cx, o_cn = L.o_cn, o_cmx = L.o_cmx, o_cmn = L.o_cmn;
    uint8_t* ws = nullptr;
    const size_t stride = L.stride;
    // more scratch than the budget (4 GiB from the library's pool; the caller's buffer otherwise): the batch goes through in groups of
    // streams, one chain of launches each (its scratch is the one the group in front of it used: the launches are stream-ordered).
    // NOTE the piece size follows the group's bytes, so a group is laid out again by the recursive call.
    const size_t BUDGET = w.caller ? w.bytes : (size_t)8 << 30;
    if (stride > BUDGET) return hipSuccess;                     // not even one stream: the caller's other paths
    if (nstr > 1u && stride * (size_t)nstr > BUDGET) {
        uint32_t gs = (uint32_t)(BUDGET / stride);
        // (a smaller group may be cut into smaller pieces with more lists per byte: shrink until the group's own layout fits)
        while (gs > 1u && layout_of(zn, gs, a.out_pitch, a.flags, batch).stride * (size_t)gs > BUDGET) gs = gs * 3u / 4u;
        if (gs == 0u || layout_of(zn, gs, a.out_pitch, a.flags, batch).stride * (size_t)gs > BUDGET) return hipSuccess;
        if (gs < nstr) {
            for (uint32_t s0 = 0; s0 < nstr; s0 += gs) {
                InflateArgs g = a;
                if (a.in_off) g.in_off = a.in_off + s0;
                else g.in = a.in + (uint64_t)s0 * a.in_pitch;
                g.out = a.out + (uint64_t)s0 * a.out_pitch;
                g.out_len = a.out_len + s0;
                g.status = a.status + s0;
                g.nstreams = nstr - s0 < gs ? nstr - s0 : gs;
                // (the last, smaller group could be laid out with smaller pieces and need more per stream than fits)
                if (layout_of(zn, (uint32_t)g.nstreams, a.out_pitch, a.flags, batch).stride * (size_t)g.nstreams > BUDGET) {
                    if (s0 == 0) return hipSuccess;
                    // hand the rest to the batch kernels: mark them as the chain's give-ups
                    hipLaunchKernelGGL(k_par_flag_rest, dim3((unsigned)((g.nstreams + 63u) / 64u)), dim3(64), 0, stream, g.out_len, g.status, (uint32_t)g.nstreams);
                    hipError_t er = hipGetLastError();
                    if (er == hipSuccess) er = launch_inflate_dyn_flagged(g, stream);
                    if (er != hipSuccess) return er;
                    continue;
                }
                bool u = false;
                const hipError_t eg = launch_inflate_par_group(g, stream, &u, w, batch);
                if (eg != hipSuccess) return eg;
                // (a first group without scratch: the caller's batch kernels redo the whole batch, which is harmless)
                if (!u) {
                    if (s0 == 0) return hipSuccess;
                    hipLaunchKernelGGL(k_par_flag_rest, dim3((unsigned)((g.nstreams + 63u) / 64u)), dim3(64), 0, stream, g.out_len, g.status, (uint32_t)g.nstreams);
                    hipError_t er = hipGetLastError();
                    if (er == hipSuccess) er = launch_inflate_dyn_flagged(g, stream);
                    if (er != hipSuccess) return er;
                }
            }
            *used = true;
            return hipSuccess;
        }
    }
    hipError_t e = w.get(stride * nstr, stream, &ws);
    if (e != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }      // no scratch: the caller goes on with the serial decoder
    {
        ParArgs p{a.in, zn, a.flags, a.obsize, a.out, (uint32_t)cap64, (uint32_t)srcn, a.out_len, a.status, nchunks, chbits,
                  reinterpret_cast<uint32_t*>(ws + o_ctl), ws + o_ex, reinterpret_cast<uint32_t*>(ws + o_nb), ws + o_en,
                  reinterpret_cast<uint32_t*>(ws + o_op), ws + o_gx, ws + o_gs, reinterpret_cast<uint32_t*>(ws + o_gn), ws + o_ge,
                  reinterpret_cast<uint32_t*>(ws + o_go), reinterpret_cast<uint32_t*>(ws + o_tk), tmax_of(chbits / sub),
                  reinterpret_cast<uint32_t*>(ws + o_nt), reinterpret_cast<uint32_t*>(ws + o_sa), sub, ws + o_mx,
                  reinterpret_cast<uint32_t*>(ws + o_mn), (uint32_t)C_NUSED, reinterpret_cast<uint32_t*>(ws + o_me),
                  reinterpret_cast<uint32_t*>(ws + L.o_cr), reinterpret_cast<uint32_t*>(ws + L.o_nf),
                  a.in_pitch, a.out_pitch, a.in_off, stride, nstr > 1u ? 1u : 0u};
        // fork: the chain for any block types on the side stream, beside this one
        const uint32_t* actl = nullptr;
        uint32_t apasses = 0;
        hipEvent_t ev_join = nullptr;
        if (L.any_bytes != 0u) {
            hipStream_t side = side_stream();
            hipEvent_t ev_fork = nullptr;
            bool forked = side && hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess &&
                          hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) == hipSuccess &&
                          hipEventRecord(ev_fork, stream) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess;
            if (!forked) { (void)hipGetLastError(); side = stream; }
            e = launch_inflate_any(a, nstr, ws, stride, L.o_any, o_sa, (uint32_t)srcn, (uint32_t)cap64, side, &apasses);
            actl = reinterpret_cast<const uint32_t*>(ws + L.o_any);
            if (forked && e == hipSuccess) e = hipEventRecord(ev_join, side);
            if (!forked && ev_join) { (void)hipEventDestroy(ev_join); ev_join = nullptr; }
            if (ev_fork) (void)hipEventDestroy(ev_fork);
            if (e != hipSuccess) { if (ev_join) (void)hipEventDestroy(ev_join); const hipError_t e2 = w.put(ws, stream); (void)e2; return e; }
        }
        hipLaunchKernelGGL(k_par_zero, dim3(1, nstr), dim3(64), 0, stream, p);
        // the same arguments at sub-piece granularity: what the real decode and the emit work on
        ParArgs pf = p;
        pf.nchunks = nchunks * sub; pf.chbits = chbits / sub; pf.cnu = C_FNUSED;
        pf.entry8 = ws + o_fe; pf.opos = reinterpret_cast<uint32_t*>(ws + o_fo);
        const uint32_t passes = passes_for(nchunks);
#ifdef HDLZ_PAR_SPEC32
        if (sub > 1u) hipLaunchKernelGGL(k_par_spec<true>, dim3((nchunks + 1u) / 2u, nstr), dim3(64), 0, stream, p);
        else hipLaunchKernelGGL(k_par_spec<false>, dim3((nchunks + 1u) / 2u, nstr), dim3(64), 0, stream, p);
#else
        Chains ch{reinterpret_cast<uint32_t*>(ws + o_rp), reinterpret_cast<uint32_t*>(ws + o_cp), ws + o_cx, reinterpret_cast<uint32_t*>(ws + o_cn),
                  ws + o_cmx, reinterpret_cast<uint32_t*>(ws + o_cmn)};
        hipLaunchKernelGGL(k_par_head, dim3((nchunks + 2u * HEAD_WAVES - 1u) / (2u * HEAD_WAVES), nstr), dim3(64 * HEAD_WAVES), 0, stream, p, ch);
        hipLaunchKernelGGL(k_par_tail, dim3((nchunks * 32u + 63u) / 64u, nstr), dim3(64), 0, stream, p, ch);
        hipLaunchKernelGGL(k_par_resolve, dim3((nchunks * 32u + 255u) / 256u, nstr), dim3(256), 0, stream, p, ch);
#endif
        hipLaunchKernelGGL(k_par_scan_groups, dim3(ngroups, nstr), dim3(64), 0, stream, p);
        hipLaunchKernelGGL(k_par_scan_top, dim3(1, nstr), dim3(256), 0, stream, p);
        hipLaunchKernelGGL(k_par_scan_pieces, dim3(ngroups, nstr), dim3(64), 0, stream, p, pf.entry8, pf.opos);
        if (pf.chbits <= CH_BITS_MAX / 2u)
            hipLaunchKernelGGL(k_par_tokens<true>, dim3((pf.nchunks + 63u) / 64u, nstr), dim3(64), 256u * (pf.chbits / 32u + 6u), stream, pf);
        else hipLaunchKernelGGL(k_par_tokens<false>, dim3((pf.nchunks + 63u) / 64u, nstr), dim3(64), 0, stream, pf);
        hipLaunchKernelGGL(k_par_ends, dim3(1, nstr), dim3(256), 0, stream, p);
        ParArgs pe = p;                                                 // the emit: pieces, reading the sub-pieces' token lists
        pe.tokens = pf.tokens; pe.ntok = pf.ntok;
        hipLaunchKernelGGL(k_par_emit<false>, dim3(nchunks, nstr), dim3(64), 0, stream, pe);
        for (uint32_t j = 0; j < passes; j++) hipLaunchKernelGGL(k_par_jump, dim3(j == 0u ? nchunks : later_grid(nchunks, nstr), nstr), dim3(64), 0, stream, p, j);
        e = hipGetLastError();
        // join: the verdict looks at both chains' control words
        if (ev_join) {
            if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev_join, 0);
            (void)hipEventDestroy(ev_join);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_par_finish, dim3(1, nstr), dim3(64), 0, stream, p, passes, actl, apasses);
            e = hipGetLastError();
        }
        // ONE stream: the serial decoder returns at once when ctl[C_OK] >= 1; several: it redoes the streams k_par_finish flagged
        if (e == hipSuccess) e = nstr == 1u ? launch_inflate_dyn(a, stream, true, p.ctl + C_OK, 1u) : launch_inflate_dyn_flagged(a, stream);
        *used = true;
    }
    const hipError_t e2 = w.put(ws, stream);
    return e != hipSuccess ? e : e2;
}

}  // namespace hdlz

#ifdef HDLZ_DEBUG_EXPORTS      // (lib/libhdlz_dbg.so, tools/dev_any.py: where the control words of the two chains lie in the caller's scratch)
extern "C" size_t hdlz_debug_par_offsets(uint32_t in_len, uint64_t nstreams, uint64_t out_pitch, uint32_t flags, size_t* o_ctl, size_t* o_any) {
    const hdlz::par::Layout L = hdlz::par::layout_of(in_len, (uint32_t)nstreams, out_pitch, flags);
    *o_ctl = L.o_ctl; *o_any = L.o_any;
    return L.ok ? L.stride : 0u;
}
#endif
