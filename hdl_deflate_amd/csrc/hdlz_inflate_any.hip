// hdlz_inflate_any.hip -- STARTD for ONE large stream of ANY block types (or a batch of them, blockIdx.y = the stream) on the whole GPU:
// what the reference's default build (DYNAMIC=True, /root/reference/deflate.py:32) accepts -- any stock-zlib stream: a sequence of
// stored, fixed and dynamic-tree blocks (deflate.py:656-732 HEADER loop over blocks, :1084-1202 BL/READBL, :1204-1400 HF1..SPREAD,
// :1402-1445 NEXT, :1447-1517 D_NEXT, :1519-1591 INFLATE, :1593-1659 COPY).  Round 6: up to round 5 only a stream
// that is one fixed block took the whole-GPU path (hdlz_inflate_par.hip); everything else went to ONE wave (11 MB/s).
//
// A dynamic block cannot be entered in the middle without its code, and where a block starts is only known to whoever decoded the
// block in front of it.  But a dynamic block HEADER is recognisable: BTYPE = 10, HLIT <= 29, HDIST <= 29, a COMPLETE code-length code,
// HLIT + HDIST + 258 lengths that decode without overrun, a literal/length code with an end-of-block symbol that is complete, a
// distance code that is complete (or has at most one symbol) -- the serial decoder's own acceptance rules (hdlz_inflate_dyn.hip) --
// which 1 bit position in ~10^8 of arbitrary data passes.  So:
//   1. k_any_find     every bit position of the stream is tested for the first 17 + 3 * (HCLEN + 4) bits of such a header
//                     (0.085 % of the positions pass), the survivors are listed;
//      k_any_headers  one LANE per survivor decodes the code lengths and applies the rest of the rules: the CANDIDATE blocks;
//      k_any_sort     ... ordered by position;   k_any_tables: one workgroup per candidate builds its decode tables (an 11-bit
//                     and a 9-bit look-up table + the canonical lists for longer codes), slot `maxb` holds the fixed code's;
//   2. k_any_owner    every piece of the stream (1024 bits; 2048 from 4 MiB on) belongs to the last candidate whose payload starts in
//                     front of it;
//      k_any_spec     one WAVE per piece decodes its first 256 bits with its owner's tables from the 64 bit offsets a token can start
//                     at behind its first bit (a token is at most 15 + 5 + 15 + 13 = 48 bits long); the lanes that stand at the same
//                     bit afterwards are ONE chain (up to 8 per piece are listed);  k_any_tail: one lane per listed chain decodes the
//                     rest of its piece;  k_any_resolve: exit offset / end-of-block position and the bytes produced, for every
//                     entry offset (the maps of hdlz_inflate_par.hip, per block);
//   3. k_any_walk     one WAVE per candidate walks its block: the first partial piece by decoding, then through the maps (64 pieces
//                     staged in LDS at a time) to the block's end-of-block code; behind it the blocks that cannot be found by search
//                     are followed serially -- stored blocks (a length and a jump), fixed blocks (decoded here: they are short in
//                     streams that also hold dynamic blocks; more than FIX_MAX_BITS give up) -- up to the next dynamic header, which
//                     must be a candidate (binary search): the block's successor.  A pseudo-node does the same from the stream's
//                     first header.  A walk that meets pieces decoded for ANOTHER candidate (a false positive inside its block) asks
//                     for them again with its own tables (k_any_spec2) and goes on in the next round;
//      k_any_rank     follows the successors from the pseudo-node (pointer doubling): the TRUE chain of blocks, every block's output
//                     position, the total length.  A candidate that is not on the chain (a false positive, or a header-like
//                     pattern inside stored data) is simply never visited;
//   4. k_any_tokens   one LANE per piece (and per partial piece / fixed-block piece / requested piece the walks listed) decodes for
//                     real, with the reference's checks, into a token list -- and verifies that the item ends at the bit and with
//                     the byte count the walk promised;  k_any_stored copies the stored blocks;
//      k_par_emit / k_par_jump (hdlz_inflate_par.hip) turn the token lists into bytes: history that is not there yet becomes
//                     markers, pointer jumping resolves them.
// Whatever this chain cannot do -- more candidates, items or stored blocks than its lists hold, a fixed block of more than
// FIX_MAX_BITS, a failed check, an inconsistency -- sets its fallback flag and the serial decoder (k_inflate_dyn) redoes the
// stream: statuses and bytes are the serial decoder's by construction, this path only ever reports HDLZ_OK.
// Whose stream: one that does NOT start with a fixed block (those are hdlz_inflate_par.hip's), or one that starts with a SHORT fixed block
// followed by a block of another type (k_any_zero; view()).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "hdlz_device.h"
#include "hdlz_inflate_tables.h"
#include "hdlz_inflate_par.h"

namespace hdlz {
namespace any {

using par::C_FALLBACK;
using par::C_FNUSED;
using par::C_MARK;
using par::C_NUSED;
using par::C_OK;
using par::C_TOTAL;
using par::NONE;
using par::TOK_LIT;

// bits per piece (Args::pb): 1024, and 2048 for streams of 4 MiB and more -- the real decode and the walks are lane-serial per piece
// (small streams want short pieces: 60 KB .. 1 MiB 0.63 .. 0.71 ms against 0.85 .. 0.92), the emit and the marker pass pay per item
// (large ones want fewer: 16 MiB 1.71 .. 1.89 ms against 1.73 .. 2.10)
constexpr uint32_t PB_MAX = 2048;
// a map entry: kind << 30 | offset << NBB | bytes -- offset up to pb + 15 (an end-of-block code ends that far behind the piece's first bit)
constexpr uint32_t OB = 12u, NBB = 30u - OB, NBM = (1u << NBB) - 1u, OFM = (1u << OB) - 1u;
static_assert(PB_MAX + 15u <= OFM, "map entry fields");
constexpr uint32_t LB = 11, DB = 9;           // bits of the two primary look-up tables
constexpr uint32_t FIX_MAX_BITS = 16384;      // a fixed block behind a dynamic one is decoded by ONE lane: up to this many bits
constexpr uint32_t FIRST_FIX_BITS = 4096;     // a FIRST block that is fixed: this chain's if it ends within so many bits and a block of another type follows (k_any_zero)
constexpr uint32_t NO_OWNER = 0xFFFFu;
constexpr uint32_t N_END = 0xFFFFFFFEu, N_BAD = 0xFFFFFFFFu;      // successor of a node: the stream ends / no valid successor
enum { A_NCAND = par::C_ANY0, A_NBLK, A_NX, A_NS, A_OVER, A_WHY, A_NREQ, A_NREQP, A_OPEN };       // this chain's counters in its control words; A_WHY: bits of W_* (diagnostics)
enum { W_OVER = 1, W_NODE = 2, W_NEXT = 4, W_CAP = 8, W_CYCLE = 16, W_ITEMS = 32, W_TOKEN = 64, W_TCAP = 128, W_WALK_OWNER = 256, W_WALK_MAP = 512,
       W_WALK_HDR = 1024, W_WALK_STORED = 2048, W_WALK_FIX = 4096, W_WALK_TOK = 8192, W_VERIFY = 16384 };

// decode tables of one block (global memory; a wave of k_any_spec copies its owner's into LDS)
struct __attribute__((aligned(256))) Tab {
    uint16_t ll[1u << LB];        // by the next LB stream bits: symbol | length << 9 of a literal/length code of up to LB bits; 0 = longer or none
    uint16_t dd[1u << DB];        // ... distance code of up to DB bits: symbol | length << 5
    uint16_t lsym[288], dsym[32]; // the symbols sorted by (length, value)
    uint16_t lfirst[16], lcnt[16], loff[16];      // canonical form: first code of each length (MSB first), count, index of its first symbol in lsym
    uint16_t dfirst[16], dcnt[16], doff[16];
};
static_assert(sizeof(Tab) % 256 == 0, "tables are copied in 16-byte words");

struct Blk { uint32_t hdr, pay, fin, nlen; };                         // a candidate: header bit, first payload bit, BFINAL, HLIT + 257
struct Node { uint32_t next, nbytes, ok, obase; };                    // result of its walk; obase: k_any_rank
struct XItem { uint32_t start, limit, rel, node_slot, endpos, nbytes; };      // an extra decode item: bits [start, limit), rel = bytes of its node in front; node | slot << 16;
                                                                      // endpos / nbytes: where the walk says it ends and what it makes (k_any_tokens checks both)
struct SItem { uint32_t src, len, rel, node; };                       // a stored block: `len` bytes from stream byte `src`
// A block whose walk meets pieces that were decoded for ANOTHER candidate -- a false positive inside it: a header-like bit pattern, ~1 per
// 4 MB of compressed data, almost always a degenerate code of two or three symbols -- asks for those pieces to be decoded with its own
// tables (k_any_spec2) and goes on in the next round of k_any_walk
struct Req { uint32_t node, q0, q1, slot, round; };                   // pieces [q0, q1) for candidate `node`; maps at map2[slot ..]
struct NState { uint32_t q, pos, nb, req; };                          // a waiting walk (Node::ok == 3)
constexpr uint32_t WALK_ROUNDS = 3;           // rounds of k_any_walk: up to WALK_ROUNDS - 1 false positives in a row inside one block
constexpr uint32_t REQ_MAX = 1024;            // pieces per request (more: the rest in the next round)

struct Args {
    const uint8_t* z;
    uint32_t zn, flags, obsize;
    uint8_t* out;
    uint32_t cap, srcn;
    uint64_t in_pitch, out_pitch;
    const uint64_t* in_off;
    uint8_t* ws;                  // this chain's scratch of stream 0
    size_t stride;                // bytes from a stream's scratch to the next stream's
    uint32_t* srcA;               // stream 0's marker words
    uint32_t nchunks, candcap, maxb, maxx, maxs, tcap, maxreq, mapcap, pb;
    size_t o_cand, o_blk, o_blen, o_shdr, o_spay, o_sidx, o_tab, o_owner, o_map, o_pent, o_prel, o_pnode, o_node, o_xitem, o_sitem,
           o_opos, o_ntok, o_tok, o_mext, o_req, o_map2, o_nstate, o_cpos, o_cres, o_nch, o_pmap;
};
// one stream's view
struct View {
    const uint8_t* z;
    uint32_t zn;
    uint8_t* out;
    uint32_t* ctl;
    uint32_t pb;                  // bits per piece
    bool run;                     // the gate is open and nothing has failed so far
    uint32_t* cand; Blk* blk; uint8_t* blen; uint32_t* shdr; uint32_t* spay; uint32_t* sidx; Tab* tab; uint16_t* owner; uint32_t* map;
    uint8_t* pent; uint32_t* prel; uint16_t* pnode; Node* node; XItem* xitem; SItem* sitem; uint32_t* opos; uint32_t* ntok; uint32_t* tok;
    uint32_t* mext; uint32_t* srcA; Req* req; uint32_t* map2; NState* nstate; uint32_t* cpos; uint32_t* cres; uint8_t* nch; uint32_t* pmap;
};
template <typename T> __device__ __forceinline__ T* at(uint8_t* base, size_t off) { return reinterpret_cast<T*>(base + off); }
__device__ __forceinline__ View view(const Args& a) {
    const uint32_t s = blockIdx.y;
    View v;
    if (a.in_off) {
        const uint64_t o0 = a.in_off[s], n64 = a.in_off[s + 1u] - o0;
        v.z = a.z + o0;
        v.zn = n64 > (uint64_t)a.zn ? 0u : (uint32_t)n64;          // longer than the stated bound: not this chain's (zn = 0 fails every test)
    } else { v.z = a.z + (uint64_t)s * a.in_pitch; v.zn = a.zn; }
    v.out = a.out + (uint64_t)s * a.out_pitch;
    uint8_t* w = a.ws + (size_t)s * a.stride;
    v.ctl = reinterpret_cast<uint32_t*>(w);
    v.pb = a.pb;
    // the gate: a stream that STARTS with a fixed block is the other chain's (hdlz_inflate_par.hip: starts_fixed -- the same test on
    // the same byte, so the two chains need nothing from each other and run side by side; it also takes streams of SEVERAL fixed
    // blocks) -- unless that block is SHORT and a block of another type follows it (a header written and flushed in front of the data:
    // zlib closes so small a block as a fixed one): the other chain gives those up at that block, k_any_zero opens this one
    // (A_OPEN), and the walk from the stream's first header decodes the fixed block like one between dynamic blocks.  This chain is
    // only launched for the default build's flags
    const uint32_t hdr = v.zn >= 5u ? (uint32_t)v.z[2] : 0u;
    v.run = v.zn >= 5u && (((hdr >> 1) & 3u) != 1u || v.ctl[A_OPEN] != 0u) && v.ctl[C_FALLBACK] == 0u;
    v.cand = at<uint32_t>(w, a.o_cand); v.blk = at<Blk>(w, a.o_blk); v.blen = at<uint8_t>(w, a.o_blen); v.shdr = at<uint32_t>(w, a.o_shdr);
    v.spay = at<uint32_t>(w, a.o_spay); v.sidx = at<uint32_t>(w, a.o_sidx); v.tab = at<Tab>(w, a.o_tab); v.owner = at<uint16_t>(w, a.o_owner);
    v.map = at<uint32_t>(w, a.o_map); v.pent = at<uint8_t>(w, a.o_pent); v.prel = at<uint32_t>(w, a.o_prel); v.pnode = at<uint16_t>(w, a.o_pnode);
    v.node = at<Node>(w, a.o_node); v.xitem = at<XItem>(w, a.o_xitem); v.sitem = at<SItem>(w, a.o_sitem); v.opos = at<uint32_t>(w, a.o_opos);
    v.ntok = at<uint32_t>(w, a.o_ntok); v.tok = at<uint32_t>(w, a.o_tok); v.mext = at<uint32_t>(w, a.o_mext);
    v.req = at<Req>(w, a.o_req); v.map2 = at<uint32_t>(w, a.o_map2); v.nstate = at<NState>(w, a.o_nstate);
    v.cpos = at<uint32_t>(w, a.o_cpos); v.cres = at<uint32_t>(w, a.o_cres); v.nch = at<uint8_t>(w, a.o_nch); v.pmap = at<uint32_t>(w, a.o_pmap);
    v.srcA = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(a.srcA) + (size_t)s * a.stride);
    return v;
}
__device__ __forceinline__ void give_up(const View& v) { atomicExch(&v.ctl[C_FALLBACK], 1u); }

// ---- bit reader of one lane, straight from the stream (bytes at or beyond zn read as zero)
struct Bits {
    const uint8_t* z;
    uint32_t zn, ip, bc, pos;
    uint32_t n0, n1, n2, n3;      // the four dwords behind the buffer, requested that far ahead: a lane's refill does not wait for memory
    uint64_t bb;
    __device__ __forceinline__ void init(const uint8_t* z_, uint32_t zn_, uint32_t pos_) {
        z = z_; zn = zn_; pos = pos_;
        ip = (pos >> 3) & ~3u;
        bc = 64u - (pos - 8u * ip);
        bb = (((uint64_t)tok::load32(z, ip + 4u, zn) << 32) | tok::load32(z, ip, zn)) >> (pos - 8u * ip);
        n0 = tok::load32(z, ip + 8u, zn); n1 = tok::load32(z, ip + 12u, zn); n2 = tok::load32(z, ip + 16u, zn); n3 = tok::load32(z, ip + 20u, zn);
        ip += 8u;                  // n0 = the dword at ip
    }
    __device__ __forceinline__ void refill() {          // >= 33 valid bits afterwards
        if (bc <= 32u) {
            bb |= (uint64_t)n0 << bc; bc += 32u; ip += 4u;
            n0 = n1; n1 = n2; n2 = n3; n3 = tok::load32(z, ip + 12u, zn);
        }
    }
    __device__ __forceinline__ void take(uint32_t n) { bb >>= n; bc -= n; pos += n; }
};

// ---- one symbol of a canonical code: the primary table by the next bits, the canonical lists for codes longer than it.
// -> symbol, length (0: no code starts with these bits).  x: at least 15 stream bits, LSB first.
template <uint32_t PBITS, uint32_t SHIFT, uint32_t SMASK>
__device__ __forceinline__ void sym_of(const uint16_t* prim, const uint16_t* first, const uint16_t* cnt, const uint16_t* off,
                                       const uint16_t* syms, uint32_t x, uint32_t& sym, uint32_t& len) {
    const uint32_t e = prim[x & ((1u << PBITS) - 1u)];
    sym = e & SMASK; len = e >> SHIFT;
    if (e == 0u) {
        const uint32_t rv = __builtin_bitreverse32(x) >> 17;        // the next 15 bits as an MSB-first code
#pragma unroll 1
        for (uint32_t l = PBITS + 1u; l <= 15u; l++) {
            const uint32_t idx = (rv >> (15u - l)) - (uint32_t)first[l];
            if (idx < (uint32_t)cnt[l]) { sym = syms[(uint32_t)off[l] + idx]; len = l; break; }
        }
    }
}
// one token at the reader's position, CONSUMED (an invalid one is not).  -> kind: 0 literal, 1 match, 2 end of block, 3 invalid; value /
// length / distance; nb = bits of the literal/length code, used = bits of the whole token
struct Token { uint32_t kind, lit, length, dist, nb, used; };
__device__ __forceinline__ Token token_at(const Tab* t, Bits& r) {
    Token k;
    r.refill();
    uint32_t sym, len;
    sym_of<LB, 9u, 511u>(t->ll, t->lfirst, t->lcnt, t->loff, t->lsym, (uint32_t)r.bb, sym, len);
    k.nb = len; k.used = len; k.lit = sym; k.length = 0; k.dist = 0;
    if (len == 0u) { k.kind = 3u; return k; }
    if (sym < 256u) { k.kind = 0u; r.take(len); return k; }
    if (sym == 256u) { k.kind = 2u; r.take(len); return k; }
    if (sym >= 286u) { k.kind = 3u; return k; }                     // (the fixed code has 286 / 287: BAD_SYMBOL in the serial decoder)
    uint32_t lbase, leb;
    tok::length_info(sym - 257u, lbase, leb);
    const uint32_t x1 = (uint32_t)(r.bb >> len);
    k.length = lbase + (x1 & ((1u << leb) - 1u));
    r.take(len + leb);
    r.refill();
    uint32_t ds, dl;
    sym_of<DB, 5u, 31u>(t->dd, t->dfirst, t->dcnt, t->doff, t->dsym, (uint32_t)r.bb, ds, dl);
    if (dl == 0u || ds >= 30u) { k.kind = 3u; return k; }           // (the fixed code has 30 / 31: BAD_DISTANCE)
    uint32_t dbase, deb;
    tok::dist_info(ds, dbase, deb);
    k.dist = dbase + ((uint32_t)(r.bb >> dl) & ((1u << deb) - 1u));
    r.take(dl + deb);
    k.used = len + leb + dl + deb;
    k.kind = 1u;
    return k;
}

// ================================================================================================ 1. the candidates
// the cheap part of the test: BTYPE = 10, HLIT <= 29, HDIST <= 29 (22 % of all positions pass)
__device__ __forceinline__ bool header_fields_ok(uint32_t x) {
    return ((x >> 1) & 3u) == 2u && ((x >> 3) & 31u) <= 29u && ((x >> 8) & 31u) <= 29u;
}
// ... and the rest: a COMPLETE code-length code (lo: the 64 stream bits from the header's first bit on, hi: the 32 behind them)
__device__ __forceinline__ bool header_precode_ok(uint64_t lo, uint32_t hi) {
    const uint32_t n = (((uint32_t)lo >> 13) & 15u) + 4u;
    uint64_t f = (lo >> 17) | ((uint64_t)hi << 47);
    int32_t left = 128;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t l = (uint32_t)f & 7u;
        f >>= 3;
        left -= l ? (int32_t)(128u >> l) : 0;
    }
    return left == 0;
}
constexpr uint32_t FIND_T = 256, FIND_CAP = 2048, FIND_Q = 640;
__global__ __launch_bounds__(FIND_T) void k_any_find(Args a) {
    const View v = view(a);
    __shared__ uint32_t lst[FIND_CAP];
    __shared__ uint32_t wq[FIND_T / 64][FIND_Q];        // per wave: the positions that passed the cheap part
    __shared__ uint32_t ln, gbase;
    if (!v.run) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    // (this kernel also presets the per-piece and per-item words the later ones only write where something is: no owner, no node, no tokens)
    for (uint32_t i = blockIdx.x * FIND_T + tid; i < a.nchunks + a.maxx; i += gridDim.x * FIND_T) {
        v.ntok[i] = 0u; v.mext[i] = 0u; v.opos[i] = 0u;
        if (i < a.nchunks) { v.owner[i] = (uint16_t)NO_OWNER; v.pnode[i] = (uint16_t)NO_OWNER; }
    }
    if (tid == 0u) ln = 0u;
    __syncthreads();
    const uint32_t nbits = 8u * v.zn;
    auto flush = [&]() {                                            // (called by all threads)
        const uint32_t n = min(ln, FIND_CAP);
        if (tid == 0u) gbase = n ? atomicAdd(&v.ctl[A_NCAND], n) : 0u;
        __syncthreads();
        for (uint32_t k = tid; k < n; k += FIND_T) { if (gbase + k < a.candcap) v.cand[gbase + k] = lst[k]; }
        if (tid == 0u && (gbase + n > a.candcap || ln > FIND_CAP)) v.ctl[A_OVER] = 1u;
        __syncthreads();
        if (tid == 0u) ln = 0u;
        __syncthreads();
    };
    // a thread takes the 8 bit positions of one stream byte (12 bytes cover the 7 + 17 + 57 bits behind its first bit).  Two steps per
    // wave: the cheap fields for all 512 positions, then the code-length code for the ones that passed -- packed, so that the loop over
    // up to 19 lengths runs on full waves (one step: 8 x that loop per thread for the 22 % that get there -- 224 us at 7 MB)
    for (uint32_t base = blockIdx.x * FIND_T; base < v.zn; base += gridDim.x * FIND_T) {
        const uint32_t B = base + tid;
        uint32_t mask = 0;
        if (B >= 2u && B < v.zn) {
            const uint32_t d0 = tok::load32(v.z, B, v.zn), d1 = tok::load32(v.z, B + 4u, v.zn);
            const uint64_t w01 = ((uint64_t)d1 << 32) | d0;
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++)
                mask |= (8u * B + j + 29u <= nbits && header_fields_ok((uint32_t)(w01 >> j))) ? 1u << j : 0u;
        }
        // the wave's queue: thread's entries behind those of the lanes below it
        uint32_t cntm = __builtin_popcount(mask), incl = cntm;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) { const uint32_t o = __shfl_up(incl, ofs, 64); if (lane >= (uint32_t)ofs) incl += o; }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t at_ = incl - cntm;
        for (uint32_t m = mask; m; m &= m - 1u) { if (at_ < FIND_Q) wq[wv][at_] = 8u * B + (uint32_t)__builtin_ctz(m); at_++; }
        wave_lds_order();
        __builtin_amdgcn_wave_barrier();
        if (total > FIND_Q && lane == 0u) v.ctl[A_OVER] = 1u;              // (cannot happen on 22 %: 512 positions, 640 slots)
        for (uint32_t k = lane; k < min(total, FIND_Q); k += 64u) {
            const uint32_t p = wq[wv][k], Bp = p >> 3, j = p & 7u;
            const uint32_t d0 = tok::load32(v.z, Bp, v.zn), d1 = tok::load32(v.z, Bp + 4u, v.zn), d2 = tok::load32(v.z, Bp + 8u, v.zn);
            const uint64_t w01 = ((uint64_t)d1 << 32) | d0;
            const uint64_t lo = j ? ((w01 >> j) | ((uint64_t)d2 << (64u - j))) : w01;
            if (header_precode_ok(lo, d2 >> j)) {
                const uint32_t kk = atomicAdd(&ln, 1u);
                if (kk < FIND_CAP) lst[kk] = p;
            }
        }
        __syncthreads();
        if (ln > FIND_CAP / 2u) flush();                            // (uniform: every thread reads ln behind the barrier, nobody writes it before the next one)
        else __syncthreads();
    }
    flush();
}

// one LANE per listed position: the code lengths, and the rest of the serial decoder's acceptance rules (hdlz_inflate_dyn.hip, "BL" ..
// "canon_build": over-subscribed sets rejected, incomplete ones only with a single code -- for the distance code also with none --,
// an end-of-block code must exist, the header must leave room for the reference's end-of-input margin)
// the code-length code of the header at bit p into a 7-bit table (symbol << 3 | length by the next 7 stream bits; stride: elements between
// entries -- the lanes of k_any_headers interleave theirs); -> the reader behind the 3-bit lengths, HLIT + 257, HDIST + 1, BFINAL
static __device__ const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
__device__ __forceinline__ void read_precode(Bits& r, uint8_t* clut, uint8_t* cl, uint32_t stride, uint32_t& nlen, uint32_t& ndist, uint32_t& fin) {
    fin = (uint32_t)r.bb & 1u;
    r.take(3u);
    r.refill();
    nlen = ((uint32_t)r.bb & 31u) + 257u; ndist = (((uint32_t)r.bb >> 5) & 31u) + 1u;
    const uint32_t ncode = (((uint32_t)r.bb >> 10) & 15u) + 4u;
    r.take(14u);
    for (uint32_t k = 0; k < 19u; k++) cl[k * stride] = 0;
    uint64_t cnt8 = 0;                                              // counts of the lengths 1..7, 8 bits each
    for (uint32_t k = 0; k < 19u; k++) {
        if (k < ncode) {
            r.refill();
            const uint32_t l = (uint32_t)r.bb & 7u;
            r.take(3u);
            cl[CL_ORDER[k] * stride] = (uint8_t)l;
            cnt8 += l ? (1ull << (8u * l)) : 0ull;
        }
    }
    // canonical codes (the code is complete: k_any_find checked), spread over the 7-bit table
    uint64_t next8 = 0;                                             // next code of each length, 8 bits each
    uint32_t code = 0;
    for (uint32_t l = 1; l < 8u; l++) {
        code = (code + (uint32_t)((cnt8 >> (8u * (l - 1u))) & 255u)) << 1;            // (byte 0 of cnt8 is 0)
        next8 |= (uint64_t)(code & 255u) << (8u * l);
    }
    for (uint32_t s = 0; s < 19u; s++) {
        const uint32_t l = cl[s * stride];
        if (l) {
            const uint32_t c = (uint32_t)(next8 >> (8u * l)) & 255u;
            next8 += 1ull << (8u * l);
            const uint32_t rv = __builtin_bitreverse32(c) >> (32u - l);
            for (uint32_t k = rv; k < 128u; k += 1u << l) clut[k * stride] = (uint8_t)((s << 3) | l);
        }
    }
}
// one code-length symbol (READBL / REPEAT, deflate.py:1116-1164, :1190-1202): -> value, repeat count
__device__ __forceinline__ void read_length(Bits& r, const uint8_t* clut, uint32_t stride, uint32_t prev, uint32_t& val, uint32_t& rep, uint32_t& sy) {
    r.refill();
    const uint32_t e = clut[((uint32_t)r.bb & 127u) * stride], l = e & 7u;
    sy = e >> 3;
    r.take(l);
    rep = 1u; val = sy;
    if (sy == 16u) { rep = 3u + ((uint32_t)r.bb & 3u); r.take(2u); val = prev; }
    else if (sy == 17u) { rep = 3u + ((uint32_t)r.bb & 7u); r.take(3u); val = 0u; }
    else if (sy == 18u) { rep = 11u + ((uint32_t)r.bb & 127u); r.take(7u); val = 0u; }
}
struct HdrLds {
    uint8_t clut[128][64];        // the code-length code by the next 7 bits, per lane
    uint8_t cl[19][64];
};
__global__ __launch_bounds__(64) void k_any_headers(Args a) {
    const View v = view(a);
    __shared__ HdrLds L;
    if (!v.run) return;
    if (v.ctl[A_OVER] != 0u) { if (threadIdx.x == 0u && blockIdx.x == 0u) give_up(v); return; }
    const uint32_t lane = threadIdx.x, ncand = min(v.ctl[A_NCAND], a.candcap);
    const uint32_t nbits = 8u * v.zn;
    const int32_t isize = (int32_t)v.zn - 1;
    for (uint32_t c0 = blockIdx.x * 64u; c0 < ncand; c0 += gridDim.x * 64u) {
        const uint32_t i = c0 + lane;
        bool ok = i < ncand;
        const uint32_t p = ok ? v.cand[i] : 16u;
        Bits r;
        r.init(v.z, v.zn, p);
        uint32_t nlen, ndist, fin;
        read_precode(r, &L.clut[0][lane], &L.cl[0][lane], 64u, nlen, ndist, fin);
        const uint32_t total = nlen + ndist;
        // the length list, counted and checked only -- registers: the Kraft sums of the two codes in units of 2^-15, their numbers of
        // codes, the length of the end-of-block code.  Most positions fail within a few symbols (a set that is over-subscribed
        // already is dropped at once); the lists of the few that ARE headers are decoded again where they are needed (k_any_tables)
        uint32_t idx = 0, prev = 0, kr1 = 0, kr2 = 0, nz1 = 0, nz2 = 0, eob_len = 0;
        while (ballot64(ok && idx < total) != 0ull) {
            if (ok && idx < total) {
                uint32_t val, rep, sy;
                read_length(r, &L.clut[0][lane], 64u, prev, val, rep, sy);
                if ((sy == 16u && idx == 0u) || idx + rep > total || r.pos > nbits) ok = false;
                const uint32_t n1 = idx < nlen ? min(rep, nlen - idx) : 0u, n2 = rep - n1;
                if (val) { kr1 += n1 << (15u - val); kr2 += n2 << (15u - val); nz1 += n1; nz2 += n2; }
                if (kr1 > 32768u || kr2 > 32768u) ok = false;                                // over-subscribed
                if (idx <= 256u && 256u < idx + rep) eob_len = val;
                prev = val;
                idx += rep;
            }
        }
        if (eob_len == 0u) ok = false;                                                        // no end-of-block code
        if (kr1 < 32768u && nz1 != 1u) ok = false;                                            // incomplete: only with a single code
        if (kr2 < 32768u && nz2 > 1u) ok = false;                                             // ... the distance code also with none
        if ((int32_t)(r.pos >> 3) > isize - 3) ok = false;                                    // (the serial decoder's NO EOF at the end of a header)
        if (ok) {
            const uint32_t slot = atomicAdd(&v.ctl[A_NBLK], 1u);
            if (slot < a.maxb) v.blk[slot] = Blk{p, r.pos, fin, nlen};
            else v.ctl[A_OVER] = 1u;
        }
    }
}

// the candidates ordered by position (all distinct): rank by counting, one workgroup
// the candidates by position: a rank sort (the rank of a header = the headers in front of it), one element per thread, every workgroup with
// the whole list in LDS.  (One workgroup of 1024 threads for all of them: 0.48 ms for the 5414 blocks of a 256 MiB stream.)
constexpr uint32_t SORT_T = 256;
__global__ __launch_bounds__(SORT_T) void k_any_sort(Args a) {
    const View v = view(a);
    extern __shared__ uint32_t hs[];                                // [maxb]
    if (!v.run) return;
    if (v.ctl[A_OVER] != 0u) { if (threadIdx.x == 0u && blockIdx.x == 0u) give_up(v); return; }
    const uint32_t n = min(v.ctl[A_NBLK], a.maxb);
    if (blockIdx.x * SORT_T >= n) return;
    for (uint32_t i = threadIdx.x; i < n; i += SORT_T) hs[i] = v.blk[i].hdr;
    __syncthreads();
    const uint32_t i = blockIdx.x * SORT_T + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = hs[i];
    uint32_t rank = 0;
#pragma unroll 8
    for (uint32_t j = 0; j < n; j++) rank += hs[j] < h ? 1u : 0u;
    v.shdr[rank] = h; v.spay[rank] = v.blk[i].pay; v.sidx[rank] = i;
}

// the decode tables of candidate `r` (slot r; slot maxb: the fixed code, deflate.py:1066-1073), one workgroup each
struct TabLds {
    Tab t;
    uint8_t len[320];
    uint8_t clut[128], cl[19];
    uint32_t cnt[32], first[32], off[32];
};
constexpr uint32_t TAB_T = 256;
__global__ __launch_bounds__(TAB_T) void k_any_tables(Args a) {
    const View v = view(a);
    __shared__ TabLds L;
    if (!v.run) return;
    const uint32_t tid = threadIdx.x, n = min(v.ctl[A_NBLK], a.maxb);
    for (uint32_t r = blockIdx.x; r <= n; r += gridDim.x) {
        const uint32_t slot = r < n ? r : a.maxb;
        uint32_t nlen = 288u;
        __syncthreads();
        if (r < n) {
            // the block's length list, by one thread (k_any_headers only checked it): ~300 dependent steps
            nlen = v.blk[v.sidx[r]].nlen;
            for (uint32_t k = tid; k < 320u; k += TAB_T) L.len[k] = 0;
            __syncthreads();
            if (tid == 0u) {
                Bits rd;
                rd.init(v.z, v.zn, v.shdr[r]);
                uint32_t nl, nd, fin;
                read_precode(rd, L.clut, L.cl, 1u, nl, nd, fin);
                uint32_t idx = 0, prev = 0;
                while (idx < nl + nd) {
                    uint32_t val, rep, sy;
                    read_length(rd, L.clut, 1u, prev, val, rep, sy);
                    for (uint32_t k = 0; k < rep && idx + k < 320u; k++) L.len[idx + k] = (uint8_t)val;
                    prev = val;
                    idx += rep;
                }
            }
        } else {
            for (uint32_t k = tid; k < 320u; k += TAB_T) L.len[k] = (uint8_t)(k < 144u ? 8 : k < 256u ? 9 : k < 280u ? 7 : k < 288u ? 8 : 5);
        }
        uint32_t* tw = reinterpret_cast<uint32_t*>(&L.t);
        for (uint32_t k = tid; k < sizeof(Tab) / 4u; k += TAB_T) tw[k] = 0u;
        if (tid < 32u) L.cnt[tid] = 0u;
        __syncthreads();
        const uint32_t ndist = 320u - nlen;                          // (lengths beyond HDIST + 1 are zero)
        for (uint32_t k = tid; k < 320u; k += TAB_T) { const uint32_t l = L.len[k]; if (l) atomicAdd(&L.cnt[(k < nlen ? 0u : 16u) + l], 1u); }
        __syncthreads();
        if (tid < 2u) {
            uint32_t code = 0, o = 0;
            const uint32_t b = 16u * tid;
            L.first[b] = 0u; L.off[b] = 0u;
            for (uint32_t l = 1; l < 16u; l++) {
                code = (code + (l > 1u ? L.cnt[b + l - 1u] : 0u)) << 1;
                L.first[b + l] = code; L.off[b + l] = o;
                o += L.cnt[b + l];
            }
        }
        __syncthreads();
        if (tid < 16u) {
            L.t.lfirst[tid] = (uint16_t)L.first[tid]; L.t.lcnt[tid] = (uint16_t)(tid ? L.cnt[tid] : 0u); L.t.loff[tid] = (uint16_t)L.off[tid];
            L.t.dfirst[tid] = (uint16_t)L.first[16u + tid]; L.t.dcnt[tid] = (uint16_t)(tid ? L.cnt[16u + tid] : 0u); L.t.doff[tid] = (uint16_t)L.off[16u + tid];
        }
        // every symbol: its place among the codes of its length (by value), its canonical code, its table entries
        for (uint32_t k = tid; k < 320u; k += TAB_T) {
            const uint32_t l = L.len[k];
            if (l == 0u) continue;
            const bool isd = k >= nlen;
            const uint32_t k0 = isd ? nlen : 0u, sym = k - k0, b = isd ? 16u : 0u;
            uint32_t rank = 0;
            for (uint32_t j = k0; j < k; j++) rank += L.len[j] == l ? 1u : 0u;
            const uint32_t code = L.first[b + l] + rank;
            const uint32_t rv = __builtin_bitreverse32(code) >> (32u - l);
            if (isd) {
                L.t.dsym[L.off[b + l] + rank] = (uint16_t)sym;
                if (l <= DB) for (uint32_t e = rv; e < (1u << DB); e += 1u << l) L.t.dd[e] = (uint16_t)(sym | (l << 5));
            } else {
                L.t.lsym[L.off[b + l] + rank] = (uint16_t)sym;
                if (l <= LB) for (uint32_t e = rv; e < (1u << LB); e += 1u << l) L.t.ll[e] = (uint16_t)(sym | (l << 9));
            }
        }
        (void)ndist;
        __syncthreads();
        uint32_t* dst = reinterpret_cast<uint32_t*>(&v.tab[slot]);
        for (uint32_t k = tid; k < sizeof(Tab) / 4u; k += TAB_T) dst[k] = tw[k];
    }
}

// ================================================================================================ 2. pieces and their maps
// piece q = the stream bits [q v.pb, (q + 1) v.pb); its owner: the last candidate whose payload starts at or in front of its first bit
__global__ __launch_bounds__(64) void k_any_owner(Args a) {
    const View v = view(a);
    if (!v.run) return;
    const uint32_t n = min(v.ctl[A_NBLK], a.maxb);
    for (uint32_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint32_t q0 = (v.spay[r] + v.pb - 1u) / v.pb;
        const uint32_t q1 = r + 1u < n ? min((v.spay[r + 1u] + v.pb - 1u) / v.pb, a.nchunks) : a.nchunks;
        for (uint32_t q = q0 + threadIdx.x; q < q1; q += 64u) v.owner[q] = (uint16_t)r;
    }
}

// a map entry: kind << 30 | offset << 19 | bytes.  kind 0: the chain leaves the piece `offset` bits behind its end; 2: it meets the
// block's end-of-block code, which ends `offset` bits behind the piece's FIRST bit; 3: a bit pattern that is no code
constexpr uint32_t SPEC_W = 4;                // waves per workgroup, each with its own tables
struct SpecLds {
    Tab t[SPEC_W];
    uint32_t win[SPEC_W][PB_MAX / 32 + 8];
};
__device__ __forceinline__ void load_tab(Tab* dst_, const Tab* src_, uint32_t lane) {
    const tok::u32x4* src = reinterpret_cast<const tok::u32x4*>(src_);
    tok::u32x4* dst = reinterpret_cast<tok::u32x4*>(dst_);
    static_assert(sizeof(Tab) % 1024 == 0, "whole rounds of 64 lanes x 16 bytes");
    tok::u32x4 t[sizeof(Tab) / 1024];                               // all the loads first: ONE memory latency, not one per round
#pragma unroll
    for (uint32_t k = 0; k < sizeof(Tab) / 1024u; k++) t[k] = src[k * 64u + lane];
#pragma unroll
    for (uint32_t k = 0; k < sizeof(Tab) / 1024u; k++) dst[k * 64u + lane] = t[k];
}
// one token of a chain, lengths and byte counts only (x: the next 64 stream bits).  -> false when the chain ends here: res = the map
// entry (end-of-block: kind 2, the bit behind the code relative to b0; no code: kind 3)
__device__ __forceinline__ bool spec_step(const Tab* t, uint64_t x, uint32_t b0, uint32_t& pos, uint32_t& nbytes, uint32_t& res) {
    uint32_t sym, len;
    sym_of<LB, 9u, 511u>(t->ll, t->lfirst, t->lcnt, t->loff, t->lsym, (uint32_t)x, sym, len);
    if (len == 0u) { res = 3u << 30; return false; }
    if (sym < 256u) { pos += len; nbytes += 1u; return true; }
    if (sym == 256u) { res = (2u << 30) | ((pos + len - b0) << NBB) | nbytes; return false; }
    uint32_t lbase, leb, ds, dl;
    tok::length_info(sym - 257u, lbase, leb);
    const uint64_t x1 = x >> len;
    const uint32_t tl = lbase + ((uint32_t)x1 & ((1u << leb) - 1u));
    sym_of<DB, 5u, 31u>(t->dd, t->dfirst, t->dcnt, t->doff, t->dsym, (uint32_t)(x1 >> leb), ds, dl);
    if (dl == 0u) { res = 3u << 30; return false; }
    const uint32_t deb = ds < 4u ? 0u : (ds >> 1) - 1u;
    pos += len + leb + dl + deb;
    nbytes += tl;
    if (nbytes >= (1u << NBB)) { res = 3u << 30; return false; }       // (cannot happen: 512 tokens of 258 bytes)
    return true;
}
__device__ __forceinline__ void stage_piece(const View& v, uint32_t* win, uint32_t q, uint32_t lane) {
    for (uint32_t k = lane; k < v.pb / 32u + 8u; k += 64u) win[k] = tok::load32(v.z, ((q * v.pb) >> 3) + 4u * k, v.zn);
    wave_lds_order();
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint64_t win_bits(const uint32_t* win, uint32_t rel) {
    const uint32_t w = rel >> 5, sh = rel & 31u;
    const uint32_t d0 = win[w], d1 = win[w + 1u], d2 = win[w + 2u];
    return (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32);
}
// the chains of all lanes from where they stand to the end of the piece -> res
__device__ __forceinline__ void spec_finish(const Tab* t, const uint32_t* win, uint32_t b0, uint32_t pb, uint32_t& pos, uint32_t& nbytes, uint32_t& res, bool& run) {
    const uint32_t end = b0 + pb;
    while (ballot64(run) != 0ull) {
        if (run) {
            run = spec_step(t, win_bits(win, pos - b0), b0, pos, nbytes, res);
            if (run && pos >= end) { res = ((pos - end) << NBB) | nbytes; run = false; }
        }
    }
}
// piece q with the tables at t (LDS), one wave: lane e starts at the piece's bit e -> mp[e]
__device__ __forceinline__ void spec_piece(const View& v, const Tab* t, uint32_t* win, uint32_t q, uint32_t* mp, uint32_t lane) {
    const uint32_t b0 = q * v.pb;
    stage_piece(v, win, q, lane);
    uint32_t pos = b0 + lane, nbytes = 0, res = 0;
    bool run = true;
    spec_finish(t, win, b0, v.pb, pos, nbytes, res, run);
    mp[lane] = res;
    __builtin_amdgcn_wave_barrier();
}
// ---- the maps with the 64-fold work only where it is needed (as k_par_head / _tail / _resolve do for the fixed code): chains that start
// at different offsets of a piece fall into step at the first token boundary they share and are ONE chain from there on.
//   k_any_spec     all 64 offsets of a piece decode its first HEAD bits only; the lanes that stand at the same bit afterwards are one
//                  chain: up to CH_MAX of them per piece are listed (more -- a period of a few tokens -- : the wave decodes the piece
//                  to its end right here);  a lane's entry: final, or kind 1 = pending: chain rank << 19 | bytes of the head
//   k_any_tail     one LANE per listed chain decodes the rest of its piece;   k_any_resolve: pending entries take their chain's result
constexpr uint32_t HEAD = 256, CH_MAX = 8;
__global__ __launch_bounds__(64 * SPEC_W) void k_any_spec(Args a) {
    const View v = view(a);
    __shared__ SpecLds L;
    __shared__ uint32_t first[SPEC_W][64], slotof[SPEC_W][64];
    if (!v.run) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t nw = gridDim.x * SPEC_W, g = blockIdx.x * SPEC_W + wv;
    const uint32_t per = (a.nchunks + nw - 1u) / nw;
    const uint32_t q0 = g * per, q1 = min(q0 + per, a.nchunks);
    uint32_t cur = NO_OWNER;
    for (uint32_t q = q0; q < q1; q++) {
        const uint32_t r = v.owner[q];
        if (r == NO_OWNER) { if (lane == 0u) v.nch[q] = 0; continue; }
        if (r != cur) { load_tab(&L.t[wv], &v.tab[r], lane); cur = r; }
        const Tab* t = &L.t[wv];
        const uint32_t* win = L.win[wv];
        const uint32_t b0 = q * v.pb, hb = b0 + HEAD;
        stage_piece(v, L.win[wv], q, lane);
        first[wv][lane] = 0xFFFFFFFFu;
        uint32_t pos = b0 + lane, nbytes = 0, res = 0;
        bool run = true;
        while (ballot64(run && pos < hb) != 0ull) {
            if (run && pos < hb) run = spec_step(t, win_bits(win, pos - b0), b0, pos, nbytes, res);
        }
        wave_lds_order();
        const uint32_t key = (pos - hb) & 63u;                   // (a token is at most 48 bits long)
        if (run) atomicMin(&first[wv][key], lane);
        wave_lds_order();
        __builtin_amdgcn_wave_barrier();
        const bool leader = run && first[wv][key] == lane;
        const uint64_t lm = ballot64(leader);
        const uint32_t nlead = (uint32_t)__popcll(lm);
        if (nlead > CH_MAX) {                                    // (rare) too many distinct chains: to the end, here
            spec_finish(t, win, b0, v.pb, pos, nbytes, res, run);
            if (lane == 0u) v.nch[q] = 0;
        } else {
            if (leader) {
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                slotof[wv][key] = rk;
                v.cpos[(size_t)q * CH_MAX + rk] = pos;
            }
            wave_lds_order();
            __builtin_amdgcn_wave_barrier();
            if (run) res = (1u << 30) | (slotof[wv][key] << NBB) | nbytes;
            if (lane == 0u) v.nch[q] = (uint8_t)nlead;
        }
        v.map[(size_t)q * 64u + lane] = res;
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ __launch_bounds__(64) void k_any_tail(Args a) {
    const View v = view(a);
    __shared__ Tab lt;
    if (!v.run) return;
    const uint32_t lane = threadIdx.x;
    uint32_t cur = NO_OWNER;
    for (uint32_t c0 = blockIdx.x * 64u; c0 < a.nchunks * CH_MAX; c0 += gridDim.x * 64u) {
        const uint32_t c = c0 + lane, q = c / CH_MAX, rk = c % CH_MAX;
        const bool have = q < a.nchunks && rk < v.nch[q];
        const uint32_t r = have ? (uint32_t)v.owner[q] : NO_OWNER;
        const uint64_t hm = ballot64(have);
        if (hm == 0ull) continue;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, (int)__builtin_ctzll(hm));
        const bool uniform = ballot64(have && r != r0) == 0ull;       // (8 consecutive pieces: almost always one block)
        if (uniform && r0 != cur) {
            __builtin_amdgcn_wave_barrier();
            load_tab(&lt, &v.tab[r0], lane);
            wave_lds_order();
            __builtin_amdgcn_wave_barrier();
            cur = r0;
        }
        const uint32_t b0 = q * v.pb, end = b0 + v.pb;
        Bits rd;
        rd.init(v.z, v.zn, have ? v.cpos[c] : 16u);
        uint32_t pos = rd.pos, nbytes = 0, res = 0;
        bool run = have;
        auto go = [&](const Tab* t) {
            while (run) {
                // (a token is at most 48 bits: the buffer's >= 33 bits + the dword behind it make the 64-bit window whole)
                rd.refill();
                const uint64_t x = rd.bc < 64u ? (rd.bb | ((uint64_t)rd.n0 << rd.bc)) : rd.bb;
                const uint32_t p0 = pos;
                run = spec_step(t, x, b0, pos, nbytes, res);
                if (run) {
                    uint32_t used = pos - p0;                    // (up to 48 bits, the buffer holds >= 33: in two steps)
                    rd.pos = p0;
                    if (used > 32u) { rd.take(32u); rd.refill(); used -= 32u; }
                    rd.take(used);
                    if (pos >= end) { res = ((pos - end) << NBB) | nbytes; run = false; }
                }
            }
        };
        if (uniform) go(&lt);
        else if (have) go(&v.tab[r]);
        if (have) v.cres[c] = res;
    }
}
__global__ __launch_bounds__(256) void k_any_resolve(Args a) {
    const View v = view(a);
    if (!v.run) return;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < a.nchunks * 64u; t += gridDim.x * 256u) {
        const uint32_t q = t >> 6;
        if (v.owner[q] == NO_OWNER) continue;
        const uint32_t m = v.map[t];
        if ((m >> 30) != 1u) continue;
        const uint32_t c = v.cres[(size_t)q * CH_MAX + ((m >> NBB) & 7u)], nb = (m & NBM) + (c & NBM);
        v.map[t] = nb >= (1u << NBB) ? (3u << 30) : ((c & ~NBM) | nb);
    }
}
// the pieces the walks of round `round` - 1 asked for, with the asking candidate's tables
__global__ __launch_bounds__(64 * SPEC_W) void k_any_spec2(Args a, uint32_t round) {
    const View v = view(a);
    __shared__ SpecLds L;
    if (!v.run) return;
    const uint32_t nreq = min(v.ctl[A_NREQ], a.maxreq);
    if (nreq == 0u) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t nw = gridDim.x * SPEC_W, g = blockIdx.x * SPEC_W + wv;
    uint32_t cur = NO_OWNER;
    for (uint32_t r = 0; r < nreq; r++) {
        const Req rq = v.req[r];
        if (rq.round + 1u != round) continue;
        for (uint32_t k = g; k < rq.q1 - rq.q0; k += nw) {
            if (rq.node != cur) { load_tab(&L.t[wv], &v.tab[rq.node], lane); cur = rq.node; }
            spec_piece(v, &L.t[wv], L.win[wv], rq.q0 + k, v.map2 + (size_t)(rq.slot + k) * 64u, lane);
        }
    }
}

// ================================================================================================ 3. the chain of blocks
__device__ __forceinline__ uint32_t find_rank(const uint32_t* shdr, uint32_t n, uint32_t hdr) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (shdr[m] < hdr) lo = m + 1u; else hi = m; }
    return lo < n && shdr[lo] == hdr ? lo : N_BAD;
}
// ONE WAVE per node.  The walk is one serial chain, kept identical in all 64 lanes (every lane computes the same values from the same
// addresses; what has a side effect -- a counter, a list entry, the node's result -- is done by lane 0 and broadcast); what the other
// lanes are for: the maps of 64 pieces at a time are staged in LDS by the whole wave (16 KB, coalesced), so a step of the chain costs
// an LDS round trip instead of two dependent global loads (one lane per node straight from memory: 260 us for the ~235 pieces of a
// 16 K-symbol block)
constexpr uint32_t WALK_STAGE = 64;           // pieces staged at a time
__device__ __forceinline__ void walk_node(const Args& a, const View& v, uint32_t round, uint32_t n, uint32_t i, uint32_t lane, uint32_t* smap, uint32_t* sown, Tab* ltp);
__global__ __launch_bounds__(64) void k_any_walk(Args a, uint32_t round) {
    const View v = view(a);
    __shared__ uint32_t smap[WALK_STAGE * 64];
    __shared__ uint32_t sown[WALK_STAGE];
    __shared__ Tab lt;                       // the tables the lane-serial parts decode with (the block's; later the fixed code's)
    if (!v.run) return;
    const uint32_t n = min(v.ctl[A_NBLK], a.maxb);
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i <= n; i += gridDim.x) walk_node(a, v, round, n, i, lane, smap, sown, &lt);
}
// node i: candidate rank i < n, or the pseudo-node i == n (stored at index maxb)
__device__ __forceinline__ void walk_node(const Args& a, const View& v, uint32_t round, uint32_t n, uint32_t i, uint32_t lane, uint32_t* smap, uint32_t* sown, Tab* ltp) {
    Tab& lt = *ltp;
    const uint32_t nbits = 8u * v.zn;
    const int32_t isize = (int32_t)v.zn - 1;
    const uint32_t len_mask = a.obsize ? ((1u << (31u - (uint32_t)__builtin_clz(a.obsize))) - 1u) : 0xFFFFu;     // deflate.py:329,:714
    const uint32_t node_id = i < n ? i : a.maxb;
    if (round != 0u && v.node[node_id].ok != 3u) return;              // later rounds: the walks that wait for pieces of their own
    uint64_t nb = 0;                      // bytes of this node so far
    uint32_t ebit = 16u, fin = 0u, next = N_BAD, why = 0u;
    bool ok = true;
    Bits rd;
    auto bump = [&](uint32_t* ctr, uint32_t by) -> uint32_t {        // atomicAdd by lane 0, the old value in every lane
        uint32_t k = 0;
        if (lane == 0u) k = atomicAdd(ctr, by);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    };
    // (endpos / made: where the item ends and the bytes it makes, as the walk knows them)
    auto add_xitem = [&](uint32_t start, uint32_t limit, uint32_t slot, uint64_t nb0, uint32_t endpos, uint32_t made) {
        const uint32_t k = bump(&v.ctl[A_NX], 1u);
        if (k < a.maxx) { if (lane == 0u) v.xitem[k] = XItem{start, limit, (uint32_t)nb0, node_id | (slot << 16), endpos, made}; }
        else { if (lane == 0u) v.ctl[A_OVER] = 1u; ok = false; }
    };
    // bits [rd.pos, limit) with the tables of `slot`, counting bytes; -> true at the end-of-block code (consumed)
    auto run_to = [&](const Tab* t, uint32_t limit, bool& eob) {
        eob = false;
        while (ok && rd.pos < limit) {
            const Token k = token_at(t, rd);
            if (k.kind == 3u) { ok = false; why |= W_WALK_TOK; break; }
            if (k.kind == 2u) { eob = true; break; }
            nb += k.kind == 0u ? 1u : k.length;
        }
    };
    if (i < n) {
        const Blk b = v.blk[v.sidx[i]];
        fin = b.fin;
        bool eob = false;
        uint32_t q, pos, rq0 = 0, rq1 = 0, rslot = 0;
        if (round == 0u) {
            rd.init(v.z, v.zn, b.pay);
            q = b.pay / v.pb;
            if (b.pay % v.pb != 0u) {                                   // the first, partial piece: decoded here
                const uint32_t limit = (q + 1u) * v.pb;
                load_tab(&lt, &v.tab[i], lane);
                wave_lds_order();
                __builtin_amdgcn_wave_barrier();
                run_to(&lt, limit, eob);
                add_xitem(b.pay, limit, i, 0u, rd.pos, (uint32_t)nb);
                q++;
            }
            pos = rd.pos;
        } else {
            const NState st = v.nstate[i];
            q = st.q; pos = st.pos; nb = st.nb;
            const Req rq = v.req[st.req];
            rq0 = rq.q0; rq1 = rq.q1; rslot = rq.slot;
        }
        uint32_t sq0 = 0;                                             // first staged piece
        bool staged = false;
        while (ok && !eob) {                                          // whole pieces: through the maps
            if (q >= a.nchunks || pos - q * v.pb >= 64u) { ok = false; why |= W_WALK_OWNER; break; }
            const uint32_t e = pos - q * v.pb;
            if (!staged || q - sq0 >= WALK_STAGE) {
                sq0 = q; staged = true;
                const uint32_t cnt = min(WALK_STAGE, a.nchunks - sq0);
                __builtin_amdgcn_wave_barrier();
                const tok::u32x4* src = reinterpret_cast<const tok::u32x4*>(v.map + (size_t)sq0 * 64u);
                tok::u32x4* dst = reinterpret_cast<tok::u32x4*>(smap);
                tok::u32x4 t[WALK_STAGE / 4];                         // (all 16 loads in flight at once; the rows behind the last piece: re-read of the last one)
#pragma unroll
                for (uint32_t k = 0; k < WALK_STAGE / 4u; k++) t[k] = src[min(k * 64u + lane, cnt * 16u - 1u)];
#pragma unroll
                for (uint32_t k = 0; k < WALK_STAGE / 4u; k++) dst[k * 64u + lane] = t[k];
                if (lane < cnt) sown[lane] = v.owner[sq0 + lane];
                wave_lds_order();
                __builtin_amdgcn_wave_barrier();
            }
            uint32_t m;
            const uint32_t own = sown[q - sq0];
            if (own == i) {
                m = smap[(q - sq0) * 64u + e];
                if (lane == 0u) { v.pent[q] = (uint8_t)e; v.prel[q] = (uint32_t)nb; v.pnode[q] = (uint16_t)i; v.pmap[q] = m; }
            } else if (q >= rq0 && q < rq1) {                         // a piece decoded for this block on request: an item of its own
                m = v.map2[(size_t)(rslot + (q - rq0)) * 64u + e];
                const uint32_t k_ = m >> 30, o_ = (m >> NBB) & OFM;
                add_xitem(q * v.pb + e, (q + 1u) * v.pb, i, nb, k_ == 0u ? (q + 1u) * v.pb + o_ : q * v.pb + o_, m & NBM);
            } else {
                // decoded for another candidate (a false positive inside this block, if this block is a true one): ask for the pieces up
                // to the next candidate behind that one with this block's tables, go on in the next round
                const uint32_t f = own;
                if (f == NO_OWNER || round + 1u >= WALK_ROUNDS) { ok = false; why |= W_WALK_OWNER; break; }
                uint32_t qe = f + 1u < n ? min((v.spay[f + 1u] + v.pb - 1u) / v.pb, a.nchunks) : a.nchunks;
                qe = min(qe, q + REQ_MAX);
                const uint32_t slot = bump(&v.ctl[A_NREQP], qe - q), r = bump(&v.ctl[A_NREQ], 1u);
                if (slot + (qe - q) > a.mapcap || r >= a.maxreq) { ok = false; why |= W_WALK_OWNER; break; }
                if (lane == 0u) {
                    v.req[r] = Req{i, q, qe, slot, round};
                    v.nstate[i] = NState{q, pos, (uint32_t)nb, r};
                    v.node[node_id] = Node{N_BAD, 0u, 3u, 0u};
                }
                return;
            }
            nb += m & NBM;
            const uint32_t kind = m >> 30, off = (m >> NBB) & OFM;
            if (kind == 0u) { pos = (q + 1u) * v.pb + off; q++; }
            else if (kind == 2u) { pos = q * v.pb + off; eob = true; }
            else { ok = false; why |= W_WALK_MAP; }
            if (nb > 0xFFFFFFFFull) ok = false;
        }
        ebit = pos;
    }
    // behind the block: the blocks that cannot be found by search, up to the next dynamic header
    uint32_t guard = 0;
    while (ok) {
        if (fin) { next = N_END; break; }
        if (ebit + 3u > nbits || ++guard > 0x100000u) { ok = false; why |= W_WALK_HDR; break; }
        rd.init(v.z, v.zn, ebit);
        const uint32_t f = (uint32_t)rd.bb & 1u, ty = ((uint32_t)rd.bb >> 1) & 3u;
        if (ty == 2u) { next = find_rank(v.shdr, n, ebit); ok = next != N_BAD; if (!ok) why |= W_NEXT; break; }
        if (ty == 3u) { ok = false; why |= W_WALK_HDR; break; }
        if (ty == 0u) {
            // stored (deflate.py:709-717, :1603-1626): LEN sits `skip` bits behind the header's first bit, NLEN is not checked (D2)
            const uint32_t dio = ebit & 7u;
            uint32_t skip = 8u - dio;
            if (skip <= 2u) skip = 16u - dio;
            const uint32_t length = (uint32_t)(rd.bb >> skip) & 0xFFFFu & len_mask;
            const uint32_t p0 = (ebit + skip + 32u) >> 3;
            const uint32_t i_noeof = (int32_t)p0 >= isize ? 0u : (uint32_t)isize - p0;
            if (length > i_noeof || (int32_t)(p0 + length) >= isize) { ok = false; why |= W_WALK_STORED; break; }
            if (length) {
                const uint32_t k = bump(&v.ctl[A_NS], 1u);
                if (k < a.maxs) { if (lane == 0u) v.sitem[k] = SItem{p0, length, (uint32_t)nb, node_id}; }
                else { if (lane == 0u) v.ctl[A_OVER] = 1u; ok = false; break; }
            }
            nb += length;
            ebit = 8u * (p0 + length);
        } else {
            // fixed: decoded by this lane, an item per piece it touches
            rd.take(3u);
            const uint32_t stop = rd.pos + FIX_MAX_BITS;
            bool eob = false;
            __builtin_amdgcn_wave_barrier();
            load_tab(&lt, &v.tab[a.maxb], lane);
            wave_lds_order();
            __builtin_amdgcn_wave_barrier();
            while (ok && !eob) {
                const uint32_t limit = (rd.pos / v.pb + 1u) * v.pb, st0 = rd.pos;
                if (rd.pos >= stop) { ok = false; why |= W_WALK_FIX; break; }
                const uint64_t nb0 = nb;
                run_to(&lt, limit, eob);
                add_xitem(st0, limit, a.maxb, nb0, rd.pos, (uint32_t)(nb - nb0));
            }
            ebit = rd.pos;
        }
        if (nb > 0xFFFFFFFFull) { ok = false; break; }
        fin = f;
    }
    if (lane == 0u) v.node[node_id] = Node{next, (uint32_t)nb, ok ? 1u : 0u, why};        // (obase of a failed node: why its walk failed -- read by k_any_rank if it is on the chain)
}

// the successors from the pseudo-node on: the true chain, the output position of every block on it, the total
// (One thread following the successors: 0.86 ms for the 5414 blocks of a 256 MiB stream, 45 us for the 340 of 16 MiB.  Now the order of
//  the chain by pointer doubling -- A[p + 2^r] = J_r[A[p]], J_{r+1} = J_r o J_r: the node at every position of the chain after
//  log2(n) rounds --, the output positions by a scan along that order.  J and A live in the candidate list, which is dead by now.)
constexpr uint32_t RANK_T = 1024;
__global__ __launch_bounds__(RANK_T) void k_any_rank(Args a) {
    const View v = view(a);
    extern __shared__ uint32_t nl[];                                // [maxb + 1][2]: next, bytes (a failed node: N_BAD, why its walk failed)
    __shared__ unsigned long long wtot[RANK_T / 64];
    __shared__ uint32_t s_len, s_end;
    if (!v.run) return;
    const uint32_t n = min(v.ctl[A_NBLK], a.maxb), tid = threadIdx.x;
    if (v.ctl[A_OVER] != 0u) { if (tid == 0u) give_up(v); return; }
    // nodes 0 .. n-1: the candidates; n: the pseudo-node (the stream's first header); E / B: the stream ends / no valid successor
    const uint32_t E = n + 1u, B = n + 2u, M = n + 3u;
    uint32_t* J0 = v.cand;                                          // (candcap >= 1024 + 64 * maxb words: room for 3 * (maxb + 3))
    uint32_t* J1 = v.cand + M;
    uint32_t* A = v.cand + 2u * M;
    for (uint32_t k = tid; k < M; k += RANK_T) {
        uint32_t j = k;                                             // (E and B lead to themselves)
        if (k <= n) {
            const Node nd = v.node[k < n ? k : a.maxb];
            const uint32_t nx = nd.ok == 1u ? nd.next : N_BAD;
            nl[2u * k] = nx; nl[2u * k + 1u] = nd.ok == 1u ? nd.nbytes : nd.obase;
            j = nx == N_END ? E : nx < n ? nx : B;
        }
        J0[k] = j;
        A[k] = B;
    }
    if (tid == 0u) { A[0] = n; s_len = 0xFFFFFFFFu; s_end = 0u; }
    __syncthreads();
    for (uint32_t len = 1u; len < M; len <<= 1) {
        for (uint32_t p = tid; p < len && p + len < M; p += RANK_T) A[p + len] = J0[A[p]];
        for (uint32_t k = tid; k < M; k += RANK_T) J1[k] = J0[J0[k]];
        __syncthreads();
        uint32_t* t = J0; J0 = J1; J1 = t;
    }
    // the chain: A[0 .. Lc), Lc = the first position that holds E or B
    for (uint32_t p = tid; p < M; p += RANK_T) { if (A[p] >= E) atomicMin(&s_len, p); }
    __syncthreads();
    const uint32_t Lc = s_len;
    bool good = true;
    if (Lc == 0xFFFFFFFFu) { good = false; if (tid == 0u) atomicOr(&v.ctl[A_WHY], (uint32_t)W_CYCLE); }      // never ends: a cycle
    else if (A[Lc] == B) {
        good = false;
        if (tid == 0u) {
            const uint32_t last = A[Lc - 1u];                       // (Lc >= 1: A[0] is the pseudo-node)
            atomicOr(&v.ctl[A_WHY], nl[2u * last] == N_BAD ? ((uint32_t)W_NODE | nl[2u * last + 1u]) : (uint32_t)W_CYCLE);
        }
    }
    unsigned long long total = 0;
    if (good) {
        // output positions: an exclusive scan of the nodes' bytes along the chain (a contiguous run of positions per thread)
        const uint32_t per = (Lc + RANK_T - 1u) / RANK_T, p0 = min(tid * per, Lc), p1 = min(p0 + per, Lc);
        unsigned long long mine = 0;
        for (uint32_t p = p0; p < p1; p++) mine += nl[2u * A[p] + 1u];
        unsigned long long incl = mine;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) {
            const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, ofs, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), ofs, 64);
            if ((tid & 63u) >= (uint32_t)ofs) incl += ((unsigned long long)hi << 32) | lo;
        }
        if ((tid & 63u) == 63u) wtot[tid >> 6] = incl;
        __syncthreads();
        unsigned long long before = incl - mine;
        for (uint32_t w = 0; w < (tid >> 6); w++) before += wtot[w];
        for (uint32_t w = 0; w < RANK_T / 64u; w++) total += wtot[w];
        if (total > (unsigned long long)a.cap || total > (unsigned long long)a.srcn) { good = false; if (tid == 0u) atomicOr(&v.ctl[A_WHY], (uint32_t)W_CAP); }
        else {
            for (uint32_t p = p0; p < p1; p++) {
                const uint32_t k = A[p];
                Node* nd = &v.node[k < n ? k : a.maxb];
                nd->obase = (uint32_t)before; nd->ok = 2u;
                before += nl[2u * k + 1u];
            }
        }
    }
    if (tid != 0u) return;
    const uint32_t nx_items = v.ctl[A_NX], ns_items = v.ctl[A_NS];
    if (nx_items > a.maxx || ns_items > a.maxs) { good = false; atomicOr(&v.ctl[A_WHY], (uint32_t)W_ITEMS); }
    if (!good) { give_up(v); return; }
    v.ctl[C_TOTAL] = (uint32_t)total;
    v.ctl[C_NUSED] = a.nchunks + nx_items;
    v.ctl[C_FNUSED] = a.nchunks + nx_items;
    v.ctl[C_OK] = 1u;
}

// ================================================================================================ 4. the real decode
// one LANE per item: the pieces the walks went through (entry offset and position from the walk) and the extra items; the reference's
// checks (deflate.py:1409-1445, :1519-1591, :1447-1517, :1600) are all evaluated -- WHICH one failed does not matter here, any failure
// hands the stream to the serial decoder, which reports the reference's status in the reference's order
__global__ __launch_bounds__(64) void k_any_tokens(Args a) {
    const View v = view(a);
    __shared__ Tab lt;                       // the tables of the wave's items when they all belong to one block (most waves: 64 consecutive pieces)
    if (!v.run || v.ctl[C_OK] == 0u) return;
    const uint32_t nitems = v.ctl[C_NUSED];
    const int32_t isize = (int32_t)v.zn - 1;
    const uint32_t obsize = a.obsize ? a.obsize : 32768u;
    const uint32_t lane = threadIdx.x;
    bool bad = false;
    uint32_t cur = 0xFFFFFFFFu;
    for (uint32_t i0 = blockIdx.x * 64u; i0 < nitems; i0 += gridDim.x * 64u) {
        const uint32_t i = i0 + lane;
        uint32_t start = 0, limit = 0, rel = 0, nid = NO_OWNER, slot = 0, want_end = 0, want_made = 0;
        bool have = i < nitems;
        if (have) {
            if (i < a.nchunks) {
                nid = v.pnode[i];
                if (nid != NO_OWNER) {
                    start = i * v.pb + v.pent[i]; limit = (i + 1u) * v.pb; rel = v.prel[i]; slot = nid;
                    const uint32_t m = v.pmap[i];
                    want_made = m & NBM;
                    want_end = (m >> 30) == 0u ? limit + ((m >> NBB) & OFM) : i * v.pb + ((m >> NBB) & OFM);
                }
            } else {
                const XItem x = v.xitem[i - a.nchunks];
                start = x.start; limit = x.limit; rel = x.rel; nid = x.node_slot & 0xFFFFu; slot = x.node_slot >> 16;
                want_end = x.endpos; want_made = x.nbytes;
            }
        }
        Node nd = Node{0u, 0u, 0u, 0u};
        if (have && nid != NO_OWNER) nd = v.node[nid];
        have = have && nid != NO_OWNER && nd.ok == 2u;                // (not on the chain: nothing to do)
        const uint64_t hm = ballot64(have);
        if (hm == 0ull) continue;
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)slot, (int)__builtin_ctzll(hm));
        const bool uniform = ballot64(have && slot != s0) == 0ull;
        if (uniform && s0 != cur) {
            __builtin_amdgcn_wave_barrier();
            load_tab(&lt, &v.tab[s0], lane);
            wave_lds_order();
            __builtin_amdgcn_wave_barrier();
            cur = s0;
        }
        uint32_t P = nd.obase + rel, nt = 0, icur = i;
        uint32_t* tk = v.tok + (size_t)i * a.tcap;
        if (have) v.opos[i] = P;
        Bits rd;
        rd.init(v.z, v.zn, have ? start : 16u);
        uint4 q = make_uint4(0, 0, 0, 0);                            // the last (up to) four tokens
        auto decode = [&](const Tab* t) {
            while (have && rd.pos < limit) {
                const uint32_t p0 = rd.pos;
                const Token k = token_at(t, rd);
                bool f = k.kind == 3u || (int32_t)((p0 + k.nb) >> 3) > isize - 3;                  // no code; NO EOF (deflate.py:1535-1539)
                if (k.kind == 2u) { bad |= f; break; }
                const uint32_t made = k.kind == 0u ? 1u : k.length;
                f |= (uint64_t)P + made > a.cap;
                f |= k.kind == 1u && (k.dist > P || k.dist > obsize || (int32_t)((p0 + k.used) >> 3) >= isize - 2);     // D8; COPY hold (:1600)
                if (!f && nt >= a.tcap) {
                    // the list is full (tokens of two or three bits: long runs): the rest of the item goes on as an item of its own -- the
                    // emit takes items in any order, each with its own position
                    const uint32_t k2 = atomicAdd(&v.ctl[A_NX], 1u);
                    if (k2 >= a.maxx) { f = true; atomicOr(&v.ctl[A_WHY], (uint32_t)W_TCAP); }
                    else {
                        v.ntok[icur] = nt;
                        icur = a.nchunks + k2; nt = 0;
                        tk = v.tok + (size_t)icur * a.tcap;
                        v.opos[icur] = P;
                        atomicMax(&v.ctl[C_NUSED], icur + 1u); atomicMax(&v.ctl[C_FNUSED], icur + 1u);
                    }
                }
                if (f) { bad = true; atomicOr(&v.ctl[A_WHY], (uint32_t)W_TOKEN); break; }
                {   // (four tokens per store, as k_par_tokens: tcap is a multiple of 4, so a list that is full has just been written)
                    const uint32_t w = k.kind == 0u ? (TOK_LIT | k.lit) : (k.length | (k.dist << 9)), k4 = nt & 3u;
                    q.x = k4 == 0u ? w : q.x; q.y = k4 == 1u ? w : q.y; q.z = k4 == 2u ? w : q.z; q.w = k4 == 3u ? w : q.w;
                    if (k4 == 3u) *reinterpret_cast<uint4*>(tk + (nt & ~3u)) = q;
                    nt++;
                }
                P += made;
            }
        };
        if (uniform) decode(&lt);
        else decode(&v.tab[slot]);
        // the item must end where the walk -- the speculative maps -- said it would, with the bytes it said: whatever goes wrong in the
        // speculation can cost the fallback, never a wrong byte
        if (have && !bad && (rd.pos != want_end || P - (nd.obase + rel) != want_made)) { bad = true; atomicOr(&v.ctl[A_WHY], (uint32_t)W_VERIFY); }
        if (have && (nt & 3u) != 0u) *reinterpret_cast<uint4*>(tk + (nt & ~3u)) = q;
        if (have) v.ntok[icur] = nt;
    }
    if (ballot64(bad) != 0ull && lane == 0u) give_up(v);
}

// the stored blocks: straight copies; their bytes are there (no marker)
__global__ __launch_bounds__(256) void k_any_stored(Args a) {
    const View v = view(a);
    if (!v.run || v.ctl[C_OK] == 0u) return;
    const uint32_t ns = min(v.ctl[A_NS], a.maxs);
    for (uint32_t k = blockIdx.x; k < ns; k += gridDim.x) {
        const SItem s = v.sitem[k];
        const Node nd = v.node[s.node];
        if (nd.ok != 2u) continue;
        const uint32_t P = nd.obase + s.rel;
        for (uint32_t j = threadIdx.x; j < s.len; j += 256u) { v.out[P + j] = v.z[s.src + j]; v.srcA[P + j] = NONE; }
    }
}

// a stream whose FIRST block is fixed with more blocks behind it (BFINAL = 0): does that block end within FIRST_FIX_BITS, in front of a
// stored or a dynamic block?  One lane reads the fixed code by its arithmetic (RFC 1951 3.2.6: 7 bits 256..279, 8 bits 0..143 and
// 280..287, 9 bits 144..255; distances 5 bits) -- lengths only, nothing is produced or checked: whatever it gets wrong costs the
// fallback (the real decode checks everything), never a byte
__device__ bool short_fixed_block_first(const uint8_t* z, uint32_t zn) {
    const uint32_t nbits = 8u * zn;
    Bits rd;
    rd.init(z, zn, 16u + 3u);
    while (rd.pos < 16u + 3u + FIRST_FIX_BITS && rd.pos + 7u <= nbits) {
        rd.refill();
        const uint32_t r9 = __builtin_bitreverse32((uint32_t)rd.bb) >> 23;        // the next 9 bits as an MSB-first code
        uint32_t sym, n;
        if ((r9 >> 2) < 24u) { sym = 256u + (r9 >> 2); n = 7u; }
        else if ((r9 >> 1) < 0xC0u) { sym = (r9 >> 1) - 0x30u; n = 8u; }
        else if ((r9 >> 1) < 0xC8u) { sym = 280u + (r9 >> 1) - 0xC0u; n = 8u; }
        else { sym = 144u + r9 - 0x190u; n = 9u; }
        if (sym < 256u) { rd.take(n); continue; }
        if (sym == 256u) {
            rd.take(n);
            if (rd.pos + 3u > nbits) return false;
            rd.refill();
            const uint32_t btype = ((uint32_t)rd.bb >> 1) & 3u;
            return btype == 0u || btype == 2u;
        }
        if (sym > 285u) return false;
        const uint32_t ls = sym - 257u, leb = (ls < 8u || ls == 28u) ? 0u : (ls >> 2) - 1u;
        rd.take(n + leb);
        rd.refill();
        const uint32_t d = __builtin_bitreverse32((uint32_t)rd.bb) >> 27;
        if (d > 29u) return false;
        rd.take(5u + (d < 4u ? 0u : (d >> 1) - 1u));
    }
    return false;
}

__global__ __launch_bounds__(64) void k_any_zero(Args a) {
    const uint32_t s = blockIdx.y;
    uint32_t* ctl = reinterpret_cast<uint32_t*>(a.ws + (size_t)s * a.stride);
    uint32_t open = 0u;
    if (threadIdx.x == 0u) {
        const uint8_t* z;
        uint32_t zn;
        if (a.in_off) {
            const uint64_t o0 = a.in_off[s], n64 = a.in_off[s + 1u] - o0;
            z = a.z + o0; zn = n64 > (uint64_t)a.zn ? 0u : (uint32_t)n64;
        } else { z = a.z + (uint64_t)s * a.in_pitch; zn = a.zn; }
        if (zn >= 5u && (z[2] & 7u) == 2u) open = short_fixed_block_first(z, zn) ? 1u : 0u;      // BFINAL = 0, BTYPE = 01
    }
    open = (uint32_t)__builtin_amdgcn_readfirstlane((int)open);
    if (threadIdx.x < par::C_WORDS) ctl[threadIdx.x] = threadIdx.x == (uint32_t)A_OPEN ? open : 0u;
}

// streams below ANY_MIN bytes stay with the serial decoder: the chain of launches takes 0.55 ms whatever the stream, one wave 0.2 ms per
// KB of compressed bytes (0.96 ms at 4.7 KB, 1.64 at 8.9 KB).  Streams per call (any_work_bytes): the more streams, the longer each
// must be for the chain to beat a wave per stream (profiles/r06_any_batches.txt)
constexpr uint32_t ANY_MIN = 4096;
__host__ inline uint32_t any_batch_max(uint32_t in_len) { return in_len >= (96u << 10) ? 1024u : in_len >= 16384u ? 512u : in_len >= 8192u ? 256u : 64u; }
struct Lay {
    uint32_t nchunks, candcap, maxb, maxx, maxs, tcap, maxreq, mapcap, pb;
    size_t o_cand, o_blk, o_blen, o_shdr, o_spay, o_sidx, o_tab, o_owner, o_map, o_pent, o_prel, o_pnode, o_node, o_xitem, o_sitem,
           o_opos, o_ntok, o_tok, o_mext, o_req, o_map2, o_nstate, o_cpos, o_cres, o_nch, o_pmap, bytes;
};
static Lay lay_of(uint32_t zn) {
    Lay L;
    memset(&L, 0, sizeof(L));
    const uint64_t nbits = 8ull * zn;
    L.pb = zn >= (4u << 20) ? 2048u : 1024u;       // (1024 for all: base64 / hex / float32 / text at 16 MiB 3.15 / 2.52 / 3.13 / 2.06 ms against 2.62 / 2.16 / 2.59 / 1.83, 256 MiB 21.6 against 18.7)
    L.nchunks = (uint32_t)((nbits + L.pb - 1u) / L.pb);
    L.candcap = (uint32_t)(nbits / 128u) + 1024u;                   // (0.085 % of arbitrary bit positions pass k_any_find; periodic streams far more)
    L.maxb = zn / 1024u < 64u ? 64u : zn / 1024u > 8000u ? 8000u : zn / 1024u;      // (zlib's blocks hold 16 K symbols, ~20 KB; small memLevels and flushes make many small ones)
    L.maxreq = 2u * L.maxb;
    L.mapcap = L.nchunks / 16u + 1024u;                             // pieces decoded a second time, on request (false positives: ~one per 4 MB, ~120 pieces each)
    if (L.mapcap > 2u * L.nchunks + 64u) L.mapcap = 2u * L.nchunks + 64u;      // (a small stream cannot ask for more than its pieces, once per round that asks)
    // extra items: every piece decoded on request is one, + a partial piece per block, + the fixed blocks' pieces -- and one per `tcap`
    // tokens a piece holds beyond its own list.  A piece of literals with 6-bit codes (base64, float data) is 341 tokens per 2048 bits:
    // with lists of 256 tokens EVERY piece of such a stream needed a second list and the stream fell back to the one-wave decoder
    // (7.6 MB/s; round 6, first cut).  So a list holds a token per 2.67 bits of its piece, and an eighth of the pieces may go on beyond
    // that (items are launched as workgroups whether used or not: as many more as there are pieces cost the other streams 8 %)
    L.maxx = 2u * L.maxb + 256u + L.mapcap + L.nchunks / 8u;
    L.maxs = zn / 512u + 256u;
    static_assert((3u * 1024u / 8u) % 4u == 0u, "token lists are written 16 bytes at a time");
    L.tcap = 3u * L.pb / 8u;                                        // (hex text: 16 symbols, codes of 4 bits and a few of 5 -- a list per 4 bits overflowed in every third piece)
    size_t off = 256;                                               // the control words in front
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255u) & ~(size_t)255u; return o; };
    const size_t items = (size_t)L.nchunks + L.maxx;
    L.o_cand = take(4u * (size_t)L.candcap); L.o_blk = take(sizeof(Blk) * L.maxb); L.o_blen = take(320u * (size_t)L.maxb);
    L.o_shdr = take(4u * L.maxb); L.o_spay = take(4u * L.maxb); L.o_sidx = take(4u * L.maxb);
    L.o_tab = take(sizeof(Tab) * ((size_t)L.maxb + 1u)); L.o_owner = take(2u * (size_t)L.nchunks); L.o_map = take(256u * (size_t)L.nchunks);
    L.o_pent = take(L.nchunks); L.o_prel = take(4u * (size_t)L.nchunks); L.o_pnode = take(2u * (size_t)L.nchunks);
    L.o_node = take(sizeof(Node) * ((size_t)L.maxb + 1u)); L.o_xitem = take(sizeof(XItem) * L.maxx); L.o_sitem = take(sizeof(SItem) * L.maxs);
    L.o_opos = take(4u * items); L.o_ntok = take(4u * items); L.o_tok = take(4u * (size_t)L.tcap * items); L.o_mext = take(4u * items);
    L.o_req = take(sizeof(Req) * L.maxreq); L.o_map2 = take(256u * (size_t)L.mapcap); L.o_nstate = take(sizeof(NState) * L.maxb);
    L.o_cpos = take(4u * 8u * (size_t)L.nchunks); L.o_cres = take(4u * 8u * (size_t)L.nchunks); L.o_nch = take(L.nchunks); L.o_pmap = take(4u * (size_t)L.nchunks);
    L.bytes = off;
    return L;
}

}  // namespace any

// 0 = this chain is not launched for the call.  `nstreams`: the streams of the whole call.  The chain inflates 2.4 .. 3.7 GB/s of streams
// of 8 .. 16 KiB, 7 .. 9 of 48 .. 64 KiB, 15 of 256 KiB, 19 of 1 MiB whatever their number, a wave per stream (what its give-ups get)
// takes ~90 us per KiB of ONE stream up to a few thousand of them: from ~200 streams of 8 KiB, ~400 of 16 KiB, ~550 of 48 KiB, ~750 of
// 64 KiB, ~1300 of 256 KiB, ~1700 of 1 MiB on the waves win (profiles/r06_any_batches.txt) -- above any_batch_max() a batch of zlib
// streams is decoded as it was before round 6
size_t any_work_bytes(uint32_t in_len, uint64_t out_pitch, uint32_t flags, uint32_t nstreams) {
    (void)out_pitch;
#ifdef HDLZ_ANY_OFF                            // (A/B build: what the chain costs a stream that is not its)
    return 0;
#endif
    // (those builds read every block as fixed / stop at the first one; the hint: the caller knows the streams are single fixed blocks)
    if (in_len < any::ANY_MIN || (flags & (HDLZ_INFLATE_ASSUME_FIXED | HDLZ_INFLATE_ONEBLOCK | HDLZ_INFLATE_ONE_FIXED_BLOCK))) return 0;
    if (nstreams > any::any_batch_max(in_len)) return 0;
    return any::lay_of(in_len).bytes;
}

hipError_t launch_inflate_any(const InflateArgs& a, uint32_t nstr, uint8_t* ws, size_t ws_stride, size_t ws_off, size_t sa_off,
                              uint32_t srcn, uint32_t cap, hipStream_t stream, uint32_t* passes_out) {
    using namespace any;
    const Lay L = lay_of(a.in_len);
    Args g;
    memset(&g, 0, sizeof(g));
    g.z = a.in; g.zn = a.in_len; g.flags = a.flags; g.obsize = a.obsize; g.out = a.out; g.cap = cap; g.srcn = srcn;
    g.in_pitch = a.in_pitch; g.out_pitch = a.out_pitch; g.in_off = a.in_off;
    g.ws = ws + ws_off; g.stride = ws_stride; g.srcA = reinterpret_cast<uint32_t*>(ws + sa_off);
    g.nchunks = L.nchunks; g.candcap = L.candcap; g.maxb = L.maxb; g.maxx = L.maxx; g.maxs = L.maxs; g.tcap = L.tcap;
    g.maxreq = L.maxreq; g.mapcap = L.mapcap; g.pb = L.pb; g.o_req = L.o_req; g.o_map2 = L.o_map2; g.o_nstate = L.o_nstate;
    g.o_cpos = L.o_cpos; g.o_cres = L.o_cres; g.o_nch = L.o_nch; g.o_pmap = L.o_pmap;
    g.o_cand = L.o_cand; g.o_blk = L.o_blk; g.o_blen = L.o_blen; g.o_shdr = L.o_shdr; g.o_spay = L.o_spay; g.o_sidx = L.o_sidx; g.o_tab = L.o_tab;
    g.o_owner = L.o_owner; g.o_map = L.o_map; g.o_pent = L.o_pent; g.o_prel = L.o_prel; g.o_pnode = L.o_pnode; g.o_node = L.o_node;
    g.o_xitem = L.o_xitem; g.o_sitem = L.o_sitem; g.o_opos = L.o_opos; g.o_ntok = L.o_ntok; g.o_tok = L.o_tok; g.o_mext = L.o_mext;
    const uint32_t nitems = L.nchunks + L.maxx;
    auto gx = [&](uint64_t work, uint32_t cap_) { const uint32_t c = par::grid_cap(work, nstr); return (unsigned)(c > cap_ ? cap_ : c); };
    hipLaunchKernelGGL(k_any_zero, dim3(1, nstr), dim3(64), 0, stream, g);
    hipLaunchKernelGGL(k_any_find, dim3(gx((a.in_len + FIND_T - 1u) / FIND_T, 4096u), nstr), dim3(FIND_T), 0, stream, g);
    hipLaunchKernelGGL(k_any_headers, dim3(gx((L.candcap + 63u) / 64u, 1280u), nstr), dim3(64), 0, stream, g);
    hipLaunchKernelGGL(k_any_sort, dim3((L.maxb + SORT_T - 1u) / SORT_T, nstr), dim3(SORT_T), 4u * L.maxb, stream, g);
    hipLaunchKernelGGL(k_any_tables, dim3(gx(L.maxb + 1u, 1024u), nstr), dim3(TAB_T), 0, stream, g);
    hipLaunchKernelGGL(k_any_owner, dim3(gx(L.maxb, 1024u), nstr), dim3(64), 0, stream, g);
    hipLaunchKernelGGL(k_any_spec, dim3(gx((L.nchunks + SPEC_W - 1u) / SPEC_W, 1536u), nstr), dim3(64 * SPEC_W), 0, stream, g);
    hipLaunchKernelGGL(k_any_tail, dim3(gx(((size_t)L.nchunks * CH_MAX + 63u) / 64u, 8192u), nstr), dim3(64), 0, stream, g);
    hipLaunchKernelGGL(k_any_resolve, dim3(gx(((size_t)L.nchunks * 64u + 255u) / 256u, 4096u), nstr), dim3(256), 0, stream, g);
    hipLaunchKernelGGL(k_any_walk, dim3(gx(L.maxb + 1u, 8192u), nstr), dim3(64), 0, stream, g, 0u);
    for (uint32_t round = 1; round < WALK_ROUNDS; round++) {        // (return at once when no walk met a false positive)
        hipLaunchKernelGGL(k_any_spec2, dim3(gx(256u, 256u), nstr), dim3(64 * SPEC_W), 0, stream, g, round);
        hipLaunchKernelGGL(k_any_walk, dim3(gx(L.maxb + 1u, 8192u), nstr), dim3(64), 0, stream, g, round);
    }
    hipLaunchKernelGGL(k_any_rank, dim3(1, nstr), dim3(RANK_T), 8u * (L.maxb + 1u), stream, g);
    hipLaunchKernelGGL(k_any_tokens, dim3(gx((nitems + 63u) / 64u, 8192u), nstr), dim3(64), 0, stream, g);
    hipLaunchKernelGGL(k_any_stored, dim3(gx(L.maxs, 1024u), nstr), dim3(256), 0, stream, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the bytes: the kernels of hdlz_inflate_par.hip on this chain's items (one "piece" each, no sub-pieces)
    par::ParArgs p;
    memset(&p, 0, sizeof(p));
    p.z = a.in; p.zn = a.in_len; p.flags = a.flags; p.obsize = a.obsize; p.out = a.out; p.cap = cap; p.srcn = srcn;
    p.out_len = a.out_len; p.status = a.status; p.nchunks = nitems; p.chbits = L.pb;
    p.ctl = reinterpret_cast<uint32_t*>(g.ws);
    p.opos = reinterpret_cast<uint32_t*>(g.ws + L.o_opos); p.tokens = reinterpret_cast<uint32_t*>(g.ws + L.o_tok); p.tcap = L.tcap;
    p.ntok = reinterpret_cast<uint32_t*>(g.ws + L.o_ntok); p.srcA = g.srcA; p.sub = 1u; p.cnu = par::C_NUSED;
    p.mext = reinterpret_cast<uint32_t*>(g.ws + L.o_mext);
    p.in_pitch = a.in_pitch; p.out_pitch = a.out_pitch; p.in_off = a.in_off; p.ws_stride = ws_stride; p.batch = nstr > 1u ? 1u : 0u;
    const uint32_t passes = par::passes_for(nitems);
    *passes_out = passes;
    return par::par_launch_emit_jump(p, nitems, passes, nstr, stream);
}

}  // namespace hdlz
