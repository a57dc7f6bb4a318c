// hdlz_inflate_par.h -- what the two chains of the whole-GPU inflate share: hdlz_inflate_par.hip (a stream that is ONE fixed-Huffman
// block: pieces cut anywhere) and hdlz_inflate_any.hip (any sequence of stored / fixed / dynamic blocks: blocks found first, pieces
// inside them) end in the same token lists, and the same kernels turn those into bytes (k_par_emit, k_par_jump).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"

namespace hdlz {
namespace par {

constexpr uint32_t CH_BITS_MAX = 8192;        // a piece: 1 KiB of the stream -- 512 bytes for streams below 24 MiB, 256 below 3 MiB, 128 below 1.25 MiB: a piece is ONE
                                              // wave's (lane's) serial chain in k_par_spec and k_par_tokens, and 16 MiB in 1 KiB pieces do not fill the
                                              // GPU twice (16 MiB: 1.37 -> 1.24 ms, 1 MiB: 0.76 -> 0.45 ms with 512-byte pieces)
constexpr uint32_t WIN_DW = CH_BITS_MAX / 32 + 8; // its staged window: the piece, the 31 + 64 bits a token starting at its end may read
constexpr uint32_t FIRST_BIT = 19;            // 2 zlib header bytes, BFINAL, BTYPE
constexpr uint32_t X_EOB = 0x40, X_BAD = 0x80;
constexpr uint32_t SUB = 4;                   // sub-pieces per piece: the granularity of the real decode and the emit (k_par_spec)
constexpr uint32_t NONE = 0xFFFFFFFFu;
enum { C_FALLBACK = 0, C_NUSED = 1, C_TOTAL = 2, C_OK = 3, C_MARK = 4, C_FNUSED = 5, C_NCHAIN = 6, C_NCROSS = 36, C_FAILF = 37,
       C_NOTFIXED = 7,         // the stream is not ONE fixed block: the gate of the chain for any block types (hdlz_inflate_any.hip)
       C_PASS0 = 8, C_ANY0 = 40 /* .. 63: that chain's own counters */, C_WORDS = 64 };

struct ParArgs {
    const uint8_t* z;
    uint32_t zn;
    uint32_t flags, obsize;
    uint8_t* out;
    uint32_t cap;               // output capacity (bytes)
    uint32_t srcn;              // entries of srcA
    uint32_t* out_len;
    uint32_t* status;
    uint32_t nchunks;
    uint32_t chbits;            // bits per piece
    uint32_t* ctl;              // C_WORDS control words (zeroed)
    uint8_t* exit8;             // [nchunks][32]
    uint32_t* nb32;             // [nchunks][32]
    uint8_t* entry8;            // [nchunks]
    uint32_t* opos;             // [nchunks]
    uint8_t* gexit8;            // [ngroups][32]  the same maps for groups of 64 pieces
    uint8_t* gstop8;            // [ngroups][32]  piece of the group in which the chain ends
    uint32_t* gnb32;            // [ngroups][32]
    uint8_t* gentry8;           // [ngroups]
    uint32_t* gopos;            // [ngroups]
    uint32_t* tokens;           // [nchunks][tcap]  the tokens of every piece
    uint32_t tcap;              // words per token list
    uint32_t* ntok;             // [nchunks]
    uint32_t* srcA;             // [srcn]  marker of every output byte: the absolute position it comes from; NONE / ROOT | r: the byte is there
    uint32_t sub;               // k_par_spec: sub-pieces per piece (SUB), whose boundaries get maps of their own
    uint8_t* mexit8;            // [nchunks][SUB-1][32]  offset behind sub-boundary s for entry offset e (X_EOB: the chain ended in front of it)
    uint32_t* mnb32;            // [nchunks][SUB-1][32]  bytes of the tokens that start in front of that boundary
    uint32_t cnu;               // the control word that holds the number of pieces in use at THIS granularity (C_NUSED / C_FNUSED)
    uint32_t* mext;             // [nchunks]  bytes from a piece's first output byte to behind its LAST marker (0: it has none)
    uint32_t* cross;            // [MAXCROSS][4]  the end-of-block codes the real decode passed: bit, output position, sub-piece, token index | header << 24 | goes on << 27
    uint32_t* nfail;            // [nchunks * sub]  token index of a sub-piece's first failed check (NONE: none)
    // SEVERAL streams in the same launches (round 5): blockIdx.y is the stream; stream s reads z + s * in_pitch, writes out + s * out_pitch,
    // out_len[s], status[s], and owns the scratch ws_stride bytes behind stream s - 1's (every array above, same layout)
    uint64_t in_pitch, out_pitch;
    const uint64_t* in_off;     // nullable.  Ragged input: stream s is z[in_off[s] .. in_off[s + 1]); zn is then the caller's BOUND on the lengths (the
                                // pieces are laid out for it; bytes behind a stream's own end read as zero, like the padding of a pitched row)
    size_t ws_stride;
    uint32_t batch;             // != 0: a stream the path gives up on is FLAGGED for the serial pass (status HDLZ_E_DYNAMIC_UNSUPPORTED)
};
constexpr uint32_t MAXCROSS = 4096;           // end-of-block codes (blocks) of a stream of fixed blocks the chain can list
constexpr uint32_t TOK_LIT = 0x80000000u;     // a token: TOK_LIT | byte, or length | distance << 9
__host__ __device__ inline uint32_t tmax_of(uint32_t chbits) { return (chbits / 8u + 2u + 3u) & ~3u; }      // fixed blocks: the shortest token is 8 bits long (a multiple of 4: the lists are written 16 bytes at a time)
#ifndef HDLZ_HOPS
#define HDLZ_HOPS 32
#endif
constexpr uint32_t HOPS = HDLZ_HOPS;           // marker chain steps per pass of k_par_jump

// the last kernels of either chain, for the items (pieces) the arguments describe: bytes + markers, the marker passes
hipError_t par_launch_emit_jump(const ParArgs& p, uint32_t nitems, uint32_t passes, uint32_t nstr, hipStream_t stream);
// workgroups per stream of a grid-stride launch over `work` units: all of them for one stream, fewer per stream the more streams there are
// (>= 32, ~2^14 workgroups per launch in all: they fill the GPU twice over, and a launch whose workgroups all turn away at the door --
// the chain that is not the streams' -- stays cheap beside the other chain's real work)
__host__ inline uint32_t grid_cap(uint64_t work, uint32_t nstr) {
    const uint64_t cap = nstr <= 1u ? work : ((1u << 14) / nstr < 32u ? 32u : (1u << 14) / nstr);
    return (uint32_t)(work < 1u ? 1u : work < cap ? work : cap);
}
__host__ inline uint32_t passes_for(uint32_t nitems) {                 // chains of up to `nitems` hops, HOPS-fold shorter per pass
    uint32_t passes = 1;
    for (uint64_t reach = 1; reach < (uint64_t)nitems + 1u; reach *= HOPS) passes++;
    return passes > (uint32_t)(C_ANY0 - C_PASS0) ? (uint32_t)(C_ANY0 - C_PASS0) : passes;       // (a counter per pass: control words C_PASS0 .. C_ANY0 - 1)
}

}  // namespace par

// the chain for streams of any block types (hdlz_inflate_any.hip).  Scratch of ONE stream: any_work_bytes (0: not for this shape);
// its kernels return at once for a stream that is one fixed block (the other chain's: the same test on the stream's third byte), and
// leave their verdict in their own control words (C_FALLBACK / C_OK / C_TOTAL / C_MARK / C_PASS0 ..)
size_t any_work_bytes(uint32_t in_len, uint64_t out_pitch, uint32_t flags, uint32_t nstreams);
// (ws: stream 0's scratch; ws_off / sa_off: where this chain's part and the marker words lie in it; its control words are the first
//  par::C_WORDS words of its part -- read by k_par_finish)
hipError_t launch_inflate_any(const InflateArgs& a, uint32_t nstr, uint8_t* ws, size_t ws_stride, size_t ws_off, size_t sa_off,
                              uint32_t srcn, uint32_t cap, hipStream_t stream, uint32_t* passes_out);
}  // namespace hdlz
