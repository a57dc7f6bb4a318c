// hdlz_compress_common.h -- constants and device helpers shared by the compress kernels
// (hdlz_compress.hip: one block per wave, any size; hdlz_compress_small.hip: several small blocks per wave-tile)
#pragma once
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
constexpr int LUT_LIT = 256;        // [byte]                                  at LUT word 0
constexpr int LUT_MATCH = 256;      // [len-3][dist-1] (CWINDOW <= 32) or [dist-1]  at LUT word 256
constexpr uint32_t LUT_MATCH_BYTE = 4u * LUT_LIT;
constexpr uint32_t ADLER_MOD = 65521u;
constexpr uint32_t NB_SHIFT = 27;   // LUT entry = code (27 bits) | nbits << 27
constexpr uint32_t CODE_MASK = (1u << NB_SHIFT) - 1u;

struct __attribute__((aligned(16))) WaveLds {
    uint32_t in[IN_BYTES / 4];      // byte index = position - tile_start + HALO
    uint32_t out[OUT_WORDS];        // bit buffer of the current tile
    uint32_t lut[LUT_LIT + LUT_MATCH];
};

// fence for the instruction scheduler + value fences: keep independent phases from being overlapped
// (that blew the VGPR budget to 239 and spilled ~200 SGPR lane masks in the first version)
#define PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
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

__device__ __forceinline__ uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);      // bytes [sh, sh+4) of hi:lo
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t t = a < b ? a : b;
    return t < c ? t : c;
}
// index of the lowest set bit, 0xFFFFFFFF for 0 (v_ffbl_b32)
__device__ __forceinline__ uint32_t ffbl(uint32_t x) { return (uint32_t)(__builtin_ffs((int)x) - 1); }

// key of the 3-byte string at byte J of d[]: bytes 1..3 = the string, byte 0 = tag (4 * window index)
template <int J>
__device__ __forceinline__ uint32_t key3(const uint32_t* d, uint32_t tag) {
    constexpr int w = J >> 2, sh = J & 3;
    if constexpr (sh == 0) return (d[w] << 8) | tag;
    else if constexpr (sh == 1) return (d[w] & 0xFFFFFF00u) | tag;
    else return (alignbyte(d[w + 1], d[w], sh - 1) & 0xFFFFFF00u) | tag;
}

// ---- fixed Huffman token bits (used to fill the LUTs) ---------------------------------------
// literal (R7, deflate.py:1005-1016 + out_codes :112-149): sym<144 -> 8 bits rev8(0x30+sym),
// else 9 bits rev9(0x100+sym)
__device__ __forceinline__ uint32_t literal_entry(uint32_t b) {
    const bool big = b >= 144u;
    const uint32_t v = big ? (0x100u + b) : (0x30u + b);
    const uint32_t nb = big ? 9u : 8u;
    return (__builtin_bitreverse32(v) >> (32u - nb)) | (nb << NB_SHIFT);
}
// distance part of a match token (R6, deflate.py:836-882): rev5(dist code) | extra<<5 in 5+eb bits,
// placed behind the 7-bit length code; nbits = 12 + eb
__device__ __forceinline__ uint32_t dist_entry(uint32_t d) {
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
    const uint32_t dcode = __builtin_bitreverse32(c) >> 27;
    return ((dcode | (extra << 5)) << 7) | ((12u + eb) << NB_SHIFT);
}
// 7-bit code of length symbol 254+m (no extra bits for m <= 10)
__device__ __forceinline__ uint32_t length_code(uint32_t m) { return __builtin_bitreverse32(m - 2u) >> 25; }


}  // namespace hdlz
