// hdlz_inflate_tables.h -- the fixed-Huffman decode tables and the bit-level helpers shared by the lane-per-stream inflate kernel
// (hdlz_inflate_tok.hip) and the parallel single-stream inflate (hdlz_inflate_par.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hdlz {
namespace tok {

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
struct __attribute__((packed, aligned(1))) u128_unaligned { uint64_t lo, hi; };

__device__ __forceinline__ uint32_t rev(uint32_t v, uint32_t nbits) { return __builtin_bitreverse32(v) >> (32u - nbits); }

__device__ __forceinline__ uint32_t load32(const uint8_t* __restrict__ z, uint32_t ip, uint32_t zn) {
    if (ip + 4u <= zn) return *reinterpret_cast<const u32_unaligned*>(z + ip);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4u; k++)
        if (ip + k < zn) v |= (uint32_t)z[ip + k] << (8u * k);
    return v;
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
// the widened stat_leaves (deflate.py:151-216): nbits[3:0] | sym[12:4] | type[14:13] | lbase[24:16] | leb[27:25]
enum { T_LIT = 0, T_LEN = 1, T_EOB = 2, T_BAD = 3 };
__device__ __forceinline__ uint32_t lit_entry(uint32_t c, bool zero_leaf) {
    uint32_t sym, nb;
    const uint32_t r7 = rev(c & 127u, 7), r8 = rev(c & 255u, 8), r9 = rev(c, 9);
    if (r7 < 24u) { sym = 256u + r7; nb = 7; }
    else if (r8 >= 0x30u && r8 < 0xC0u) { sym = r8 - 0x30u; nb = 8; }
    else if (r8 >= 0xC0u && r8 < 0xC8u) { sym = 280u + (r8 - 0xC0u); nb = 8; }
    else { sym = r9 - 256u; nb = 9; }
    uint32_t type = sym < 256u ? T_LIT : sym == 256u ? T_EOB : sym <= 285u ? T_LEN : T_BAD;
    uint32_t lbase = 0, leb = 0;
    if (type == T_LEN) length_info(sym - 257u, lbase, leb);
    // the reference's ONE zero leaf, index 483 of the DYNAMIC=False build's stat_leaves (deflate.py:212; zero_leaf = ASSUME_FIXED):
    // symbol 287's other slot (227) is an ordinary leaf -- its code fails in INFLATE --, and so are both in a DYNAMIC=True build,
    // which decodes fixed blocks through leaves built from the fixed lengths (deflate.py:1066-1073)
    if (c == 483u && zero_leaf) nb = 0;
    return nb | (sym << 4) | (type << 13) | (lbase << 16) | (leb << 25);
}
__device__ __forceinline__ uint32_t dst_entry(uint32_t raw5) {
    const uint32_t dc = rev(raw5, 5);
    if (dc >= 30u) return 0xFFFFFFFFu;
    uint32_t dbase, deb;
    dist_info(dc, dbase, deb);
    return dbase | (deb << 16);
}

}  // namespace tok
}  // namespace hdlz
