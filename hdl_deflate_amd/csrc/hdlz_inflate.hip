// hdlz_inflate.hip -- STARTD for a batch of independent zlib streams on gfx950.
//
// Replaces the reference's inflate FSM for stored + fixed-Huffman blocks
// (/root/reference/deflate.py:635-732 IDLE/HEADER, :1402-1445 NEXT, :1519-1591 INFLATE,
// :1593-1659 COPY, :517-533 get4/adv).  Rule names D0..D8 are SURVEY.md 8(a)'s.
//
// Inflate is serial per stream, so the parallelism is ACROSS streams: one lane per stream,
// 64 streams per wave.  The 512-entry fixed-tree leaf table (the reference's stat_leaves,
// deflate.py:151-216: leaf = (sym << 4) | nbits, indexed by the next 9 bits) lives in LDS and is
// generated arithmetically from RFC1951 3.2.6 at kernel start.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hdlz_device.h"

namespace hdlz {

__device__ __forceinline__ uint32_t alignbyte_i(uint32_t hi, uint32_t lo, uint32_t sh) {
    return __builtin_amdgcn_alignbyte(hi, lo, sh);
}

// 8 bytes of the stream starting at byte `idx` (little endian), zeros past `zn`
// (the reference's b41 window, deflate.py:348, widened to 64 bits)
__device__ __forceinline__ uint64_t window64(const uint8_t* __restrict__ z, uint32_t idx, uint32_t zn) {
    if (idx >= zn) return 0;
    const uintptr_t addr = reinterpret_cast<uintptr_t>(z) + idx;
    const uint32_t sh = (uint32_t)(addr & 3u);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(addr - sh);
    const uint32_t valid = zn - idx;                 // >= 1
    const uint32_t nd = (valid + sh + 3u) >> 2;      // dwords holding valid bytes
    const uint32_t a0 = q[0];
    const uint32_t a1 = nd > 1 ? q[1] : 0u;
    const uint32_t a2 = nd > 2 ? q[2] : 0u;
    uint64_t w = ((uint64_t)alignbyte_i(a2, a1, sh) << 32) | alignbyte_i(a1, a0, sh);
    if (valid < 8u) w &= (1ull << (8u * valid)) - 1ull;
    return w;
}

__device__ __forceinline__ uint32_t rev(uint32_t v, uint32_t nbits) { return __builtin_bitreverse32(v) >> (32u - nbits); }

// RFC1951 tables in closed form (deflate.py:100-110)
__device__ __forceinline__ void length_info(uint32_t token, uint32_t& base, uint32_t& eb) {
    // token = sym - 257 in [0, 28]
    if (token < 8u) { base = 3u + token; eb = 0; }
    else if (token == 28u) { base = 258u; eb = 0; }
    else { eb = (token >> 2) - 1u; base = 3u + ((4u + (token & 3u)) << eb); }
}
__device__ __forceinline__ void dist_info(uint32_t dc, uint32_t& base, uint32_t& eb) {
    // dc in [0, 29]
    if (dc < 4u) { base = 1u + dc; eb = 0; }
    else { eb = (dc >> 1) - 1u; base = 1u + ((2u + (dc & 1u)) << eb); }
}

__global__ __launch_bounds__(64) void k_inflate(InflateArgs a) {
    __shared__ uint16_t leaves[512];
    const uint32_t lane = threadIdx.x;
    // ---- fixed-tree leaf table: index = next 9 stream bits (LSB first)
    for (uint32_t c = lane; c < 512u; c += 64u) {
        uint32_t sym, nb;
        const uint32_t r7 = rev(c & 127u, 7);
        const uint32_t r8 = rev(c & 255u, 8);
        const uint32_t r9 = rev(c, 9);
        if (r7 < 24u) { sym = 256u + r7; nb = 7; }                     // 0000000..0010111
        else if (r8 >= 0x30u && r8 < 0xC0u) { sym = r8 - 0x30u; nb = 8; } // 00110000..10111111
        else if (r8 >= 0xC0u && r8 < 0xC8u) { sym = 280u + (r8 - 0xC0u); nb = 8; }
        else { sym = r9 - 256u; nb = 9; }                              // 110010000..111111111 -> 144..255
        uint32_t leaf = (sym << 4) | nb;
        if (sym == 287u) leaf = 0;      // the reference's table holds 0 there (deflate.py:212) -> "< 1 bits"
        leaves[c] = (uint16_t)leaf;
    }
    __syncthreads();

    const uint64_t sid = (uint64_t)blockIdx.x * 64u + lane;
    if (sid >= a.nstreams) return;
    uint64_t off;
    uint32_t zn;
    if (a.in_off) {
        off = a.in_off[sid];
        zn = (uint32_t)(a.in_off[sid + 1] - off);
    } else {
        off = sid * a.in_pitch;
        zn = a.in_len;
    }
    const uint8_t* __restrict__ z = a.in + off;
    uint8_t* __restrict__ out = a.out + sid * a.out_pitch;
    const uint64_t cap = a.out_pitch;
    // obsize != 0: reference-exact OBSIZE build -- the stored LEN register is LOBSIZE bits wide
    // (deflate.py:329,:714), so LEN is taken mod 2^floor(log2(obsize)); obsize == 0: RFC behaviour.
    const uint32_t obsize = a.obsize ? a.obsize : 32768u;
    const uint32_t len_mask = a.obsize ? ((1u << (31u - (uint32_t)__builtin_clz(a.obsize))) - 1u) : 0xFFFFu;
    const bool assume_fixed = (a.flags & HDLZ_INFLATE_ASSUME_FIXED) != 0;

    uint32_t status = HDLZ_OK;
    uint32_t dout = 0;
    if (zn < 5u) {
        status = HDLZ_E_SHORT_INPUT;
    } else {
        const int32_t isize = (int32_t)zn - 1;            // deflate.py:605
        uint32_t bitpos = 16;                             // D0: zlib header skipped unvalidated
        for (;;) {
            // HEADER (deflate.py:677-732)
            uint64_t w = window64(z, bitpos >> 3, zn) >> (bitpos & 7u);
            const uint32_t final = (uint32_t)w & 1u;
            const uint32_t hm = assume_fixed ? 1u : ((uint32_t)(w >> 1) & 3u);
            if (hm == 3u) { status = HDLZ_E_BAD_BTYPE; break; }
            if (hm == 2u) { status = HDLZ_E_DYNAMIC_UNSUPPORTED; break; }
            if (hm == 0u) {
                // stored (deflate.py:709-717, COPY :1603-1626)
                const uint32_t dio = bitpos & 7u;
                uint32_t skip = 8u - dio;
                if (skip <= 2u) skip = 16u - dio;
                const uint32_t length = (uint32_t)(w >> skip) & 0xFFFFu & len_mask;
                bitpos += skip + 16u;                     // at NLEN (unchecked, D2); data at di+2
                int32_t di = (int32_t)(bitpos >> 3);
                for (uint32_t i = 0; i < length; i++) {
                    if (di >= isize - 2) { status = HDLZ_E_NO_EOF; break; }
                    if (dout >= cap) { status = HDLZ_E_OUT_CAPACITY; break; }
                    out[dout++] = z[di + 2];
                    di++;
                }
                if (status != HDLZ_OK) break;
                if (di >= isize - 2) { status = HDLZ_E_NO_EOF; break; }
                if (final) break;
                bitpos = (uint32_t)(di + 2) * 8u;
                continue;
            }
            bitpos += 3u;
            // NEXT / INFLATE
            for (;;) {
                w = window64(z, bitpos >> 3, zn) >> (bitpos & 7u);
                const uint32_t leaf = leaves[(uint32_t)w & 511u];
                const uint32_t nb = leaf & 15u, code = leaf >> 4;
                if (nb < 1u) { status = HDLZ_E_BAD_SYMBOL; break; }
                bitpos += nb;
                w >>= nb;
                if ((int32_t)(bitpos >> 3) > isize - 3) { status = HDLZ_E_NO_EOF; break; }   // deflate.py:1535
                if (code == 256u) break;
                if (code < 256u) {
                    if (dout >= cap) { status = HDLZ_E_OUT_CAPACITY; break; }
                    out[dout++] = (uint8_t)code;
                    continue;
                }
                const uint32_t token = code - 257u;
                if (token >= 29u) { status = HDLZ_E_BAD_SYMBOL; break; }
                uint32_t lbase, leb, dbase, deb;
                length_info(token, lbase, leb);
                const uint32_t tlength = lbase + ((uint32_t)w & ((1u << leb) - 1u));
                w >>= leb;
                const uint32_t dc = rev((uint32_t)w & 31u, 5);
                w >>= 5;
                if (dc >= 30u) { status = HDLZ_E_BAD_DISTANCE; break; }
                dist_info(dc, dbase, deb);
                const uint32_t distance = dbase + ((uint32_t)w & ((1u << deb) - 1u));
                bitpos += leb + 5u + deb;
                if (distance > dout || distance > obsize) { status = HDLZ_E_BAD_DISTANCE; break; }
                if ((int32_t)(bitpos >> 3) >= isize - 2) { status = HDLZ_E_NO_EOF; break; }
                if ((uint64_t)dout + tlength > cap) { status = HDLZ_E_OUT_CAPACITY; break; }
                for (uint32_t i = 0; i < tlength; i++, dout++) out[dout] = out[dout - distance];
            }
            if (status != HDLZ_OK || final) break;
        }
    }
    a.out_len[sid] = status == HDLZ_OK ? dout : 0u;
    a.status[sid] = status;
}

hipError_t launch_inflate(const InflateArgs& a, hipStream_t stream) {
    if (a.nstreams == 0) return hipSuccess;
    const dim3 grid((unsigned)((a.nstreams + 63u) / 64u)), block(64);
    hipLaunchKernelGGL(k_inflate, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace hdlz
