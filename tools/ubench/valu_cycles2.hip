// valu_cycles2.hip -- issue cost in SHADER CYCLES (s_memtime) of the wave64 VALU instructions valu_cycles.hip left out: the select
// forms (v_cndmask with VCC / with an SGPR pair, v_bfi), three-operand integer ops, 64-bit shifts, the f32 min/max family, lane ops.
// Same method: W waves per SIMD run 64 independent instructions per iteration; cycles = ticks of a wave / (instructions * W).
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_cycles2 tools/ubench/valu_cycles2.hip && tools/ubench/valu_cycles2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x

template <int OP>
__global__ void k(uint64_t* ticks, uint32_t* sink, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
    uint32_t b = seed * 2654435761u + 12345u + threadIdx.x, c = seed | 1u;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    const uint64_t msk = 0x5555AAAA3333CCCCull ^ seed;      // a lane mask in an SGPR pair
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#define R8(ins) asm volatile(ins(%0) ins(%1) ins(%2) ins(%3) ins(%4) ins(%5) ins(%6) ins(%7) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(msk) : "vcc");
#define Q4(ins) asm volatile(ins(%0) ins(%1) ins(%2) ins(%3) ins(%0) ins(%1) ins(%2) ins(%3) \
        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b), "v"(c));
#define I_CNDVCC(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n\t"
#define I_CNDS(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, %10\n\t"
#define I_BFI(r) "v_bfi_b32 " #r ", %9, " #r ", %8\n\t"
#define I_ASHR(r) "v_ashrrev_i32 " #r ", 31, " #r "\n\t"
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9\n\t"
#define I_ANDOR(r) "v_and_or_b32 " #r ", " #r ", %8, %9\n\t"
#define I_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n\t"
#define I_LSHLOR(r) "v_lshl_or_b32 " #r ", " #r ", 3, %9\n\t"
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 3, %9\n\t"
#define I_OR3(r) "v_or3_b32 " #r ", " #r ", %8, %9\n\t"
#define I_XAD(r) "v_xad_u32 " #r ", " #r ", %8, %9\n\t"
#define I_BITOP3(r) "v_bitop3_b32 " #r ", " #r ", %8, %9 bitop3:0xde\n\t"
#define I_MINF(r) "v_min_f32 " #r ", " #r ", %8\n\t"
#define I_MAXF(r) "v_max_f32 " #r ", " #r ", %8\n\t"
#define I_MIN3F(r) "v_min3_f32 " #r ", " #r ", %8, %9\n\t"
#define I_MAX3F(r) "v_max3_f32 " #r ", " #r ", %8, %9\n\t"
#define I_MED3F(r) "v_med3_f32 " #r ", " #r ", %8, %9\n\t"
#define I_SUBF(r) "v_sub_f32 " #r ", " #r ", %8\n\t"
#define I_MAX3U(r) "v_max3_u32 " #r ", " #r ", %8, %9\n\t"
#define I_MAXU(r) "v_max_u32 " #r ", " #r ", %8\n\t"
#define I_MAXI16(r) "v_max_i16 " #r ", " #r ", %8\n\t"
#define I_SAD8(r) "v_sad_u8 " #r ", " #r ", %8, %9\n\t"
#define I_DOT4(r) "v_dot4_u32_u8 " #r ", " #r ", %8, %9\n\t"
#define I_MBCNT(r) "v_mbcnt_lo_u32_b32 " #r ", " #r ", %8\n\t"
#define I_BCNT(r) "v_bcnt_u32_b32 " #r ", " #r ", %8\n\t"
#define I_READLANE(r) "v_readlane_b32 s20, " #r ", 5\n\t"
#define I_CMPS(r) "v_cmp_lt_u32_e64 s[20:21], " #r ", %8\n\t"
#define I_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8\n\t"
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n\t"
#define I_SUBREV_SDWA(r) "v_lshlrev_b32_sdwa " #r ", %9, " #r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
#define I_MOVDPP(r) "v_mov_b32_dpp " #r ", " #r " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_ADDDPP(r) "v_add_u32_dpp " #r ", " #r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_LSHR64(r) "v_lshrrev_b64 " #r ", %5, " #r "\n\t"
#define I_LSHL64(r) "v_lshlrev_b64 " #r ", 4, " #r "\n\t"
#define I_ALIGNBIT(r) "v_alignbit_b32 " #r ", " #r ", %8, %9\n\t"
#define I_CVTPK(r) "v_cvt_pk_u16_u32 " #r ", " #r ", %8\n\t"
#define I_PKMINU16(r) "v_pk_min_u16 " #r ", " #r ", %8\n\t"
        if (OP == 0) { REP8(R8(I_CNDVCC)) }
        if (OP == 1) { REP8(R8(I_CNDS)) }
        if (OP == 2) { REP8(R8(I_BFI)) }
        if (OP == 3) { REP8(R8(I_ASHR)) }
        if (OP == 4) { REP8(R8(I_PERM)) }
        if (OP == 5) { REP8(R8(I_ANDOR)) }
        if (OP == 6) { REP8(R8(I_ADD3)) }
        if (OP == 7) { REP8(R8(I_LSHLOR)) }
        if (OP == 8) { REP8(R8(I_LSHLADD)) }
        if (OP == 9) { REP8(R8(I_OR3)) }
        if (OP == 10) { REP8(R8(I_XAD)) }
        if (OP == 11) { REP8(R8(I_BITOP3)) }
        if (OP == 12) { REP8(R8(I_MINF)) }
        if (OP == 13) { REP8(R8(I_MAXF)) }
        if (OP == 14) { REP8(R8(I_MIN3F)) }
        if (OP == 15) { REP8(R8(I_MAX3F)) }
        if (OP == 16) { REP8(R8(I_MED3F)) }
        if (OP == 17) { REP8(R8(I_SUBF)) }
        if (OP == 18) { REP8(R8(I_MAX3U)) }
        if (OP == 19) { REP8(R8(I_MAXU)) }
        if (OP == 20) { REP8(R8(I_MAXI16)) }
        if (OP == 21) { REP8(R8(I_SAD8)) }
        if (OP == 22) { REP8(R8(I_DOT4)) }
        if (OP == 23) { REP8(R8(I_MBCNT)) }
        if (OP == 24) { REP8(R8(I_BCNT)) }
        if (OP == 25) { REP8(asm volatile(I_READLANE(%0) I_READLANE(%1) I_READLANE(%2) I_READLANE(%3) I_READLANE(%4) I_READLANE(%5) I_READLANE(%6) I_READLANE(%7)
                                        :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20");) }
        if (OP == 26) { REP8(asm volatile(I_CMPS(%0) I_CMPS(%1) I_CMPS(%2) I_CMPS(%3) I_CMPS(%4) I_CMPS(%5) I_CMPS(%6) I_CMPS(%7)
                                        :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(b) : "s20", "s21");) }
        if (OP == 27) { REP8(R8(I_MUL24)) }
        if (OP == 28) { REP8(R8(I_MULLO)) }
        if (OP == 29) { REP8(R8(I_SUBREV_SDWA)) }
        if (OP == 30) { REP8(R8(I_MOVDPP)) }
        if (OP == 31) { REP8(R8(I_ADDDPP)) }
        if (OP == 32) { REP8(Q4(I_LSHR64)) }
        if (OP == 33) { REP8(Q4(I_LSHL64)) }
        if (OP == 34) { REP8(R8(I_ALIGNBIT)) }
        if (OP == 35) { REP8(R8(I_CVTPK)) }
        if (OP == 36) { REP8(R8(I_PKMINU16)) }
        // cmp -> cndmask pairs as the compress kernel has them (the mask through an SGPR pair and an s_and)
        if (OP == 37) { REP8(asm volatile(
            "v_cmp_le_u32 vcc, %0, %8\n\ts_and_b64 s[20:21], vcc, %10\n\tv_cndmask_b32_e64 %1, %1, %9, s[20:21]\n\t"
            "v_cmp_le_u32 vcc, %2, %8\n\ts_and_b64 s[22:23], vcc, %10\n\tv_cndmask_b32_e64 %3, %3, %9, s[22:23]\n\t"
            "v_cmp_le_u32 vcc, %4, %8\n\ts_and_b64 s[20:21], vcc, %10\n\tv_cndmask_b32_e64 %5, %5, %9, s[20:21]\n\t"
            "v_cmp_le_u32 vcc, %6, %8\n\ts_and_b64 s[22:23], vcc, %10\n\tv_cndmask_b32_e64 %7, %7, %9, s[22:23]\n\t"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(msk) : "vcc", "s20", "s21", "s22", "s23");) }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63u) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3)) == 0x12345u) sink[0] = a0;
}

template <int OP>
void run(const char* name, int waves_per_simd, int insts_per_iter) {
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;
    uint64_t* d; uint32_t* sink;
    hipMalloc(&d, sizeof(uint64_t) * blocks * 16); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, sink, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, sink, iters, 7u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks * threads / 64);
    hipMemcpy(h.data(), d, sizeof(uint64_t) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double per = med / ((double)iters * insts_per_iter * waves_per_simd);
    printf("%-34s waves/SIMD %d: %5.2f ticks per wave64 instr per SIMD   [%.3f ms wall -> %.2f G ticks/s]\n", name, waves_per_simd, per, ms, med / (ms * 1e6));
    hipFree(d); hipFree(sink);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("v_cndmask_b32 (vcc)", w, 64); run<1>("v_cndmask_b32_e64 (sgpr pair)", w, 64); run<2>("v_bfi_b32", w, 64); run<3>("v_ashrrev_i32", w, 64);
        run<4>("v_perm_b32", w, 64); run<5>("v_and_or_b32", w, 64); run<6>("v_add3_u32", w, 64); run<7>("v_lshl_or_b32", w, 64);
        run<8>("v_lshl_add_u32", w, 64); run<9>("v_or3_b32", w, 64); run<10>("v_xad_u32", w, 64); run<11>("v_bitop3_b32", w, 64);
        run<12>("v_min_f32", w, 64); run<13>("v_max_f32", w, 64); run<14>("v_min3_f32", w, 64); run<15>("v_max3_f32", w, 64);
        run<16>("v_med3_f32", w, 64); run<17>("v_sub_f32", w, 64); run<18>("v_max3_u32", w, 64); run<19>("v_max_u32", w, 64);
        run<20>("v_max_i16", w, 64); run<21>("v_sad_u8", w, 64); run<22>("v_dot4_u32_u8", w, 64); run<23>("v_mbcnt_lo_u32_b32", w, 64);
        run<24>("v_bcnt_u32_b32", w, 64); run<25>("v_readlane_b32", w, 64); run<26>("v_cmp_lt_u32_e64 (sgpr pair)", w, 64);
        run<27>("v_mul_u32_u24", w, 64); run<28>("v_mul_lo_u32", w, 64); run<29>("v_lshlrev_b32_sdwa", w, 64); run<30>("v_mov_b32_dpp row_shr", w, 64);
        run<31>("v_add_u32_dpp row_shr", w, 64); run<32>("v_lshrrev_b64 (var)", w, 64); run<33>("v_lshlrev_b64 (4)", w, 64);
        run<34>("v_alignbit_b32", w, 64); run<35>("v_cvt_pk_u16_u32", w, 64); run<36>("v_pk_min_u16", w, 64);
        run<37>("cmp + s_and + cndmask_e64 (x4)", w, 64);      // 8 VALU + 4 SALU per group of 12: ticks are per VALU-or-SALU slot / 64 * 96 ...
    }
    return 0;
}
