// round 2: issue cost of the candidates for a cheaper match search / extension in k_compress
// (packed SAD family, packed 16-bit ops, 3-operand bit ops, 64-bit shifts).  Same method as valu_rate3.hip:
// 8 independent accumulators per lane, 8 waves per SIMD, inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 1024;
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define DEFK(NAME, BODY, NOPS)                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t* out, uint32_t seed) {             \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,   \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed ^ 0x55, c = seed + 9;        \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;        \
        uint64_t s64 = ((uint64_t)a3 << 32) | a5;                                                \
        u4 w0 = {a0, a1, a2, a3}, w1 = {a1, a2, a3, a4}, w2 = {a2, a3, a4, a5}, w3 = {a3, a4, a5, a6}; \
        for (int it = 0; it < ITER; it++) { BODY BODY BODY BODY }                                \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ w0.x ^ w1.y ^ w2.z ^ w3.w ^ (uint32_t)s64; \
    }                                                                                           \
    static const int nops_##NAME = NOPS;
// 8 x 32-bit accumulate ops "OP d, d, b" style given a format with %0 accumulator, %8 = b, %9 = c
#define A32_8(F) asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#define A64_8(F) asm volatile(F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(b), "v"(c), "v"(s64));
#define A128_4(F) asm volatile(F(0) F(1) F(2) F(3) : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(s64));

#define F_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define F_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define F_SAD8(i) "v_sad_u8 %" #i ", %" #i ", %8, %9\n"
#define F_MSAD8(i) "v_msad_u8 %" #i ", %" #i ", %8, %9\n"
#define F_SADU32(i) "v_sad_u32 %" #i ", %" #i ", %8, %9\n"
#define F_PKSUB(i) "v_pk_sub_u16 %" #i ", %" #i ", %8\n"
#define F_PKMIN(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define F_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define F_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define F_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, %9\n"
#define F_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, %9\n"
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", %8, 5\n"
#define F_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define F_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define F_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_XAD(i) "v_xad_u32 %" #i ", %" #i ", %8, %9\n"
#define F_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %9\n"
#define F_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %9\n"
#define F_ADDLSHL(i) "v_add_lshl_u32 %" #i ", %" #i ", %8, 2\n"
#define F_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define F_DOT4(i) "v_dot4_u32_u8 %" #i ", %" #i ", %8, %9\n"
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define F_FFBL(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define F_FFBH(i) "v_ffbh_u32 %" #i ", %" #i "\n"
#define F_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define F_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %8\n"
#define F_CMPS(i) "v_cmp_eq_u32_e64 s[20:21], %" #i ", %8\n"
#define F_CMPCND(i) "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
#define F_MINU16(i) "v_min_u16 %" #i ", %" #i ", %8\n"
#define F_MAX3(i) "v_max3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_MED3(i) "v_med3_u32 %" #i ", %" #i ", %8, %9\n"
#define F_LSHL64(i) "v_lshlrev_b64 %" #i ", %8, %" #i "\n"
#define F_LSHR64(i) "v_lshrrev_b64 %" #i ", %8, %" #i "\n"
#define F_QSAD(i) "v_qsad_pk_u16_u8 %" #i ", %10, %8, %" #i "\n"
#define F_MQSADPK(i) "v_mqsad_pk_u16_u8 %" #i ", %10, %8, %" #i "\n"
#define F_MQSAD32(i) "v_mqsad_u32_u8 %" #i ", %5, %4, %" #i "\n"
#define F_PKMAD(i) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9\n"
#define F_PKLSHR(i) "v_pk_lshrrev_b16 %" #i ", %8, %" #i "\n"

DEFK(min3, A32_8(F_MIN3), 8) DEFK(min, A32_8(F_MIN), 8) DEFK(sub, A32_8(F_SUB), 8) DEFK(sad_u8, A32_8(F_SAD8), 8)
DEFK(msad_u8, A32_8(F_MSAD8), 8) DEFK(sad_u32, A32_8(F_SADU32), 8) DEFK(pk_sub_u16, A32_8(F_PKSUB), 8) DEFK(pk_min_u16, A32_8(F_PKMIN), 8)
DEFK(pk_add_u16, A32_8(F_PKADD), 8) DEFK(perm, A32_8(F_PERM), 8) DEFK(alignbyte, A32_8(F_ALIGNBYTE), 8) DEFK(alignbit, A32_8(F_ALIGNBIT), 8)
DEFK(bfe, A32_8(F_BFE), 8) DEFK(and_or, A32_8(F_ANDOR), 8) DEFK(or3, A32_8(F_OR3), 8) DEFK(add3, A32_8(F_ADD3), 8) DEFK(xad, A32_8(F_XAD), 8)
DEFK(lshl_or, A32_8(F_LSHLOR), 8) DEFK(lshl_add, A32_8(F_LSHLADD), 8) DEFK(add_lshl, A32_8(F_ADDLSHL), 8) DEFK(bfi, A32_8(F_BFI), 8)
DEFK(dot4, A32_8(F_DOT4), 8) DEFK(mad_u24, A32_8(F_MAD24), 8) DEFK(ffbl, A32_8(F_FFBL), 8) DEFK(ffbh, A32_8(F_FFBH), 8) DEFK(bcnt, A32_8(F_BCNT), 8)
DEFK(mbcnt, A32_8(F_MBCNT), 8) DEFK(cmp_sgpr, A32_8(F_CMPS), 8) DEFK(cmp_cnd, A32_8(F_CMPCND), 16) DEFK(min_u16, A32_8(F_MINU16), 8)
DEFK(max3, A32_8(F_MAX3), 8) DEFK(med3, A32_8(F_MED3), 8) DEFK(lshl64, A64_8(F_LSHL64), 8) DEFK(lshr64, A64_8(F_LSHR64), 8)
DEFK(qsad_pk_u16_u8, A64_8(F_QSAD), 8) DEFK(mqsad_pk_u16_u8, A64_8(F_MQSADPK), 8) DEFK(mqsad_u32_u8, A128_4(F_MQSAD32), 4)
DEFK(pk_mad_u16, A32_8(F_PKMAD), 8) DEFK(pk_lshr_b16, A32_8(F_PKLSHR), 8)

template <class K>
void run(const char* name, K kern, int nops) {
    const int blocks = 256 * 8;
    uint32_t* d;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 3u + r);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double waveinstr = (double)blocks * 4 * ITER * 4 * nops;
    double cyc = (best * 1e-3) * 2.4e9 * 1024.0 / waveinstr;
    printf("%-18s %8.3f ms  %7.2f T lane-ops/s  %5.2f cycles/wave64-instr @2.4GHz\n", name, best,
           waveinstr * 64 / (best * 1e-3) / 1e12, cyc);
    hipFree(d);
}
#define RUN(NAME) run(#NAME, k_##NAME, nops_##NAME);
int main() {
    RUN(sub) RUN(min) RUN(min3) RUN(max3) RUN(med3) RUN(min_u16) RUN(sad_u8) RUN(msad_u8) RUN(sad_u32) RUN(qsad_pk_u16_u8) RUN(mqsad_pk_u16_u8) RUN(mqsad_u32_u8)
    RUN(pk_sub_u16) RUN(pk_min_u16) RUN(pk_add_u16) RUN(pk_mad_u16) RUN(pk_lshr_b16) RUN(perm) RUN(alignbyte) RUN(alignbit) RUN(bfe) RUN(and_or) RUN(or3) RUN(add3) RUN(xad)
    RUN(lshl_or) RUN(lshl_add) RUN(add_lshl) RUN(bfi) RUN(dot4) RUN(mad_u24) RUN(ffbl) RUN(ffbh) RUN(bcnt) RUN(mbcnt) RUN(cmp_sgpr) RUN(cmp_cnd)
    RUN(lshl64) RUN(lshr64)
    return 0;
}
