// semantics check of LDS-DMA loads on gfx950: where do global_load_lds_dword / _dwordx4 put each lane's data?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int X4>
__global__ __launch_bounds__(64) void k(const uint32_t* src, uint32_t* out) {
    __shared__ uint32_t q[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) q[i] = 0xDEAD0000u + i;
    __syncthreads();
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(&q[64]));
    uint32_t save;
    if (X4) {
        const uint32_t* p = src + threadIdx.x * 4;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(save) : "v"(p), "s"(base) : "memory");
    } else {
        const uint32_t* p = src + threadIdx.x;
        if (threadIdx.x & 1)       // odd lanes only: masked lanes must leave their slot alone
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off offset:16\n\ts_mov_b32 m0, %0" : "=&s"(save) : "v"(p), "s"(base) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = ((volatile uint32_t*)q)[i];
}

int main() {
    uint32_t h[1024], *d, *o, r[1024];
    for (int i = 0; i < 1024; i++) h[i] = i;
    CHECK(hipMalloc(&d, 4096)); CHECK(hipMalloc(&o, 4096));
    CHECK(hipMemcpy(d, h, 4096, hipMemcpyHostToDevice));
    for (int x4 = 0; x4 < 2; x4++) {
        if (x4) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, o); else hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, o);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost));
        printf("%s: LDS dwords that changed (index: value):\n", x4 ? "dwordx4" : "dword (odd lanes, offset:16)");
        int shown = 0;
        for (int i = 0; i < 1024 && shown < 24; i++) if (r[i] != 0xDEAD0000u + i) { printf(" %d:%u", i, r[i]); shown++; }
        int cnt = 0; for (int i = 0; i < 1024; i++) cnt += r[i] != 0xDEAD0000u + i;
        printf("\n total changed %d\n", cnt);
    }
    return 0;
}
