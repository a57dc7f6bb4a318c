// wave_launch.hip -- what does it cost to REPLACE a wave?  k_compress queues 256 single-wave workgroups per CU (20 resident at 5 waves
// per SIMD); tools/exp_tile_timing.py sees 4.35 of the 5 slots of a SIMD filled on average.  Here: workgroups of WG threads with the
// same residency (8 KB of LDS per wave: 20 waves per CU) that do nothing but wait `cyc` s_memtime ticks, 256 waves queued per CU.  The launch
// times of two waits give the counter's clock and the cost of replacing a wave.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_launch tools/ubench/wave_launch.hip && /tmp/wave_launch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int WG>
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_wait(uint32_t cyc, uint64_t* t_start, uint64_t* t_end) {
    __shared__ uint32_t lds[(WG / 64) * 2048];                 // 8 KB per wave: 20 waves per CU, as k_compress has by its VGPRs
    const uint64_t t0 = __builtin_readcyclecounter();
    lds[threadIdx.x] = (uint32_t)t0;
    uint64_t t;
    do {
        __builtin_amdgcn_s_sleep(8);
        t = __builtin_readcyclecounter();
    } while (t - t0 < cyc);
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t w = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
        t_start[w] = t0 + (lds[threadIdx.x] & 0u);
        t_end[w] = t;
    }
}

template <int WG>
static float run(uint32_t cyc, uint64_t* d_s, uint64_t* d_e, int nw) {
    const int grid = nw / (WG / 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_wait<WG>, dim3(grid), dim3(WG), 0, 0, cyc, d_s, d_e);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

// launch = generations x (wait / f + c): two waits give the clock f of the counter and the cost c of replacing a wave
// (the s_memtime counters of the XCDs have different origins: start / end stamps of different waves cannot be compared)
template <int WG>
static void fit(int ncu, uint64_t* d_s, uint64_t* d_e, int nw) {
    const uint32_t w0 = 100000u, w1 = 750000u;
    const float m0 = run<WG>(w0, d_s, d_e, nw), m1 = run<WG>(w1, d_s, d_e, nw);
    const int gens = (nw / (ncu * 4) + 4) / 5;
    const double f = gens * (double)(w1 - w0) / ((m1 - m0) * 1e-3);
    const double c = m0 * 1e-3 / gens - w0 / f;
    printf("workgroups of %3d threads: %d generations of waves per slot; wait %u ticks: %.3f ms, wait %u ticks: %.3f ms -> counter %.3f GHz, "
           "%.2f us to replace a wave\n", WG, gens, w0, m0, w1, m1, f * 1e-9, c * 1e6);
}

int main() {
    int ncu = 256;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int nw = ncu * 256;
    uint64_t *d_s, *d_e;
    hipMalloc(&d_s, 8 * nw); hipMalloc(&d_e, 8 * nw);
    fit<64>(ncu, d_s, d_e, nw);
    fit<128>(ncu, d_s, d_e, nw);
    fit<256>(ncu, d_s, d_e, nw);
    return 0;
}
