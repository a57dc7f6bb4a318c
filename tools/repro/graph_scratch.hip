// graph_scratch.hip -- is the hipGraphLaunch abort of round 5 the RUNTIME's?  No libhdlz in here: a captured graph
// of NCALL "calls", each = stream-ordered scratch (hipMallocFromPoolAsync / hipMallocAsync) -> a kernel chain writing and checking the
// scratch -> hipFreeAsync, replayed REPS times.     hipcc --offload-arch=gfx950 -O2 graph_scratch.hip -o graph_scratch
//   usage: graph_scratch [pool|default|none] [ncall] [reps] [mib_of_first] [grid_y] [sync_every]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ void k_fill(uint32_t* p, size_t n, uint32_t tag) {
    for (size_t i = (blockIdx.y * gridDim.x + blockIdx.x) * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * gridDim.y * blockDim.x) p[i] = tag ^ (uint32_t)i;
}
__global__ void k_check(const uint32_t* p, size_t n, uint32_t tag, uint32_t* bad) {       // bad[0] = count; bad[1 + 3j ..]: index, expected tag, tag found
    for (size_t i = (blockIdx.y * gridDim.x + blockIdx.x) * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * gridDim.y * blockDim.x)
        if (p[i] != (tag ^ (uint32_t)i)) { const uint32_t j = atomicAdd(bad, 1u); if (j < 8u) { bad[1 + 3 * j] = (uint32_t)i; bad[2 + 3 * j] = tag; bad[3 + 3 * j] = p[i] ^ (uint32_t)i; } }
}
int main(int argc, char** argv) {
    const bool pool_mode = argc < 2 || !strcmp(argv[1], "pool"), none_mode = argc > 1 && !strcmp(argv[1], "none");   // none: ONE hipMalloc'd buffer, no mem nodes (control)
    const int sync_every = argc > 6 ? atoi(argv[6]) : 3;
    const int ncall = argc > 2 ? atoi(argv[2]) : 4, reps = argc > 3 ? atoi(argv[3]) : 200;
    const size_t mib = argc > 4 ? atoi(argv[4]) : 16;
    const unsigned gy = argc > 5 ? atoi(argv[5]) : 512;
    hipMemPool_t pool = nullptr;
    if (pool_mode) {
        hipMemPoolProps pr; memset(&pr, 0, sizeof(pr));
        pr.allocType = hipMemAllocationTypePinned; pr.handleTypes = hipMemHandleTypeNone; pr.location.type = hipMemLocationTypeDevice; pr.location.id = 0;
        CK(hipMemPoolCreate(&pool, &pr));
        uint64_t keep = 256ull << 20; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    uint32_t* bad; CK(hipMalloc(&bad, 128)); CK(hipMemset(bad, 0, 128));
    uint32_t* fixed = nullptr; if (none_mode) CK(hipMalloc((void**)&fixed, (mib << 20) + 4096));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int c = 0; c < ncall; c++) {
        const size_t n = ((mib << 20) >> (2 * (c % 3))) / 4 + 64 * c;            // 16 MiB, 4 MiB, 1 MiB, 16 MiB + ...: the sizes differ per call
        uint32_t* w = nullptr;
        if (none_mode) w = fixed; else CK(pool_mode ? hipMallocFromPoolAsync((void**)&w, 4 * n, pool, s) : hipMallocAsync((void**)&w, 4 * n, s));
        printf("  call %d: %zu words at %p\n", c, n, (void*)w);
        for (int k = 0; k < 6; k++) {
            hipLaunchKernelGGL(k_fill, dim3(8, gy), dim3(64), 0, s, w, n, 0x1234u * (c + 1) + k);
            hipLaunchKernelGGL(k_check, dim3(8, gy), dim3(64), 0, s, w, n, 0x1234u * (c + 1) + k, bad);
        }
        if (!none_mode) CK(hipFreeAsync(w, s));
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int first_bad = -1;
    for (int r = 0; r < reps; r++) {
        CK(hipGraphLaunch(ge, s));
        if (r % sync_every == 0) { CK(hipStreamSynchronize(s)); uint32_t b = 0; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); if (b && first_bad < 0) first_bad = r; }
    }
    CK(hipStreamSynchronize(s));
    uint32_t hb[32]; CK(hipMemcpy(hb, bad, 128, hipMemcpyDeviceToHost));
    printf("%s: %d calls x %d launches (sync every %d): %u bad words, first seen after launch %d\n", argc > 1 ? argv[1] : "pool", ncall, reps, sync_every, hb[0], first_bad);
    for (uint32_t j = 0; j < 8u && j < hb[0]; j++) printf("    word %u: expected tag %#x, found tag %#x\n", hb[1 + 3 * j], hb[2 + 3 * j], hb[3 + 3 * j]);
    return hb[0] ? 1 : 0;
}
