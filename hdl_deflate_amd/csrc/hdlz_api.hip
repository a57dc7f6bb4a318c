// hdlz_api.hip -- the extern "C" surface of libhdlz.so (declared in include/hdlz.h).
// Host-side parameter checks + kernel launches; no CPU compute path exists here by design.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include "hdlz_device.h"

namespace {
thread_local char g_err[256] = "";

__global__ void k_set_result(uint32_t* out_len, uint32_t* status, uint32_t code) {
    *out_len = 0;
    *status = code;
}

int fail_param(const char* what) {
    snprintf(g_err, sizeof(g_err), "bad parameter: %s", what);
    return HDLZ_E_BAD_PARAM;
}
// The entry points that draw scratch from the library's pool (hipMallocFromPoolAsync / hipFreeAsync on the caller's stream) refuse a
// stream that is being captured: the calls would become graph memory nodes, and on ROCm 7.2 the memory such a node hands out does not
// keep what the graph's own kernel nodes write to it (tools/repro/graph_scratch.hip: a kernel node reads ZEROS where the node in front
// of it wrote, 2-3 million words in 60 replays, no libhdlz involved; with libhdlz: wrong bytes, a hang, and -- round 5 -- an abort in
// hipGraphLaunch when k_par_* followed garbage lists out of bounds; profiles/r06_graph_abort_cause.txt).  The _ws entry points
// allocate nothing and are capturable.
int refuse_capture(void* stream, const char* use) {
#ifdef HDLZ_ALLOW_POOL_IN_CAPTURE                         // (lib/libhdlz_poolcap.so: tools/repro/graph_abort.py shows what then happens)
    return HDLZ_OK;
#endif
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const hipError_t e = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs);
    if (e != hipSuccess) { (void)hipGetLastError(); return HDLZ_OK; }
    if (cs == hipStreamCaptureStatusNone) return HDLZ_OK;
    snprintf(g_err, sizeof(g_err), "bad parameter: the stream is being captured into a graph and this entry point allocates "
                                   "stream-ordered scratch: use %s (caller-owned scratch)", use);
    return HDLZ_E_BAD_PARAM;
}
int fail_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return HDLZ_E_HIP;
}
int check_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail_hip(e, "hipGetDeviceCount");
    if (n <= 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device visible: libhdlz has no CPU path");
        return HDLZ_E_HIP;
    }
    return HDLZ_OK;
}
}  // namespace

namespace hdlz {
__global__ __launch_bounds__(64) void k_zero_words(uint32_t* p, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += 64u) p[i] = 0u;
}
hipError_t zero_words(uint32_t* p, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, stream, p, n);
    return hipGetLastError();
}
// The library's own per-device pool for stream-ordered scratch.  Release threshold: SCRATCH_KEEP bytes stay cached between calls (the
// per-call lists of a 2^20-stream batch are 4 MB, the markers of a 16 MiB port stream 128 MB: neither pays a device allocation per
// call); anything above it goes back to the device at the next synchronisation point of the stream, so one 256 MiB single-stream
// inflate (2 GiB of markers) does not keep 2 GiB of HBM away from the caller's own allocator for the life of the process.
// hdlz_release_scratch() trims the rest.
constexpr uint64_t SCRATCH_KEEP = 256ull << 20;
static hipMemPool_t g_pools[64] = {nullptr};
static bool g_pool_tried[64] = {false};
static std::mutex g_pool_mu;                             // (callers may drive several streams / devices from several threads)
hipError_t scratch_alloc(void** p, size_t bytes, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (dev >= 0 && dev < 64 && !g_pool_tried[dev]) {
        g_pool_tried[dev] = true;
        hipMemPoolProps props;
        memset(&props, 0, sizeof(props));
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess) {
            uint64_t keep = SCRATCH_KEEP;
            if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess) g_pools[dev] = pool;
            else (void)hipMemPoolDestroy(pool);
        }
        (void)hipGetLastError();
    }
    if (dev >= 0 && dev < 64 && g_pools[dev]) {
        const hipError_t e = hipMallocFromPoolAsync(p, bytes, g_pools[dev], stream);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    return hipMallocAsync(p, bytes, stream);
}
hipError_t scratch_release() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (dev >= 0 && dev < 64 && g_pools[dev]) return hipMemPoolTrimTo(g_pools[dev], 0);
    return hipSuccess;
}
hipError_t Work::get(size_t need, hipStream_t stream, uint8_t** p) const {
    if (caller) {                                         // the caller's buffer or nothing: this call allocates no memory
        if (!base || need > bytes) return hipErrorOutOfMemory;
        *p = base;
        return hipSuccess;
    }
    return scratch_alloc(reinterpret_cast<void**>(p), need, stream);
}
hipError_t Work::put(uint8_t* p, hipStream_t stream) const { return caller ? hipSuccess : hipFreeAsync(p, stream); }
}  // namespace hdlz

extern "C" {

int hdlz_version(void) { return HDLZ_VERSION; }

const char* hdlz_last_error(void) { return g_err; }

const char* hdlz_status_string(int s) {
    switch (s) {
        case HDLZ_OK: return "OK";
        case HDLZ_E_SHORT_INPUT: return "SHORT_INPUT (N < 5)";
        case HDLZ_E_OUT_CAPACITY: return "OUT_CAPACITY";
        case HDLZ_E_BAD_BTYPE: return "BAD_BTYPE";
        case HDLZ_E_BAD_DISTANCE: return "BAD_DISTANCE";
        case HDLZ_E_NO_EOF: return "NO_EOF";
        case HDLZ_E_DYNAMIC_UNSUPPORTED: return "DYNAMIC_UNSUPPORTED";
        case HDLZ_E_BAD_SYMBOL: return "BAD_SYMBOL";
        case HDLZ_E_BAD_PARAM: return "BAD_PARAM";
        case HDLZ_E_HIP: return "HIP_ERROR";
        case HDLZ_E_BAD_TREE: return "BAD_TREE";
        default: return "?";
    }
}

int hdlz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int good = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) good++;
    }
    return good;
}

size_t hdlz_out_bound(size_t n) { return 6 + (9 * n + 10 + 7) / 8; }

int hdlz_release_scratch(void) {
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    const hipError_t e = hdlz::scratch_release();
    if (e != hipSuccess) return fail_hip(e, "hipMemPoolTrimTo");
    return HDLZ_OK;
}

int hdlz_compress_batch(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                        uint64_t nblocks, int cwindow, int maxmatch, uint8_t* d_out, uint64_t out_pitch,
                        uint32_t* d_out_len, uint32_t* d_status, void* stream) {
    if (cwindow < 1 || cwindow > 256) return fail_param("cwindow must be in [1,256]");
    if (maxmatch != 5 && maxmatch != 10) return fail_param("maxmatch must be 5 (MATCH10=False) or 10 (MATCH10=True)");
    if (nblocks > 0x7FFFFFFFull) return fail_param("nblocks too large for one launch");
    if (nblocks && (!d_in || !d_out || !d_out_len || !d_status)) return fail_param("null device pointer");
    if ((out_pitch & 3u) || (reinterpret_cast<uintptr_t>(d_out) & 3u)) return fail_param("d_out / out_pitch must be 4-byte aligned");
    if (!d_in_off && in_len >= 0x80000000u) return fail_param("in_len too large");
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    hdlz::CompressArgs a{d_in, d_in_off, in_pitch, in_len, nblocks, cwindow, maxmatch, d_out, out_pitch, d_out_len, d_status};
    hipError_t e = hdlz::launch_compress(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "launch k_compress");
    return HDLZ_OK;
}

#ifndef HDLZ_PAR_BATCH_SHORT_MAX
#define HDLZ_PAR_BATCH_SHORT_MAX HDLZ_INFLATE_PAR_BATCH_SHORT_MAX
#endif
#ifndef HDLZ_PAR_BATCH_MAX
#define HDLZ_PAR_BATCH_MAX HDLZ_INFLATE_PAR_BATCH_MAX      // (A/B builds override it: tools/bench_few_large_inflate.py)
#endif
}  // extern "C"

namespace {
// does a batch of this shape take the whole-GPU path (hdlz_inflate_par.hip)?  ONE large stream: cut into pieces and decoded by the whole
// GPU.  A FEW large streams: the same chain with every kernel launched once for all of them (blockIdx.y = the stream): the batch kernels
// decode a stream as ONE serial chain (64 KiB: 5.9 ms, 1 MiB: 94 ms, however few there are); this path costs the launch chain once plus
// the streams' bytes at the rate of the single-stream path (profiles/r05_inflate_mapping.txt).  Streams below HDLZ_INFLATE_PAR_LONG bytes:
// up to HDLZ_INFLATE_PAR_BATCH_SHORT_MAX of them.  Ragged input: in_len is the caller's upper bound on the stream lengths (0 = not stated).
// The explicit mapping hints keep the batch kernels.
bool par_applies(uint64_t nstreams, uint32_t in_len, uint32_t flags) {
    if (in_len < HDLZ_INFLATE_PAR_MIN) return false;
    if (flags & (HDLZ_INFLATE_LANE_PER_STREAM | HDLZ_INFLATE_WAVE_PER_STREAM)) return false;
    if (nstreams == 1) return true;
    if (flags & HDLZ_INFLATE_GROUP_PER_STREAM) return false;
    return nstreams <= (in_len >= HDLZ_INFLATE_PAR_LONG ? HDLZ_PAR_BATCH_MAX : HDLZ_PAR_BATCH_SHORT_MAX);
}

int inflate_batch_impl(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                       uint64_t nstreams, uint32_t flags, uint32_t obsize, uint8_t* d_out, uint64_t out_pitch,
                       uint32_t* d_out_len, uint32_t* d_status, const hdlz::Work& w, void* stream) {
    if (nstreams > 0x7FFFFFFFull * 32) return fail_param("nstreams too large for one launch");
    if (nstreams && (!d_in || !d_out || !d_out_len || !d_status)) return fail_param("null device pointer");
    if (flags & ~(HDLZ_INFLATE_ASSUME_FIXED | HDLZ_INFLATE_LANE_PER_STREAM | HDLZ_INFLATE_WAVE_PER_STREAM | HDLZ_INFLATE_ONEBLOCK |
                  HDLZ_INFLATE_GROUP_PER_STREAM | HDLZ_INFLATE_ONE_FIXED_BLOCK))
        return fail_param("unknown flag");
    // the kernels keep stream lengths and bit positions in 32 bits (8 * length must not wrap)
    if (!d_in_off && in_len >= 0x10000000u) return fail_param("in_len too large (streams are limited to 256 MiB - 1)");
    if (d_in_off && in_len >= 0x10000000u) in_len = 0;      // ragged: a bound nobody can use is no bound
    if (!d_in_off && nstreams > 1 && in_pitch < in_len) return fail_param("in_pitch < in_len");
    {
        const uint32_t mf = flags & (HDLZ_INFLATE_LANE_PER_STREAM | HDLZ_INFLATE_WAVE_PER_STREAM | HDLZ_INFLATE_GROUP_PER_STREAM);
        if (mf & (mf - 1u)) return fail_param("contradictory mapping flags");
    }
    if ((out_pitch & 3u) || (reinterpret_cast<uintptr_t>(d_out) & 3u)) return fail_param("d_out / out_pitch must be 4-byte aligned");
    if (w.caller && w.base && (reinterpret_cast<uintptr_t>(w.base) & 255u)) return fail_param("d_work must be 256-byte aligned");
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    if (nstreams == 0) return HDLZ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hdlz::InflateArgs a{d_in, d_in_off, in_pitch, in_len, nstreams, flags, obsize, d_out, out_pitch, d_out_len, d_status};
    // mapping: one LANE per stream (k_inflate_tok, 64 streams in lockstep per wave) needs ~10^5 streams to fill the GPU;
    // below HDLZ_INFLATE_WAVE_THRESHOLD streams one WAVE per stream (k_inflate_dyn, window decode) is faster, for any block type;
    // in between: 16 lanes per stream (k_inflate_grp) -- from HDLZ_INFLATE_GROUP_MIN streams on it beats a wave per stream,
    // up to HDLZ_INFLATE_GROUP_MAX a lane per stream; the streams it flags (dynamic-tree blocks) take the second pass below
    const uint32_t hint = flags & (HDLZ_INFLATE_LANE_PER_STREAM | HDLZ_INFLATE_WAVE_PER_STREAM | HDLZ_INFLATE_GROUP_PER_STREAM);
    if ((flags & HDLZ_INFLATE_GROUP_PER_STREAM) && nstreams > 0x7FFFFFFFull * 16) return fail_param("nstreams too large for the 16-lanes-per-stream mapping");
    const bool group = (flags & HDLZ_INFLATE_GROUP_PER_STREAM) ||
                       (hint == 0u && nstreams >= HDLZ_INFLATE_GROUP_MIN && nstreams <= HDLZ_INFLATE_GROUP_MAX);
    const bool wave_all = !group && ((flags & HDLZ_INFLATE_WAVE_PER_STREAM) ||
                                     (!(flags & HDLZ_INFLATE_LANE_PER_STREAM) && nstreams <= HDLZ_INFLATE_WAVE_THRESHOLD));
    if (par_applies(nstreams, in_len, flags)) {
        bool used = false;
        hipError_t e = hdlz::launch_inflate_par(a, st, &used, w);
        if (e != hipSuccess) return fail_hip(e, "launch the whole-GPU inflate (k_par_*)");
        if (used) return HDLZ_OK;
        // (no scratch, or a shape the path leaves alone: the batch kernels below)
    }
    if (wave_all) {
        hipError_t e = hdlz::launch_inflate_dyn(a, st, true);
        if (e != hipSuccess) return fail_hip(e, "launch k_inflate_dyn");
        return HDLZ_OK;
    }
    // lane per stream, one TOKEN group per round (k_inflate_tok), or 16 lanes per stream
    hipError_t e = group ? hdlz::launch_inflate_grp(a, st) : hdlz::launch_inflate_tok(a, st, w);
    if (e != hipSuccess) return fail_hip(e, "launch k_inflate");
    // second pass, same stream: streams in which pass 1 met a dynamic-tree block (status 6) are redone, again one lane each
    // (k_inflate_tok<true>); everything else is left untouched
    e = nstreams > 0xFFFFFFFFull ? hdlz::launch_inflate_dyn(a, st, false) : hdlz::launch_inflate_tok_dyn(a, st, false, w);
    if (e != hipSuccess) return fail_hip(e, "launch the dynamic-tree pass");
    return HDLZ_OK;
}
}  // namespace

extern "C" {

size_t hdlz_inflate_work_bytes(uint64_t nstreams, uint32_t in_len, uint64_t out_pitch, uint32_t flags, int ragged) {
    if (nstreams == 0) return 0;
    size_t need = hdlz::inflate_tok_work_bytes(nstreams, ragged != 0);
    if (ragged && in_len >= 0x10000000u) in_len = 0;
    if (par_applies(nstreams, in_len, flags)) {
        const size_t p = hdlz::inflate_par_work_bytes(in_len, nstreams, out_pitch, flags);
        if (p > need) need = p;
    }
    return need < 256u ? 256u : need;
}

int hdlz_inflate_batch_ws(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                          uint64_t nstreams, uint32_t flags, uint32_t obsize, uint8_t* d_out, uint64_t out_pitch,
                          uint32_t* d_out_len, uint32_t* d_status, void* d_work, size_t work_bytes, void* stream) {
    const hdlz::Work w{static_cast<uint8_t*>(d_work), d_work ? work_bytes : 0u, true};
    return inflate_batch_impl(d_in, d_in_off, in_pitch, in_len, nstreams, flags, obsize, d_out, out_pitch, d_out_len, d_status, w, stream);
}

int hdlz_inflate_batch(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                       uint64_t nstreams, uint32_t flags, uint32_t obsize, uint8_t* d_out, uint64_t out_pitch,
                       uint32_t* d_out_len, uint32_t* d_status, void* stream) {
    if (nstreams) { const int rc = refuse_capture(stream, "hdlz_inflate_batch_ws"); if (rc != HDLZ_OK) return rc; }
    const hdlz::Work w{nullptr, 0u, false};                  // scratch from the library's stream-ordered pool
    return inflate_batch_impl(d_in, d_in_off, in_pitch, in_len, nstreams, flags, obsize, d_out, out_pitch, d_out_len, d_status, w, stream);
}

int hdlz_compact_batch(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, const uint64_t* d_off,
                       uint64_t nblocks, uint8_t* d_archive, void* stream) {
    if (nblocks && (!d_rows || !d_len || !d_off || !d_archive)) return fail_param("null device pointer");
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    hipError_t e = hdlz::launch_compact(d_rows, row_pitch, d_len, d_off, nblocks, d_archive, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "launch k_compact");
    return HDLZ_OK;
}

static int archive_impl(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, uint64_t nblocks,
                        uint8_t* d_archive, uint64_t archive_cap, uint64_t* d_off, const hdlz::Work& w, void* stream) {
    if (!d_off || (nblocks && (!d_rows || !d_len || !d_archive))) return fail_param("null device pointer");
    if (nblocks > 0x7FFFFFFFull) return fail_param("nblocks too large for one launch (2^31 - 1 rows)");   // (32-bit tile and word counts)
    if (w.caller && nblocks && (!w.base || w.bytes < hdlz::archive_work_bytes(nblocks))) return fail_param("d_work smaller than hdlz_archive_work_bytes(nblocks)");
    if (w.caller && (reinterpret_cast<uintptr_t>(w.base) & 7u)) return fail_param("d_work must be 8-byte aligned");
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    hipError_t e = hdlz::launch_archive(d_rows, row_pitch, d_len, nblocks, d_archive, archive_cap, d_off, static_cast<hipStream_t>(stream), w);
    if (e != hipSuccess) return fail_hip(e, "launch k_archive");
    return HDLZ_OK;
}

size_t hdlz_archive_work_bytes(uint64_t nblocks) { return nblocks > 0x7FFFFFFFull ? 0u : hdlz::archive_work_bytes(nblocks); }

int hdlz_archive_batch_ws(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, uint64_t nblocks,
                          uint8_t* d_archive, uint64_t archive_cap, uint64_t* d_off, void* d_work, size_t work_bytes, void* stream) {
    const hdlz::Work w{static_cast<uint8_t*>(d_work), d_work ? work_bytes : 0u, true};
    return archive_impl(d_rows, row_pitch, d_len, nblocks, d_archive, archive_cap, d_off, w, stream);
}

int hdlz_archive_batch(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, uint64_t nblocks,
                       uint8_t* d_archive, uint64_t archive_cap, uint64_t* d_off, void* stream) {
    if (nblocks) { const int rc = refuse_capture(stream, "hdlz_archive_batch_ws"); if (rc != HDLZ_OK) return rc; }
    const hdlz::Work w{nullptr, 0u, false};
    return archive_impl(d_rows, row_pitch, d_len, nblocks, d_archive, archive_cap, d_off, w, stream);
}

size_t hdlz_stream_work_bytes(size_t in_len) { return hdlz_streams_work_bytes(in_len, 1); }

size_t hdlz_streams_work_bytes(size_t in_len, uint64_t nblocks) {
    if (in_len >= 0x80000000ull || nblocks == 0 || nblocks > 0xFFFFFFFFull) return 0;
    if (((in_len + 2047) / 2048) * nblocks >= 0x80000000ull) return 0;          // tile indices are 32-bit
    return hdlz::stream_work_bytes((uint32_t)in_len, (uint32_t)nblocks);
}

int hdlz_compress_stream(const uint8_t* d_in, uint32_t in_len, int cwindow, int maxmatch, uint8_t* d_out,
                         uint64_t out_cap, uint32_t* d_out_len, uint32_t* d_status, void* d_work,
                         size_t work_bytes, void* stream) {
    return hdlz_compress_streams(d_in, 0, in_len, 1, cwindow, maxmatch, d_out, out_cap, d_out_len, d_status, d_work,
                                 work_bytes, stream);
}

int hdlz_compress_streams(const uint8_t* d_in, uint64_t in_pitch, uint32_t in_len, uint64_t nblocks, int cwindow,
                          int maxmatch, uint8_t* d_out, uint64_t out_pitch, uint32_t* d_out_len, uint32_t* d_status,
                          void* d_work, size_t work_bytes, void* stream) {
    if (cwindow < 1 || cwindow > 256) return fail_param("cwindow must be in [1,256]");
    if (maxmatch != 5 && maxmatch != 10) return fail_param("maxmatch must be 5 (MATCH10=False) or 10 (MATCH10=True)");
    if (nblocks == 0) return HDLZ_OK;
    if (!d_in || !d_out || !d_out_len || !d_status || !d_work) return fail_param("null device pointer");
    if (in_len >= 0x80000000u) return fail_param("in_len too large");
    if (reinterpret_cast<uintptr_t>(d_out) & 3u) return fail_param("d_out must be 4-byte aligned");
    if (nblocks > 1 && (out_pitch & 3u)) return fail_param("out_pitch must be a multiple of 4");
    if (reinterpret_cast<uintptr_t>(d_work) & 7u) return fail_param("d_work must be 8-byte aligned");
    const size_t need_work = hdlz_streams_work_bytes(in_len, nblocks);
    if (need_work == 0) return fail_param("in_len x nblocks too large for one call");
    if (work_bytes < need_work) return fail_param("d_work smaller than hdlz_streams_work_bytes(in_len, nblocks)");
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint64_t need = ((uint64_t)hdlz::out_bound(in_len) + 3u) & ~3ull;
    if (in_len < 5u || out_pitch < need) {            // R0 / capacity: same per-stream status as the batch call
        if (nblocks > 1) return fail_param(in_len < 5u ? "in_len < 5" : "out_pitch smaller than hdlz_out_bound(in_len)");
        // (a kernel, not hipMemsetD32Async: 32-bit memsets did not survive HIP-graph capture on this stack)
        hipLaunchKernelGGL(k_set_result, dim3(1), dim3(1), 0, st, d_out_len, d_status,
                           in_len < 5u ? (uint32_t)HDLZ_E_SHORT_INPUT : (uint32_t)HDLZ_E_OUT_CAPACITY);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "launch k_set_result");
        return HDLZ_OK;
    }
    hipError_t e = hdlz::launch_compress_streams(d_in, in_pitch, in_len, (uint32_t)nblocks, cwindow, maxmatch, d_out, out_pitch,
                                                 d_out_len, d_status, d_work, st);
    if (e != hipSuccess) return fail_hip(e, "launch k_stream_*");
    return HDLZ_OK;
}

int hdlz_compress_chunk(const uint8_t* d_in, uint32_t in_len, uint32_t q_end, int final, int cwindow, int maxmatch,
                        uint8_t* d_out, uint64_t out_cap, void* d_state, void* stream) {
    if (cwindow < 1 || cwindow > 256) return fail_param("cwindow must be in [1,256]");
    if (maxmatch != 5 && maxmatch != 10) return fail_param("maxmatch must be 5 (MATCH10=False) or 10 (MATCH10=True)");
    if (!d_in || !d_out || !d_state) return fail_param("null device pointer");
    if (in_len >= 0x80000000u) return fail_param("in_len too large");
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) || (reinterpret_cast<uintptr_t>(d_state) & 3u)) return fail_param("d_out / d_state must be 4-byte aligned");
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    hipError_t e = hdlz::launch_compress_chunk(d_in, in_len, q_end, final, cwindow, maxmatch, d_out, out_cap, d_state,
                                               static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "launch k_compress_chunk");
    return HDLZ_OK;
}

int hdlz_inflate_chunk(const uint8_t* d_in, uint32_t in_len, int final, uint32_t flags, uint32_t obsize, uint8_t* d_out,
                       uint64_t out_cap, uint32_t out_limit, void* d_state, void* stream) {
    if (flags & ~(HDLZ_INFLATE_ASSUME_FIXED | HDLZ_INFLATE_ONEBLOCK)) return fail_param("unknown flag");
    if (!d_in || !d_out || !d_state) return fail_param("null device pointer");
    if (in_len >= 0x10000000u) return fail_param("in_len too large (streams are limited to 256 MiB - 1)");
    if ((reinterpret_cast<uintptr_t>(d_out) & 3u) || (reinterpret_cast<uintptr_t>(d_state) & 3u)) return fail_param("d_out / d_state must be 4-byte aligned");
    int rc = check_device();
    if (rc != HDLZ_OK) return rc;
    hipError_t e = hdlz::launch_inflate_chunk(d_in, in_len, final, flags, obsize, d_out, out_cap, out_limit, d_state,
                                              static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "launch k_inflate_dyn<stream>");
    return HDLZ_OK;
}

}  // extern "C"
