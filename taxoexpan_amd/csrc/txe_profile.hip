// Optional per-kernel timing with HIP events on the launch stream (OFF by default).
// bench.py switches it on for one instrumented pass to obtain each kernel's live launch duration together with the
// algorithmic work (flops or bytes) the launcher attributes to that launch -- the `roofline` numbers of the bench line.
// With txe_stream_order's event ring below this is the library's only process-global state; with profiling off the launch path
// does not touch it.
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "txe_common.h"

namespace txe {

struct ProfRec {
    char name[64];
    hipEvent_t a, b;
    double work;
    int kind;  // 0 = flops, 1 = bytes
    hipStream_t stream;
};

static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

bool prof_enabled() { return g_on; }

// compute-unit count of the current device (256 on MI355X), cached per process; 256 if no device is visible (CPU-only
// build hosts only ever size workspaces with it).
int device_cu_count() {
    static int cached = 0;
    if (cached > 0) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
        cached = n;
    else
        return 256;
    return cached;
}

static hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

int prof_begin(const char* name, hipStream_t s, double work, int kind) {
    ProfRec r;
    strncpy(r.name, name, sizeof(r.name) - 1); r.name[sizeof(r.name) - 1] = 0; r.work = work; r.kind = kind;
    r.a = take_event(); r.b = take_event();
    r.stream = s;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}
void prof_end(int id, hipStream_t s) { (void)hipEventRecord(g_recs[id].b, s); }

// dst[i] = src[i] with 16-byte loads and stores, four independent vectors in flight per thread: the streaming-copy ceiling of the
// box (MI355X_MICROARCH.md: ~6.3 TB/s read + written), the yardstick the HBM-bound sweeps are shown against beside the 8 TB/s spec
template <int U>
__global__ __launch_bounds__(256) void copy_f4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    // a workgroup owns U * 256 consecutive vectors: U coalesced 4 KB loads in flight per wave, then the stores
    const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long j = base + u * 256; v[u] = src[j < n4 ? j : n4 - 1]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const long long j = base + u * 256; if (j < n4) dst[j] = v[u]; }
}

}  // namespace txe

using namespace txe;

extern "C" {

// device-to-device streaming copy of n_bytes (a multiple of 16; 16-byte aligned pointers): measurement yardstick, see copy_f4_kernel
int txe_copy_stream(const void* src, void* dst, long long n_bytes, void* stream) {
    if (!src || !dst || n_bytes < 0 || (n_bytes & 15) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return TXE_ERR_ARG;
    if (n_bytes == 0) return TXE_OK;
    const long long n4 = n_bytes / 16;
    constexpr int U = 4;
    const long long nb = (n4 + 256 * U - 1) / (256 * U);
    if (nb > 0x7fffffffLL) return TXE_ERR_ARG;
    hipLaunchKernelGGL(copy_f4_kernel<U>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst, n4);
    TXE_CHECK_LAUNCH();
    return TXE_OK;
}

int txe_profile_enable(int on) {
    g_on = (on != 0);
    return TXE_OK;
}

int txe_profile_reset(void) {
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    return TXE_OK;
}

int txe_profile_count(void) { return (int)g_recs.size(); }

// record i: kernel name (copied into name_buf), elapsed ms between its two events (synchronises on them), the
// algorithmic work of the launch and its kind (0 flops / 1 bytes).
int txe_profile_get(int i, char* name_buf, int buf_len, float* ms, double* work, int* kind) {
    if (i < 0 || i >= (int)g_recs.size() || !name_buf || buf_len < 2 || !ms || !work || !kind) return TXE_ERR_ARG;
    const ProfRec& r = g_recs[i];
    if (hipEventSynchronize(r.b) != hipSuccess) return TXE_ERR_LAUNCH;
    if (hipEventElapsedTime(ms, r.a, r.b) != hipSuccess) return TXE_ERR_LAUNCH;
    strncpy(name_buf, r.name, buf_len - 1);
    name_buf[buf_len - 1] = 0;
    *work = r.work;
    *kind = r.kind;
    return TXE_OK;
}

// stream record i was launched on (a caller that overlaps kernels on a second stream tells the critical path's launches from the
// ones whose duration is stretched by sharing the machine)
int txe_profile_stream(int i, void** stream) {
    if (i < 0 || i >= (int)g_recs.size() || !stream) return TXE_ERR_ARG;
    *stream = (void*)g_recs[i].stream;
    return TXE_OK;
}

// Stream ordering without the system-scope fence: work submitted to `then` after this call starts only when everything submitted to
// `first` before it has completed.  The library overlaps kernels of ONE device on two streams; the events torch (or a plain
// hipEventCreate) records for that carry a system-scope release -- an L2 write-back and invalidate in front of the recording
// stream's next kernel, a 6.5 us bubble on the critical path, six per training step.  These events are created with
// hipEventDisableTiming | hipEventDisableSystemFence: kernel boundaries already order device memory at agent scope, which is all two
// streams of the same device need (host readers synchronise through hipMemcpy / hipStreamSynchronize as before).
// A ring of events per process; an event is re-recorded only after 64 later orderings, long after its wait was consumed.
int txe_stream_order(void* first, void* then) {
    constexpr int MAXDEV = 16;
    static hipEvent_t ring[MAXDEV][64];                 // (events belong to the device that was current when they were created)
    static bool made[MAXDEV];                           // a device's 64 events are created together, once, under its lock
    static std::mutex make_lock[MAXDEV];
    static std::atomic<unsigned> next{0};
    if (first == then) return TXE_OK;
    int cur = 0;
    hipDevice_t dev = 0;                                // the streams' device (a null stream: the current device's)
    if (hipGetDevice(&cur) != hipSuccess) return TXE_ERR_LAUNCH;
    if (hipStreamGetDevice((hipStream_t)(first ? first : then), &dev) != hipSuccess) dev = cur;
    if (dev < 0 || dev >= MAXDEV) return TXE_ERR_ARG;
    const unsigned i = next.fetch_add(1) & 63u;
    {
        std::lock_guard<std::mutex> g(make_lock[dev]);  // (two host threads may arrive here for the same device at once)
        if (!made[dev]) {
            if (dev != cur && hipSetDevice(dev) != hipSuccess) return TXE_ERR_LAUNCH;
            hipError_t e = hipSuccess;
            for (int k = 0; k < 64 && e == hipSuccess; ++k)
                e = hipEventCreateWithFlags(&ring[dev][k], hipEventDisableTiming | hipEventDisableSystemFence);
            if (dev != cur) (void)hipSetDevice(cur);
            if (e != hipSuccess) return TXE_ERR_LAUNCH;
            made[dev] = true;
        }
    }
    if (hipEventRecord(ring[dev][i], (hipStream_t)first) != hipSuccess) return TXE_ERR_LAUNCH;
    if (hipStreamWaitEvent((hipStream_t)then, ring[dev][i], 0) != hipSuccess) return TXE_ERR_LAUNCH;
    return TXE_OK;
}

}  // extern "C"
