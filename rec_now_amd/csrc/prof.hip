#include "prof.hpp"
#include <vector>

static bool g_on = false;
static std::vector<RnProfRecord> g_pool;
static size_t g_used = 0;
static int g_every = 1;          // time one launch in g_every (timing events serialise the stream around the launch)
static unsigned g_seq = 0;
static int g_dropped = 0;         // records wanted while the pool was full (since the last enable / collect)

bool rn_prof_on() { return g_on; }

RnProfRecord* rn_prof_begin(int tag, double flops, double bytes, hipStream_t st) {
    if (!g_on) return nullptr;
    if (tag >= RN_TAG_FIRST_PHASE && g_every != 1) return nullptr;      // phase tags: every-launch mode only (prof.hpp)
    if ((g_seq++ % (unsigned)g_every) != 0) return nullptr;
    if (g_used >= g_pool.size()) { ++g_dropped; return nullptr; }
    RnProfRecord* r = &g_pool[g_used++];
    r->closed = false;
    r->tag = tag;
    r->flops = flops;
    r->bytes = bytes;
    (void)hipEventRecord(r->e0, st);
    return r;
}
void rn_prof_end(RnProfRecord* r, hipStream_t st) {
    if (r) { (void)hipEventRecord(r->e1, st); r->closed = true; }
}

// capacity > 0: (re)arm with that many launch slots; capacity == 0: disable and free.
extern "C" int recnow_prof_enable(int capacity) {
    for (auto& r : g_pool) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_pool.clear();
    g_used = 0;
    g_dropped = 0;
    g_on = false;
    if (capacity <= 0) return RECNOW_OK;
    g_pool.resize((size_t)capacity);
    for (auto& r : g_pool) {
        RN_HIP(hipEventCreate(&r.e0));
        RN_HIP(hipEventCreate(&r.e1));
    }
    g_on = true;
    return RECNOW_OK;
}

// Time every n-th GEMM launch only (n >= 1; choose n coprime to the launches per step so every launch position is sampled).
extern "C" int recnow_prof_sample_every(int n) {
    if (n < 1) return RECNOW_EINVAL;
    g_every = n;
    g_seq = 0;
    return RECNOW_OK;
}

// Synchronises, then fills per-tag totals (arrays of RN_TAG_MAX entries, HOST memory) and rewinds the pool.
extern "C" int recnow_prof_collect(int* count_host, double* ms_host, double* flops_host, double* bytes_host) {
    if (!count_host || !ms_host || !flops_host) return RECNOW_EINVAL;
    for (int t = 0; t < RN_TAG_MAX; ++t) {
        count_host[t] = 0;
        ms_host[t] = 0.0;
        flops_host[t] = 0.0;
        if (bytes_host) bytes_host[t] = 0.0;
    }
    for (size_t i = 0; i < g_used; ++i) {
        RnProfRecord& r = g_pool[i];
        if (!r.closed) continue;             // (an error return between begin and end: e1 was never recorded)
        RN_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        RN_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        if (r.tag >= 0 && r.tag < RN_TAG_MAX) {
            count_host[r.tag] += 1;
            ms_host[r.tag] += ms;
            flops_host[r.tag] += r.flops;
            if (bytes_host) bytes_host[r.tag] += r.bytes;
        }
    }
    g_used = 0;
    return RECNOW_OK;
}

// Entries of the per-tag arrays recnow_prof_collect fills (a caller sizes its host arrays from this instead of a literal), and the records that
// found the pool full since the last recnow_prof_enable / the last call of this function (a non-zero value: the totals under-report).
extern "C" int recnow_prof_tag_count(void) { return RN_TAG_MAX; }
extern "C" int recnow_prof_dropped(void) { const int d = g_dropped; g_dropped = 0; return d; }


// Synchronises and returns the recorded launches one by one: tag and [t0, t1] in milliseconds after the FIRST record's start (events of
// different streams of one device share a clock).  With recnow_prof_sample_every(1) this is the timeline of every hooked launch and phase:
// what overlaps under a second stream can then be told apart (bench.py: exclusive time per kernel family).  Rewinds the pool like
// recnow_prof_collect (use one or the other per measurement).  Returns the number of records written (<= capacity) or a negative code.
extern "C" int recnow_prof_intervals(int* tag_host, double* t0_ms_host, double* t1_ms_host, int capacity) {
    if (!tag_host || !t0_ms_host || !t1_ms_host || capacity < 0) return RECNOW_EINVAL;
    int n = 0;
    for (size_t i = 0; i < g_used && n < capacity; ++i) {
        RnProfRecord& r = g_pool[i];
        if (!r.closed) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) return RECNOW_EINVAL;
        float a = 0.f, b = 0.f;
        if (i > 0 && hipEventElapsedTime(&a, g_pool[0].e0, r.e0) != hipSuccess) return RECNOW_EINVAL;
        if (hipEventElapsedTime(&b, g_pool[0].e0, r.e1) != hipSuccess) return RECNOW_EINVAL;
        tag_host[n] = r.tag; t0_ms_host[n] = a; t1_ms_host[n] = b;
        ++n;
    }
    g_used = 0;
    return n;
}


// ---- events owned through the C ABI (the per-layer "gradients issued" events of recnow_dcn_mix_score_bwd) -----------------
extern "C" int recnow_event_create(void** event_out) {
    if (!event_out) return RECNOW_EINVAL;
    hipEvent_t e = nullptr;
    RN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event_out = (void*)e;
    return RECNOW_OK;
}
extern "C" int recnow_event_destroy(void* event) {
    if (event) RN_HIP(hipEventDestroy((hipEvent_t)event));
    return RECNOW_OK;
}
extern "C" int recnow_event_record(void* event, void* stream) {
    if (!event) return RECNOW_EINVAL;
    RN_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return RECNOW_OK;
}
extern "C" int recnow_stream_wait_event(void* stream, void* event) {
    if (!event) return RECNOW_EINVAL;
    RN_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return RECNOW_OK;
}


// x[i] *= 1 / (count[0] + eps)  (the gradient buckets of dp.LayerwiseReducer: count = the global pair count, a device scalar)
__global__ void __launch_bounds__(256) k_scale_by_inv_count(float* __restrict__ x, int64_t n, const float* __restrict__ count, float eps,
                                                            const float* __restrict__ loss_sum, float* __restrict__ stats_out) {
    const float inv = 1.f / (count[0] + eps);
    if (stats_out && blockIdx.x == 0 && threadIdx.x == 0) {      // (global mean loss, global count): what the step returns
        stats_out[0] = loss_sum[0] * inv;
        stats_out[1] = count[0];
    }
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = reinterpret_cast<float4*>(x)[i];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        reinterpret_cast<float4*>(x)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[n4 * 4 + threadIdx.x] *= inv;
}
extern "C" int recnow_scale_by_inv_count(float* x, int64_t n, const float* count, float eps, const float* loss_sum, float* stats_out,
                                         void* stream) {
    if (n < 0 || (stats_out && !loss_sum)) return RECNOW_EINVAL;
    if (n == 0 && !stats_out) return RECNOW_OK;
    if (!x || !count || ((uintptr_t)x & 15)) return RECNOW_EINVAL;
    int64_t g = (n / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_scale_by_inv_count, (int)g, 256, 0, (hipStream_t)stream, x, n, count, eps, loss_sum, stats_out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
