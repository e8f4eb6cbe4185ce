// Grouping machinery (integer path): canonical keys, stable LSD radix sort of row indices, device-wide scans,
// segment detection.  Replaces the reference's dense (B,B) same-group mask
// (/root/reference/rec_now/rec_block/pairwise_loss_from_batch.py:16-40,43-74) and tf.unique_with_counts
// (/root/reference/rec_now/rec_block/listwise_loss_from_batch.py:109).
//
// HBM-bound integer work on <= a few MB: everything here is about few launches, coalesced 4-B streams and
// LDS-resident counters, not MFMA.
#include <atomic>
#include "common.hpp"
#include "scan.hpp"
#include "group_small.hpp"
#include "dcnmix_tile.hpp"
#define RN_FRONT_PACK_WGS 255      // pack workgroups of the step's front kernels (k_front_small / k_front_mid)

// ------------------------------------------------------------------------------------------------
// keys
// ------------------------------------------------------------------------------------------------
__global__ void k_keys_f32(const float* __restrict__ g, int64_t B, uint32_t* __restrict__ w0, uint8_t* __restrict__ solo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float v = g[i];
    uint32_t u = __float_as_uint(v);
    if (v == 0.0f) u = 0u;                       // -0.0 == +0.0
    if (!(fabsf(v) < INFINITY)) solo[i] = 1;      // NaN, +-inf: v - v is NaN -> equals nothing
    w0[i] = u;
}
__global__ void k_keys_f64(const double* __restrict__ g, int64_t B, uint32_t* __restrict__ w, uint8_t* __restrict__ solo) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double v = g[i];
    uint64_t u = (uint64_t)__double_as_longlong(v);
    if (v == 0.0) u = 0ull;
    if (!(fabs(v) < (double)INFINITY)) solo[i] = 1;
    w[i] = (uint32_t)(u >> 32);
    w[B + i] = (uint32_t)u;
}
__global__ void k_keys_i32(const int32_t* __restrict__ g, int64_t B, uint32_t* __restrict__ w0) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) w0[i] = (uint32_t)g[i];
}
__global__ void k_keys_i64(const int64_t* __restrict__ g, int64_t B, uint32_t* __restrict__ w) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    uint64_t u = (uint64_t)g[i];
    w[i] = (uint32_t)(u >> 32);
    w[B + i] = (uint32_t)u;
}

extern "C" int recnow_key_words(int dtype) {
    switch (dtype) {
        case RECNOW_KEY_F32: case RECNOW_KEY_I32: return 1;
        case RECNOW_KEY_F64: case RECNOW_KEY_I64: return 2;
        default: return RECNOW_EINVAL;
    }
}

extern "C" int recnow_group_keys(const void* group, int dtype, int64_t B, uint32_t* words, uint8_t* solo, void* stream) {
    if (B < 0 || recnow_key_words(dtype) < 0) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!group || !words || !solo) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int T = 256, G = rn_cdiv(B, T);
    switch (dtype) {
        case RECNOW_KEY_F32: hipLaunchKernelGGL(k_keys_f32, G, T, 0, st, (const float*)group, B, words, solo); break;
        case RECNOW_KEY_F64: hipLaunchKernelGGL(k_keys_f64, G, T, 0, st, (const double*)group, B, words, solo); break;
        case RECNOW_KEY_I32: hipLaunchKernelGGL(k_keys_i32, G, T, 0, st, (const int32_t*)group, B, words); break;
        case RECNOW_KEY_I64: hipLaunchKernelGGL(k_keys_i64, G, T, 0, st, (const int64_t*)group, B, words); break;
    }
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// ------------------------------------------------------------------------------------------------
// radix sort of row indices by n_words x 32-bit keys, 8-bit digits, LSD, stable.
// The index array is permuted, and beside it the values of the key word being sorted travel in the same order, so that
// only the first pass on a word gathers it through the index (for millions of keys every gathered 4-byte word costs a
// 64-byte sector; the first pass of all reads through the identity order, i.e. coalesced).
// Passes whose digit is constant over all rows are skipped on the device (plan), so float-encoded small ids
// cost 2-3 passes instead of 4.
// ------------------------------------------------------------------------------------------------
#define RN_MAX_WORDS 8
#define RN_MAX_PASS (4 * RN_MAX_WORDS)
#define RN_TILE 2048          // keys per block per pass (256 threads x 8)

struct SortPlan {
    int trivial[RN_MAX_PASS];
    int src[RN_MAX_PASS];     // which index buffer (0/1) pass p reads
    int carried[RN_MAX_PASS]; // the key buffer beside that index buffer already holds pass p's word in that order
    int final_buf;            // buffer holding the sorted order after the last pass
    int final_word;           // key word whose values lie beside that order in the key buffer (-1: none)
    int word_const[RN_MAX_WORDS];   // word w has one value over the whole batch (all four digits trivial)
    int pad[2];
};

// Which digits vary over the batch?  A pass is skippable iff its digit is the same in every key, i.e. iff no bit of that
// byte is 1 in some key and 0 in another: mix[w] = OR of the keys' word w, mix[n_words + w] = OR of its complement; the
// bits set in both are the varying ones.  (Full 256-bin histograms of every digit answered the same question with eight
// LDS atomics per key -- 0.15 ms for 6.5 M ids whose hot values serialise on one bin.)  Also writes the identity order.
__global__ void __launch_bounds__(256)
k_sort_ghist(const uint32_t* __restrict__ words, int64_t B, int n_words, unsigned* __restrict__ mix, int32_t* __restrict__ idx0) {
    __shared__ unsigned part[4][2 * RN_MAX_WORDS];
    unsigned o[RN_MAX_WORDS], z[RN_MAX_WORDS];
#pragma unroll
    for (int w = 0; w < RN_MAX_WORDS; ++w) o[w] = z[w] = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        idx0[i] = (int32_t)i;                   // the identity order the first pass starts from
#pragma unroll
        for (int w = 0; w < RN_MAX_WORDS; ++w)
            if (w < n_words) {
                const uint32_t k = words[(int64_t)w * B + i];
                o[w] |= k;
                z[w] |= ~k;
            }
    }
#pragma unroll
    for (int w = 0; w < RN_MAX_WORDS; ++w)
        if (w < n_words) {
            unsigned a = o[w], c = z[w];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) {
                a |= __shfl_xor(a, s, 64);
                c |= __shfl_xor(c, s, 64);
            }
            if ((threadIdx.x & 63) == 0) {
                part[threadIdx.x >> 6][w] = a;
                part[threadIdx.x >> 6][RN_MAX_WORDS + w] = c;
            }
        }
    __syncthreads();
    // one pair of atomics per word and WORKGROUP: thousands of atomics on the same few addresses serialise in the L2
    if ((int)threadIdx.x < 2 * n_words) {
        const int w = threadIdx.x % n_words, half = threadIdx.x / n_words;
        const int col = half * RN_MAX_WORDS + w;
        atomicOr(&mix[half * n_words + w], part[0][col] | part[1][col] | part[2][col] | part[3][col]);
    }
}

// pass p sorts by word w = n_words-1 - p/4, digit d = p%4 (LSD overall)
__global__ void k_sort_plan(const unsigned* __restrict__ mix, int64_t B, int n_words, SortPlan* plan) {
    __shared__ int triv[RN_MAX_PASS];
    const int np = n_words * 4;
    if (threadIdx.x < RN_MAX_PASS) triv[threadIdx.x] = 0;
    __syncthreads();
    if ((int)threadIdx.x < np) {
        const int p = threadIdx.x, w = n_words - 1 - p / 4, d = p % 4;
        triv[p] = (((mix[w] & mix[n_words + w]) >> (8 * d)) & 255u) == 0u;           // no bit of the byte differs between keys
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int cur = 0, word_in_buf = -1;
        for (int p = 0; p < np; ++p) {
            const int w = n_words - 1 - p / 4;
            plan->trivial[p] = triv[p];
            plan->src[p] = cur;
            plan->carried[p] = word_in_buf == w;
            if (!triv[p]) {                      // the scatter of pass p leaves word w beside the permuted indices
                cur ^= 1;
                word_in_buf = w;
            }
        }
        plan->final_buf = cur;
        plan->final_word = word_in_buf;
        for (int w = 0; w < n_words; ++w) {
            int c = 1;
            for (int d = 0; d < 4; ++d) c &= triv[(n_words - 1 - w) * 4 + d];
            plan->word_const[w] = c;
        }
    }
}

__global__ void __launch_bounds__(256)
k_sort_blockhist(const uint32_t* __restrict__ words, int64_t B, int n_words, int pass, const SortPlan* __restrict__ plan,
                 const int32_t* __restrict__ idx0, const int32_t* __restrict__ idx1, const uint32_t* __restrict__ key0,
                 const uint32_t* __restrict__ key1, unsigned* __restrict__ blockhist, int nblk) {
    if (plan->trivial[pass]) return;
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int32_t* src = plan->src[pass] ? idx1 : idx0;
    const uint32_t* ksrc = plan->src[pass] ? key1 : key0;
    const bool carried = plan->carried[pass];
    const uint32_t* wk = words + (int64_t)(n_words - 1 - pass / 4) * B;
    const int shift = 8 * (pass % 4);
    const int64_t base = (int64_t)blockIdx.x * RN_TILE;
    // One LDS atomic per DISTINCT digit of a wave's 64 keys (the lanes that share a digit are found with eight ballots, the lowest of them adds
    // their count): heavy-tailed keys -- pooled embedding ids, where one id owns a quarter of the entries -- put most lanes of every wave on ONE
    // bin, and 64 atomics on one LDS address serialise (round 5: 38 -> ~20 us per pass at 6.5 M ids; uniform digits cost the ballots, ~1 us).
    const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1ull;
    uint32_t kv[RN_TILE / 256];
#pragma unroll
    for (int r = 0; r < RN_TILE / 256; ++r) {           // all loads of the tile in flight before the first ballot
        const int64_t e = base + r * 256 + threadIdx.x;
        kv[r] = e < B ? (carried ? ksrc[e] : wk[src[e]]) : 0u;
    }
#pragma unroll
    for (int r = 0; r < RN_TILE / 256; ++r) {
        const int64_t e = base + r * 256 + threadIdx.x;
        const bool ok = e < B;
        const unsigned d = (kv[r] >> shift) & 255u;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        if (ok && (peers & lt) == 0ull) atomicAdd(&h[d], (unsigned)__popcll(peers));
    }
    __syncthreads();
    blockhist[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];   // digit-major
}

// stable scatter: wave w of the block owns RN_TILE/4 consecutive keys, 8 rounds of 64 consecutive keys.
__global__ void __launch_bounds__(256)
k_sort_scatter(const uint32_t* __restrict__ words, int64_t B, int n_words, int pass, const SortPlan* __restrict__ plan,
               int32_t* __restrict__ idx0, int32_t* __restrict__ idx1, uint32_t* __restrict__ key0, uint32_t* __restrict__ key1,
               const unsigned* __restrict__ blockoff, int nblk, int scan_here) {
    if (plan->trivial[pass]) return;
    __shared__ unsigned wcnt[4][256];
    __shared__ unsigned boff[256];          // scan_here: this block's first output position of every digit
    __shared__ unsigned wtot[4];
    for (int t = threadIdx.x; t < 1024; t += 256) (&wcnt[0][0])[t] = 0;
    if (scan_here) {
        // `blockoff` holds the RAW digit-major counts [256][nblk]: thread d adds up digit d's row (the part before this block
        // separately), then the 256 row totals are scanned -- one launch less per pass than a separate scan kernel, which
        // matters because a grouping call is a chain of ~30 few-microsecond launches
        const unsigned* row = blockoff + (int64_t)threadIdx.x * nblk;
        unsigned before = 0, total = 0;
        for (int b = 0; b < nblk; ++b) {
            const unsigned v = row[b];
            before += b < (int)blockIdx.x ? v : 0u;
            total += v;
        }
        const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
        unsigned inc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o, 64);
            if (ln >= o) inc += t;
        }
        if (ln == 63) wtot[wv] = inc;
        __syncthreads();
        unsigned woff = 0;
        for (int i = 0; i < wv; ++i) woff += wtot[i];
        boff[threadIdx.x] = woff + inc - total + before;
    }
    __syncthreads();
    const int sb = plan->src[pass];
    const int32_t* src = sb ? idx1 : idx0;
    int32_t* dst = sb ? idx0 : idx1;
    const uint32_t* ksrc = sb ? key1 : key0;
    uint32_t* kdst = sb ? key0 : key1;
    const bool carried = plan->carried[pass];
    const uint32_t* wk = words + (int64_t)(n_words - 1 - pass / 4) * B;
    const int shift = 8 * (pass % 4);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wbase = (int64_t)blockIdx.x * RN_TILE + w * (RN_TILE / 4);
    const unsigned long long lt = (1ull << lane) - 1ull;
    int32_t my_idx[RN_TILE / 256];
    uint32_t my_key[RN_TILE / 256];
    unsigned my_dr[RN_TILE / 256];        // digit | (rank_in_wave << 8)
#pragma unroll
    for (int r = 0; r < RN_TILE / 256; ++r) {
        const int64_t e = wbase + r * 64 + lane;
        const bool ok = e < B;
        int32_t id = ok ? src[e] : 0;
        const uint32_t kv = ok ? (carried ? ksrc[e] : wk[id]) : 0u;
        unsigned d = (kv >> shift) & 255u;
        unsigned long long peers = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        unsigned prior = ok ? wcnt[w][d] : 0u;                 // every lane reads before any leader adds
        const unsigned rank = prior + (unsigned)__popcll(peers & lt);
        if (ok && (peers & lt) == 0ull) wcnt[w][d] = prior + (unsigned)__popcll(peers);   // leader = lowest peer
        my_idx[r] = id;
        my_key[r] = kv;
        my_dr[r] = d | (rank << 8);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RN_TILE / 256; ++r) {
        const int64_t e = wbase + r * 64 + lane;
        if (e < B) {
            const unsigned d = my_dr[r] & 255u, rank = my_dr[r] >> 8;
            unsigned off = (scan_here ? boff[d] : blockoff[(int64_t)d * nblk + blockIdx.x]) + rank;
            for (int pw = 0; pw < w; ++pw) off += wcnt[pw][d];
            dst[off] = my_idx[r];
            kdst[off] = my_key[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// segments
// ------------------------------------------------------------------------------------------------
__global__ void k_seg_heads(const uint32_t* __restrict__ words, const uint8_t* __restrict__ solo, int64_t B, int n_words,
                            int n_words_first, const SortPlan* __restrict__ plan, const int32_t* __restrict__ idx0,
                            const int32_t* __restrict__ idx1, const uint32_t* __restrict__ key0, const uint32_t* __restrict__ key1,
                            int32_t* __restrict__ order, int32_t* __restrict__ head, int32_t* __restrict__ shead) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int32_t* fin = plan->final_buf ? idx1 : idx0;
    const uint32_t* kfin = plan->final_buf ? key1 : key0;       // values of word plan->final_word in sorted order
    const int32_t i = fin[k];
    order[k] = i;
    int h = 1, sh = 1;
    if (k > 0) {
        const int32_t j = fin[k - 1];
        const bool s = solo[i] | solo[j];
        bool diff_first = false, diff_any = false;
        for (int w = 0; w < n_words; ++w) {
            if (plan->word_const[w]) continue;                   // one value over the batch: neighbours cannot differ in it
            const bool d = w == plan->final_word ? kfin[k] != kfin[k - 1]
                                                 : words[(int64_t)w * B + i] != words[(int64_t)w * B + j];
            diff_any |= d;
            if (w < n_words_first) diff_first |= d;
        }
        h = (s || diff_any) ? 1 : 0;
        sh = (s || diff_first) ? 1 : 0;
    }
    head[k] = h;
    shead[k] = sh;
}

__global__ void k_seg_finish(const int32_t* __restrict__ head, const int32_t* __restrict__ seg_incl,
                             const int32_t* __restrict__ super_incl, int64_t B, int32_t* __restrict__ seg_id,
                             int32_t* __restrict__ seg_first, int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int32_t g = seg_incl[k] - 1;
    seg_id[k] = g;
    super_id[k] = super_incl[k] - 1;
    if (head[k]) seg_first[g] = (int32_t)k;
    if (k == B - 1) {
        seg_first[g + 1] = (int32_t)B;
        n_seg[0] = g + 1;
        n_seg[1] = super_incl[k];
    }
}

__global__ void k_seg_empty(int32_t* seg_first, int32_t* n_seg) {
    seg_first[0] = 0;
    n_seg[0] = 0;
    n_seg[1] = 0;
}


// ------------------------------------------------------------------------------------------------
// Small batches (B <= 8192, one 32-bit key word): the whole grouping -- stable LSD radix sort, segment heads, segment ids --
// in ONE workgroup on LDS-resident keys.  The general path above is ~24 launches of a few microseconds each, which is all
// of the time at BASELINE config 2 (B = 8192: 0.24 ms per loss, launch-latency bound).
//   1024 threads, thread t owns the 8 consecutive positions [8t, 8t+8) of the current order (so ranks are stable);
//   4-bit digits: a thread's 16 digit counters (each <= 8) pack into one 64-bit register; passes whose digit is constant
//   over the batch are skipped (OR/AND of all keys); counters [16][1024] u16 are scanned block-wide in digit-major order.
// ------------------------------------------------------------------------------------------------
// RAW = 0: canonical key words + solo flags from recnow_group_keys (the C ABI's two-call form).  RAW = 1 / 2 (round 5, the step's GROUP phase at
// shard sizes): `words` IS the caller's float32 / int32 id tensor -- the canonical key (-0.0 -> +0.0) and the solo flag (NaN, +-inf) of
// k_keys_f32 are formed here and the flags live in an LDS bit set, so that the phase is ONE launch instead of a fill, a key kernel and this one
// (kernel trace at 8192 rows: 5.2 + 5.0 + 31.8 us in front of the forward launch).
// diagnostic build (tools/build_variant.py gstrace -DRN_GS_TRACE, tools/gs_trace.py): wall-clock stamps (100 MHz) of thread 0
#ifdef RN_GS_TRACE
__device__ long long g_gs_trace[16];
#define GS_STAMP(i) do { if (threadIdx.x == 0) g_gs_trace[(i)] = wall_clock64(); } while (0)
__device__ long long g_gm_trace[64];
__device__ int g_gm_n;
#define GM_STAMP() do { if (g == GM_TRACE_WG && threadIdx.x == 0) { g_gm_trace[gm_n < 63 ? gm_n : 63] = wall_clock64(); ++gm_n; g_gm_n = gm_n; } } while (0)
#ifndef GM_TRACE_WG
#define GM_TRACE_WG 0
#endif
extern "C" int recnow_debug_gm_trace(long long* out) { int n = 0; hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_gm_n), sizeof(int)); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gm_trace), sizeof(long long) * 64); return n; }
extern "C" int recnow_debug_gs_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gs_trace), sizeof(long long) * 16); }
#else
#define GS_STAMP(i) do { } while (0)
#define GM_STAMP() do { } while (0)
#endif
template <int RAW>
__device__ __forceinline__ void
group_small_body(const uint32_t* __restrict__ words, const uint8_t* __restrict__ solo, int B, int32_t* __restrict__ order,
                 int32_t* __restrict__ seg_id, int32_t* __restrict__ seg_first, int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_lds[];
    uint32_t* key0 = reinterpret_cast<uint32_t*>(gs_lds);
    uint32_t* key1 = key0 + GS_MAXB;
    uint16_t* idx0 = reinterpret_cast<uint16_t*>(key1 + GS_MAXB);
    uint16_t* idx1 = idx0 + GS_MAXB;
    uint16_t* cnt = idx1 + GS_MAXB;                       // [16][GS_T]
    unsigned* wsum = reinterpret_cast<unsigned*>(cnt + 16 * GS_T);   // [16] + 2
    __shared__ unsigned s_solo[GS_MAXB / 32];             // RAW: bit i = row i pairs with nobody
    const int tid = threadIdx.x;
    GS_STAMP(0);
    if (RAW == 1) {
        for (int i = tid; i < GS_MAXB / 32; i += GS_T) s_solo[i] = 0u;
        __syncthreads();
    }
    auto is_solo = [&](int row) -> bool { return RAW == 0 ? solo[row] != 0 : RAW == 1 ? ((s_solo[row >> 5] >> (row & 31)) & 1u) != 0u : false; };
    unsigned vor = 0, vand = 0xffffffffu, ior = 0, iand = 0xffffffffu, bad = 0;
    for (int i = tid; i < B; i += GS_T) {
        uint32_t k = words[i];
        if (RAW == 1) {
            const float v = __uint_as_float(k);
            if (v == 0.0f) k = 0u;                        // -0.0 == +0.0
            if (!(fabsf(v) < INFINITY)) atomicOr(&s_solo[i >> 5], 1u << (i & 31));
        }
        key0[i] = k;
        idx0[i] = (uint16_t)i;
        vor |= k;
        vand &= k;
        uint32_t im;
        bad |= gs_int_key(k, &im) ? 0u : 1u;
        ior |= im;
        iand &= im;
    }
    GS_STAMP(1);
    const unsigned varying = gs_varying_bits(key0, B, vor, vand, ior, iand, bad, wsum);      // (keys rewritten to their small-integer images when all have one)
    GS_STAMP(2);
    uint32_t* ka = key0; uint32_t* kb = key1;
    uint16_t* ia = idx0; uint16_t* ib = idx1;
    const int lo = tid * GS_KPT, hi = min(B, lo + GS_KPT);
    gs_radix_sort_lds(ka, kb, ia, ib, cnt, wsum, B, varying);
    GS_STAMP(3);
    // segment heads over the sorted order; solo rows (NaN / inf ids) are segments of their own
    unsigned heads = 0, nh = 0;
    for (int i = lo; i < hi; ++i) {
        bool h = true;
        if (i > 0) h = (ka[i] != ka[i - 1]) || is_solo(ia[i]) || is_solo(ia[i - 1]);
        heads |= (h ? 1u : 0u) << (i - lo);
        nh += h ? 1u : 0u;
    }
    unsigned total = 0;
    unsigned g = gs_block_exclusive_scan(nh, wsum, &total);
    GS_STAMP(4);
    for (int i = lo; i < hi; ++i) {
        const bool h = (heads >> (i - lo)) & 1u;
        if (h) { seg_first[g] = i; ++g; }
        order[i] = (int32_t)ia[i];
        seg_id[i] = (int32_t)g - 1;
        super_id[i] = (int32_t)g - 1;
    }
    if (tid == 0) {
        seg_first[total] = B;
        n_seg[0] = (int32_t)total;
        n_seg[1] = (int32_t)total;
    }
    GS_STAMP(5);
#ifdef RN_GS_TRACE
    if (threadIdx.x == 0) g_gs_trace[6] = (long long)varying;
#endif
}
template <int RAW>
__global__ void __launch_bounds__(GS_T)
k_group_small(const uint32_t* __restrict__ words, const uint8_t* __restrict__ solo, int B, int32_t* __restrict__ order,
              int32_t* __restrict__ seg_id, int32_t* __restrict__ seg_first, int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg) {
    group_small_body<RAW>(words, solo, B, order, seg_id, seg_first, super_id, n_seg);
}
// Front kernel of recnow_dcn_mix_step at shard sizes (round 5): workgroup 0 groups the batch, the others write the weight packs of the row-block
// kernels (dcnmix_tile.hpp) -- two launches that depend on nothing but the step's inputs, one of which keeps ONE compute unit busy for ~20 us.
template <int RAW>
__global__ void __launch_bounds__(GS_T)
k_front_small(const uint32_t* __restrict__ words, int B, int32_t* __restrict__ order, int32_t* __restrict__ seg_id, int32_t* __restrict__ seg_first,
              int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg, const RnTileFwd pack, unsigned long long* __restrict__ zero1) {
    if (blockIdx.x == 0) {
        if (zero1 && threadIdx.x == 0) *zero1 = 0ull;      // the step's pair counter (its loss stage adds into it: dcnmix.hip)
        group_small_body<RAW>(words, nullptr, B, order, seg_id, seg_first, super_id, n_seg);
        return;
    }
    tl_pack_range(pack, (int64_t)(blockIdx.x - 1) * GS_T + threadIdx.x, (int64_t)(gridDim.x - 1) * GS_T);
}


// ------------------------------------------------------------------------------------------------
// Mid-size batches (8192 < B <= 524288, or several key words): the whole grouping in ONE cooperative launch.
// The general path above is a chain of ~35 launches (per digit: histogram, scan, scatter; then heads, two device-wide scans,
// finish) of a few microseconds of work each -- at BASELINE config 3 (B = 65536) 0.38 ms of kernel time stretched over the side
// stream.  Here G = ceil(B / 2048) <= 256 workgroups (all co-resident: at most one per CU) walk the same phases separated by a
// grid barrier: identity order + varying-bit words | per digit: local histogram | offsets + stable scatter | segment heads + local
// counts | ids.  2 + 2 * (non-trivial digits) barriers of ~5 us each (MI355X_MICROARCH.md, barrier-counter row).
// Barrier = the guide's R1 hand-off: every wave drains its stores, workgroup barrier, lane 0: agent-scope release, arrive on one
// monotonic counter, relaxed agent-scope poll, agent-scope acquire, workgroup barrier, plain loads.  The poll is bounded (2 s):
// on a timeout the error word is set, every later barrier returns at once, the kernel leaves the identity grouping (every row
// its own group: consumers stay in bounds) and n_seg = -1 -- the grid always drains.  The host only takes this route when
// `gm_max_coresident()` says all workgroups fit on the device at once.
// ------------------------------------------------------------------------------------------------
#define GM_MAXG 256
// batches from this size on run 4096 keys per workgroup: half the workgroups at every grid barrier and in every histogram row (phase stamps, tools/gs_trace.py, one box:
// 262 144 rows 79.0 -> 74.6 us, 65 536 rows 44.5 -> 52.8 us -- the phases of a workgroup take longer than what the cheaper barriers save there)
#define GM_TILE4096_FROM 262144ll
struct GroupMidCtl {          // zeroed by the host before the launch
    unsigned bar;
    int err;
    unsigned mix[2 * RN_MAX_WORDS];
    unsigned imix[2];         // one key word: OR of the keys' small-integer images and OR of their complements (see gs_int_key)
    unsigned notint;          // != 0: some key has no small-integer image (the keys are sorted as they are)
    unsigned anysolo;         // != 0: some row carries a solo flag (NaN / +-inf id): the heads phase gathers the flags only then
    unsigned pad[10];
    unsigned grp[8 * 16];     // hierarchical barrier: arrivals of the workgroups with index % 8 == q (one 64-byte line each)
    unsigned gen[8 * 16];     // ... and the epoch those workgroups poll
};

// Returns false (to every thread of the workgroup) once the error word is set: the caller then leaves through GM_BAIL -- what the other
// workgroups have published so far may be stale, and offsets derived from it could scatter out of bounds.
// Round 5 (-DRN_GM_HIER_BARRIER, A/B): the same hand-off with the arrivals spread over eight counters -- workgroup g arrives on counter g % 8 (round-robin
// dispatch: its XCD's, which is a matter of speed only), the last arrival of a counter arrives on the top counter, the last of those publishes the epoch on
// eight words, one per group, which the group's workgroups poll: 16 instead of 128 workgroups hammer one line.  e = number of this barrier (1-based), G
// workgroups take part, this one is g.  Measured (phase stamps, one box, two alternating repetitions): 262 144 rows (64 workgroups) 63.2 -> 62.2 us, 65 536 rows (32)
// 41.8 -> 45.3 us, 16 384 rows (8) 22.5 -> 24.8 us -- the second hop costs more than the contention it removes at these grid sizes; the flat counter stays.
__device__ __forceinline__ bool gm_barrier_hier(GroupMidCtl* ctl, unsigned e, int G, int g) {
    __shared__ int s_err2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int q = g & 7, ngrp = G < 8 ? G : 8, in_grp = (G - q + 7) >> 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned a = __hip_atomic_fetch_add(&ctl->grp[q * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1u == e * (unsigned)in_grp) {
            const unsigned t = __hip_atomic_fetch_add(&ctl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1u == e * (unsigned)ngrp)
                for (int i = 0; i < ngrp; ++i) __hip_atomic_store(&ctl->gen[i * 16], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&ctl->gen[q * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
            if (__hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (wall_clock64() - t0 > 200000000LL) {          // 2 s of the 100 MHz clock
                __hip_atomic_store(&ctl->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_err2 = __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_err2 == 0;
}
__device__ __forceinline__ bool gm_barrier(GroupMidCtl* ctl, unsigned target) {
    __shared__ int s_err;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&ctl->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&ctl->bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            if (wall_clock64() - t0 > 200000000LL) {          // 2 s of the 100 MHz clock
                __hip_atomic_store(&ctl->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_err = __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_err == 0;
}
// A barrier timed out (the workgroups were not co-resident: the host gates this route on the occupancy query, so this is a last line of
// defence; RECNOW_DEBUG_GROUP_TIMEOUT=1 forces it for the tests): this workgroup leaves the identity grouping in ITS ranges -- every row its own
// group -- so that whatever the other workgroups have written or still write, every index in the arrays stays in [0, B] (in-bounds walks for
// every consumer; NOT necessarily a consistent grouping when only some workgroups bail), and n_seg = -1, which `Segments.num_segments()` /
// `recnow_group_segments_status` report and which makes the one-call losses return NaN (k_pair_norm_grad, k_step_dscore, k_lw_norm).  Workgroup 0
// publishes the segment count last and re-reads the error word before it does (end of the kernel).
#ifdef RN_GM_HIER_BARRIER
#define GM_BAR() gm_barrier_hier(ctl, ++nbar, G, g)
#else
#define GM_BAR() gm_barrier(ctl, (++nbar) * G)
#endif
#define GM_BAIL()                                                                                                    \
    do {                                                                                                             \
        for (int q_ = 0; q_ < TILE / 256; ++q_) {                                                                 \
            const int64_t k_ = (int64_t)g * TILE + q_ * 256 + threadIdx.x;                                      \
            if (k_ < B) { order[k_] = (int32_t)k_; seg_id[k_] = (int32_t)k_; seg_first[k_] = (int32_t)k_; super_id[k_] = (int32_t)k_; } \
        }                                                                                                            \
        if (g == 0 && threadIdx.x == 0) { seg_first[B] = (int32_t)B; n_seg[0] = -1; n_seg[1] = -1; }              \
        return;                                                                                                      \
    } while (0)

// TILE keys per workgroup (256 threads x TILE / 256): 2048 is what the host launches; 512 / 1024 are A/B instantiations (measured slower, see the host).
// RAW = 0: canonical key words + solo flags from recnow_group_keys.  RAW = 1 / 2 (round 5, one float32 / int32 id tensor, n_words = 1: the GROUP phase
// of recnow_dcn_mix_step above 8192 rows): `words` IS the id tensor -- every read of it forms the canonical key (-0.0 -> +0.0) on the way, and phase 0
// writes the solo flags (NaN, +-inf) itself, one grid barrier before anybody gathers them: the fill of `solo` and the key kernel in front of this launch go.
template <int RAW>
__device__ __forceinline__ uint32_t gm_canon(uint32_t k) {
    if (RAW == 1) return __uint_as_float(k) == 0.0f ? 0u : k;
    return k;
}
// The scattered (index, key) pairs of a digit pass are read by OTHER workgroups -- on other XCDs, i.e. through other L2s -- right behind the next grid barrier.
// Tried (round 5, -DRN_GM_WT_SCATTER): written through (sc1), so that the barrier's release fence finds nothing dirty (MI355X_MICROARCH.md, publish-large).  For
// RANDOM 4-byte stores that loses: every store becomes a partial write to memory -- scatter phase 5.7 -> 11.9 us, the barrier behind it 7.2 -> 10.8 us, the
// launch 62 -> 72-74 us at 262 144 rows (phase stamps, one box, twice).  Plain stores (merged in L2, written back by the fence) are the default.
template <typename T>
__device__ __forceinline__ void gm_store_wt(T* p, T v) {
    static_assert(sizeof(T) == 4, "one dword");
#ifdef RN_GM_WT_SCATTER
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");      // (covered by gm_barrier's own `s_waitcnt vmcnt(0)`)
#else
    *p = v;
#endif
}
template <int TILE, int RAW>
__device__ __forceinline__ void
group_mid_body(const uint32_t* __restrict__ words, uint8_t* __restrict__ solo, int64_t B, int n_words, int n_words_first,
               GroupMidCtl* __restrict__ ctl, int32_t* __restrict__ idx0, int32_t* __restrict__ idx1, uint32_t* __restrict__ key0,
               uint32_t* __restrict__ key1, unsigned* __restrict__ blockhist, int* __restrict__ headcnt, int32_t* __restrict__ order,
               int32_t* __restrict__ seg_id, int32_t* __restrict__ seg_first, int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg,
               const int G, const int g) {
    __shared__ unsigned h[256];
    __shared__ unsigned wcnt[4][256];
    __shared__ unsigned boff[256];
    __shared__ unsigned wtot[4];
    __shared__ unsigned part[4][2 * RN_MAX_WORDS];
    __shared__ unsigned ipart[4][3];
    __shared__ int s_useint, s_anysolo, s_spec;
    __shared__ int s_triv[RN_MAX_PASS], s_src[RN_MAX_PASS], s_carried[RN_MAX_PASS], s_wconst[RN_MAX_WORDS], s_final[2];
    __shared__ int s_cnt[4][2];
    __shared__ int s_red[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;      // G workgroups take part, this one is g
    const int Gp = (G + 3) & ~3;
    const int64_t base = (int64_t)g * TILE;
    unsigned nbar = 0;
#ifdef RN_GS_TRACE
    int gm_n = 0;
#endif
    GM_STAMP();
    // ---- phase 0: identity order, bits that vary over the batch ----------------------------------------------------------
    // (round 5) ... and, for one key word, the histogram of the FIRST digit pass on the guess that every key has a small-integer image and that its lowest
    // byte varies (float ids 0 .. 2^24: the usual batch): when the plan behind the barrier agrees (s_spec), pass 0 starts at its offsets -- one grid
    // barrier and one histogram phase less (~8 us of 66 at 262 144 rows).  A wrong guess costs the 2048 LDS atomics; the pass then counts as before.
    h[tid] = 0;
    __syncthreads();
    {
        unsigned o[RN_MAX_WORDS], z[RN_MAX_WORDS];
        unsigned io = 0u, iz = 0u, bad = 0u;                 // small-integer images of the keys (one key word only)
        unsigned anys = 0u;                                   // this thread saw a solo row
#pragma unroll
        for (int w = 0; w < RN_MAX_WORDS; ++w) o[w] = z[w] = 0u;
        // loads from clamped indices, masked afterwards: no load sits under a lane condition (each would end in an `s_waitcnt vmcnt(0)`
        // at its merge, i.e. the eight rounds of a thread would be eight round trips in series)
        for (int w = 0; w < n_words; ++w) {                   // block-uniform trip count
            uint32_t kk[TILE / 256];
#pragma unroll
            for (int r = 0; r < TILE / 256; ++r) {
                const int64_t i = base + r * 256 + tid;
                kk[r] = gm_canon<RAW>(words[(int64_t)w * B + (i < B ? i : B - 1)]);
            }
            if (RAW == 1) {                                   // the solo flags of this workgroup's keys (read by others only behind a grid barrier)
#pragma unroll
                for (int r = 0; r < TILE / 256; ++r) {
                    const int64_t i = base + r * 256 + tid;
                    const bool fl = !(fabsf(__uint_as_float(kk[r])) < INFINITY);
                    if (i < B) solo[i] = fl ? 1 : 0;
                    anys |= (i < B && fl) ? 1u : 0u;
                }
            } else if (RAW == 2) {
#pragma unroll
                for (int r = 0; r < TILE / 256; ++r) {
                    const int64_t i = base + r * 256 + tid;
                    if (i < B) solo[i] = 0;
                }
            }
            unsigned oo = 0u, zz = 0u;
#pragma unroll
            for (int r = 0; r < TILE / 256; ++r) {
                const bool ok = base + r * 256 + tid < B;
                oo |= ok ? kk[r] : 0u;
                zz |= ok ? ~kk[r] : 0u;
                uint32_t im;
                const bool isint = gs_int_key(kk[r], &im);
                io |= ok ? im : 0u;
                iz |= ok ? ~im : 0u;
                bad |= (ok && !isint) ? 1u : 0u;
                if (ok && n_words == 1) atomicAdd(&h[im & 255u], 1u);
            }
#pragma unroll
            for (int q = 0; q < RN_MAX_WORDS; ++q)
                if (q == w) { o[q] = oo; z[q] = zz; }
        }
        if (RAW == 0) {
#pragma unroll
            for (int r = 0; r < TILE / 256; ++r) {
                const int64_t i = base + r * 256 + tid;
                anys |= (solo[i < B ? i : B - 1] != 0) ? 1u : 0u;
            }
        }
        if (__ballot(anys != 0u) != 0ull && lane == 0) atomicOr(&ctl->anysolo, 1u);
#pragma unroll
        for (int r = 0; r < TILE / 256; ++r) {
            const int64_t i = base + r * 256 + tid;
            if (i < B) idx0[i] = (int32_t)i;
        }
#pragma unroll
        for (int w = 0; w < RN_MAX_WORDS; ++w)
            if (w < n_words) {
                unsigned a = o[w], c = z[w];
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) {
                    a |= __shfl_xor(a, sft, 64);
                    c |= __shfl_xor(c, sft, 64);
                }
                if (lane == 0) { part[wv][w] = a; part[wv][RN_MAX_WORDS + w] = c; }
            }
        if (n_words == 1) {                                   // block-uniform
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) {
                io |= __shfl_xor(io, sft, 64);
                iz |= __shfl_xor(iz, sft, 64);
                bad |= __shfl_xor(bad, sft, 64);
            }
            if (lane == 0) { ipart[wv][0] = io; ipart[wv][1] = iz; ipart[wv][2] = bad; }
        }
        __syncthreads();
        if (tid < 2 * n_words) {
            const int w = tid % n_words, half = tid / n_words, col = half * RN_MAX_WORDS + w;
            atomicOr(&ctl->mix[half * n_words + w], part[0][col] | part[1][col] | part[2][col] | part[3][col]);
        }
        if (n_words == 1 && tid >= 64 && tid < 67) {
            const int c = tid - 64;
            const unsigned v = ipart[0][c] | ipart[1][c] | ipart[2][c] | ipart[3][c];
            atomicOr(c == 2 ? &ctl->notint : &ctl->imix[c], v);
        }
        if (n_words == 1) blockhist[(int64_t)tid * Gp + g] = h[tid];      // (behind the barrier above: the histogram is complete)
    }
    GM_STAMP();
    if (!GM_BAR()) GM_BAIL();
    GM_STAMP();
    const int np = n_words * 4;
    if (tid == 0) {          // the pass plan, as k_sort_plan builds it (every workgroup derives the same one)
        int cur = 0, word_in_buf = -1;
        const bool useint = n_words == 1 && __hip_atomic_load(&ctl->notint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
        s_useint = useint ? 1 : 0;
        s_anysolo = __hip_atomic_load(&ctl->anysolo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ? 1 : 0;
        for (int p = 0; p < np; ++p) {
            const int w = n_words - 1 - p / 4, d = p % 4;
            const unsigned both = useint ? (__hip_atomic_load(&ctl->imix[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &
                                            __hip_atomic_load(&ctl->imix[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                         : (__hip_atomic_load(&ctl->mix[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &
                                            __hip_atomic_load(&ctl->mix[n_words + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int tv = ((both >> (8 * d)) & 255u) == 0u;
            s_triv[p] = tv;
            s_src[p] = cur;
            s_carried[p] = word_in_buf == w;
            if (!tv) { cur ^= 1; word_in_buf = w; }
        }
        s_final[0] = cur;
        s_final[1] = word_in_buf;
        s_spec = (useint && !s_triv[0]) ? 1 : 0;             // phase 0 guessed pass 0: integer images, lowest byte
        for (int w = 0; w < n_words; ++w) {
            int c = 1;
            for (int d = 0; d < 4; ++d) c &= s_triv[(n_words - 1 - w) * 4 + d];
            s_wconst[w] = c;
        }
    }
    __syncthreads();
    // ---- digits ---------------------------------------------------------------------------------------------------------------
    for (int p = 0; p < np; ++p) {
        if (s_triv[p]) continue;                              // block-uniform (and the same in every workgroup)
        const int sb = s_src[p];
        const int32_t* src = sb ? idx1 : idx0;
        int32_t* dst = sb ? idx0 : idx1;
        const uint32_t* ksrc = sb ? key1 : key0;
        uint32_t* kdst = sb ? key0 : key1;
        const bool carried = s_carried[p];
        const uint32_t* wk = words + (int64_t)(n_words - 1 - p / 4) * B;
        const int shift = 8 * (p % 4);
        // this tile's keys (kept in registers for the scatter) and its digit histogram
        int32_t my_idx[TILE / 256];
        uint32_t my_key[TILE / 256];
        const bool spec = p == 0 && s_spec != 0;                // block-uniform: blockhist already holds this pass's counts (phase 0)
        h[tid] = 0;
        for (int t = tid; t < 1024; t += 256) (&wcnt[0][0])[t] = 0;
        __syncthreads();
        const int64_t wbase = base + wv * (TILE / 4);
        // all eight rounds' loads in flight together (clamped indices; the block-uniform `carried` chooses between two straight-line
        // load sequences instead of sitting inside every round): two dependent round trips per pass, not sixteen
        if (carried) {
#pragma unroll
            for (int r = 0; r < TILE / 256; ++r) {
                const int64_t e = wbase + r * 64 + lane, ec = e < B ? e : B - 1;
                my_idx[r] = src[ec];
                my_key[r] = ksrc[ec];
            }
        } else {
#pragma unroll
            for (int r = 0; r < TILE / 256; ++r) {
                const int64_t e = wbase + r * 64 + lane;
                my_idx[r] = src[e < B ? e : B - 1];
            }
#pragma unroll
            for (int r = 0; r < TILE / 256; ++r) my_key[r] = gm_canon<RAW>(wk[my_idx[r]]);
            if (s_useint) {                                   // block-uniform: the keys' small-integer images are what is sorted (and carried)
#pragma unroll
                for (int r = 0; r < TILE / 256; ++r) my_key[r] = (uint32_t)__uint_as_float(my_key[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < TILE / 256; ++r) {
            const bool ok = wbase + r * 64 + lane < B;
            my_idx[r] = ok ? my_idx[r] : 0;
            my_key[r] = ok ? my_key[r] : 0u;
            if (ok && !spec) atomicAdd(&h[(my_key[r] >> shift) & 255u], 1u);
        }
        if (!spec) {
            __syncthreads();
            blockhist[(int64_t)tid * Gp + g] = h[tid];        // digit-major, rows of Gp = G rounded up to 4 entries (16-byte row reads below)
            GM_STAMP();
            if (!GM_BAR()) GM_BAIL();
            GM_STAMP();
        }
        {   // first output position of every digit for this workgroup: counts of the workgroups before it + the digits below
            // (round 5: the row is read as 16-byte pieces, sixteen in flight -- one entry per iteration was a chain of G dependent waits: 9.6 us of a
            // pass at G = 128, phase stamps of tools/gs_trace.py; the pad entries of a row are never written and are masked here)
            const uint4* row4 = reinterpret_cast<const uint4*>(blockhist + (int64_t)tid * Gp);
            const int n4 = Gp >> 2;
            unsigned before = 0, total = 0;
            for (int c0 = 0; c0 < n4; c0 += 16) {
                uint4 v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = row4[c0 + j < n4 ? c0 + j : n4 - 1];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int b = (c0 + j) * 4;
                    const unsigned e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        total += (c0 + j < n4 && b + q < G) ? e[q] : 0u;
                        before += (c0 + j < n4 && b + q < g) ? e[q] : 0u;
                    }
                }
            }
            unsigned inc = total;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            if (lane == 63) wtot[wv] = inc;
            __syncthreads();
            unsigned woff = 0;
            for (int i = 0; i < wv; ++i) woff += wtot[i];
            boff[tid] = woff + inc - total + before;
        }
        __syncthreads();
        GM_STAMP();
        // stable ranks: wave w owns TILE / 4 consecutive keys, 8 rounds of 64 consecutive keys (as k_sort_scatter)
        const unsigned long long lt = (1ull << lane) - 1ull;
        unsigned my_dr[TILE / 256];
#pragma unroll
        for (int r = 0; r < TILE / 256; ++r) {
            const int64_t e = wbase + r * 64 + lane;
            const bool ok = e < B;
            const unsigned d = (my_key[r] >> shift) & 255u;
            unsigned long long peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const unsigned prior = ok ? wcnt[wv][d] : 0u;
            const unsigned rank = prior + (unsigned)__popcll(peers & lt);
            if (ok && (peers & lt) == 0ull) wcnt[wv][d] = prior + (unsigned)__popcll(peers);
            my_dr[r] = d | (rank << 8);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < TILE / 256; ++r) {
            const int64_t e = wbase + r * 64 + lane;
            if (e < B) {
                const unsigned d = my_dr[r] & 255u, rank = my_dr[r] >> 8;
                unsigned off = boff[d] + rank;
                for (int pw = 0; pw < wv; ++pw) off += wcnt[pw][d];
                gm_store_wt(dst + off, my_idx[r]);
                gm_store_wt(kdst + off, my_key[r]);
            }
        }
        GM_STAMP();
        if (!GM_BAR()) GM_BAIL();
        GM_STAMP();
    }
    // ---- segments: heads of this tile's sorted positions ------------------------------------------------------------------------------------
    // Round 5: thread (round r, tid) owns position base + r * 256 + tid, so every array access of this phase is coalesced (with TILE / 256 CONSECUTIVE
    // positions per thread a wave instruction read 64 pieces 32-64 bytes apart: 9.4 us of 74 at 262 144 rows, phase stamps of tools/gs_trace.py).  A
    // position's predecessor is the lane below; across waves and rounds the edge values go through LDS, the block's own predecessor is a uniform load.
    // The solo flags are gathered only when the batch has any (ctl->anysolo, set in phase 0).
    const int32_t* fin = s_final[0] ? idx1 : idx0;
    const uint32_t* kfin = s_final[0] ? key1 : key0;
    const int final_word = s_final[1];
    constexpr int R = TILE / 256;
    __shared__ uint32_t s_edge[R][4];                         // the value of lane 63 of (round, wave)
    __shared__ int s_hc[R][4], s_sc[R][4];                    // heads of (round, wave); then their exclusive prefix in (round, wave) order
    const unsigned long long ltm = (1ull << lane) - 1ull;
    const int64_t kprev = base > 0 ? base - 1 : 0;            // the block's predecessor (position 0 has none: it is a head by definition)
    int32_t f[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t k = base + r * 256 + tid;
        f[r] = fin[k < B ? k : B - 1];
    }
    const int32_t fprev = fin[kprev];
    // pv[r] = the value of position k - 1 for this thread's position of round r (vb: the value in front of the block)
    auto prev_of = [&](const uint32_t (&v)[R], uint32_t vb, uint32_t (&pv)[R]) {
        if (lane == 63) {
#pragma unroll
            for (int r = 0; r < R; ++r) s_edge[r][wv] = v[r];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t pr = __shfl_up(v[r], 1, 64);
            if (lane == 0) pr = wv > 0 ? s_edge[r][wv - 1] : (r > 0 ? s_edge[r > 0 ? r - 1 : 0][3] : vb);
            pv[r] = pr;
        }
        __syncthreads();
    };
    unsigned dany = 0u, dfirst = 0u, sob = 0u;                // bit r: position k and k - 1 differ in some / in a groups[0] word; one of them is solo
    if (s_anysolo) {                                          // block-uniform (and the same in every workgroup)
        uint32_t sv[R], sp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sv[r] = solo[f[r]];
        const uint32_t vb = solo[fprev];
        prev_of(sv, vb, sp);
#pragma unroll
        for (int r = 0; r < R; ++r) sob |= ((sv[r] | sp[r]) != 0u ? 1u : 0u) << r;
    }
    for (int w = 0; w < n_words; ++w) {                       // block-uniform
        if (s_wconst[w]) continue;
        uint32_t v[R], pv[R], vb;
        if (w == final_word) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t k = base + r * 256 + tid;
                v[r] = kfin[k < B ? k : B - 1];
            }
            vb = kfin[kprev];
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = gm_canon<RAW>(words[(int64_t)w * B + f[r]]);
            vb = gm_canon<RAW>(words[(int64_t)w * B + fprev]);
        }
        prev_of(v, vb, pv);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned df = v[r] != pv[r] ? 1u : 0u;
            dany |= df << r;
            if (w < n_words_first) dfirst |= df << r;
        }
    }
    unsigned hb = 0, sbm = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t k = base + r * 256 + tid;
        const bool valid = k < B;
        const bool so = (sob >> r) & 1u;
        const bool hd = valid && (k == 0 || so || ((dany >> r) & 1u));
        const bool sh = valid && (k == 0 || so || ((dfirst >> r) & 1u));
        hb |= (hd ? 1u : 0u) << r;
        sbm |= (sh ? 1u : 0u) << r;
        const unsigned long long mh = __ballot(hd), ms = __ballot(sh);
        if (lane == 0) { s_hc[r][wv] = (int)__popcll(mh); s_sc[r][wv] = (int)__popcll(ms); }
        if (valid) order[k] = f[r];
    }
    __syncthreads();
    if (wv == 0) {                                            // exclusive prefix of the 4 R <= 64 counts in (round, wave) order
        int* hc = &s_hc[0][0];
        int* sc = &s_sc[0][0];
        const int a = lane < 4 * R ? hc[lane] : 0, c = lane < 4 * R ? sc[lane] : 0;
        int ia = a, ic = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t1 = __shfl_up(ia, o, 64), t2 = __shfl_up(ic, o, 64);
            if (lane >= o) { ia += t1; ic += t2; }
        }
        if (lane < 4 * R) { hc[lane] = ia - a; sc[lane] = ic - c; }
        if (lane == 63) { s_cnt[0][0] = ia; s_cnt[0][1] = ic; }
    }
    __syncthreads();
    if (tid == 0) { headcnt[2 * g] = s_cnt[0][0]; headcnt[2 * g + 1] = s_cnt[0][1]; }
    GM_STAMP();
    if (!GM_BAR()) GM_BAIL();
    GM_STAMP();
    int preh = 0, pres = 0, allh = 0, alls = 0;              // heads in the workgroups before this one / in all of them
    {   // thread b reads workgroup b's pair (G <= 256 = the block), block reduction (round 5: every thread walked all G pairs: 9 us at G = 128)
        const int bc = tid < G ? tid : G - 1;
        const int a0 = headcnt[2 * bc], c0 = headcnt[2 * bc + 1];
        int r4[4] = {tid < g ? a0 : 0, tid < g ? c0 : 0, tid < G ? a0 : 0, tid < G ? c0 : 0};
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r4[q] += __shfl_xor(r4[q], o, 64);
        }
        if (lane == 0) { s_red[wv][0] = r4[0]; s_red[wv][1] = r4[1]; s_red[wv][2] = r4[2]; s_red[wv][3] = r4[3]; }
        __syncthreads();
        for (int i = 0; i < 4; ++i) { preh += s_red[i][0]; pres += s_red[i][1]; allh += s_red[i][2]; alls += s_red[i][3]; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t k = base + r * 256 + tid;
        const bool hd = (hb >> r) & 1u, sh = (sbm >> r) & 1u;
        const unsigned long long mh = __ballot(hd), ms = __ballot(sh);
        const int eh = preh + s_hc[r][wv] + (int)__popcll(mh & ltm);      // heads in front of position k
        const int es = pres + s_sc[r][wv] + (int)__popcll(ms & ltm);
        if (k < B) {
            if (hd) seg_first[eh] = (int32_t)k;
            seg_id[k] = hd ? eh : eh - 1;
            super_id[k] = sh ? es : es - 1;
        }
    }
    if (g == 0 && tid == 0) {
        // a workgroup can time out at the LAST barrier and set ctl->err just as the last arrival completes the count; its peers
        // may have read err before that and passed.  The bailing workgroup then rewrites ITS ranges with the identity grouping while the others
        // write real segments -- every value stays in [0, B] (walks stay in bounds), but the arrays are not a grouping.  So the error word is
        // read once more here, after this workgroup's own writes, and the count is published as -1 (every consumer poisons its result on
        // n_seg < 0) whenever any workgroup has reported a time-out by now.
        const int late = __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seg_first[allh] = (int32_t)B;
        n_seg[0] = late ? -1 : allh;
        n_seg[1] = late ? -1 : alls;
    }
    GM_STAMP();
}
template <int TILE, int RAW = 0>
__global__ void __launch_bounds__(256)
k_group_mid(const uint32_t* __restrict__ words, uint8_t* __restrict__ solo, int64_t B, int n_words, int n_words_first,
            GroupMidCtl* __restrict__ ctl, int32_t* __restrict__ idx0, int32_t* __restrict__ idx1, uint32_t* __restrict__ key0,
            uint32_t* __restrict__ key1, unsigned* __restrict__ blockhist, int* __restrict__ headcnt, int32_t* __restrict__ order,
            int32_t* __restrict__ seg_id, int32_t* __restrict__ seg_first, int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg) {
    group_mid_body<TILE, RAW>(words, solo, B, n_words, n_words_first, ctl, idx0, idx1, key0, key1, blockhist, headcnt, order, seg_id, seg_first, super_id,
                              n_seg, (int)gridDim.x, (int)blockIdx.x);
}
// Front kernel of recnow_dcn_mix_step above GS_MAXB rows: the first G workgroups (dispatched first: co-resident as before) group the batch, the others
// write the weight packs of the row-block kernels (see k_front_small).
template <int RAW>
__global__ void __launch_bounds__(256)
k_front_mid(const uint32_t* __restrict__ words, uint8_t* __restrict__ solo, int64_t B, GroupMidCtl* __restrict__ ctl, int32_t* __restrict__ idx0,
            int32_t* __restrict__ idx1, uint32_t* __restrict__ key0, uint32_t* __restrict__ key1, unsigned* __restrict__ blockhist,
            int* __restrict__ headcnt, int32_t* __restrict__ order, int32_t* __restrict__ seg_id, int32_t* __restrict__ seg_first,
            int32_t* __restrict__ super_id, int32_t* __restrict__ n_seg, const int G, const RnTileFwd pack, unsigned long long* __restrict__ zero1) {
    if (zero1 && blockIdx.x == 0 && threadIdx.x == 0) *zero1 = 0ull;
    if ((int)blockIdx.x < G) {
        group_mid_body<RN_TILE, RAW>(words, solo, B, 1, 1, ctl, idx0, idx1, key0, key1, blockhist, headcnt, order, seg_id, seg_first, super_id, n_seg, G,
                                     (int)blockIdx.x);
        return;
    }
    tl_pack_range(pack, (int64_t)((int)blockIdx.x - G) * 256 + threadIdx.x, (int64_t)((int)gridDim.x - G) * 256);
}

// The grouping of a small batch (B <= 8192) straight from ONE float32 / int32 id tensor in one launch (library-internal: the GROUP phase of
// recnow_dcn_mix_step).  Returns RECNOW_EUNSUPPORTED for other shapes: the caller then takes recnow_group_keys + recnow_group_segments.
int rn_group_small_raw(const void* group, int dtype, int64_t B, int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id,
                       int32_t* n_seg, hipStream_t st, const RnTileFwd* pack, unsigned long long* zero1, int* zeroed, int* packed) {
    if (packed) *packed = 0;            // set only where the launch really wrote the weight packs (a grouping launch without them must not mark them current)
    if (B < 1 || B > GS_MAXB || (dtype != RECNOW_KEY_F32 && dtype != RECNOW_KEY_I32)) return RECNOW_EUNSUPPORTED;
    static std::atomic<bool> raised[64];
    int dev = 0;
    RN_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !raised[dev].load(std::memory_order_acquire)) {
        RN_HIP(hipFuncSetAttribute((const void*)k_group_small<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes()));
        RN_HIP(hipFuncSetAttribute((const void*)k_group_small<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes()));
        RN_HIP(hipFuncSetAttribute((const void*)k_front_small<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes()));
        RN_HIP(hipFuncSetAttribute((const void*)k_front_small<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes()));
        if (dev >= 0 && dev < 64) raised[dev].store(true, std::memory_order_release);
    }
    if (pack) {      // + the weight packs: one pack workgroup per remaining compute unit (every workgroup of this kernel holds 112 KB of LDS)
        const int grid = 1 + RN_FRONT_PACK_WGS;
        if (dtype == RECNOW_KEY_F32)
            hipLaunchKernelGGL(k_front_small<1>, grid, GS_T, gs_lds_bytes(), st, (const uint32_t*)group, (int)B, order, seg_id, seg_first, super_id, n_seg, *pack, zero1);
        else
            hipLaunchKernelGGL(k_front_small<2>, grid, GS_T, gs_lds_bytes(), st, (const uint32_t*)group, (int)B, order, seg_id, seg_first, super_id, n_seg, *pack, zero1);
        RN_LAUNCH_CHECK();
        if (zeroed) *zeroed = zero1 ? 1 : 0;
        if (packed) *packed = 1;
        return RECNOW_OK;
    }
    if (dtype == RECNOW_KEY_F32)
        hipLaunchKernelGGL(k_group_small<1>, 1, GS_T, gs_lds_bytes(), st, (const uint32_t*)group, (const uint8_t*)nullptr, (int)B, order, seg_id, seg_first, super_id, n_seg);
    else
        hipLaunchKernelGGL(k_group_small<2>, 1, GS_T, gs_lds_bytes(), st, (const uint32_t*)group, (const uint8_t*)nullptr, (int)B, order, seg_id, seg_first, super_id, n_seg);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// Largest grid of k_group_mid whose workgroups are all resident at once on the CURRENT device: occupancy x compute units, queried
// once per device (a CU mask, a smaller partition (CPX) or a register-hungrier build all show up here).  The hand-rolled grid
// barrier is only correct below it; anything larger takes the multi-launch chain.
static int gm_max_coresident() {
    static int cached[64];
    static bool have[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!have[dev]) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_group_mid<2048>, 256, 0) != hipSuccess) per_cu = 0;
        // ... and of the front kernels, which carry the same barrier among their first G workgroups (their pack workgroups never wait: they end
        // without looking at the barrier, so residency of the G grouping workgroups is all the barrier needs)
        int f1 = 0, f2 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&f1, (const void*)k_front_mid<1>, 256, 0) != hipSuccess) f1 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&f2, (const void*)k_front_mid<2>, 256, 0) != hipSuccess) f2 = 0;
        if (f1 < 1 || f2 < 1) per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        // one workgroup per CU is what the route counts on (the step's GEMMs hold the other slots)
        cached[dev] = per_cu > 0 ? cus : 0;
        have[dev] = true;
    }
    return cached[dev];
}

extern "C" size_t recnow_group_segments_workspace_bytes(int64_t B, int n_words) {
    if (B < 0 || n_words < 1 || n_words > RN_MAX_WORDS) return 0;
    const int nblk = rn_cdiv(B > 0 ? B : 1, RN_TILE);
    size_t s = 0;
    s += rn_align(sizeof(SortPlan));
    s += rn_align((size_t)n_words * 4 * 256 * sizeof(unsigned));    // varying-bit words (2 per key word; sized as before)
    s += 4 * rn_align((size_t)(B + 1) * sizeof(int32_t));           // idx0, idx1, key0, key1
    s += rn_align((size_t)256 * (nblk > GM_MAXG ? nblk : GM_MAXG) * sizeof(unsigned));      // blockhist (the cooperative route runs up to GM_MAXG smaller tiles)
    s += 4 * rn_align((size_t)(B + 1) * sizeof(int32_t));           // head, shead, seg_incl, super_incl
    s += rn_scan_ws_bytes(B);
    s += rn_align(sizeof(GroupMidCtl)) + rn_align((size_t)2 * GM_MAXG * sizeof(int));      // cooperative mid-size path
    return s;
}

// The cooperative route (one launch, all workgroups co-resident: <= one per CU) over the workspace of recnow_group_segments; RECNOW_EUNSUPPORTED when the
// batch does not fit it (the caller then takes the multi-launch chain).  raw = 0: canonical key words + solo flags; raw = 1 / 2: `words` is a float32 / int32
// id tensor (n_words = 1), the kernel forms keys and solo flags itself (`solo` is written, not read, by the caller's side).  zero1 / zeroed: the front
// kernels (pack != NULL) also clear one 64-bit word for the caller and say so; the other forms leave *zeroed alone.
static int rn_group_coop(int raw, const uint32_t* words, uint8_t* solo, int64_t B, int n_words, int n_words_first, int32_t* order, int32_t* seg_id,
                         int32_t* seg_first, int32_t* super_id, int32_t* n_seg, void* ws, size_t ws_bytes, hipStream_t st, const RnTileFwd* pack = nullptr,
                         unsigned long long* zero1 = nullptr, int* zeroed = nullptr, int* packed = nullptr) {
    if (packed) *packed = 0;
    static const bool coop = []() { const char* e = getenv("RECNOW_GROUP_COOP"); return !e || e[0] != '0'; }();      // A/B switch
    if (!coop) return RECNOW_EUNSUPPORTED;
    const int nblk = rn_cdiv(B, RN_TILE);
    RnCarver c(ws, ws_bytes);
    c.take<SortPlan>(1);
    c.take<unsigned>((size_t)n_words * 4 * 256);
    int32_t* idx0 = c.take<int32_t>(B + 1);
    int32_t* idx1 = c.take<int32_t>(B + 1);
    uint32_t* key0 = c.take<uint32_t>(B + 1);
    uint32_t* key1 = c.take<uint32_t>(B + 1);
    unsigned* blockhist = c.take<unsigned>((size_t)256 * (nblk > GM_MAXG ? nblk : GM_MAXG));
    c.take<int32_t>(B + 1);
    c.take<int32_t>(B + 1);
    c.take<int32_t>(B + 1);
    c.take<int32_t>(B + 1);
    // Keys per workgroup: 2048.  Smaller tiles (more workgroups, less
    // per-key work between two grid barriers; RECNOW_GROUP_TILE=512 / 1024 selects them where the grid stays inside the bound) measured SLOWER:
    // B = 65 536 on 128 workgroups of 512 keys 114 us against 89 us on 32 of 2048, B = 262 144 on 256 of 1024 keys 207 against 147 us per fused
    // loss -- every digit pass makes each thread scan one histogram row over ALL workgroups, and a barrier costs more the more arrive.
    const int maxg = gm_max_coresident() < GM_MAXG ? gm_max_coresident() : GM_MAXG;
    static const int tile_env = []() { const char* e = getenv("RECNOW_GROUP_TILE"); return e ? atoi(e) : 0; }();      // A/B switch: 512 / 1024 / 2048
    int tile = RN_TILE;
    if (raw == 0 && (tile_env == 512 || tile_env == 1024) && rn_cdiv(B, tile_env) <= maxg) tile = tile_env;
    if (tile_env == 4096 || (tile_env == 0 && B >= GM_TILE4096_FROM)) tile = 4096;
    if (rn_cdiv(B, tile) > maxg) return RECNOW_EUNSUPPORTED;
    const size_t scan_bytes = rn_scan_ws_bytes(B);
    char* tail = c.base + c.off + scan_bytes;
    GroupMidCtl* ctl = (GroupMidCtl*)tail;
    int* headcnt = (int*)(tail + rn_align(sizeof(GroupMidCtl)));
    RN_HIP(hipMemsetAsync(ctl, 0, sizeof(GroupMidCtl), st));
    static const bool dbg_timeout = []() { const char* e = getenv("RECNOW_DEBUG_GROUP_TIMEOUT"); return e && e[0] == '1'; }();
    if (dbg_timeout) RN_HIP(hipMemsetAsync(&ctl->err, 1, sizeof(int), st));      // tests: every barrier reports the time-out at once
    const int g = rn_cdiv(B, tile);
#define GM_LAUNCH(T, R)                                                                                                                          \
    hipLaunchKernelGGL((k_group_mid<T, R>), g, 256, 0, st, words, solo, B, n_words, n_words_first, ctl, idx0, idx1, key0, key1, blockhist, headcnt, \
                       order, seg_id, seg_first, super_id, n_seg)
    if (raw && pack && tile == RN_TILE) {      // + the weight packs on workgroups behind the grouping's
        if (raw == 1)
            hipLaunchKernelGGL(k_front_mid<1>, g + RN_FRONT_PACK_WGS, 256, 0, st, words, solo, B, ctl, idx0, idx1, key0, key1, blockhist, headcnt, order, seg_id,
                               seg_first, super_id, n_seg, g, *pack, zero1);
        else
            hipLaunchKernelGGL(k_front_mid<2>, g + RN_FRONT_PACK_WGS, 256, 0, st, words, solo, B, ctl, idx0, idx1, key0, key1, blockhist, headcnt, order, seg_id,
                               seg_first, super_id, n_seg, g, *pack, zero1);
        if (zeroed) *zeroed = zero1 ? 1 : 0;
        if (packed) *packed = 1;
    } else if (raw == 1 && tile == 4096) GM_LAUNCH(4096, 1);
    else if (raw == 2 && tile == 4096) GM_LAUNCH(4096, 2);
    else if (raw == 1) GM_LAUNCH(2048, 1);
    else if (raw == 2) GM_LAUNCH(2048, 2);
    else if (tile == 512) GM_LAUNCH(512, 0);
    else if (tile == 1024) GM_LAUNCH(1024, 0);
    else if (tile == 4096) GM_LAUNCH(4096, 0);
    else GM_LAUNCH(2048, 0);
#undef GM_LAUNCH
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// Grouping above GS_MAXB rows straight from ONE float32 / int32 id tensor (the step's GROUP phase, the one-call pairwise and listwise losses): the cooperative
// launch forms keys and solo flags itself -- no fill of `solo`, no key kernel.
// RECNOW_EUNSUPPORTED: other id types, or a batch beyond the co-resident grid (the caller takes recnow_group_keys + recnow_group_segments).
int rn_group_mid_raw(const void* group, int dtype, int64_t B, uint8_t* solo, int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id,
                     int32_t* n_seg, void* ws, size_t ws_bytes, hipStream_t st, const RnTileFwd* pack, unsigned long long* zero1, int* zeroed, int* packed) {
    if (packed) *packed = 0;
    static const bool on = []() { const char* e = getenv("RECNOW_GROUP_RAW"); return !e || e[0] != '0'; }();      // A/B switch
    if (!on || B <= GS_MAXB || (dtype != RECNOW_KEY_F32 && dtype != RECNOW_KEY_I32) || !solo || !ws) return RECNOW_EUNSUPPORTED;
    if (ws_bytes < recnow_group_segments_workspace_bytes(B, 1)) return RECNOW_EWORKSPACE;
    return rn_group_coop(dtype == RECNOW_KEY_F32 ? 1 : 2, (const uint32_t*)group, solo, B, 1, 1, order, seg_id, seg_first, super_id, n_seg, ws, ws_bytes, st, pack,
                         zero1, zeroed, packed);
}

extern "C" int recnow_group_segments(const uint32_t* words, const uint8_t* solo, int64_t B, int n_words, int n_words_first,
                                     int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id, int32_t* n_seg,
                                     void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || B > 0x7fffff00ll || n_words < 1 || n_words > RN_MAX_WORDS || n_words_first < 1 || n_words_first > n_words)
        return RECNOW_EINVAL;
    if (!seg_first || !n_seg) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        hipLaunchKernelGGL(k_seg_empty, 1, 1, 0, st, seg_first, n_seg);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (!words || !solo || !order || !seg_id || !super_id || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_group_segments_workspace_bytes(B, n_words)) return RECNOW_EWORKSPACE;
    if (B <= GS_MAXB && n_words == 1) {           // small batch, one key word: everything in one workgroup
        static bool lds_raised[64];      // per device, once (not a stream operation: kept out of every later call and of stream captures)
        int dev = 0;
        RN_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !lds_raised[dev]) {
            RN_HIP(hipFuncSetAttribute((const void*)k_group_small<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes()));
            RN_HIP(hipFuncSetAttribute((const void*)k_group_small<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes()));
            RN_HIP(hipFuncSetAttribute((const void*)k_group_small<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gs_lds_bytes()));
            if (dev >= 0 && dev < 64) lds_raised[dev] = true;
        }
        hipLaunchKernelGGL(k_group_small<0>, 1, GS_T, gs_lds_bytes(), st, words, solo, (int)B, order, seg_id, seg_first, super_id, n_seg);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    const int nblk = rn_cdiv(B, RN_TILE);
    int rc0 = rn_group_coop(0, words, const_cast<uint8_t*>(solo), B, n_words, n_words_first, order, seg_id, seg_first, super_id, n_seg, ws, ws_bytes, st);
    if (rc0 != RECNOW_EUNSUPPORTED) return rc0;
    RnCarver c(ws, ws_bytes);
    SortPlan* plan = c.take<SortPlan>(1);
    unsigned* ghist = c.take<unsigned>((size_t)n_words * 4 * 256);
    int32_t* idx0 = c.take<int32_t>(B + 1);
    int32_t* idx1 = c.take<int32_t>(B + 1);
    uint32_t* key0 = c.take<uint32_t>(B + 1);
    uint32_t* key1 = c.take<uint32_t>(B + 1);
    unsigned* blockhist = c.take<unsigned>((size_t)256 * (nblk > GM_MAXG ? nblk : GM_MAXG));
    int32_t* head = c.take<int32_t>(B + 1);
    int32_t* shead = c.take<int32_t>(B + 1);
    int32_t* seg_incl = c.take<int32_t>(B + 1);
    int32_t* super_incl = c.take<int32_t>(B + 1);
    void* scan_ws = (void*)(c.base + c.off);
    size_t scan_ws_bytes = ws_bytes - c.off;

    RN_HIP(hipMemsetAsync(ghist, 0, (size_t)2 * n_words * sizeof(unsigned), st));
    const int T = 256, G = rn_cdiv(B, T);
    {
        int gh = rn_cdiv(B, 256 * 8);
        if (gh > 1024) gh = 1024;
        hipLaunchKernelGGL(k_sort_ghist, gh, 256, 0, st, words, B, n_words, ghist, idx0);
    }
    hipLaunchKernelGGL(k_sort_plan, 1, 256, 0, st, ghist, B, n_words, plan);
    for (int p = 0; p < n_words * 4; ++p) {
        hipLaunchKernelGGL(k_sort_blockhist, nblk, 256, 0, st, words, B, n_words, p, plan, idx0, idx1, key0, key1, blockhist, nblk);
        const int scan_here = nblk <= 128;      // few blocks: every scatter workgroup scans the raw counts itself
        if (!scan_here) {   // millions of keys (pooled embedding ids): device-wide scan, in place (a skipped pass scans stale counts, unused)
            int rc2 = rn_scan<unsigned, unsigned, 0>(blockhist, blockhist, 256 * (int64_t)nblk, 0, scan_ws, scan_ws_bytes, st);
            if (rc2) return rc2;
        }
        hipLaunchKernelGGL(k_sort_scatter, nblk, 256, 0, st, words, B, n_words, p, plan, idx0, idx1, key0, key1, blockhist, nblk, scan_here);
    }
    hipLaunchKernelGGL(k_seg_heads, G, T, 0, st, words, solo, B, n_words, n_words_first, plan, idx0, idx1, key0, key1, order, head, shead);
    RN_LAUNCH_CHECK();
    int rc = rn_inclusive_scan_i32(head, seg_incl, B, scan_ws, scan_ws_bytes, st);
    if (rc) return rc;
    // every word belongs to groups[0] (the pooled lookup's ids, any single id tensor): a super-segment IS a segment -- k_seg_heads wrote equal flags --
    // and the second device-wide scan (37 us of the 6.5 M ids of the pooled lookup's backward) is the first one
    const bool same = n_words_first == n_words;
    if (!same) {
        rc = rn_inclusive_scan_i32(shead, super_incl, B, scan_ws, scan_ws_bytes, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_seg_finish, G, T, 0, st, head, seg_incl, same ? seg_incl : super_incl, B, seg_id, seg_first, super_id, n_seg);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
