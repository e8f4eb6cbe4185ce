// Single-workgroup grouping of a small batch (B <= 8192 rows, one 32-bit key word) on LDS-resident keys: shared by
// k_group_small (scan_sort.hip) and the fused small pairwise loss (pairwise.hip).
#pragma once
#include "common.hpp"

#define GS_T 1024
#define GS_KPT 8
#define GS_MAXB (GS_T * GS_KPT)

__device__ __forceinline__ unsigned gs_block_exclusive_scan(unsigned v, unsigned* wsum /* [16] */, unsigned* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    unsigned off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned x = wsum[i];
        if (i < w) off += x;
        tot += x;
    }
    if (total) *total = tot;
    return off + inc - v;
}


// Group ids usually arrive as float32 holding small non-negative integers (the reference casts its ids to float: rec_block/
// pairwise_loss_from_batch.py:33-35 compares g_i - g_j == 0.0).  Their bit patterns vary in the exponent AND the top mantissa bits: ids 0..4095
// differ in 19 bits = 3 digit passes, while their integer values differ in 12 bits = 2 passes (ids 0..127: 1 pass instead of 3).  Grouping only
// needs equal keys to end up adjacent, so any INJECTIVE image of the keys sorts as well: when EVERY key of the batch is a non-negative
// integer-valued float below 2^24 (a batch-wide AND computed beside the varying-bit masks), the passes sort the integer values instead.
__device__ __forceinline__ bool gs_int_key(uint32_t w, uint32_t* image) {
    const float f = __uint_as_float(w);
    const bool small = (w >> 31) == 0u && f < 16777216.f;      // NaN compares false
    const uint32_t i = small ? (uint32_t)f : 0u;
    *image = i;
    return small && (float)i == f;
}


// Bits that differ somewhere in the batch, from the threads' partial OR / AND of the keys (vor, vand) and of their small-integer images (ior,
// iand; bad != 0: this thread saw a key without an image).  When every key has an image the keys in LDS are REPLACED by their images and the
// images' varying bits are returned: float ids 0..127 then cost two 4-bit passes instead of four.  wsum: >= 32 words.  Ends with a barrier.
__device__ __forceinline__ unsigned gs_varying_bits(uint32_t* key0, int B, unsigned vor, unsigned vand, unsigned ior, unsigned iand, unsigned bad,
                                                    unsigned* wsum) {
    const bool plain = __syncthreads_or((int)bad) != 0;    // block-uniform
    unsigned a = plain ? vor : ior, b = plain ? vand : iand;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a |= __shfl_xor(a, o, 64);
        b &= __shfl_xor(b, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = a; wsum[16 + (threadIdx.x >> 6)] = b; }
    if (!plain)
        for (int i = threadIdx.x; i < B; i += GS_T) key0[i] = (uint32_t)__uint_as_float(key0[i]);
    __syncthreads();
    a = 0; b = 0xffffffffu;
    for (int i = 0; i < 16; ++i) { a |= wsum[i]; b &= wsum[16 + i]; }
    __syncthreads();
    return a ^ b;
}

// Stable LSD radix sort (4-bit digits, constant digits skipped) of B <= GS_MAXB keys held in LDS, by one 1024-thread workgroup.
// ka/ia hold keys / original positions on entry and the sorted order on return (the pointers are swapped in place); kb/ib are
// the ping-pong buffers, cnt [16][GS_T] u16, wsum >= 34 words.  Thread t owns positions [8t, 8t + 8) of the current order.
__device__ __forceinline__ void gs_radix_sort_lds(uint32_t*& ka, uint32_t*& kb, uint16_t*& ia, uint16_t*& ib, uint16_t* cnt,
                                                  unsigned* wsum, int B, unsigned varying) {
    const int tid = threadIdx.x;
    const int lo = tid * GS_KPT, hi = min(B, lo + GS_KPT);
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 4 * pass;
        if (((varying >> shift) & 15u) == 0) continue;    // block-uniform: this digit is the same for every key
        unsigned long long c64 = 0;
        for (int i = lo; i < hi; ++i) c64 += 1ull << (4 * ((ka[i] >> shift) & 15u));
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d * GS_T + tid] = (uint16_t)((c64 >> (4 * d)) & 15u);
        __syncthreads();
        // exclusive scan of the flattened [digit][thread] counters: thread t owns entries [16t, 16t + 16)
        unsigned loc[16], sum = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { loc[j] = cnt[tid * 16 + j]; sum += loc[j]; }
        unsigned run = gs_block_exclusive_scan(sum, wsum, nullptr);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) { cnt[tid * 16 + j] = (uint16_t)run; run += loc[j]; }
        __syncthreads();
        unsigned long long r64 = 0;                       // running per-digit rank inside this thread's chunk
        for (int i = lo; i < hi; ++i) {
            const uint32_t k = ka[i];
            const unsigned d = (k >> shift) & 15u;
            const unsigned dst = cnt[d * GS_T + tid] + (unsigned)((r64 >> (4 * d)) & 15u);
            r64 += 1ull << (4 * d);
            kb[dst] = k;
            ib[dst] = ia[i];
        }
        __syncthreads();
        uint32_t* tk = ka; ka = kb; kb = tk;
        uint16_t* ti = ia; ia = ib; ib = ti;
    }
}

static inline size_t gs_lds_bytes() { return (size_t)GS_MAXB * (4 + 4 + 2 + 2) + (size_t)16 * GS_T * 2 + 34 * sizeof(unsigned); }
