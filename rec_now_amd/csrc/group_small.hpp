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

// Stable LSD radix sort of B <= GS_MAXB keys held in LDS, by one 1024-thread workgroup (round 5: 8-bit digits ranked by wave-level multi-split; the
// 4-bit passes with per-thread counters took 5.2 us each -- phase stamps of tools/gs_trace.py -- i.e. 10.4 us for ids 0..127, 15.2 us for ids 0..1023).
// ka/ia hold keys / original positions on entry and the sorted order on return (the pointers are swapped in place); kb/ib are the ping-pong buffers,
// cnt >= 256 * 18 u16 (the callers give [16][GS_T]), wsum >= 34 words.
// Per pass (digits whose bits are constant over the batch are skipped, and only the VARYING bits of a digit are matched): wave w owns positions
// [512 w, 512 w + 512) in 8 rounds of 64 consecutive keys; in a round the lanes with equal digits find each other by one ballot per varying bit, the
// lowest of them advances the wave's counter of that digit (cnt[digit][wave], LDS operations of a wave complete in order) and every lane keeps
// (count before the round) + (equal digits in lower lanes) = its rank among the wave's keys of that digit.  An exclusive scan over cnt in
// (digit, wave) order turns the counters into output offsets; destination = offset + rank: stable.
__device__ __forceinline__ void gs_radix_sort_lds(uint32_t*& ka, uint32_t*& kb, uint16_t*& ia, uint16_t*& ib, uint16_t* cnt,
                                                  unsigned* wsum, int B, unsigned varying) {
    constexpr int WS = 18;                                // u16 per digit row (16 used): 9 words -- the rows of 32 consecutive digits start in 32 different banks
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* cnt32 = reinterpret_cast<uint32_t*>(cnt);
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        const unsigned vb = (varying >> shift) & 255u;
        if (vb == 0) continue;                            // block-uniform: this digit is the same for every key
        for (int i = tid; i < 256 * WS / 2; i += GS_T) cnt32[i] = 0u;
        __syncthreads();
        uint32_t kreg[GS_KPT];
        unsigned ireg[GS_KPT], lr[GS_KPT];
#pragma unroll
        for (int r = 0; r < GS_KPT; ++r) {
            const int pos = w * (64 * GS_KPT) + r * 64 + lane;
            const bool ok = pos < B;
            const int pc = ok ? pos : 0;
            const uint32_t k = ka[pc];
            kreg[r] = k;
            ireg[r] = ia[pc];
            const unsigned d = (k >> shift) & 255u;
            unsigned long long peers = __ballot(ok);
            for (unsigned bits = vb; bits; bits &= bits - 1u) {      // block-uniform trip count
                const bool bit = (d >> (__ffs(bits) - 1)) & 1u;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const unsigned below = (unsigned)__popcll(peers & lt);
            const unsigned prior = cnt[d * WS + w];
            if (ok && below == 0u) cnt[d * WS + w] = (uint16_t)(prior + (unsigned)__popcll(peers));
            lr[r] = prior + below;
        }
        __syncthreads();
        // exclusive scan of the counters in (digit, wave) order: thread t owns digit t / 4, waves 4 (t % 4) .. + 3
        const int sd = (tid >> 2) * WS + (tid & 3) * 4;
        unsigned loc[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { loc[j] = cnt[sd + j]; sum += loc[j]; }
        unsigned run = gs_block_exclusive_scan(sum, wsum, nullptr);
#pragma unroll
        for (int j = 0; j < 4; ++j) { cnt[sd + j] = (uint16_t)run; run += loc[j]; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GS_KPT; ++r) {
            const int pos = w * (64 * GS_KPT) + r * 64 + lane;
            if (pos < B) {
                const unsigned dst = cnt[((kreg[r] >> shift) & 255u) * WS + w] + lr[r];
                kb[dst] = kreg[r];
                ib[dst] = (uint16_t)ireg[r];
            }
        }
        __syncthreads();
        uint32_t* tk = ka; ka = kb; kb = tk;
        uint16_t* ti = ia; ia = ib; ib = ti;
    }
}

static inline size_t gs_lds_bytes() { return (size_t)GS_MAXB * (4 + 4 + 2 + 2) + (size_t)16 * GS_T * 2 + 34 * sizeof(unsigned); }
