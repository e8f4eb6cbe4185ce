// Device-wide prefix sums (three launches: block sums, scan of block sums, block scan + offset).
// Inputs here are <= a few MB of int32, so this is launch-latency work; kept simple and deterministic.
#pragma once
#include "common.hpp"

#define RN_SCAN_T 1024
#define RN_SCAN_ITEMS 4
#define RN_SCAN_TILE (RN_SCAN_T * RN_SCAN_ITEMS)

static inline size_t rn_scan_ws_bytes(int64_t n) {
    const size_t nblk = (size_t)rn_cdiv(n > 0 ? n : 1, RN_SCAN_TILE);
    return rn_align((nblk + 1) * sizeof(long long));
}

template <typename TI>
static __global__ void __launch_bounds__(RN_SCAN_T)
k_scan_blocksum(const TI* __restrict__ in, int64_t n, long long* __restrict__ bsum) {
    __shared__ long long red[16];
    const int64_t base = (int64_t)blockIdx.x * RN_SCAN_TILE + (int64_t)threadIdx.x * RN_SCAN_ITEMS;
    long long s = 0;
#pragma unroll
    for (int i = 0; i < RN_SCAN_ITEMS; ++i)
        if (base + i < n) s += (long long)in[base + i];
    s = block_sum<long long>(s, red);
    if (threadIdx.x == 0) bsum[blockIdx.x] = s;
}

// exclusive scan of bsum[nblk] in place; bsum[nblk] = total
static __global__ void __launch_bounds__(RN_SCAN_T)
k_scan_blockoffsets(long long* __restrict__ bsum, int nblk) {
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nblk; base += RN_SCAN_T) {
        const int i = base + threadIdx.x;
        const long long v = i < nblk ? bsum[i] : 0;
        long long inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            long long t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        long long woff = 0;
        for (int j = 0; j < w; ++j) woff += wsum[j];
        const long long carry = carry_s;
        if (i < nblk) bsum[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == RN_SCAN_T - 1) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[nblk] = carry_s;
}

// INCLUSIVE != 0: out[i] = sum in[0..i];  else out[i] = sum in[0..i-1] and (if write_total) out[n] = total
template <typename TI, typename TO, int INCLUSIVE>
static __global__ void __launch_bounds__(RN_SCAN_T)
k_scan_final(const TI* __restrict__ in, TO* __restrict__ out, int64_t n, const long long* __restrict__ boff, int write_total) {
    __shared__ long long wsum[16];
    const int64_t base = (int64_t)blockIdx.x * RN_SCAN_TILE + (int64_t)threadIdx.x * RN_SCAN_ITEMS;
    long long v[RN_SCAN_ITEMS];
    long long s = 0;
#pragma unroll
    for (int i = 0; i < RN_SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? (long long)in[base + i] : 0;
        s += v[i];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    long long run = boff[blockIdx.x] + inc - s;
    for (int j = 0; j < w; ++j) run += wsum[j];
#pragma unroll
    for (int i = 0; i < RN_SCAN_ITEMS; ++i) {
        if (base + i < n) {
            if (INCLUSIVE) out[base + i] = (TO)(run + v[i]);
            else out[base + i] = (TO)run;
        }
        run += v[i];
    }
    if (!INCLUSIVE && write_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = (TO)boff[gridDim.x];
}

template <typename TI, typename TO, int INCLUSIVE>
static inline int rn_scan(const TI* in, TO* out, int64_t n, int write_total, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0) return RECNOW_OK;
    if (ws_bytes < rn_scan_ws_bytes(n)) return RECNOW_EWORKSPACE;
    const int nblk = rn_cdiv(n, RN_SCAN_TILE);
    long long* bsum = (long long*)ws;
    hipLaunchKernelGGL((k_scan_blocksum<TI>), nblk, RN_SCAN_T, 0, st, in, n, bsum);
    hipLaunchKernelGGL(k_scan_blockoffsets, 1, RN_SCAN_T, 0, st, bsum, nblk);
    hipLaunchKernelGGL((k_scan_final<TI, TO, INCLUSIVE>), nblk, RN_SCAN_T, 0, st, in, out, n, bsum, write_total);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

static inline int rn_inclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes, hipStream_t st) {
    return rn_scan<int32_t, int32_t, 1>(in, out, n, 0, ws, ws_bytes, st);
}
// out has n+1 entries; out[n] = total
static inline int rn_exclusive_scan_i32_i64(const int32_t* in, int64_t* out, int64_t n, void* ws, size_t ws_bytes, hipStream_t st) {
    return rn_scan<int32_t, int64_t, 0>(in, out, n, 1, ws, ws_bytes, st);
}
