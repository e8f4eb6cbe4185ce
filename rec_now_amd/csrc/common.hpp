// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of librecnow_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/recnow.h"

#define RN_WAVE 64

// Forward a HIP error (positive hipError_t) across the C ABI; no exceptions.
#define RN_HIP(expr)                                   \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return (int)_e;          \
    } while (0)

#define RN_LAUNCH_CHECK() RN_HIP(hipGetLastError())

static inline size_t rn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int rn_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Workspace carving: every carve is 256-B aligned so vector accesses stay aligned.
// internal flag bits of rn_pair_bpr_onepass (pairwise.hip; set by the step's loss stage, dcnmix.hip; never part of the C ABI's flags):
#define RN_PAIR_UNPACKED (1 << 30)         // no pack launch: the walk's workgroups fill their LDS stages from (scores, labels, mask) through `order`
#define RN_PAIR_NPAIR_ZEROED (1 << 29)     // *n_pair has been cleared by an earlier launch of the caller (the step's front kernel)
struct RnCarver {
    char* base;
    size_t off;
    size_t cap;
    RnCarver(void* p, size_t c) : base((char*)p), off(0), cap(c) {}
    template <typename T>
    T* take(size_t n) {
        T* r = (T*)(base + off);
        off += rn_align(n * sizeof(T));
        return r;
    }
    bool ok() const { return off <= cap; }
};

// ---- wave64 reductions -------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `red` = >= 16 elements of LDS.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    T r = (T)0;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// ---- activations (enum recnow_act) -------------------------------------------------------------
__device__ __forceinline__ float rn_sigmoid(float x) {
    // stable logistic
    if (x >= 0.f) {
        float e = __expf(-x);
        return 1.f / (1.f + e);
    }
    float e = __expf(x);
    return e / (1.f + e);
}
// tanh through the hardware exp2 / rcp: (1 - e) / (1 + e) with e = exp(-2|x|), sign restored.  |error| <= ~2e-7 absolute (e carries the
// rounding of its argument and one ulp of v_exp_f32; 1 + e is in (1, 2]), on outputs of magnitude <= 1 -- two orders below the 1e-5
// parity bar; libm's tanhf is ~45 instructions with branches, and the DCN-v2 step applies it to 50 M elements in epilogues that no
// MFMA work overlaps.
// Small arguments take the odd series instead (selected, not branched): (1 - e) cancels there, an ABSOLUTE 2e-7 is a RELATIVE 1e-4
// on tanh(2e-3) -- layers with small weights (default initialisers at D = 1024) run entirely in that range, and the full-size
// parity test sat at 1.0e-5 because of it.  |x| < 0.25: x (1 - x^2/3 + 2 x^4/15 - 17 x^6/315 + 62 x^8/2835), truncation < 1e-8 relative.
__device__ __forceinline__ float rn_tanh(float x) {
    const float a = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(-2.885390082f * a);
    const float t = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
    const float x2 = x * x;
    const float s = a * (1.f + x2 * (-0.33333333f + x2 * (0.13333333f + x2 * (-0.053968254f + x2 * 0.021869489f))));
    return copysignf(a < 0.25f ? s : t, x);
}

__device__ __forceinline__ float rn_act(float x, int act) {
    switch (act) {
        case RECNOW_ACT_RELU: return x > 0.f ? x : 0.f;
        case RECNOW_ACT_TANH: return rn_tanh(x);
        case RECNOW_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        default: return x;
    }
}
// derivative expressed through the OUTPUT y = act(z)
__device__ __forceinline__ float rn_act_grad_from_out(float y, int act) {
    switch (act) {
        case RECNOW_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case RECNOW_ACT_TANH: return 1.f - y * y;
        case RECNOW_ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

// ---- pointers fetched from device-resident pointer arrays (lists of field tensors) ---------------
// To the compiler such a pointer is GENERIC: loads through it become flat_load, which count on lgkmcnt as well as vmcnt, so
// the s_waitcnt lgkmcnt(0) behind every scalar fetch of the NEXT pointer also drains the vector loads in flight (FM forward
// issued its 8 loads per lane one at a time).  These types state that the target is global memory: global_load / global_store,
// vmcnt only.  Native vectors, because HIP's float4 struct cannot be copied out of a non-generic address space.
#define RN_GLOBAL __attribute__((address_space(1)))
typedef float rn_f4 __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(1)))* rn_gcf;
typedef float __attribute__((address_space(1)))* rn_gf;
typedef const rn_f4 __attribute__((address_space(1)))* rn_gcf4;
typedef rn_f4 __attribute__((address_space(1)))* rn_gf4;
// Streamed-once global accesses (every byte of the tensor is touched once by the launch): non-temporal loads / stores.  Round 4 measured them on the
// HBM-bound layer kernels (DCNLayer, FMLayer: +10..13 % on one box, tools/micro/stream_bench.py); -DRN_STREAM_PLAIN (tools/build_variant.py) builds
// the plain-access variant for A/B runs.  `p` is a (possibly address-space-qualified) pointer to a native vector or scalar.
#ifndef RN_STREAM_PLAIN
#define RN_LD_STREAM(p) __builtin_nontemporal_load(p)
#define RN_ST_STREAM(p, v) __builtin_nontemporal_store((v), (p))
#else
#define RN_LD_STREAM(p) (*(p))
#define RN_ST_STREAM(p, v) (*(p) = (v))
#endif

