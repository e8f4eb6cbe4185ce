// MultiDenseLayer (batched GEMM + bias + activation) and the softmax-gate mixing of MMoE / PLE.
//   /root/reference/rec_now/layers/multi_dense_layer.py:80-94
//   /root/reference/rec_now/layers/mmoe_layer.py:109-117, /root/reference/rec_now/layers/ple_layer.py:274-293
#include <stdlib.h>
#include "gemm.hpp"


// ---- N*U == 1 (one output column, the scoring head): three streaming kernels at HBM speed ---------------------------
// y[b] = act(x[b,:] . w + bias);   one wave per row, w held in registers across the rows a wave owns (D <= 4096)
template <int NV>
__global__ void __launch_bounds__(256)
k_head_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int64_t B, int D, int act,
           float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    float4 wv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = (i * 64 + lane) * 4;
        wv[i] = d < D ? *reinterpret_cast<const float4*>(w + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float bv = bias ? bias[0] : 0.f;
    // two rows per step: 2 * NV independent 16-byte loads per lane are in flight before the first reduction starts
    const int64_t step = (int64_t)gridDim.x * 4;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += 2 * step) {
        const int64_t b2 = b + step;
        const bool two = b2 < B;                              // wave-uniform
        const float* x2 = x + (two ? b2 : b) * D;             // no second row: re-read the first (result unused)
        float4 v[NV], u[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int d = (i * 64 + lane) * 4;
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            u[i] = v[i];
            if (d < D) {
                v[i] = *reinterpret_cast<const float4*>(x + b * D + d);
                u[i] = *reinterpret_cast<const float4*>(x2 + d);
            }
        }
        float acc = 0.f, acc2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            acc += v[i].x * wv[i].x + v[i].y * wv[i].y + v[i].z * wv[i].z + v[i].w * wv[i].w;
            acc2 += u[i].x * wv[i].x + u[i].y * wv[i].y + u[i].z * wv[i].z + u[i].w * wv[i].w;
        }
        acc = wave_sum(acc);
        acc2 = wave_sum(acc2);
        if (lane == 0) {
            y[b] = rn_act(acc + bv, act);
            if (two) y[b2] = rn_act(acc2 + bv, act);
        }
    }
}
// dx[b][:] = dz[b] * w,   dz = dy * act'(y)
__global__ void __launch_bounds__(256)
k_head_dx(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ w, int64_t B, int D, int act,
          float* __restrict__ dx) {
    const int64_t nq = B * (D / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % (D / 4)) * 4;
        const int64_t b = i / (D / 4);
        const float dz = dy[b] * rn_act_grad_from_out(y[b], act);
        const float4 wv = *reinterpret_cast<const float4*>(w + d);
        *reinterpret_cast<float4*>(dx + b * D + d) = make_float4(dz * wv.x, dz * wv.y, dz * wv.z, dz * wv.w);
    }
}
// part[chunk][d] = sum_{b in chunk} x[b][d] * dz[b];  a workgroup walks HEAD_ROWS full rows, thread = float4 column(s)
#define HEAD_ROWS 128
__global__ void __launch_bounds__(256)
k_head_dw_partial(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y, int64_t B, int D, int act,
                  float* __restrict__ part) {
    const int64_t b0 = (int64_t)blockIdx.x * HEAD_ROWS, b1 = min(B, b0 + HEAD_ROWS);
    for (int d = threadIdx.x * 4; d < D; d += 1024) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t b = b0;
        for (; b + 8 <= b1; b += 8) {                 // eight rows requested before the first is used; adds stay in row order
            float4 v[8];
            float dz[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = *reinterpret_cast<const float4*>(x + (b + r) * D + d);
#pragma unroll
            for (int r = 0; r < 8; ++r) dz[r] = dy[b + r] * rn_act_grad_from_out(y[b + r], act);
#pragma unroll
            for (int r = 0; r < 8; ++r) { acc.x += v[r].x * dz[r]; acc.y += v[r].y * dz[r]; acc.z += v[r].z * dz[r]; acc.w += v[r].w * dz[r]; }
        }
        for (; b < b1; ++b) {
            const float dz = dy[b] * rn_act_grad_from_out(y[b], act);
            const float4 v = *reinterpret_cast<const float4*>(x + b * D + d);
            acc.x += v.x * dz; acc.y += v.y * dz; acc.z += v.z * dz; acc.w += v.w * dz;
        }
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * D + d) = acc;
    }
}
static inline bool head_ok(const float* x, const float* w, int D, int U, int N) {
    return N == 1 && U == 1 && D % 4 == 0 && D >= 64 && D <= 4096 && ((((uintptr_t)x | (uintptr_t)w) & 15) == 0);
}
static inline size_t head_ws_bytes(int64_t B, int D) {
    const int64_t nchunk = rn_cdiv(B > 0 ? B : 1, HEAD_ROWS);
    return rn_align((size_t)nchunk * D * sizeof(float)) + rn_colsum_ws_bytes(nchunk, D);
}

// ---- weight gradient of a NARROW Dense layer ----------------------------------------------------------------------
// dkernel[D][U] = x^T dZ with D*U <= 4096 (SENET's excitation MLP: 64 x 32 and 32 x 64) and B in the 10^5.  As a GEMM this
// is one output tile with K = B: the split-K route writes a padded 256x64 partial tile per split and reduces ~60 MB for 8 KB
// of result (52 + 47 us per layer at B = 131 072).  Here a workgroup streams a slab of rows through LDS, every thread keeps
// D*U/256 outputs in registers, and the per-workgroup partials are added by the fixed-order column sum.
#define XTY_ROWS 32
static inline bool xty_ok(int64_t B, int D, int U, int N) {
    const int64_t du = (int64_t)D * U;
    if (!(N == 1 && B >= 4096 && D <= 128 && U <= 128 && du % 256 == 0)) return false;
    const int nq = (int)(du / 256);            // outputs per thread: consecutive columns of ONE row of dkernel
    return (nq == 1 || nq == 2 || nq == 4 || nq == 8 || nq == 16) && U % nq == 0;
}
static inline int xty_blocks(int64_t B) {
    int64_t g = (B + 127) / 128;          // >= 128 rows per workgroup: ~4 workgroups per CU at B ~ 10^5
    return (int)(g > 1024 ? 1024 : g);
}
static inline size_t xty_ws_bytes(int64_t B, int D, int U) {
    return rn_align((size_t)xty_blocks(B) * D * U * sizeof(float)) + rn_align((size_t)xty_blocks(B) * U * sizeof(float)) +
           rn_colsum_ws_bytes(xty_blocks(B), (int64_t)D * U);
}
// thread t owns outputs [t*NQ, t*NQ + NQ) of the row-major (D, U) result: one row m, NQ consecutive columns -- per batch
// row that is one x value and NQ/4 float4 pieces of dZ from LDS for NQ FMAs
template <int NQ>
__global__ void __launch_bounds__(256)
k_small_xty(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y, int zmode, int act, int64_t B,
            int D, int U, float* __restrict__ part, float* __restrict__ part_b /* optional [blocks][U]: column sums of dZ (bias gradient) */) {
    extern __shared__ __attribute__((aligned(16))) float xty_lds[];
    float* zs = xty_lds;                       // [XTY_ROWS][U]   (first: 16-byte aligned rows, U % 4 == 0 whenever NQ >= 4)
    float* xs = zs + XTY_ROWS * U;             // [XTY_ROWS][D]
    const int64_t per = (B + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(B, r0 + per);
    const int o0 = threadIdx.x * NQ, m = o0 / U, n0 = o0 % U;
    float acc[NQ], bz[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = bz[q] = 0.f;
    const float first = (part_b && m == 0) ? 1.f : 0.f;          // the threads of row 0 of dkernel also add up their dZ columns
    for (int64_t c0 = r0; c0 < r1; c0 += XTY_ROWS) {
        const int rows = (int)min((int64_t)XTY_ROWS, r1 - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < rows * D; i += 256) xs[i] = x[c0 * D + i];
        for (int i = threadIdx.x; i < rows * U; i += 256) {
            float v = dy[c0 * U + i];
            if (zmode == RECNOW_OPMODE_ACTGRAD) v *= rn_act_grad_from_out(y[c0 * U + i], act);
            zs[i] = v;
        }
        for (int i = rows * U + threadIdx.x; i < XTY_ROWS * U; i += 256) zs[i] = 0.f;       // short last chunk: zero rows add nothing
        for (int i = rows * D + threadIdx.x; i < XTY_ROWS * D; i += 256) xs[i] = 0.f;
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < XTY_ROWS; ++r) {
            const float xv = xs[r * D + m];
            if (NQ >= 4) {
#pragma unroll
                for (int q = 0; q < NQ; q += 4) {
                    const float4 z = *reinterpret_cast<const float4*>(zs + r * U + n0 + q);
                    acc[q] += xv * z.x; acc[q + 1] += xv * z.y; acc[q + 2] += xv * z.z; acc[q + 3] += xv * z.w;
                    bz[q] += first * z.x; bz[q + 1] += first * z.y; bz[q + 2] += first * z.z; bz[q + 3] += first * z.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float z = zs[r * U + n0 + q];
                    acc[q] += xv * z;
                    bz[q] += first * z;
                }
            }
        }
    }
    float* dst = part + (int64_t)blockIdx.x * D * U + o0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) dst[q] = acc[q];
    if (part_b && m == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) part_b[(int64_t)blockIdx.x * U + n0 + q] = bz[q];
    }
}

// Forward and input gradient of the same narrow layers: out[B][NO] = in'[B][KI] * Wm (+ bias, activation), with
//   forward : in' = x,                    Wm[k][n] = kernel[k][n]        (KI = D, NO = U)
//   dx      : in' = dy * act'(y) (zmode), Wm[k][n] = kernel[n][k]        (KI = U, NO = D)
// The (B, 64) x (64, 16)-sized products moved 40 MB in 46-64 us as one-column-tile GEMMs; here a workgroup takes 64 rows:
// the input tile and the whole weight matrix sit in LDS, a thread owns NO/4 outputs of one row.
#define ND_ROWS 64
static inline bool narrow_ok(int64_t B, int D, int U, int N) {
    return N == 1 && B >= 4096 && D % 4 == 0 && U % 4 == 0 && D <= 128 && U <= 128 && (int64_t)D * U <= 4096;
}
template <int NQ>                                  // outputs per thread = NO / 4, NO = 4 * NQ
__global__ void __launch_bounds__(256)
k_narrow_dense(const float* __restrict__ in, const float* __restrict__ in2, int zmode, int in_act, const float* __restrict__ w,
               int w_trans, const float* __restrict__ bias, int act, int64_t B, int KI, float* __restrict__ out) {
    constexpr int NO = 4 * NQ;
    extern __shared__ __attribute__((aligned(16))) float nd_lds[];
    float* ws = nd_lds;                            // [KI][NO]
    float* xs = ws + KI * NO;                      // [ND_ROWS][KI + 1], later the [ND_ROWS][NO] results (KI * NO % 4 == 0: 16-byte aligned)
    const int LDX = KI + 1;
    for (int i = threadIdx.x; i < KI * NO; i += 256) {
        const int k = i / NO, n = i - k * NO;
        ws[i] = w_trans ? w[n * KI + k] : w[i];
    }
    const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * NQ;
    float bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) bv[q] = bias ? bias[c0 + q] : 0.f;
    for (int64_t b0 = (int64_t)blockIdx.x * ND_ROWS; b0 < B; b0 += (int64_t)gridDim.x * ND_ROWS) {
        const int rows = (int)min((int64_t)ND_ROWS, B - b0);
        __syncthreads();
        for (int i = threadIdx.x; i < ND_ROWS * KI; i += 256) {
            const int rr = i / KI, k = i - rr * KI;
            float v = 0.f;
            if (rr < rows) {
                v = in[b0 * KI + i];
                if (zmode == RECNOW_OPMODE_ACTGRAD) v *= rn_act_grad_from_out(in2[b0 * KI + i], in_act);
            }
            xs[rr * LDX + k] = v;
        }
        __syncthreads();
        float acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = bv[q];
        for (int k = 0; k < KI; ++k) {
            const float xv = xs[r * LDX + k];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] += xv * ws[k * NO + c0 + q];
        }
        // the 64 x NO results are ONE contiguous block of `out`: hand them over through LDS (over the dead input tile) so that
        // the stores are whole coalesced runs instead of NQ dwords per lane at a 4*NO-byte stride
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) xs[r * NO + c0 + q] = rn_act(acc[q], act);
        __syncthreads();
        float* o = out + b0 * NO;
        for (int i = threadIdx.x * 4; i < rows * NO; i += 1024) *reinterpret_cast<float4*>(o + i) = *reinterpret_cast<const float4*>(xs + i);
    }
}
static int narrow_dense(const float* in, const float* in2, int zmode, int in_act, const float* w, int w_trans, const float* bias,
                        int act, int64_t B, int KI, int NO, float* out, hipStream_t st) {
    int64_t g = (B + ND_ROWS - 1) / ND_ROWS;
    if (g > 2048) g = 2048;
    const size_t lds = ((size_t)KI * NO + (size_t)ND_ROWS * (KI + 1 > NO ? KI + 1 : NO)) * sizeof(float);
#define ND_CASE(Q) case Q: hipLaunchKernelGGL(k_narrow_dense<Q>, (int)g, 256, lds, st, in, in2, zmode, in_act, w, w_trans, bias, act, B, KI, out); break;
    switch (NO / 4) {
        ND_CASE(1) ND_CASE(2) ND_CASE(3) ND_CASE(4) ND_CASE(5) ND_CASE(6) ND_CASE(7) ND_CASE(8) ND_CASE(9) ND_CASE(10) ND_CASE(11)
        ND_CASE(12) ND_CASE(13) ND_CASE(14) ND_CASE(15) ND_CASE(16) ND_CASE(17) ND_CASE(18) ND_CASE(19) ND_CASE(20) ND_CASE(21)
        ND_CASE(22) ND_CASE(23) ND_CASE(24) ND_CASE(25) ND_CASE(26) ND_CASE(27) ND_CASE(28) ND_CASE(29) ND_CASE(30) ND_CASE(31)
        ND_CASE(32)
        default: return RECNOW_EUNSUPPORTED;
    }
#undef ND_CASE
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// ---- MultiDense ------------------------------------------------------------------------------------------
extern "C" size_t recnow_multi_dense_workspace_bytes(int64_t B, int D, int U, int N) {
    if (B <= 0 || D <= 0 || U <= 0 || N <= 0) return 256;
    // largest split-K slab set among {dkernel: (D x U), K = B} plus the colsum slabs
    recnow_gemm_desc d = rn_gemm_desc_zero();
    d.M = D; d.N = U; d.K = (int)B; d.batch = N;
    size_t s = rn_gemm_ws_bytes(&d);
    d.M = (int)B; d.N = U; d.K = D; d.batch = N;
    size_t t = rn_gemm_ws_bytes(&d);
    if (t > s) s = t;
    d.M = (int)B; d.N = D; d.K = U; d.batch = N;
    t = rn_gemm_ws_bytes(&d);
    if (t > s) s = t;
    size_t sk = (N == 1 && U == 1) ? head_ws_bytes(B, D) : 0;
    if (xty_ok(B, D, U, N) && xty_ws_bytes(B, D, U) > sk) sk = xty_ws_bytes(B, D, U);
    return (s > sk ? s : sk) + rn_colsum_ws_bytes(B, U) * (size_t)(N > 0 ? N : 1) + 256;      // (the column sums of all N experts' bias gradients run as one launch)
}

extern "C" int recnow_multi_dense_fwd(const float* x, int x_batched, const float* kernel, const float* bias, int64_t B, int D,
                                      int U, int N, int act, float* y, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || D < 1 || U < 1 || N < 1 || B > 0x7fffffffll) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!x || !kernel || !y) return RECNOW_EINVAL;
    if (head_ok(x, kernel, D, U, N)) {
        int g = rn_cdiv(B, 4);
        if (g > 2048) g = 2048;
        hipStream_t hs = (hipStream_t)stream;
        if (D <= 256) hipLaunchKernelGGL(k_head_fwd<1>, g, 256, 0, hs, x, kernel, bias, B, D, act, y);
        else if (D <= 1024) hipLaunchKernelGGL(k_head_fwd<4>, g, 256, 0, hs, x, kernel, bias, B, D, act, y);
        else hipLaunchKernelGGL(k_head_fwd<16>, g, 256, 0, hs, x, kernel, bias, B, D, act, y);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (narrow_ok(B, D, U, N)) return narrow_dense(x, nullptr, RECNOW_OPMODE_NONE, 0, kernel, 0, bias, act, B, D, U, y, (hipStream_t)stream);
    recnow_gemm_desc d = rn_gemm_desc_zero();
    d.A = x; d.lda = D; d.a_batch_stride = x_batched ? B * D : 0; d.a_trans = 0;
    d.B = kernel; d.ldb = U; d.b_batch_stride = (int64_t)D * U; d.b_trans = 0;
    d.C = y; d.ldc = U; d.c_batch_stride = B * U;
    d.M = (int)B; d.N = U; d.K = D; d.batch = N;
    d.bias = bias; d.bias_batch_stride = U;
    d.act = act;
    return rn_gemm(&d, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int recnow_multi_dense_bwd(const float* x, int x_batched, const float* kernel, const float* y, const float* dy,
                                      int64_t B, int D, int U, int N, int act, float* dx, float* dkernel, float* dbias, void* ws,
                                      size_t ws_bytes, void* stream) {
    if (B < 0 || D < 1 || U < 1 || N < 1 || B > 0x7fffffffll) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (dkernel) RN_HIP(hipMemsetAsync(dkernel, 0, (size_t)N * D * U * sizeof(float), st));
        if (dbias) RN_HIP(hipMemsetAsync(dbias, 0, (size_t)N * U * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!x || !kernel || !y || !dy) return RECNOW_EINVAL;
    const int zmode = (act == RECNOW_ACT_LINEAR) ? RECNOW_OPMODE_NONE : RECNOW_OPMODE_ACTGRAD;   // dZ = dy * act'(y)
    int rc;
    if (head_ok(x, kernel, D, U, N) && (!dx || (((uintptr_t)dx & 15) == 0))) {
        if (dkernel) {
            if (ws_bytes < head_ws_bytes(B, D)) return RECNOW_EWORKSPACE;
            const int nchunk = rn_cdiv(B, HEAD_ROWS);
            float* part = (float*)ws;
            char* cws = (char*)ws + rn_align((size_t)nchunk * D * sizeof(float));
            hipLaunchKernelGGL(k_head_dw_partial, nchunk, 256, 0, st, x, dy, y, B, D, act, part);
            RN_LAUNCH_CHECK();
            if ((rc = rn_colsum(part, nullptr, 0, 0, nchunk, D, D, dkernel, 0, cws, ws_bytes - (size_t)(cws - (char*)ws), st))) return rc;
        }
        if (dx) {
            int g = rn_cdiv(B * (D / 4), 256);
            if (g > 8192) g = 8192;
            hipLaunchKernelGGL(k_head_dx, g, 256, 0, st, dy, y, kernel, B, D, act, dx);
            RN_LAUNCH_CHECK();
        }
        if (dbias && (rc = rn_colsum(dy, y, zmode, act, B, 1, 1, dbias, 0, ws, ws_bytes, st))) return rc;
        return RECNOW_OK;
    }
    if (dkernel && xty_ok(B, D, U, N)) {          // narrow layer: per-workgroup register tiles + fixed-order sum of the partials
        if (ws_bytes < xty_ws_bytes(B, D, U)) return RECNOW_EWORKSPACE;
        const int nb = xty_blocks(B);
        float* part = (float*)ws;
        float* part_b = dbias ? (float*)((char*)ws + rn_align((size_t)nb * D * U * sizeof(float))) : nullptr;
        char* cws = (char*)ws + rn_align((size_t)nb * D * U * sizeof(float)) + rn_align((size_t)nb * U * sizeof(float));
        const size_t lds = (size_t)XTY_ROWS * (D + U) * sizeof(float);
        switch (D * U / 256) {
            case 1: hipLaunchKernelGGL(k_small_xty<1>, nb, 256, lds, st, x, dy, y, zmode, act, B, D, U, part, part_b); break;
            case 2: hipLaunchKernelGGL(k_small_xty<2>, nb, 256, lds, st, x, dy, y, zmode, act, B, D, U, part, part_b); break;
            case 4: hipLaunchKernelGGL(k_small_xty<4>, nb, 256, lds, st, x, dy, y, zmode, act, B, D, U, part, part_b); break;
            case 8: hipLaunchKernelGGL(k_small_xty<8>, nb, 256, lds, st, x, dy, y, zmode, act, B, D, U, part, part_b); break;
            default: hipLaunchKernelGGL(k_small_xty<16>, nb, 256, lds, st, x, dy, y, zmode, act, B, D, U, part, part_b); break;
        }
        RN_LAUNCH_CHECK();
        if ((rc = rn_colsum(part, nullptr, 0, 0, nb, (int64_t)D * U, (int64_t)D * U, dkernel, 0, cws, ws_bytes - (size_t)(cws - (char*)ws), st)))
            return rc;
        if (dbias) {                              // the same pass produced per-workgroup column sums of dZ
            if ((rc = rn_colsum(part_b, nullptr, 0, 0, nb, U, U, dbias, 0, cws, ws_bytes - (size_t)(cws - (char*)ws), st))) return rc;
            dbias = nullptr;
        }
    } else if (dkernel) {   // dkernel[n] = x[n]^T dZ[n]      (D x U), K = B, split-K
        recnow_gemm_desc d = rn_gemm_desc_zero();
        d.A = x; d.lda = D; d.a_batch_stride = x_batched ? B * D : 0; d.a_trans = 1;
        d.B = dy; d.B2 = y; d.b_mode = zmode; d.b_act = act; d.ldb = U; d.b_batch_stride = B * U; d.b_trans = 0;
        d.C = dkernel; d.ldc = U; d.c_batch_stride = (int64_t)D * U;
        d.M = D; d.N = U; d.K = (int)B; d.batch = N;
        if ((rc = rn_gemm(&d, ws, ws_bytes, st))) return rc;
    }
    if (dbias) {
        // the bias gradients of all N experts in one pair of launches (N > 1, U >= 64); else one column sum per expert
        rc = N > 1 ? rn_colsum_batched(dy, y, zmode, act, B, U, U, N, (int64_t)B * U, dbias, U, ws, ws_bytes, st) : RECNOW_EUNSUPPORTED;
        if (rc == RECNOW_EUNSUPPORTED) {
            rc = RECNOW_OK;
            for (int n = 0; n < N && !rc; ++n)
                rc = rn_colsum(dy + (int64_t)n * B * U, y + (int64_t)n * B * U, zmode, act, B, U, U, dbias + (int64_t)n * U, 0, ws, ws_bytes, st);
        }
        if (rc) return rc;
    }
    if (dx && narrow_ok(B, D, U, N)) {
        if ((rc = narrow_dense(dy, y, zmode, act, kernel, 1, nullptr, RECNOW_ACT_LINEAR, B, U, D, dx, st))) return rc;
    } else if (dx) {        // dx[n] = dZ[n] kernel[n]^T
        recnow_gemm_desc d = rn_gemm_desc_zero();
        d.A = dy; d.A2 = y; d.a_mode = zmode; d.a_act = act; d.lda = U; d.a_trans = 0;
        d.B = kernel; d.ldb = U; d.b_trans = 1;            // kernel[n] stored [D][U] = [N_out][K]
        d.C = dx; d.ldc = D;
        d.M = (int)B; d.N = D; d.K = U;
        if (x_batched) {
            d.a_batch_stride = B * U; d.b_batch_stride = (int64_t)D * U; d.c_batch_stride = B * D; d.batch = N;
            if ((rc = rn_gemm(&d, ws, ws_bytes, st))) return rc;
        } else {     // broadcast input: sum over n, in order
            d.batch = 1;
            for (int n = 0; n < N; ++n) {
                d.A = dy + (int64_t)n * B * U; d.A2 = y + (int64_t)n * B * U;
                d.B = kernel + (int64_t)n * D * U;
                d.accumulate = n > 0;
                if ((rc = rn_gemm(&d, ws, ws_bytes, st))) return rc;
            }
        }
    }
    return RECNOW_OK;
}

// ---- gate softmax + expert mixing ----------------------------------------------------------------------------
#define MOE_MAX_N 64

// one wave per (t, b) row: softmax over N logits in registers of lane n, then every lane walks the U columns
__global__ void __launch_bounds__(256)
k_moe_mix_fwd(const float* __restrict__ logits, const float* const* __restrict__ experts, int T, int64_t B, int N, int U,
              float* __restrict__ gates, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t nrow = (int64_t)T * B;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrow; row += (int64_t)gridDim.x * 4) {
        const int64_t b = row % B;
        float lg = lane < N ? logits[row * N + lane] : -INFINITY;
        const float mx = wave_max(lg);
        const float e = lane < N ? expf(lg - mx) : 0.f;
        const float g = e / wave_sum(e);
        if (lane < N) gates[row * N + lane] = g;
        // wave-uniform trip count: a shuffle must never run with its source lane masked off (it would read 0)
        for (int u0 = 0; u0 < U; u0 += 64) {
            const int u = u0 + lane;
            float acc = 0.f;
            for (int n = 0; n < N; ++n) {
                const float gn = __shfl(g, n, 64);
                if (u < U) acc += gn * ((rn_gcf)experts[n])[b * U + u];      // global address space: see common.hpp
            }
            if (u < U) out[row * U + u] = acc;
        }
    }
}

// one wave per batch row b, all T gates of that row: dexperts and dlogits in one pass over dout / experts
__global__ void __launch_bounds__(256)
k_moe_mix_bwd(const float* __restrict__ gates, const float* const* __restrict__ experts, const float* __restrict__ dout, int T,
              int64_t B, int N, int U, float* __restrict__ dlogits, float* const* __restrict__ dexperts, int accumulate) {
    const int lane = threadIdx.x & 63;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += (int64_t)gridDim.x * 4) {
        // dexperts[n][b][u] = sum_t g[t][b][n] * dout[t][b][u]
        if (dexperts) {
            for (int u = lane; u < U; u += 64) {
                for (int n = 0; n < N; ++n) {
                    float acc = 0.f;
                    for (int t = 0; t < T; ++t) acc += gates[((int64_t)t * B + b) * N + n] * dout[((int64_t)t * B + b) * U + u];
                    const rn_gf de = (rn_gf)dexperts[n] + b * U + u;
                    *de = accumulate ? (*de + acc) : acc;
                }
            }
        }
        if (dlogits) {
            for (int t = 0; t < T; ++t) {
                const int64_t row = (int64_t)t * B + b;
                // dg[n] = dout[row] . E_n[b]; lane n ends up owning dg[n]
                float dg_mine = 0.f;
                for (int n = 0; n < N; ++n) {
                    float p = 0.f;
                    for (int u = lane; u < U; u += 64) p += dout[row * U + u] * ((rn_gcf)experts[n])[b * U + u];
                    p = wave_sum(p);
                    if (lane == n) dg_mine = p;
                }
                const float g = lane < N ? gates[row * N + lane] : 0.f;
                const float dot = wave_sum(g * dg_mine);
                if (lane < N) dlogits[row * N + lane] = g * (dg_mine - dot);
            }
        }
    }
}

static inline int moe_grid(int64_t rows) {
    int64_t g = (rows + 3) / 4;
    if (g > 256 * 16) g = 256 * 16;
    return (int)(g > 0 ? g : 1);
}

extern "C" int recnow_moe_mix_fwd(const float* logits, const float* const* experts, int T, int64_t B, int N, int U, float* gates,
                                  float* out, void* stream) {
    if (T < 1 || B < 0 || N < 1 || U < 1) return RECNOW_EINVAL;
    if (N > MOE_MAX_N) return RECNOW_EUNSUPPORTED;
    if (B == 0) return RECNOW_OK;
    if (!logits || !experts || !gates || !out) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_moe_mix_fwd, moe_grid((int64_t)T * B), 256, 0, (hipStream_t)stream, logits, experts, T, B, N, U, gates, out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_moe_mix_bwd(const float* gates, const float* const* experts, const float* dout, int T, int64_t B, int N, int U,
                                  float* dlogits, float* const* dexperts, int accumulate_dexperts, void* stream) {
    if (T < 1 || B < 0 || N < 1 || U < 1) return RECNOW_EINVAL;
    if (N > MOE_MAX_N) return RECNOW_EUNSUPPORTED;
    if (B == 0) return RECNOW_OK;
    if (!gates || !experts || !dout) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_moe_mix_bwd, moe_grid(B), 256, 0, (hipStream_t)stream, gates, experts, dout, T, B, N, U, dlogits, dexperts,
                       accumulate_dexperts);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
