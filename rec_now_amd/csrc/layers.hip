// MultiDenseLayer (batched GEMM + bias + activation) and the softmax-gate mixing of MMoE / PLE.
//   /root/reference/rec_now/layers/multi_dense_layer.py:80-94
//   /root/reference/rec_now/layers/mmoe_layer.py:109-117, /root/reference/rec_now/layers/ple_layer.py:274-293
#include "gemm.hpp"

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
    return s + rn_colsum_ws_bytes(B, U) + 256;
}

extern "C" int recnow_multi_dense_fwd(const float* x, int x_batched, const float* kernel, const float* bias, int64_t B, int D,
                                      int U, int N, int act, float* y, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || D < 1 || U < 1 || N < 1 || B > 0x7fffffffll) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!x || !kernel || !y) return RECNOW_EINVAL;
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
    if (dkernel) {   // dkernel[n] = x[n]^T dZ[n]      (D x U), K = B, split-K
        recnow_gemm_desc d = rn_gemm_desc_zero();
        d.A = x; d.lda = D; d.a_batch_stride = x_batched ? B * D : 0; d.a_trans = 1;
        d.B = dy; d.B2 = y; d.b_mode = zmode; d.b_act = act; d.ldb = U; d.b_batch_stride = B * U; d.b_trans = 0;
        d.C = dkernel; d.ldc = U; d.c_batch_stride = (int64_t)D * U;
        d.M = D; d.N = U; d.K = (int)B; d.batch = N;
        if ((rc = rn_gemm(&d, ws, ws_bytes, st))) return rc;
    }
    if (dbias) {
        for (int n = 0; n < N; ++n)
            if ((rc = rn_colsum(dy + (int64_t)n * B * U, y + (int64_t)n * B * U, zmode, act, B, U, U, dbias + (int64_t)n * U, 0, ws,
                                ws_bytes, st)))
                return rc;
    }
    if (dx) {        // dx[n] = dZ[n] kernel[n]^T
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
                if (u < U) acc += gn * experts[n][b * U + u];
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
                    float* de = dexperts[n] + b * U + u;
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
                    for (int u = lane; u < U; u += 64) p += dout[row * U + u] * experts[n][b * U + u];
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
