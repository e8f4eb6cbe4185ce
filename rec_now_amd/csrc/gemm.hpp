// Internal interface of the exact-fp32 MFMA GEMM (gemm.hip).  The public C struct lives in include/recnow.h.
#pragma once
#include "common.hpp"
#include <string.h>

// C[b] = epilogue( opA(A[b]) * opB(B[b]) ), b = 0..batch-1, see recnow_gemm_desc in include/recnow.h
int rn_gemm(const recnow_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st);
size_t rn_gemm_ws_bytes(const recnow_gemm_desc* d);
// 0 = exact fp32 MFMA (default), 1 = bf16x3 split for the products that have a split kernel (gemm_split.hip)
int rn_gemm_precision();
int rn_gemm_set_precision(int mode);

static inline recnow_gemm_desc rn_gemm_desc_zero() {
    recnow_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.batch = 1;
    return d;
}

// out[n] (+)= sum_m X[m][n] * (mode ? f(X2[m][n]) : 1)   -- deterministic two-stage column sum
int rn_colsum(const float* X, const float* X2, int mode, int act, int64_t M, int64_t N, int64_t ld, float* out,
              int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
size_t rn_colsum_ws_bytes(int64_t M, int64_t N);
