// Internal interface of the exact-fp32 MFMA GEMM (gemm.hip).  The public C struct lives in include/recnow.h.
#pragma once
#include "common.hpp"
#include <string.h>

// C[b] = epilogue( opA(A[b]) * opB(B[b]) ), b = 0..batch-1, see recnow_gemm_desc in include/recnow.h
int rn_gemm(const recnow_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st);
size_t rn_gemm_ws_bytes(const recnow_gemm_desc* d);
// A split-K product whose slab reduction is left to the caller: rn_gemm_deferred launches the product only and describes the
// reduction; rn_layer_end_reduce then runs up to two such reductions and the dV partial sum of the sub-space backward kernel as ONE
// launch (DCN-v2: the dW, dU and dV of a cross layer end in one launch instead of three).  `valid == 0`: the product was not
// split, its output is final.  The slabs (`ws`) must stay untouched until the reduction has run.
struct RnDeferredReduce {
    alignas(16) char k[1024];      // the launch's GemmK
    int variant;                   // k_gemm_splitk_reduce<VEC, QUAD>: 0 <false,false>, 1 <true,false>, 2 <true,true>
    int blocks;
    int valid;
};
int rn_gemm_deferred(const recnow_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st, RnDeferredReduce* out);
// K-split target of the following launches of this host thread: workgroup slots to fill (0 = default 512); products launched as concurrent pairs ask for 256
void rn_gemm_split_slots(int slots);
// the slabs of a deferred (valid) product: slab s, row m, column c at partial[s * stride + m * ld + c]; side columns behind the N main ones
void rn_deferred_slabs(const RnDeferredReduce* r, const float** partial, int* nsplit, int* ld, int64_t* stride);
int rn_layer_end_reduce(const RnDeferredReduce* a, const RnDeferredReduce* b, const float* dv_part, int dv_nparts, int dv_total, float* dV,
                        hipStream_t st);
// 0 = exact fp32 MFMA (default), 1 = bf16x3 split for the products that have a split kernel (gemm_split.hip)
int rn_gemm_precision();
int rn_gemm_set_precision(int mode);

// Split-precision piece planes (gemm_split.hip).  rn_split_planes_multi: the B operands of several products in ONE launch (the packed weights
// of every cross layer, once per step); planes of one job: rn_gemm_split_planes_bytes(K, N) bytes, layout [piece][K / 8][N] units of 8 bf16.
#define RN_SPLIT_MAX_JOBS 16
struct RnSplitJob { const float* B; int64_t ldb; int b_kc, K, N; char* planes; };
struct RnSplitJobs { RnSplitJob job[RN_SPLIT_MAX_JOBS]; int n; };
int rn_split_planes_multi(const RnSplitJobs& jobs, hipStream_t st);
size_t rn_gemm_split_planes_bytes(int K, int N);
// The next rn_gemm / rn_gemm_deferred call of THIS host thread finds the piece planes of its B operand at `planes` (the layout and size its own
// split would write) and skips that split; consumed (or dropped) by that one call.
void rn_gemm_planes_hint(const void* planes);

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
int rn_colsum_batched(const float* X, const float* X2, int mode, int act, int64_t M, int64_t N, int64_t ld, int batch, int64_t x_bs, float* out,
                      int64_t out_bs, void* ws, size_t ws_bytes, hipStream_t st);
