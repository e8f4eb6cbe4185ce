// Fused sub-space stage of DCNMixLayer (dcnmix_mid.hip): S x S expert products + softmax gate, one kernel per direction.
#pragma once
#include "common.hpp"

bool rn_mix_mid_supported(int S, int N, int LDT);
// Split-K slabs of the product that feeds a sub-space kernel (GEMM1 -> T1, or the dT2g product): slab s, row m, column c at
// p[s * stride + m * ld + c], the N side columns (gate logits / their gradients) behind the NS main columns.  At shard sizes where
// those products are split (B <= 32 768 at D = 1024) the sub-space kernel sums the slabs while it loads its tile -- the separate
// reduction launch (7-9 us, a quarter of the product itself at 8192 rows) disappears.
struct RnSlabs {
    const float* p;
    int n, ld;
    int64_t stride;
    int stagger;        // backward fast kernel, experiment (RECNOW_MID_STAGGER, 10 ns ticks): the second resident round of workgroups starts this late
};
bool rn_mix_mid_absorbs_slabs(int64_t B, int S, int N, int LDT);
size_t rn_mix_mid_bwd_ws_bytes(int64_t B, int S, int N);
// T1 = [H1 | logits | pad] (B x LDT)  ->  T2 = [H2 | G | 0],  T2g = [G*H2 | G | 0]
// slabs != NULL (rn_mix_mid_absorbs_slabs shapes only): T1 is an OUTPUT -- [act_inner(sum of the slabs' main columns) | summed side columns]
int rn_mix_mid_fwd(const float* T1, const float* V, float* T2, float* T2g, int64_t B, int S, int N, int LDT, int act_outer, hipStream_t st,
                   const RnSlabs* slabs = nullptr, int act_inner = 0);
// dT2g, T2, T1 (B x LDT), V (N,S,S)  ->  dT1 = [dA | dlogits | 0] (B x LDT),  dV (N,S,S);  rscale (B, optional): dT2g rows are scaled by it on load
int rn_mix_mid_bwd(const float* dT2g, const float* T2, const float* T1, const float* V, float* dT1, float* dV, int64_t B, int S, int N,
                   int LDT, int act_inner, int act_outer, void* ws, size_t ws_bytes, hipStream_t st, const float* rscale = nullptr,
                   bool defer_dv = false, const RnSlabs* slabs = nullptr);      // slabs != NULL: dT2g is not read, the slabs' sum is
// defer_dv: the per-workgroup dV partials are left in `ws` (rn_mix_mid_bwd_nparts(B) blocks of N*S*S floats) for the caller to sum
// (rn_layer_end_reduce) instead of a reduction launch of their own
int rn_mix_mid_bwd_nparts(int64_t B);
