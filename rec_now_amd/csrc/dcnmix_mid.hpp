// Fused sub-space stage of DCNMixLayer (dcnmix_mid.hip): S x S expert products + softmax gate, one kernel per direction.
#pragma once
#include "common.hpp"

bool rn_mix_mid_supported(int S, int N, int LDT);
size_t rn_mix_mid_bwd_ws_bytes(int64_t B, int S, int N);
// T1 = [H1 | logits | pad] (B x LDT)  ->  T2 = [H2 | G | 0],  T2g = [G*H2 | G | 0]
int rn_mix_mid_fwd(const float* T1, const float* V, float* T2, float* T2g, int64_t B, int S, int N, int LDT, int act_outer, hipStream_t st);
// dT2g, T2, T1 (B x LDT), V (N,S,S)  ->  dT1 = [dA | dlogits | 0] (B x LDT),  dV (N,S,S);  rscale (B, optional): dT2g rows are scaled by it on load
int rn_mix_mid_bwd(const float* dT2g, const float* T2, const float* T1, const float* V, float* dT1, float* dV, int64_t B, int S, int N,
                   int LDT, int act_inner, int act_outer, void* ws, size_t ws_bytes, hipStream_t st, const float* rscale = nullptr);
