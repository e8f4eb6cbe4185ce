// Chained products of DCNMixLayer (csrc/dcnmix_chain.hip): the K = N*S + N deep product that leaves a cross layer and the
// K = D deep product that enters the next one, in one kernel per 128-row block of the batch.
#pragma once
#include "common.hpp"

// forward: out = x0 * (T2g [W; b]) (and O = T2g [W; b] when O != nullptr), T1n = [act_inner(out U_next) | out K_next]
bool rn_mix_chain_fwd_supported(int64_t B, int D, int S, int N, int LDT);
int rn_mix_chain_fwd(const float* T2g, const float* Wc2, const float* x0, float* out, float* O, const float* Wc1_next,
                     const float* gate_next, float* T1_next, int64_t B, int D, int LDT, int act_inner, hipStream_t st);
