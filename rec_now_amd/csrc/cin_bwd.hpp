// CINLayer backward: dX_{k-1} and dx0 of one layer from ONE forward-sized product (csrc/cin_bwd.hip).
#pragma once
#include "common.hpp"

// M rows (= B * D) a multiple of 128, Hk a multiple of 32, Hp (= H_{k-1}) 64 or 128, F * Hp a multiple of 128, F a multiple of 4 up to what LDS holds
bool rn_cin_bwd_fused_supported(int64_t M, int Hk, int Hp, int F);
// dXp[M][Hp] += sum_f T * x0,  dx0t[M][F] += sum_h T * Xp,  T = dXk [M][Hk] x W [Hk][F * Hp]; dXp may be dx0t itself (Hp == F, Xp == x0t)
int rn_cin_bwd_fused(const float* dXk, const float* W, const float* x0t, const float* Xp, float* dXp, float* dx0t, int64_t M, int Hk, int Hp,
                     int F, hipStream_t st);
