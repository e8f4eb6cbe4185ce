// Lean (interior-only) instantiations of the 256x64 tile family: the per-expert (B,S)x(S,S) products of DCN-v2.
#include "gemm_kernel.hpp"

int rn_gemm_launch_lean64(const GemmK& k, bool a_kc, bool b_kc, int a2k, int b2k, dim3 grid, hipStream_t st) {
    if (a2k != 0 || b2k != 0 || !a_kc) return RECNOW_EUNSUPPORTED;
    if (b_kc) rn_gemm_launch_one<256, 64, 4, 1, 32, true, true, false, 0, 0>(k, grid, st);
    else rn_gemm_launch_one<256, 64, 4, 1, 32, true, false, false, 0, 0>(k, grid, st);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
