// Lean (interior-only) instantiations of the 256x64 tile family: (B,S)x(S,S)-like products and CIN's N = F product.
#include "gemm_kernel.hpp"

int rn_gemm_launch_lean64(const GemmK& k, bool a_kc, bool b_kc, int a2k, int b2k, dim3 grid, hipStream_t st) {
    if (a2k == RECNOW_OPMODE_OUTER && b2k == 0 && a_kc && !b_kc) {       // CIN: dx0t += (dX_k (x) X_{k-1}) W_k  (N = F <= 64)
        rn_gemm_launch_one<256, 64, 4, 1, 32, true, false, false, 3, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a2k != 0 || b2k != 0 || !a_kc) return RECNOW_EUNSUPPORTED;
    if (b_kc) rn_gemm_launch_one<256, 64, 4, 1, 32, true, true, false, 0, 0>(k, grid, st);
    else rn_gemm_launch_one<256, 64, 4, 1, 32, true, false, false, 0, 0>(k, grid, st);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
