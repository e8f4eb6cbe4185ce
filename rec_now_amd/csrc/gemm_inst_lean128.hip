// Lean (interior-only, compile-time operand kinds) instantiations of the 128x128 tile family.
#include "gemm_kernel.hpp"

int rn_gemm_launch_lean128(const GemmK& k, bool a_kc, bool b_kc, int bk, int a2k, int b2k, dim3 grid, hipStream_t st) {
    if (a_kc == true && b_kc == false && a2k == 0 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, true, false, false, 0, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, true, false, false, 0, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == true && b_kc == false && a2k == 3 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, true, false, false, 3, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, true, false, false, 3, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == true && b_kc == true && a2k == 0 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, true, true, false, 0, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, true, true, false, 0, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == true && b_kc == true && a2k == 1 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, true, true, false, 1, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, true, true, false, 1, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == true && b_kc == true && a2k == 2 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, true, true, false, 2, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, true, true, false, 2, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == true && b_kc == true && a2k == 3 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, true, true, false, 3, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, true, true, false, 3, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == false && b_kc == false && a2k == 0 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, false, false, false, 0, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, false, false, false, 0, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == false && b_kc == false && a2k == 1 && b2k == 0) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, false, false, false, 1, 0>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, false, false, false, 1, 0>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == false && b_kc == false && a2k == 0 && b2k == 2) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, false, false, false, 0, 2>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, false, false, false, 0, 2>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (a_kc == false && b_kc == false && a2k == 0 && b2k == 3) {
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, false, false, false, 0, 3>(k, grid, st);
        else rn_gemm_launch_one<128, 128, 2, 2, 32, false, false, false, 0, 3>(k, grid, st);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    return RECNOW_EUNSUPPORTED;
}
