// General (edge-capable, run-time operand kinds) instantiations, BK = 32, all four tile families.
#include "gemm_kernel.hpp"

template <int BM, int BN, int WM, int WN>
static void edge_family(const GemmK& k, bool a_kc, bool b_kc, dim3 grid, hipStream_t st) {
    if (a_kc && b_kc) rn_gemm_launch_one<BM, BN, WM, WN, 32, true, true, true, -1, -1>(k, grid, st);
    else if (a_kc && !b_kc) rn_gemm_launch_one<BM, BN, WM, WN, 32, true, false, true, -1, -1>(k, grid, st);
    else if (!a_kc && b_kc) rn_gemm_launch_one<BM, BN, WM, WN, 32, false, true, true, -1, -1>(k, grid, st);
    else rn_gemm_launch_one<BM, BN, WM, WN, 32, false, false, true, -1, -1>(k, grid, st);
}

int rn_gemm_launch_edge(const GemmK& k, int bm, int bn, bool a_kc, bool b_kc, dim3 grid, hipStream_t st) {
    if (bm == 256 && bn == 32) edge_family<256, 32, 4, 1>(k, a_kc, b_kc, grid, st);
    else if (bm == 256 && bn == 64) edge_family<256, 64, 4, 1>(k, a_kc, b_kc, grid, st);
    else if (bn == 160) edge_family<128, 160, 4, 1>(k, a_kc, b_kc, grid, st);
    else edge_family<128, 128, 2, 2>(k, a_kc, b_kc, grid, st);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
