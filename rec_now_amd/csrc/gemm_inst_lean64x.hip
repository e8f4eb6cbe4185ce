// Lean 64x128 kernels with the two-wide VALU side product (XF = 9): the long-K products of the DCN-v2 step at the row counts of the metric's
// 4- and 8-GPU shards (B = 65 536 split over the ranks leaves 8192 rows per GPU: 64 output tiles of 128 x 128 fill a quarter of the chip's 512
// workgroup slots; with 64-row tiles the same split over K gives every CU two workgroups that cover each other's prologue and epilogue).
// Wave tile 32 x 64 (2 x 2 waves, 1 x 2 MFMA tiles): 1.5 LDS fragment reads per MFMA instead of 1 -- only chosen where the grid would
// otherwise be short (rn_gemm_impl).
#include "gemm_kernel.hpp"

int rn_gemm_launch_lean64x(const GemmK& k, bool a_kc, bool b_kc, int a2k, int b2k, int xf, dim3 grid, hipStream_t st) {
#define X(AKC, BKC, A2, B2, XFV)                                                                                     \
    if (a_kc == AKC && b_kc == BKC && a2k == A2 && b2k == B2 && xf == XFV) {                                         \
        rn_gemm_launch_one<64, 128, 2, 2, 32, AKC, BKC, false, A2, B2, XFV>(k, grid, st);                            \
        RN_LAUNCH_CHECK();                                                                                           \
        return RECNOW_OK;                                                                                            \
    }
    X(true, false, 0, 0, 9)     // GEMM1:  x_l [U | K]
    X(true, false, 1, 0, 9)     //         (x0 * O_{l-1}) [U | K]: x_l formed in the operand load (dcnmix.hip mix_xless)
    X(true, true, 1, 0, 9)      // dT2g:   (x * g) Wc2^T
    X(true, true, 0, 0, 9)      // dT2g of the top layer under the fused head
    X(false, false, 0, 0, 9)    // dU:     x_l^T dT1
    X(false, false, 1, 0, 9)    // dW^T:   (x * g)^T T2g
#undef X
    return RECNOW_EUNSUPPORTED;
}
