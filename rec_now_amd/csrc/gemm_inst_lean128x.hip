// Lean 128x128 kernels with a VALU side product (XF = 1) or a rank-R epilogue update (XF = 2): the exact-128
// formulation of DCN-v2 (dcnmix.hip).
#include "gemm_kernel.hpp"

int rn_gemm_launch_lean128x(const GemmK& k, bool a_kc, bool b_kc, int bk, int a2k, int b2k, int xf, dim3 grid, hipStream_t st) {
#define X(AKC, BKC, A2, B2, XFV)                                                                                     \
    if (a_kc == AKC && b_kc == BKC && a2k == A2 && b2k == B2 && xf == XFV) {                                         \
        if (bk == 16) rn_gemm_launch_one<128, 128, 2, 2, 16, AKC, BKC, false, A2, B2, XFV>(k, grid, st);             \
        else rn_gemm_launch_one<128, 128, 2, 2, 32, AKC, BKC, false, A2, B2, XFV>(k, grid, st);                      \
        RN_LAUNCH_CHECK();                                                                                           \
        return RECNOW_OK;                                                                                            \
    }
    X(true, false, 0, 0, 1)     // GEMM1:   x_l U           + gate logits as side product
    X(true, false, 0, 0, 9)     //          ... XF | 8: at most two side columns (two experts), two-wide side product
    X(true, true, 1, 0, 9)
    X(true, true, 0, 0, 9)
    X(false, false, 0, 0, 9)
    X(false, false, 1, 0, 9)
    X(false, false, 0, 0, 41)   // XF | 32: both operands staged by LDS-DMA (RECNOW_GEMM_GLDS=1: the A/B of DESIGN 5k)
    X(false, true, 0, 0, 25)    // GEMM1 transposed ([U | K]^T x_l^T) with the sub-space forward in its epilogue (XF | 16)
    X(false, true, 0, 1, 25)    //          ... x_l = x0 * O_{l-1} formed in the operand load (dcnmix.hip mix_xless)
    X(true, false, 1, 0, 9)     // GEMM1 (not transposed) of a layer l > 0 likewise
    X(true, false, 1, 0, 1)     //          ... four-wide form (k-tiles of 16: D <= 256)
    X(true, false, 0, 0, 2)     // GEMM3:   T2g W           + gate-weighted bias as rank-2 update
    X(true, true, 1, 0, 1)      // dT2g:    (x*g) W^T       + bias columns as side product
    X(true, true, 1, 0, 5)      //          ... and dx = g * O written from the A stream (top layer)
    X(true, true, 0, 0, 1)      // dT2g of the top layer under a fused scoring head: x (W * w_head)^T, rows scaled by dscore later
    X(true, true, 0, 0, 2)      // dxl:     dA U^T          + dlogits K^T as rank-2 update
    X(false, false, 0, 0, 1)    // dU:      x_l^T dA        + dgate as side product
    X(false, false, 1, 0, 1)    // dW^T:    (x*g)^T T2g     + dbias as side product
#undef X
    return RECNOW_EUNSUPPORTED;
}
