// Shared pieces of the split-precision (3 x bf16, six MFMA terms) products: gemm_split.hip (long-K) and gemm_shortk.hip (K <= 512).
#pragma once
#include "gemm_kernel.hpp"
#include "gemm.hpp"

typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));
typedef __bf16 bf16x2 __attribute__((__vector_size__(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SPL_BK 16
#define SPL_PLANE_H 132                                  // 16-byte units per k-half: 128 rows + 4 (the second half starts 64 B
                                                         //   into the 128-B bank row of the stores: 8-lane store groups of 4 rows x 2 halves tile it)
#define SPL_PLANE (2 * SPL_PLANE_H * 16)                 // bytes per piece plane
#define SPL_OPER (3 * SPL_PLANE)                         // bytes per operand and stage
#define SPL_STAGE (2 * SPL_OPER)
#define SPL_BX_OFF (2 * SPL_STAGE)                       // side-product weights: ring of 4 k-tiles x 16 k x 4 floats
#define SPL_BX_RING 4
#define SPL_LDS (SPL_BX_OFF + SPL_BX_RING * SPL_BK * 4 * 4)

// (u, v) -> three packed bf16 pairs (low half = piece of u, high half = piece of v)
__device__ __forceinline__ void spl_split2(float u, float v, unsigned& p1, unsigned& p2, unsigned& p3) {
    bf16x2 h = {(__bf16)u, (__bf16)v};
    p1 = __builtin_bit_cast(unsigned, h);
    f32x2 r = {u - __builtin_bit_cast(float, p1 << 16), v - __builtin_bit_cast(float, p1 & 0xffff0000u)};
    bf16x2 g = {(__bf16)r.x, (__bf16)r.y};
    p2 = __builtin_bit_cast(unsigned, g);
    r.x -= __builtin_bit_cast(float, p2 << 16);
    r.y -= __builtin_bit_cast(float, p2 & 0xffff0000u);
    bf16x2 f = {(__bf16)r.x, (__bf16)r.y};
    p3 = __builtin_bit_cast(unsigned, f);
}
// eight consecutive k of one row -> one 16-byte unit per piece
__device__ __forceinline__ void spl_split8(const float (&x)[8], u32x4 (&w)[3]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned p1, p2, p3;
        spl_split2(x[2 * e], x[2 * e + 1], p1, p2, p3);
        w[0][e] = p1; w[1][e] = p2; w[2][e] = p3;
    }
}


// B (K x N; [K][N] rows of ldb floats, or [N][K] when b_kc) -> planes[s][K/8][N] units of 8 bf16 (gemm_split.hip)
int rn_split_planes(const float* B, int64_t ldb, int b_kc, int K, int N, void* planes, hipStream_t st);
