// Split-precision variant of the lean 128x128 GEMM with a side product (the K = 1024 and K = B products of the DCN-v2 step):
// every fp32 operand element x is split, on its way into LDS, into three bf16 pieces x = x1 + x2 + x3 (round-to-nearest each,
// residual <= 2^-27 |x|), and a product a*b is formed as the six bf16 MFMA terms a_i b_j with i + j <= 4, accumulated in fp32
// (v_mfma_f32_32x32x16_bf16: products of bf16 pairs are exact in fp32).  Dropped terms: a2 b3 + a3 b2 + a3 b3 <= 2^-25 |a b|, i.e.
// below one fp32 rounding of the product; the accumulation itself is fp32, as in the exact kernels.  Six 8-pass bf16 MFMAs do the
// work of eight 16-pass fp32 MFMAs: 2.67x the fp32-MFMA rate.  OPT-IN (recnow_set_gemm_precision / RECNOW_GEMM_PRECISION=bf16x3):
// results are not bit-identical to the fp32 kernels; parity (1e-5 relative, north_star) is held by the same tests.
//
// Structure follows the sliced fp32 kernel (gemm_kernel.hpp): 256 threads = 2 x 2 waves of 64 x 64, k-tiles of 16 (= one MFMA
// k-step), two LDS stages, the registers hold k-tile t+1 while k-tile t is computed, its split + LDS writes and the loads of
// k-tile t+2 sit between the MFMA groups.  LDS image per operand and stage: 3 planes (pieces) x 2 k-halves x 136 units of 16 B
// (unit = 8 consecutive k of one row: exactly what a lane feeds to the MFMA; rows padded by one unit per 16 so that the writes of
// the [k][row]-contiguous loaders spread over the banks).  Side product (sp_r <= 4 extra columns): fp32 FMAs on the operand
// registers at LDS-write time, reduced across the threads that share a row at the end (no fp32 image in LDS to read it from).
#include "gemm_kernel.hpp"
#include "prof.hpp"

typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));
typedef __bf16 bf16x2 __attribute__((__vector_size__(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define SPL_BK 16
#define SPL_PLANE_H 136                                  // 16-byte units per k-half: 128 rows + 1 pad per 16 rows
#define SPL_PLANE (2 * SPL_PLANE_H * 16)                 // bytes per piece plane
#define SPL_OPER (3 * SPL_PLANE)                         // bytes per operand and stage
#define SPL_STAGE (2 * SPL_OPER)
#define SPL_BX_OFF (2 * SPL_STAGE)                       // side-product weights: ring of 3 k-tiles x 16 k x 4 floats
#define SPL_LDS (SPL_BX_OFF + 3 * SPL_BK * 4 * 4)

__device__ __forceinline__ int spl_pos(int r) { return r + (r >> 4); }

// (u, v) -> three packed bf16 pairs (low half = piece of u, high half = piece of v)
__device__ __forceinline__ void spl_split2(float u, float v, unsigned& p1, unsigned& p2, unsigned& p3) {
    bf16x2 h = {(__bf16)u, (__bf16)v};
    p1 = __builtin_bit_cast(unsigned, h);
    float ru = u - __builtin_bit_cast(float, p1 << 16), rv = v - __builtin_bit_cast(float, p1 & 0xffff0000u);
    bf16x2 g = {(__bf16)ru, (__bf16)rv};
    p2 = __builtin_bit_cast(unsigned, g);
    ru -= __builtin_bit_cast(float, p2 << 16);
    rv -= __builtin_bit_cast(float, p2 & 0xffff0000u);
    bf16x2 f = {(__bf16)ru, (__bf16)rv};
    p3 = __builtin_bit_cast(unsigned, f);
}

// One operand's staging registers: two float4 per thread and k-tile.
//   KC  ([row][k], k contiguous): slot i = rows (tid >> 2) + 64 i, k = 4 (tid & 3) .. +3
//   !KC ([k][row], row contiguous): slot i = k 2 (tid >> 5) + i, rows 4 (tid & 31) .. +3   (the two slots pair up along k)
template <bool KC, int K2>
struct SplTile {
    f32x4 v[2], y[2];
    __device__ __forceinline__ void issue(const float* __restrict__ p, const float* __restrict__ p2, int64_t ld, int r0, int k0) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t off = KC ? (int64_t)(r0 + (t >> 2) + 64 * i) * ld + k0 + 4 * (t & 3)
                                   : (int64_t)(k0 + 2 * (t >> 5) + i) * ld + r0 + 4 * (t & 31);
            v[i] = *reinterpret_cast<const f32x4*>(p + off);
            if (K2 != RECNOW_OPMODE_NONE) y[i] = *reinterpret_cast<const f32x4*>(p2 + off);
        }
    }
    __device__ __forceinline__ void combine(int act) {
        if (K2 != RECNOW_OPMODE_NONE) {
            v[0] = gemm_combine(v[0], y[0], K2, act);
            v[1] = gemm_combine(v[1], y[1], K2, act);
        }
    }
    // split and write to the three planes at `S` (byte address of the operand's stage image)
    __device__ __forceinline__ void store(char* __restrict__ S) const {
        const int t = threadIdx.x;
        if (KC) {
            const int k4 = 4 * (t & 3);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                char* d = S + (((k4 >> 3) * SPL_PLANE_H + spl_pos((t >> 2) + 64 * i)) * 16 + ((k4 >> 2) & 1) * 8);
                unsigned a1, a2, a3, b1, b2, b3;
                spl_split2(v[i].x, v[i].y, a1, a2, a3);
                spl_split2(v[i].z, v[i].w, b1, b2, b3);
                u32x2 w;
                w.x = a1; w.y = b1;
                *reinterpret_cast<u32x2*>(d) = w;
                w.x = a2; w.y = b2;
                *reinterpret_cast<u32x2*>(d + SPL_PLANE) = w;
                w.x = a3; w.y = b3;
                *reinterpret_cast<u32x2*>(d + 2 * SPL_PLANE) = w;
            }
        } else {
            const int k = 2 * (t >> 5), r4 = 4 * (t & 31);
            char* d = S + ((k >> 3) * SPL_PLANE_H * 16 + (k & 7) * 2);
            const float lo[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, hi[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned p1, p2, p3;
                spl_split2(lo[e], hi[e], p1, p2, p3);
                char* de = d + spl_pos(r4 + e) * 16;
                *reinterpret_cast<unsigned*>(de) = p1;
                *reinterpret_cast<unsigned*>(de + SPL_PLANE) = p2;
                *reinterpret_cast<unsigned*>(de + 2 * SPL_PLANE) = p3;
            }
        }
    }
};

template <bool A_KC, bool B_KC, int A2K, bool SP>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
k_gemm_split(const GemmK p) {
    extern __shared__ __attribute__((aligned(16))) char spl_smem[];
    float* const Bxs = reinterpret_cast<float*>(spl_smem + SPL_BX_OFF);
    int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    if (p.xcd_remap == 1) {       // as k_gemm: the row tiles of one k-slab become consecutive workgroups of one XCD
        const int gx = gridDim.x, lin = bx + gx * z, xcd = lin & 7, i = lin >> 3;
        z = xcd * ((int)gridDim.z >> 3) + i / gx;
        bx = i % gx;
    }
    const int bidx = z / p.splitk, ks = z % p.splitk;
    const int k_begin = ks * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int m0 = bx * 128, n0 = by * 128;
    const float* Ab = p.A + (int64_t)bidx * p.sA;
    const float* A2b = p.A2 ? p.A2 + (int64_t)bidx * p.sA : nullptr;
    const float* Bb = p.B + (int64_t)bidx * p.sB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = (k_end - k_begin) / SPL_BK;
    const bool sp_on = SP && by == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // side product accumulators: KC -> spacc[i] = the 4 columns of row (tid >> 2) + 64 i, partial over this thread's 4 k;
    //                            !KC -> spacc[e] = the 4 columns of row 4 (tid & 31) + e, partial over this thread's k pairs
    f32x4 spacc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) spacc[e] = mk4(0.f, 0.f, 0.f, 0.f);
    float bxr[4] = {0.f, 0.f, 0.f, 0.f};

    SplTile<A_KC, A2K> ta;
    SplTile<B_KC, RECNOW_OPMODE_NONE> tb;
    auto load_bx = [&](int tile) {      // threads < 16: the side-product weights of k-tile `tile` (one k each)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bxr[r] = r < p.sp_r ? p.bx[(int64_t)(k_begin + tile * SPL_BK + threadIdx.x) * p.bx_ks + r * p.bx_rs] : 0.f;
    };
    auto store_bx = [&](int tile) { *reinterpret_cast<f32x4*>(Bxs + (tile % 3) * SPL_BK * 4 + threadIdx.x * 4) = mk4(bxr[0], bxr[1], bxr[2], bxr[3]); };
    // side product of the A registers with the weights of k-tile `tile` (scaled by w: 0 for a surplus commit)
    auto sp_fma = [&](int tile, float w) {
        const float* bt = Bxs + (tile % 3) * SPL_BK * 4;
        const int t = threadIdx.x;
        if (A_KC) {
            const int k4 = 4 * (t & 3);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bt + k4 * 4) * w, b1 = *reinterpret_cast<const f32x4*>(bt + k4 * 4 + 4) * w,
                        b2 = *reinterpret_cast<const f32x4*>(bt + k4 * 4 + 8) * w, b3 = *reinterpret_cast<const f32x4*>(bt + k4 * 4 + 12) * w;
#pragma unroll
            for (int i = 0; i < 2; ++i) spacc[i] += ta.v[i].x * b0 + ta.v[i].y * b1 + ta.v[i].z * b2 + ta.v[i].w * b3;
        } else {
            const int k = 2 * (t >> 5);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bt + k * 4) * w, b1 = *reinterpret_cast<const f32x4*>(bt + k * 4 + 4) * w;
            spacc[0] += ta.v[0].x * b0 + ta.v[1].x * b1;
            spacc[1] += ta.v[0].y * b0 + ta.v[1].y * b1;
            spacc[2] += ta.v[0].z * b0 + ta.v[1].z * b1;
            spacc[3] += ta.v[0].w * b0 + ta.v[1].w * b1;
        }
    };

    if (ntile > 0) {
        ta.issue(Ab, A2b, p.lda, m0, k_begin);
        tb.issue(Bb, nullptr, p.ldb, n0, k_begin);
        if (SP && threadIdx.x < SPL_BK) {
            load_bx(0);
            store_bx(0);
            load_bx(min(1, ntile - 1));
            store_bx(1);
            load_bx(min(2, ntile - 1));
        }
        __syncthreads();
        ta.combine(p.a_act);
        if (sp_on) sp_fma(0, 1.f);
        ta.store(spl_smem);
        tb.store(spl_smem + SPL_OPER);
        const int k1 = k_begin + min(1, ntile - 1) * SPL_BK;
        ta.issue(Ab, A2b, p.lda, m0, k1);
        tb.issue(Bb, nullptr, p.ldb, n0, k1);
    }
    __syncthreads();

    // fragment addresses: lane (row l & 31 of the 32-row MFMA tile, k-half l >> 5) reads one 16-byte unit per piece
    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = ((lane >> 5) * SPL_PLANE_H + spl_pos(wm * 64 + i * 32 + (lane & 31))) * 16;
        b_off[i] = SPL_OPER + ((lane >> 5) * SPL_PLANE_H + spl_pos(wn * 64 + i * 32 + (lane & 31))) * 16;
    }
    for (int t = 0; t < ntile; ++t) {
        const int cur = t & 1;
        const char* S = spl_smem + cur * SPL_STAGE;
        char* Sn = spl_smem + (cur ^ 1) * SPL_STAGE;
        const int k2 = k_begin + min(t + 2, ntile - 1) * SPL_BK;
        const float spw = t + 1 < ntile ? 1.f : 0.f;           // the last iteration's commit is surplus (nobody reads it)
        if (SP && threadIdx.x < SPL_BK) {                      // weights of k-tile t+2 to the ring, t+3 requested
            store_bx(t + 2);
            load_bx(min(t + 3, ntile - 1));
        }
        bf16x8 af[3][2], bf[3][2];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[s][i] = *reinterpret_cast<const bf16x8*>(S + s * SPL_PLANE + a_off[i]);
                bf[s][i] = *reinterpret_cast<const bf16x8*>(S + s * SPL_PLANE + b_off[i]);
            }
        __builtin_amdgcn_sched_barrier(0);
        // the six terms, smallest first; the A slice after the first eight MFMAs, the B slice after the next eight
#define SPL_TERM(SA, SB)                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SA][i], bf[SB][j], acc[i][j], 0, 0, 0);
        SPL_TERM(2, 0)
        SPL_TERM(0, 2)
        ta.combine(p.a_act);
        if (sp_on) sp_fma(t + 1, spw);
        ta.store(Sn);
        ta.issue(Ab, A2b, p.lda, m0, k2);
        __builtin_amdgcn_sched_barrier(0);
        SPL_TERM(1, 1)
        SPL_TERM(1, 0)
        tb.store(Sn + SPL_OPER);
        tb.issue(Bb, nullptr, p.ldb, n0, k2);
        __builtin_amdgcn_sched_barrier(0);
        SPL_TERM(0, 1)
        SPL_TERM(0, 0)
#undef SPL_TERM
        __syncthreads();
    }

    float* smem = reinterpret_cast<float*>(spl_smem);
    if (SP) {
        if (A_KC) {
            // the four threads of a row (adjacent lanes) hold its four k-chunks
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 s = spacc[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = s[c];
                    x += __shfl_xor(x, 1);
                    x += __shfl_xor(x, 2);
                    s[c] = x;
                }
                if (sp_on && (threadIdx.x & 3) == 0) {
                    const int m = m0 + (threadIdx.x >> 2) + 64 * i;
                    for (int r = 0; r < p.sp_r; ++r) {
                        if (p.splitk > 1) p.partial[((int64_t)z * p.M + m) * p.npart + p.N + r] = s[r];
                        else p.cx[(int64_t)m * p.cx_ms + r * p.cx_rs] = s[r];
                    }
                }
            }
        } else {
            // eight thread groups (tid >> 5) hold the k-pairs of the same four rows: fixed-order sum through LDS
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<f32x4*>(smem + (((threadIdx.x >> 5) * 128) + 4 * (threadIdx.x & 31) + e) * 4) = spacc[e];
            __syncthreads();
            if (sp_on && threadIdx.x < 128) {
                f32x4 s = *reinterpret_cast<const f32x4*>(smem + threadIdx.x * 4);
#pragma unroll
                for (int g = 1; g < 8; ++g) s += *reinterpret_cast<const f32x4*>(smem + (g * 128 + threadIdx.x) * 4);
                const int m = m0 + threadIdx.x;
                for (int r = 0; r < p.sp_r; ++r) {
                    if (p.splitk > 1) p.partial[((int64_t)z * p.M + m) * p.npart + p.N + r] = s[r];
                    else p.cx[(int64_t)m * p.cx_ms + r * p.cx_rs] = s[r];
                }
            }
            __syncthreads();
        }
    }
    gemm_lean_epilogue<2, 2, 0>(p, acc, smem, m0, n0, wm, wn, lane, wave, z, bidx);
}

// Launcher: the (layout, operand kind) combinations of the DCN-v2 step.  RECNOW_EUNSUPPORTED -> the caller runs the fp32 kernel.
int rn_gemm_launch_split(const GemmK& k, bool a_kc, bool b_kc, int a2k, bool sp, dim3 grid, hipStream_t st) {
#define X(AKC, BKC, A2)                                                                                               \
    if (a_kc == AKC && b_kc == BKC && a2k == A2) {                                                                    \
        if (sp) hipLaunchKernelGGL((k_gemm_split<AKC, BKC, A2, true>), grid, GEMM_THREADS, SPL_LDS, st, k);            \
        else hipLaunchKernelGGL((k_gemm_split<AKC, BKC, A2, false>), grid, GEMM_THREADS, SPL_LDS, st, k);              \
        RN_LAUNCH_CHECK();                                                                                            \
        return RECNOW_OK;                                                                                             \
    }
    X(true, false, 0)      // GEMM1:  x_l U
    X(true, true, 1)       // dT2g:   (x*g) W^T
    X(true, true, 0)       // dT2g of the top layer under a fused scoring head
    X(false, false, 0)     // dU:     x_l^T dA
    X(false, false, 1)     // dW^T:   (x*g)^T T2g
#undef X
    return RECNOW_EUNSUPPORTED;
}
