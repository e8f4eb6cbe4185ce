// Stand-alone bench of the round-6 split-precision product kernel (k_gemm_s3) before it moved into csrc/gemm_split.hip:
//   C[m][n] = sum_k A'[m][k] W[k][n]   (N = 128, W pre-split into bf16 piece planes), Cx[m][r] = sum_k A'[m][k] bx[k][r] (r < 2)
// A' = A or A * A2, fp32 in HBM, split into three bf16 pieces on its way into LDS; six bf16 MFMA terms per product.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o split3_bench split3_bench.hip ;  run: ./split3_bench [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <algorithm>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));
typedef __bf16 bf16x2 __attribute__((__vector_size__(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

#define S3_STRIDE 132                         // 16-byte units per k-octet row of an operand plane (128 rows + 4: the second octet starts 64 B into the bank row)
#define S3_PLANE (2 * S3_STRIDE * 16)         // bytes per piece plane (two k-octets = one 16-deep k-tile)
#define S3_OPER (3 * S3_PLANE)
#define S3_STAGE (2 * S3_OPER)
#define S3_LDS (2 * S3_STAGE)
#define S3_BX_LDS (1024 * 2 * 4)        // MAP 0: the side weights of one k-chunk (<= 1024 k) behind the stages

__device__ __forceinline__ void s3_split2(float u, float v, unsigned& p1, unsigned& p2, unsigned& p3) {
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

// W (K x 128, row-major) -> planes[s][K/8][128] units of 8 bf16: unit (o, n) of piece s = piece s of W[8o .. 8o+7][n]
__global__ void k_split_planes(const float* __restrict__ W, int K, char* __restrict__ planes) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)(K / 8) * 128;
    if (u >= total) return;
    const int n = (int)(u % 128), o = (int)(u / 128);
    u32x4 w[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned p1, p2, p3;
        s3_split2(W[(int64_t)(8 * o + 2 * e) * 128 + n], W[(int64_t)(8 * o + 2 * e + 1) * 128 + n], p1, p2, p3);
        w[0][e] = p1; w[1][e] = p2; w[2][e] = p3;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(planes + s * total * 16 + u * 16) = w[s];
}

struct S3Args {
    const float* A;
    const float* A2;
    int64_t lda;
    const char* planes;         // [3][K/8][128] units
    int64_t plane_bytes;
    const float* bx;            // side weights bx[k * bx_ks + r]
    int64_t bx_ks;
    float* C;                   // [splitk][M][ldc]
    float* cx;                  // [splitk][M][2]
    int64_t ldc;
    int M, K, kchunk;
    long long* dbg;             // DBG & 8: per workgroup {cycles, 100 MHz ticks} of the k-loop
};

// AK 0: A is [row][k] (k contiguous), AK 1: A is [k][row].  MAP 0: thread -> (row = tid >> 1, octet = tid & 1); MAP 1: (row = tid & 127, octet = tid >> 7:
// wave-uniform octet, the side weights come through scalar loads).  DBG & 1: no split (raw bits to the planes: timing only), DBG & 2: no side product.
// RING: register sets of A in flight (k-tile t + 1 + RING is requested while k-tile t is computed); LB: waves per SIMD the launch bounds ask for.
// DBG & 4: no A loads in the loop (timing only: what the memory pipeline costs).
template <int AK, int MAP, bool A2MUL, int DBG, bool BXC = true, int RING = 2, int LB = 2>
__global__ void __launch_bounds__(256, LB) k_gemm_s3(const S3Args p) {
    static_assert(AK == 0 || MAP == 1, "[k][row] operands use the wave-uniform octet mapping");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int bx_ = blockIdx.x, z = blockIdx.z;
    if (gridDim.z > 1) {       // row tiles of one k-slab: consecutive workgroups of one XCD (they share the B panel)
        const int gx = gridDim.x, lin = bx_ + gx * z, xcd = lin & 7, i = lin >> 3;
        z = xcd * ((int)gridDim.z >> 3) + i / gx;
        bx_ = i % gx;
    }
    const int m0 = bx_ * 128, k_begin = z * p.kchunk, nt = p.kchunk / 16;
    const int a_row = MAP ? (tid & 127) : (tid >> 1);
    const int a_h = MAP ? __builtin_amdgcn_readfirstlane(tid >> 7) : (tid & 1);
    const int b_row = tid & 127, b_h = tid >> 7;
    const unsigned a_goff = AK == 0 ? (unsigned)(a_row * p.lda + 8 * a_h) : (unsigned)(8 * a_h * p.lda + a_row);
    const unsigned ldau = (unsigned)p.lda;
    const int a_soff = (a_h * S3_STRIDE + a_row) * 16;
    const int b_soff = S3_OPER + (b_h * S3_STRIDE + b_row) * 16;
    const int64_t bp_goff = ((int64_t)b_h * 128 + b_row) * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sp0 = 0.f, sp1 = 0.f;

    float va[RING][8], ya[A2MUL ? RING : 1][8];
    u32x4 bpl[3];
    // tile bases are block-uniform (SGPRs); the per-lane part is a 32-bit BYTE offset kept opaque so that the loads take the
    // `global_load v, voff, s[base]` form (no 64-bit vector address arithmetic in the loop)
    auto a_base = [&](int t) { return AK == 0 ? (int64_t)m0 * p.lda + k_begin + t * 16 : (int64_t)(k_begin + t * 16) * p.lda + m0; };
    auto clampt = [&](int t) { return min(t, nt - 1); };
    unsigned a_bo[AK == 0 ? 1 : 8];
    a_bo[0] = a_goff * 4u;
    if (AK == 1) {
#pragma unroll
        for (int e = 1; e < 8; ++e) a_bo[e] = (a_goff + (unsigned)e * ldau) * 4u;
    }
    auto a_issue = [&](float (&v)[8], float (&y)[8], int t) {
        const char* pa = reinterpret_cast<const char*>(p.A + a_base(t));
        const char* pa2 = reinterpret_cast<const char*>(p.A2 + a_base(t));
        if (AK == 0) {
            asm volatile("" : "+v"(a_bo[0]));
            const f32x4 a = *reinterpret_cast<const f32x4*>(pa + a_bo[0]), b = *reinterpret_cast<const f32x4*>(pa + a_bo[0] + 16);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if (A2MUL) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(pa2 + a_bo[0]), d = *reinterpret_cast<const f32x4*>(pa2 + a_bo[0] + 16);
                y[0] = c.x; y[1] = c.y; y[2] = c.z; y[3] = c.w; y[4] = d.x; y[5] = d.y; y[6] = d.z; y[7] = d.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                asm volatile("" : "+v"(a_bo[AK == 0 ? 0 : e]));
                v[e] = *reinterpret_cast<const float*>(pa + a_bo[AK == 0 ? 0 : e]);
                if (A2MUL) y[e] = *reinterpret_cast<const float*>(pa2 + a_bo[AK == 0 ? 0 : e]);
            }
        }
    };
    unsigned b_bo = (unsigned)bp_goff;
    auto b_issue = [&](int t) {
        const char* src = p.planes + (int64_t)((k_begin + t * 16) >> 3) * 128 * 16;
        asm volatile("" : "+v"(b_bo));
#pragma unroll
        for (int s = 0; s < 3; ++s) bpl[s] = *reinterpret_cast<const u32x4*>(src + s * p.plane_bytes + b_bo);
    };
    auto b_store = [&](char* S) {
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + b_soff + s * S3_PLANE) = bpl[s];
    };
    auto a_combine = [&](float (&v)[8], const float (&y)[8]) {
        if (A2MUL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= y[e];
        }
    };
    // side product of this thread's unit with the weights of k-tile t: from the copy of the chunk's weights in LDS (broadcast reads)
    const float* bxl = reinterpret_cast<const float*>(smem + S3_LDS);
    auto a_side = [&](const float (&v)[8], int t) {
        if (DBG & 2) return;
        const float* b = bxl + (t * 16 + 8 * a_h) * 2;
        f32x4 q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = *reinterpret_cast<const f32x4*>(b + 4 * e);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sp0 = fmaf(v[2 * e], q[e].x, sp0);
            sp1 = fmaf(v[2 * e], q[e].y, sp1);
            sp0 = fmaf(v[2 * e + 1], q[e].z, sp0);
            sp1 = fmaf(v[2 * e + 1], q[e].w, sp1);
        }
    };
    u32x4 wq[3];
    auto split_pairs = [&](const float (&v)[8], int e0) {
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e) {
            unsigned p1, p2, p3;
            if (DBG & 1) {
                p1 = __builtin_bit_cast(unsigned, v[2 * e]); p2 = __builtin_bit_cast(unsigned, v[2 * e + 1]); p3 = p1 ^ p2;
            } else {
                s3_split2(v[2 * e], v[2 * e + 1], p1, p2, p3);
            }
            wq[0][e] = p1; wq[1][e] = p2; wq[2][e] = p3;
        }
    };
    auto a_store = [&](char* S) {
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + a_soff + s * S3_PLANE) = wq[s];
    };

    if (!(DBG & 2)) {      // the chunk's side weights (kchunk x 2 floats) -> LDS
        for (int i = tid; i < p.kchunk; i += 256)
            *reinterpret_cast<f32x2*>(smem + S3_LDS + i * 8) = *reinterpret_cast<const f32x2*>(p.bx + (int64_t)(k_begin + i) * p.bx_ks);
        __syncthreads();
    }
    // ---- prologue: A of k-tiles 0 .. RING - 1 requested (slot = k-tile % RING), k-tile 0 -> stage 0, k-tile RING requested into its slot
    a_issue(va[0], ya[0], 0);
    b_issue(0);
#pragma unroll
    for (int u = 1; u < RING; ++u) a_issue(va[u], ya[A2MUL ? u : 0], clampt(u));
    a_combine(va[0], ya[0]);
    a_side(va[0], 0);
    split_pairs(va[0], 0);
    split_pairs(va[0], 2);
    a_store(smem);
    b_store(smem);
    a_issue(va[0], ya[0], clampt(RING));
    b_issue(clampt(1));
    __syncthreads();

    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = ((lane >> 5) * S3_STRIDE + wm * 64 + i * 32 + (lane & 31)) * 16;
        b_off[i] = S3_OPER + ((lane >> 5) * S3_STRIDE + wn * 64 + i * 32 + (lane & 31)) * 16;
    }

#define S3_TERM(SA, SB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SA][i], bf[SB][j], acc[i][j], 0, 0, 0);
    // one k-tile: t = its index; SLOT holds A of k-tile t + 1; STAGE: stage k-tile t + 1 (false: the last k-tile, compute only)
    auto ktile = [&](int t, auto slot_c, auto stage_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr bool STG = decltype(stage_c)::value;
        const char* S = smem + (t & 1) * S3_STAGE;
        char* Sn = smem + ((t & 1) ^ 1) * S3_STAGE;
        bf16x8 af[3][2], bf[3][2];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[s][i] = *reinterpret_cast<const bf16x8*>(S + s * S3_PLANE + a_off[i]);
                bf[s][i] = *reinterpret_cast<const bf16x8*>(S + s * S3_PLANE + b_off[i]);
            }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(0, 0)
        if (STG) {
            a_combine(va[SLOT], ya[A2MUL ? SLOT : 0]);
            split_pairs(va[SLOT], 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(0, 1)
        if (STG) {
            a_side(va[SLOT], t + 1);
            split_pairs(va[SLOT], 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(1, 0)
        if (STG) a_store(Sn);
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(1, 1)
        if (STG) {
            if (!(DBG & 4)) a_issue(va[SLOT], ya[A2MUL ? SLOT : 0], clampt(t + 1 + RING));
            b_store(Sn);
        }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(0, 2)
        if (STG) b_issue(clampt(t + 2));
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(2, 0)
        __syncthreads();
    };
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;
    // k-tile t + 1 sits in slot (t + 1) % RING; nt is a multiple of RING and of 2 (the caller's guarantee)
    int t = 0;
    for (; t + RING < nt; t += RING) {
        ktile(t, std::integral_constant<int, 1 % RING>(), BT());
        ktile(t + 1, std::integral_constant<int, 2 % RING>(), BT());
        if (RING == 4) {
            ktile(t + 2, std::integral_constant<int, 3 % RING>(), BT());
            ktile(t + 3, std::integral_constant<int, 0>(), BT());
        }
    }
    if (RING == 4) {
        ktile(t, std::integral_constant<int, 1 % RING>(), BT());
        ktile(t + 1, std::integral_constant<int, 2 % RING>(), BT());
        ktile(t + 2, std::integral_constant<int, 3 % RING>(), BT());
        ktile(t + 3, std::integral_constant<int, 0>(), BF());
    } else {
        ktile(t, std::integral_constant<int, 1 % RING>(), BT());
        ktile(t + 1, std::integral_constant<int, 0>(), BF());
    }
#undef S3_TERM

    // side product: the two threads of a row (its two k-octets)
    float* cxz = p.cx + (int64_t)z * p.M * 2;
    if (!(DBG & 2)) {
        if (MAP == 0) {
            sp0 += __shfl_xor(sp0, 1);
            sp1 += __shfl_xor(sp1, 1);
            if ((tid & 1) == 0) *reinterpret_cast<f32x2*>(cxz + (int64_t)(m0 + a_row) * 2) = f32x2{sp0, sp1};
        } else {
            float* sm = reinterpret_cast<float*>(smem);
            if (tid >= 128) *reinterpret_cast<f32x2*>(sm + (tid - 128) * 2) = f32x2{sp0, sp1};
            __syncthreads();
            if (tid < 128) {
                const f32x2 o = *reinterpret_cast<const f32x2*>(sm + tid * 2);
                *reinterpret_cast<f32x2*>(cxz + (int64_t)(m0 + tid) * 2) = f32x2{sp0 + o.x, sp1 + o.y};
            }
        }
    }
    // accumulators as they lie: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* Cz = p.C + (int64_t)z * p.M * p.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* cp = Cz + (int64_t)(m0 + wm * 64 + i * 32 + 4 * (lane >> 5)) * p.ldc + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][j][r];
        }
}


// ---------------------------------------------------------------------------------------------------------------------
// Version 2: THREE LDS stages, the fragments of k-tile t + 1 are read while k-tile t's MFMAs run (into the registers its retired
// fragments leave: term order (0,0) (0,1) (0,2) | (1,0) (1,1) | (2,0)), so that the first MFMA behind a barrier has its operands.
// Mapping: thread -> (row = tid & 127, octet = tid >> 7) for both operands (no padding: 64 lanes write 1 KiB contiguous).
#define S3P_PLANE (2 * 128 * 16)
#define S3P_OPER (3 * S3P_PLANE)
#define S3P_STAGE (2 * S3P_OPER)
#define S3P_LDS (3 * S3P_STAGE)
template <int AK, bool A2MUL, int DBG>
__global__ void __launch_bounds__(256, 2) k_gemm_s3p(const S3Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int bx_ = blockIdx.x, z = blockIdx.z;
    if (gridDim.z > 1) {
        const int gx = gridDim.x, lin = bx_ + gx * z, xcd = lin & 7, i = lin >> 3;
        z = xcd * ((int)gridDim.z >> 3) + i / gx;
        bx_ = i % gx;
    }
    const int m0 = bx_ * 128, k_begin = z * p.kchunk, nt = p.kchunk / 16;
    const int row = tid & 127, oct = __builtin_amdgcn_readfirstlane(tid >> 7);
    const unsigned ldau = (unsigned)p.lda;
    const unsigned a_goff = AK == 0 ? (unsigned)(row * ldau + 8 * oct) : (unsigned)(8 * oct * ldau + row);
    const int a_soff = (oct * 128 + row) * 16, b_soff = S3P_OPER + a_soff;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float sp0 = 0.f, sp1 = 0.f;
    float va[2][8], ya[A2MUL ? 2 : 1][8];
    u32x4 bpl[3], wq[3];

    auto a_base = [&](int t) { return AK == 0 ? (int64_t)m0 * p.lda + k_begin + t * 16 : (int64_t)(k_begin + t * 16) * p.lda + m0; };
    auto clampt = [&](int t) { return min(t, nt - 1); };
    unsigned a_bo[AK == 0 ? 1 : 8];
    a_bo[0] = a_goff * 4u;
    if (AK == 1) {
#pragma unroll
        for (int e = 1; e < 8; ++e) a_bo[e] = (a_goff + (unsigned)e * ldau) * 4u;
    }
    auto a_issue = [&](float (&v)[8], float (&y)[8], int t) {
        const char* pa = reinterpret_cast<const char*>(p.A + a_base(t));
        const char* pa2 = reinterpret_cast<const char*>(p.A2 + a_base(t));
        if (AK == 0) {
            asm volatile("" : "+v"(a_bo[0]));
            const f32x4 a = *reinterpret_cast<const f32x4*>(pa + a_bo[0]), b = *reinterpret_cast<const f32x4*>(pa + a_bo[0] + 16);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if (A2MUL) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(pa2 + a_bo[0]), d = *reinterpret_cast<const f32x4*>(pa2 + a_bo[0] + 16);
                y[0] = c.x; y[1] = c.y; y[2] = c.z; y[3] = c.w; y[4] = d.x; y[5] = d.y; y[6] = d.z; y[7] = d.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                asm volatile("" : "+v"(a_bo[AK == 0 ? 0 : e]));
                v[e] = *reinterpret_cast<const float*>(pa + a_bo[AK == 0 ? 0 : e]);
                if (A2MUL) y[e] = *reinterpret_cast<const float*>(pa2 + a_bo[AK == 0 ? 0 : e]);
            }
        }
    };
    unsigned b_bo = (unsigned)a_soff;      // the planes' unit (octet, row) of a k-tile sits at the same offset as in a stage
    auto b_issue = [&](int t) {
        const char* src = p.planes + (int64_t)((k_begin + t * 16) >> 3) * 128 * 16;
        asm volatile("" : "+v"(b_bo));
#pragma unroll
        for (int s = 0; s < 3; ++s) bpl[s] = *reinterpret_cast<const u32x4*>(src + s * p.plane_bytes + b_bo);
    };
    auto b_store = [&](char* S) {
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + b_soff + s * S3P_PLANE) = bpl[s];
    };
    auto a_combine = [&](float (&v)[8], const float (&y)[8]) {
        if (A2MUL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= y[e];
        }
    };
    const float* bxl = reinterpret_cast<const float*>(smem + S3P_LDS);
    auto a_side = [&](const float (&v)[8], int t) {
        if (DBG & 2) return;
        const float* b = bxl + (t * 16 + 8 * oct) * 2;
        f32x4 q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = *reinterpret_cast<const f32x4*>(b + 4 * e);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sp0 = fmaf(v[2 * e], q[e].x, sp0);
            sp1 = fmaf(v[2 * e], q[e].y, sp1);
            sp0 = fmaf(v[2 * e + 1], q[e].z, sp0);
            sp1 = fmaf(v[2 * e + 1], q[e].w, sp1);
        }
    };
    auto split_pairs = [&](const float (&v)[8], int e0) {
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e) {
            unsigned p1, p2, p3;
            if (DBG & 1) {
                p1 = __builtin_bit_cast(unsigned, v[2 * e]); p2 = __builtin_bit_cast(unsigned, v[2 * e + 1]); p3 = p1 ^ p2;
            } else {
                s3_split2(v[2 * e], v[2 * e + 1], p1, p2, p3);
            }
            wq[0][e] = p1; wq[1][e] = p2; wq[2][e] = p3;
        }
    };
    auto a_store = [&](char* S) {
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + a_soff + s * S3P_PLANE) = wq[s];
    };

    if (!(DBG & 2)) {
        for (int i = tid; i < p.kchunk; i += 256)
            *reinterpret_cast<f32x2*>(smem + S3P_LDS + i * 8) = *reinterpret_cast<const f32x2*>(p.bx + (int64_t)(k_begin + i) * p.bx_ks);
    }
    // ---- prologue: k-tiles 0 and 1 -> stages 0 and 1; A of k-tiles 2 (set 0) and 3 (set 1), B of k-tile 2 requested
    a_issue(va[0], ya[0], 0);
    b_issue(0);
    a_issue(va[1], ya[A2MUL ? 1 : 0], 1);
    __syncthreads();                       // (the side weights)
    a_combine(va[0], ya[0]);
    a_side(va[0], 0);
    split_pairs(va[0], 0);
    split_pairs(va[0], 2);
    a_store(smem);
    b_store(smem);
    b_issue(1);
    a_issue(va[0], ya[0], clampt(2));
    a_combine(va[1], ya[A2MUL ? 1 : 0]);
    a_side(va[1], 1);
    split_pairs(va[1], 0);
    split_pairs(va[1], 2);
    a_store(smem + S3P_STAGE);
    b_store(smem + S3P_STAGE);
    b_issue(clampt(2));
    a_issue(va[1], ya[A2MUL ? 1 : 0], clampt(3));
    __syncthreads();

    // fragment addresses inside a stage: lane (row l & 31 of the 32-row block, octet l >> 5)
    const int fa = ((lane >> 5) * 128 + wm * 64 + (lane & 31)) * 16;
    const int fb = S3P_OPER + ((lane >> 5) * 128 + wn * 64 + (lane & 31)) * 16;
    bf16x8 fA[2][3][2], fB[2][3][2];      // [set][piece][32-row block]
    auto rd = [&](bf16x8& dst, const char* S, int off) { dst = *reinterpret_cast<const bf16x8*>(S + off); };
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rd(fA[0][s][i], smem, fa + s * S3P_PLANE + i * 512);
            rd(fB[0][s][i], smem, fb + s * S3P_PLANE + i * 512);
        }

    int st_cur = 0;                        // stage of k-tile t
#define S3_TERM(SA, SB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA[CUR][SA][i], fB[CUR][SB][j], acc[i][j], 0, 0, 0);
    // one k-tile: t = its index (fragments in set CUR = t & 1); STG: stage k-tile t + 2 (A in register set CUR); NXT: read the fragments of k-tile t + 1
    auto ktile = [&](int t, auto cur_c, auto stage_c, auto next_c) {
        constexpr int CUR = decltype(cur_c)::value, NX = CUR ^ 1;
        constexpr bool STG = decltype(stage_c)::value, NXT = decltype(next_c)::value;
        const int st_nxt = st_cur == 2 ? 0 : st_cur + 1, st_wr = st_nxt == 2 ? 0 : st_nxt + 1;
        const char* Sn = smem + st_nxt * S3P_STAGE;
        char* Sw = smem + st_wr * S3P_STAGE;
        S3_TERM(0, 0)
        if (STG) {
            a_combine(va[CUR], ya[A2MUL ? CUR : 0]);
            split_pairs(va[CUR], 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(0, 1)
        if (STG) {
            a_side(va[CUR], t + 2);
            split_pairs(va[CUR], 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(0, 2)
        if (NXT) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rd(fA[NX][0][i], Sn, fa + i * 512);
                rd(fB[NX][0][i], Sn, fb + i * 512);
            }
        }
        if (STG) a_store(Sw);
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(1, 0)
        if (STG) {
            if (!(DBG & 4)) a_issue(va[CUR], ya[A2MUL ? CUR : 0], clampt(t + 4));
            b_store(Sw);
        }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(1, 1)
        if (NXT) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rd(fB[NX][1][i], Sn, fb + S3P_PLANE + i * 512);
                rd(fB[NX][2][i], Sn, fb + 2 * S3P_PLANE + i * 512);
            }
        }
        if (STG) b_issue(clampt(t + 3));
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(2, 0)
        if (NXT) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                rd(fA[NX][1][i], Sn, fa + S3P_PLANE + i * 512);
                rd(fA[NX][2][i], Sn, fa + 2 * S3P_PLANE + i * 512);
            }
        }
        __syncthreads();
        st_cur = st_nxt;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;
    long long c0 = 0, r0 = 0;
    if (DBG & 8) {
        c0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    int t = 0;
    if (DBG & 16) {      // MFMAs alone on the first fragments: the matrix pipe at the clock the chip holds for this mix
        for (; t < nt; ++t) {
            constexpr int CUR = 0;
            S3_TERM(0, 0) S3_TERM(0, 1) S3_TERM(0, 2) S3_TERM(1, 0) S3_TERM(1, 1) S3_TERM(2, 0)
        }
    } else {
    for (; t + 2 < nt; t += 2) {
        ktile(t, I0(), BT(), BT());
        ktile(t + 1, I1(), BT(), BT());
    }
    ktile(t, I0(), BF(), BT());
    ktile(t + 1, I1(), BF(), BF());
    }
#undef S3_TERM
    if (DBG & 8) {
        const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            p.dbg[2 * (blockIdx.x + gridDim.x * blockIdx.z)] = c1 - c0;
            p.dbg[2 * (blockIdx.x + gridDim.x * blockIdx.z) + 1] = r1 - r0;
        }
    }

    float* cxz = p.cx + (int64_t)z * p.M * 2;
    if (!(DBG & 2)) {
        float* sm = reinterpret_cast<float*>(smem);
        if (tid >= 128) *reinterpret_cast<f32x2*>(sm + (tid - 128) * 2) = f32x2{sp0, sp1};
        __syncthreads();
        if (tid < 128) {
            const f32x2 o = *reinterpret_cast<const f32x2*>(sm + tid * 2);
            *reinterpret_cast<f32x2*>(cxz + (int64_t)(m0 + tid) * 2) = f32x2{sp0 + o.x, sp1 + o.y};
        }
    }
    float* Cz = p.C + (int64_t)z * p.M * p.ldc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* cp = Cz + (int64_t)(m0 + wm * 64 + i * 32 + 4 * (lane >> 5)) * p.ldc + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * p.ldc] = acc[i][j][r];
        }
}

template <int AK, bool A2MUL, int DBG>
static float runp(const S3Args& a, dim3 grid, int reps) {
    const int lds = S3P_LDS + S3_BX_LDS;
    CK(hipFuncSetAttribute((const void*)k_gemm_s3p<AK, A2MUL, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_gemm_s3p<AK, A2MUL, DBG>), grid, 256, lds, 0, a);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gemm_s3p<AK, A2MUL, DBG>), grid, 256, lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

// ---------------------------------------------------------------------------------------------------------------------
struct Case {
    const char* name;
    int ak, M, K, splitk;
    bool a2;
};

template <int AK, int MAP, bool A2MUL, int DBG, bool BXC = true, int RING = 2, int LB = 2>
static float run(const S3Args& a, dim3 grid, int reps) {
    CK(hipFuncSetAttribute((const void*)k_gemm_s3<AK, MAP, A2MUL, DBG, BXC, RING, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, S3_LDS + S3_BX_LDS));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_gemm_s3<AK, MAP, A2MUL, DBG, BXC, RING, LB>), grid, 256, S3_LDS + S3_BX_LDS, 0, a);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gemm_s3<AK, MAP, A2MUL, DBG, BXC, RING, LB>), grid, 256, S3_LDS + S3_BX_LDS, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    const int Bt = 65536, D = 1024;
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    // one buffer set serves both layouts: X (Bt x D) row-major is [row][k] for AK 0 (M = Bt, K = D) and [k][row] for AK 1 (K = Bt, M = D)
    std::vector<float> hX((size_t)Bt * D), hX2((size_t)Bt * D), hW((size_t)D * 128), hbx((size_t)Bt * 2);
    for (auto& v : hX) v = 0.3f * nd(rng);
    for (auto& v : hX2) v = nd(rng);
    for (auto& v : hW) v = 0.05f * nd(rng);
    for (auto& v : hbx) v = 0.05f * nd(rng);
    // AK 1: B operand = (Bt x 128) activations
    std::vector<float> hT((size_t)Bt * 128);
    for (auto& v : hT) v = 0.1f * nd(rng);
    float *dX, *dX2, *dW, *dbx, *dT, *dC, *dcx;
    char *dplW, *dplT;
    CK(hipMalloc(&dX, hX.size() * 4));
    CK(hipMalloc(&dX2, hX2.size() * 4));
    CK(hipMalloc(&dW, hW.size() * 4));
    CK(hipMalloc(&dbx, hbx.size() * 4));
    CK(hipMalloc(&dT, hT.size() * 4));
    CK(hipMalloc(&dC, (size_t)Bt * 128 * 4));           // AK 0: Bt x 128; AK 1: 64 slabs x 1024 x 128 = the same
    CK(hipMalloc(&dcx, (size_t)Bt * 2 * 4));
    CK(hipMalloc(&dplW, (size_t)D / 8 * 128 * 16 * 3));
    CK(hipMalloc(&dplT, (size_t)Bt / 8 * 128 * 16 * 3));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dX2, hX2.data(), hX2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbx, hbx.data(), hbx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dT, hT.data(), hT.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_split_planes, D / 8 * 128 / 256, 256, 0, 0, dW, D, dplW);
    hipLaunchKernelGGL(k_split_planes, Bt / 8 * 128 / 256, 256, 0, 0, dT, Bt, dplT);
    CK(hipDeviceSynchronize());

    long long* ddbg;
    CK(hipMalloc(&ddbg, 1024 * 16));
    S3Args g1;      // GEMM1 shape: (Bt x D) x (D x 128)
    g1.A = dX; g1.A2 = dX2; g1.lda = D; g1.planes = dplW; g1.plane_bytes = (int64_t)D / 8 * 128 * 16; g1.bx = dbx; g1.bx_ks = 2;
    g1.dbg = ddbg; g1.C = dC; g1.cx = dcx; g1.ldc = 128; g1.M = Bt; g1.K = D; g1.kchunk = D;
    S3Args gu = g1;  // dU shape: X^T (D x Bt) x (Bt x 128), 64 k-slabs
    gu.planes = dplT; gu.plane_bytes = (int64_t)Bt / 8 * 128 * 16; gu.M = D; gu.K = Bt; gu.kchunk = Bt / 64;
    const dim3 grid1(Bt / 128, 1, 1), gridu(D / 128, 1, 64);

    // ---- correctness: rows [0, 128) and the last 128 of GEMM1 (both mappings, with and without A2); the dU shape: slab sums of 128 x 128 outputs
    std::vector<float> hC((size_t)Bt * 128), hcx((size_t)Bt * 2);
    auto check1 = [&](const char* name, bool a2) {
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hcx.data(), dcx, hcx.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0, worstx = 0, scalex = 0;
        for (int blk = 0; blk < 2; ++blk)
            for (int mm = 0; mm < 128; ++mm) {
                const int m = blk ? Bt - 128 + mm : mm;
                for (int n = 0; n < 128; n += 7) {
                    double s = 0;
                    for (int k = 0; k < D; ++k) {
                        const float a = a2 ? hX[(size_t)m * D + k] * hX2[(size_t)m * D + k] : hX[(size_t)m * D + k];
                        s += (double)a * hW[(size_t)k * 128 + n];
                    }
                    worst = fmax(worst, fabs(s - hC[(size_t)m * 128 + n]));
                    scale = fmax(scale, fabs(s));
                }
                for (int r = 0; r < 2; ++r) {
                    double s = 0;
                    for (int k = 0; k < D; ++k) {
                        const float a = a2 ? hX[(size_t)m * D + k] * hX2[(size_t)m * D + k] : hX[(size_t)m * D + k];
                        s += (double)a * hbx[(size_t)k * 2 + r];
                    }
                    worstx = fmax(worstx, fabs(s - hcx[(size_t)m * 2 + r]));
                    scalex = fmax(scalex, fabs(s));
                }
            }
        printf("check %-28s C err %.3g / scale %.3g = %.3g   side err %.3g / %.3g = %.3g\n", name, worst, scale, worst / scale, worstx, scalex, worstx / scalex);
    };
    auto checku = [&](const char* name, bool a2) {
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hcx.data(), dcx, hcx.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0, worstx = 0, scalex = 0;
        for (int m = 0; m < D; m += 37)
            for (int n = 0; n < 128; n += 13) {
                double s = 0, got = 0;
                for (int k = 0; k < Bt; ++k) {
                    const float a = a2 ? hX[(size_t)k * D + m] * hX2[(size_t)k * D + m] : hX[(size_t)k * D + m];
                    s += (double)a * hT[(size_t)k * 128 + n];
                }
                for (int zz = 0; zz < 64; ++zz) got += hC[((size_t)zz * D + m) * 128 + n];
                worst = fmax(worst, fabs(s - got));
                scale = fmax(scale, fabs(s));
            }
        for (int m = 0; m < D; m += 37)
            for (int r = 0; r < 2; ++r) {
                double s = 0, got = 0;
                for (int k = 0; k < Bt; ++k) {
                    const float a = a2 ? hX[(size_t)k * D + m] * hX2[(size_t)k * D + m] : hX[(size_t)k * D + m];
                    s += (double)a * hbx[(size_t)k * 2 + r];
                }
                for (int zz = 0; zz < 64; ++zz) got += hcx[((size_t)zz * D + m) * 2 + r];
                worstx = fmax(worstx, fabs(s - got));
                scalex = fmax(scalex, fabs(s));
            }
        printf("check %-28s C err %.3g / scale %.3g = %.3g   side err %.3g / %.3g = %.3g\n", name, worst, scale, worst / scale, worstx, scalex, worstx / scalex);
    };
    run<0, 0, false, 0>(g1, grid1, 1); check1("GEMM1 map0", false);
    run<0, 1, false, 0>(g1, grid1, 1); check1("GEMM1 map1", false);
    run<0, 0, true, 0>(g1, grid1, 1); check1("dT2g (A*A2) map0", true);
    run<0, 1, true, 0>(g1, grid1, 1); check1("dT2g (A*A2) map1", true);
    run<0, 0, false, 0, true, 4>(g1, grid1, 1); check1("GEMM1 map0 ring4", false);
    run<0, 0, true, 0, true, 4>(g1, grid1, 1); check1("dT2g map0 ring4", true);
    run<0, 0, false, 0, true, 4, 3>(g1, grid1, 1); check1("GEMM1 map0 ring4 lb3", false);
    run<1, 1, false, 0, false, 4>(gu, gridu, 1); checku("dU ring4", false);
    runp<0, false, 0>(g1, grid1, 1); check1("v2 GEMM1", false);
    runp<0, true, 0>(g1, grid1, 1); check1("v2 dT2g", true);
    runp<1, false, 0>(gu, gridu, 1); checku("v2 dU", false);
    runp<1, true, 0>(gu, gridu, 1); checku("v2 dW", true);
    run<1, 1, false, 0, false>(gu, gridu, 1); checku("dU [k][row]", false);
    run<1, 1, true, 0, false>(gu, gridu, 1); checku("dW [k][row] (A*A2)", true);

    auto clock_of = [&](const char* name, float us) {
        std::vector<long long> h(1024);
        CK(hipMemcpy(h.data(), ddbg, 1024 * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz, cyc;
        for (int i = 0; i < 512; ++i) { ghz.push_back(h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i]); }
        std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
        printf("clock %-34s %7.1f us  in-kernel clock median %.3f GHz (min %.3f max %.3f), k-loop cycles median %.0f (MFMA alone = 98304 per SIMD for two workgroups)\n", name, us, ghz[256], ghz[0], ghz[511], cyc[256]);
    };
    // ---- timing: interleaved rounds
    const double gflop = 2.0 * Bt * D * 128 * 1e-9;
    for (int rd = 0; rd < rounds; ++rd) {
        struct { const char* n; float us; } r[] = {
            {"GEMM1 map0", run<0, 0, false, 0>(g1, grid1, reps)},
            {"GEMM1 map0 noload nosplit noside", run<0, 0, false, 7>(g1, grid1, reps)},
            {"v2 GEMM1", runp<0, false, 0>(g1, grid1, reps)},
            {"v2 GEMM1 nosplit", runp<0, false, 1>(g1, grid1, reps)},
            {"v2 GEMM1 noside", runp<0, false, 2>(g1, grid1, reps)},
            {"v2 GEMM1 noload", runp<0, false, 4>(g1, grid1, reps)},
            {"v2 GEMM1 noload nosplit noside", runp<0, false, 7>(g1, grid1, reps)},
            {"dT2g map0", run<0, 0, true, 0>(g1, grid1, reps)},
            {"v2 dT2g", runp<0, true, 0>(g1, grid1, reps)},
            {"v2 dT2g noload", runp<0, true, 4>(g1, grid1, reps)},
            {"dU", run<1, 1, false, 0, false>(gu, gridu, reps)},
            {"v2 dU", runp<1, false, 0>(gu, gridu, reps)},
            {"v2 dU noload", runp<1, false, 4>(gu, gridu, reps)},
            {"dW (A*A2)", run<1, 1, true, 0, false>(gu, gridu, reps)},
            {"v2 dW", runp<1, true, 0>(gu, gridu, reps)},
        };
        for (auto& x : r) printf("round %d  %-28s %8.1f us  %6.1f TFLOP/s\n", rd, x.n, x.us, gflop / x.us * 1e3);
        // stamped builds: the clock the chip holds inside the k-loop (after the launches above: warm)
        { float us = runp<0, false, 8>(g1, grid1, reps); clock_of("v2 GEMM1", us); }
        { float us = runp<0, false, 8 | 7>(g1, grid1, reps); clock_of("v2 GEMM1 noload nosplit noside", us); }
        { float us = runp<0, false, 8 | 7 | 16>(g1, grid1, reps); clock_of("v2 MFMA only", us); }
        { float us = runp<0, true, 8>(g1, grid1, reps); clock_of("v2 dT2g", us); }
    }
    return 0;
}
