// Split-precision variant of the lean 128x128 GEMM with a side product (the K = 1024 and K = B products of the DCN-v2 step):
// every fp32 operand element x is split, on its way into LDS, into three bf16 pieces x = x1 + x2 + x3 (round-to-nearest each,
// residual <= 2^-27 |x|), and a product a*b is formed as the six bf16 MFMA terms a_i b_j with i + j <= 4, accumulated in fp32
// (v_mfma_f32_32x32x16_bf16: products of bf16 pairs are exact in fp32).  Dropped terms: a2 b3 + a3 b2 + a3 b3 <= 2^-25 |a b|, i.e.
// below one fp32 rounding of the product; the accumulation itself is fp32, as in the exact kernels.  Six 8-pass bf16 MFMAs do the
// work of eight 16-pass fp32 MFMAs: 2.67x the fp32-MFMA rate.  OPT-IN (recnow_set_gemm_precision / RECNOW_GEMM_PRECISION=bf16x3):
// results are not bit-identical to the fp32 kernels; parity (1e-5 relative, north_star) is held by the same tests.
//
// Structure: 256 threads = 2 x 2 waves of 64 x 64, k-tiles of 16 (= one MFMA k-step), two LDS stages.  The A operand is
// requested TWO k-tiles ahead (register ring of two: an iteration is ~1 us, shorter than an HBM round trip under load), the B
// operand one (weights: L2) or two (activations) ahead; the split + LDS writes of k-tile t+1 and the loads sit between the six
// MFMA groups of k-tile t.  LDS image per operand and stage: 3 planes (pieces) x 2 k-halves x 136 units of 16 B (unit = 8
// consecutive k of one row: exactly what a lane feeds to the MFMA; rows padded by one unit per 16 so that the writes of the
// [k][row]-contiguous loaders spread over the banks).  A small B operand (weights, K <= 4096) is split ONCE per launch into
// global planes of the same units by k_split_planes (every one of the 512 workgroups would otherwise split the same tile), the
// loader then copies units.  Side product (sp_r <= 4 extra columns): fp32 FMAs on the A staging registers at LDS-write time,
// reduced across the threads that share a row at the end (there is no fp32 image in LDS to read it from).
#include "gemm_kernel.hpp"
#include "prof.hpp"
#include <type_traits>

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

// B (K x N; [K][N] rows of ldb floats, or [N][K] when b_kc) -> planes[s][K/8][N] units of 8 bf16 (16 B): unit (o, n) of piece s
// holds piece s of B[8 o .. 8 o + 7][n].  One thread per unit.
__global__ void __launch_bounds__(256) k_split_planes(const float* __restrict__ B, int64_t ldb, int b_kc, int K, int N, char* __restrict__ planes) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)(K / 8) * N;
    if (u >= total) return;
    const int n = (int)(u % N), o = (int)(u / N);
    float x[8];
    if (b_kc) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(B + (int64_t)n * ldb + 8 * o), b = *reinterpret_cast<const f32x4*>(B + (int64_t)n * ldb + 8 * o + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = B[(int64_t)(8 * o + e) * ldb + n];
    }
    u32x4 w[3];
    spl_split8(x, w);
    const int64_t ps = total * 16;
#pragma unroll
    for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(planes + s * ps + u * 16) = w[s];
}

// One fp32 operand's staging registers for ONE k-tile.  Every thread owns exactly one unit = 8 consecutive k (k-half h) of one
// tile row, so that the split pieces leave as three conflict-free ds_write_b128:
//   KC  ([row][k], k contiguous): row = tid >> 1, h = tid & 1: two float4 (32 contiguous bytes)
//   !KC ([k][row], row contiguous): row = tid & 127, h = tid >> 7: eight dword loads, one per k row (a wave reads 256 contiguous
//        bytes of each: float4 loads here would leave every thread with 4 rows x 2 k, i.e. 2-byte scatters into LDS)
template <bool KC, int K2>
struct SplTile {
    float v[8], y[8];
    __device__ __forceinline__ void issue(const float* __restrict__ p, const float* __restrict__ p2, int64_t tile_off, unsigned off, int64_t ld) {
        if (KC) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + tile_off + off), b = *reinterpret_cast<const f32x4*>(p + tile_off + off + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if (K2 != RECNOW_OPMODE_NONE) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(p2 + tile_off + off), d = *reinterpret_cast<const f32x4*>(p2 + tile_off + off + 4);
                y[0] = c.x; y[1] = c.y; y[2] = c.z; y[3] = c.w; y[4] = d.x; y[5] = d.y; y[6] = d.z; y[7] = d.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = p[tile_off + e * ld + off];
                if (K2 != RECNOW_OPMODE_NONE) y[e] = p2[tile_off + e * ld + off];
            }
        }
    }
    __device__ __forceinline__ void combine(int act) {
        if (K2 == RECNOW_OPMODE_MUL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= y[e];
        } else if (K2 != RECNOW_OPMODE_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= rn_act_grad_from_out(y[e], act);
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ S) const {      // S: this thread's unit in piece plane 0
        u32x4 w[3];
        spl_split8(v, w);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + s * SPL_PLANE) = w[s];
    }
};

// BSRC: 0 = fp32 [K][N] split in the kernel, 1 = fp32 [N][K] split in the kernel, 2 = planes of k_split_planes
template <bool A_KC, int A2K, int BSRC>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
k_gemm_split(const GemmK p, const char* __restrict__ b_planes, int64_t b_plane_bytes) {
    constexpr bool B_KC = BSRC == 1;
    constexpr bool BPRE = BSRC == 2;
    extern __shared__ __attribute__((aligned(16))) char spl_smem[];
    float* const Bxs = reinterpret_cast<float*>(spl_smem + SPL_BX_OFF);
    int bx = blockIdx.x, z = blockIdx.z;
    if (p.xcd_remap == 1) {       // as k_gemm: the row tiles of one k-slab become consecutive workgroups of one XCD
        const int gx = gridDim.x, lin = bx + gx * z, xcd = lin & 7, i = lin >> 3;
        z = xcd * ((int)gridDim.z >> 3) + i / gx;
        bx = i % gx;
    }
    const int bidx = z / p.splitk, ks = z % p.splitk;
    const int k_begin = ks * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int m0 = bx * 128, n0 = blockIdx.y * 128;
    const float* Ab = p.A + (int64_t)bidx * p.sA;
    const float* A2b = p.A2 ? p.A2 + (int64_t)bidx * p.sA : nullptr;
    const float* Bb = p.B + (int64_t)bidx * p.sB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = (k_end - k_begin) / SPL_BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 spacc = mk4(0.f, 0.f, 0.f, 0.f);       // side product: the 4 columns of this thread's A row, partial over its k-half
    float bxr[4] = {0.f, 0.f, 0.f, 0.f};

    // this thread's unit (row, k-half) per operand: element offset inside a k-tile (the tile base is block-uniform) and LDS offset
    const int a_row = A_KC ? tid >> 1 : tid & 127, a_h = A_KC ? tid & 1 : tid >> 7;
    const int b_row = B_KC ? tid >> 1 : tid & 127, b_h = B_KC ? tid & 1 : tid >> 7;      // planes: as !KC
    const unsigned a_goff = A_KC ? (unsigned)(a_row * p.lda + 8 * a_h) : (unsigned)(8 * a_h * p.lda + a_row);
    const unsigned b_goff = B_KC ? (unsigned)(b_row * p.ldb + 8 * b_h) : (unsigned)(8 * b_h * p.ldb + b_row);
    const int a_soff = (a_h * SPL_PLANE_H + a_row) * 16;
    const int b_soff = SPL_OPER + (b_h * SPL_PLANE_H + b_row) * 16;
    const int64_t bp_goff = BPRE ? ((int64_t)b_h * p.N + n0 + b_row) * 16 : 0;

    SplTile<A_KC, A2K> ta[2];                      // ring of two k-tiles in flight
    SplTile<B_KC, RECNOW_OPMODE_NONE> tb[2];       // in-kernel split of B: ring of two as well
    u32x4 bpl[3];                                  // planes: one k-tile in flight (L2-resident weights)
    auto a_base = [&](int tile) { return A_KC ? (int64_t)m0 * p.lda + k_begin + tile * SPL_BK : (int64_t)(k_begin + tile * SPL_BK) * p.lda + m0; };
    auto b_base = [&](int tile) { return B_KC ? (int64_t)n0 * p.ldb + k_begin + tile * SPL_BK : (int64_t)(k_begin + tile * SPL_BK) * p.ldb + n0; };
    auto b_issue = [&](int slot, int tile) {
        if (BPRE) {
            const char* src = b_planes + (int64_t)((k_begin + tile * SPL_BK) >> 3) * p.N * 16 + bp_goff;
#pragma unroll
            for (int s = 0; s < 3; ++s) bpl[s] = *reinterpret_cast<const u32x4*>(src + s * b_plane_bytes);
        } else {
            tb[slot].issue(Bb, nullptr, b_base(tile), b_goff, p.ldb);
        }
    };
    auto b_store = [&](int slot, char* S) {
        if (BPRE) {
#pragma unroll
            for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + b_soff + s * SPL_PLANE) = bpl[s];
        } else {
            tb[slot].store(S + b_soff);
        }
    };
    auto load_bx = [&](int tile) {      // threads < 16: the side-product weights of k-tile `tile` (one k each)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bxr[r] = r < p.sp_r ? p.bx[(int64_t)(k_begin + tile * SPL_BK + tid) * p.bx_ks + r * p.bx_rs] : 0.f;
    };
    auto store_bx = [&](int tile) { *reinterpret_cast<f32x4*>(Bxs + (tile % SPL_BX_RING) * SPL_BK * 4 + tid * 4) = mk4(bxr[0], bxr[1], bxr[2], bxr[3]); };
    // side product of the A registers with the weights of k-tile `tile` (times w: 0 for a surplus commit)
    auto sp_fma = [&](int slot, int tile, float w) {
        const float* bt = Bxs + (tile % SPL_BX_RING) * SPL_BK * 4 + a_h * 32;
        f32x4 s = ta[slot].v[0] * *reinterpret_cast<const f32x4*>(bt);
#pragma unroll
        for (int e = 1; e < 8; ++e) s += ta[slot].v[e] * *reinterpret_cast<const f32x4*>(bt + 4 * e);
        spacc += w * s;
    };
    auto clampt = [&](int tile) { return min(tile, ntile - 1); };

    if (ntile > 0) {
        // k-tile 0 -> stage 0; k-tiles 1 (ring slot 1) and 2 (slot 0) requested; side-product weights of k-tiles 0..2 staged, 3 requested
        ta[0].issue(Ab, A2b, a_base(0), a_goff, p.lda);
        b_issue(0, 0);
        if (tid < SPL_BK) {
            load_bx(0);
            store_bx(0);
            load_bx(clampt(1));
            store_bx(1);
            load_bx(clampt(2));
            store_bx(2);
            load_bx(clampt(3));
        }
        ta[1].issue(Ab, A2b, a_base(clampt(1)), a_goff, p.lda);
        __syncthreads();
        ta[0].combine(p.a_act);
        sp_fma(0, 0, 1.f);
        ta[0].store(spl_smem + a_soff);
        b_store(0, spl_smem);
        ta[0].issue(Ab, A2b, a_base(clampt(2)), a_goff, p.lda);
        if (BPRE) {
            b_issue(0, clampt(1));
        } else {
            b_issue(1, clampt(1));
            b_issue(0, clampt(2));
        }
    }
    __syncthreads();

    // fragment addresses: lane (row l & 31 of the 32-row MFMA tile, k-half l >> 5) reads one 16-byte unit per piece (consecutive
    // units per half: conflict-free for the 16-lane groups of ds_read_b128)
    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = ((lane >> 5) * SPL_PLANE_H + wm * 64 + i * 32 + (lane & 31)) * 16;
        b_off[i] = SPL_OPER + ((lane >> 5) * SPL_PLANE_H + wn * 64 + i * 32 + (lane & 31)) * 16;
    }
    // one k-tile: t = its index, SLOT = the ring slot that holds k-tile t+1 (compile-time: the loop below is unrolled by two)
    auto ktile = [&](int t, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        const int cur = t & 1;
        const char* S = spl_smem + cur * SPL_STAGE;
        char* Sn = spl_smem + (cur ^ 1) * SPL_STAGE;
        const float spw = t + 1 < ntile ? 1.f : 0.f;           // the last iteration's commit is surplus (nobody reads it)
        if (tid < SPL_BK) {                                    // weights of k-tile t+3 to the ring, t+4 requested
            store_bx(t + 3);
            load_bx(clampt(t + 4));
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
        // the six terms in the order their fragments were requested (pieces 0, then 1, then 2: the first group waits for four
        // reads, not twelve); the staging work of k-tile t+1 is cut into pieces that follow the MFMA groups
#define SPL_TERM(SA, SB)                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SA][i], bf[SB][j], acc[i][j], 0, 0, 0);
        SPL_TERM(0, 0)
        ta[SLOT].combine(p.a_act);
        sp_fma(SLOT, t + 1, spw);
        __builtin_amdgcn_sched_barrier(0);
        SPL_TERM(0, 1)
        ta[SLOT].store(Sn + a_soff);
        __builtin_amdgcn_sched_barrier(0);
        SPL_TERM(1, 0)
        ta[SLOT].issue(Ab, A2b, a_base(clampt(t + 3)), a_goff, p.lda);
        __builtin_amdgcn_sched_barrier(0);
        SPL_TERM(1, 1)
        b_store(SLOT, Sn);
        __builtin_amdgcn_sched_barrier(0);
        SPL_TERM(0, 2)
        if (BPRE) b_issue(0, clampt(t + 2));
        else b_issue(SLOT, clampt(t + 3));
        __builtin_amdgcn_sched_barrier(0);
        SPL_TERM(2, 0)
#undef SPL_TERM
        __syncthreads();
    };
    // k-tile t+1 sits in ring slot (t + 1) & 1
    int t = 0;
    for (; t + 1 < ntile; t += 2) {
        ktile(t, std::integral_constant<int, 1>());
        ktile(t + 1, std::integral_constant<int, 0>());
    }
    if (t < ntile) ktile(t, std::integral_constant<int, 1>());

    // the two threads of a row (its two k-halves): adjacent lanes (KC) or 128 threads apart (!KC, through LDS)
    float* smem = reinterpret_cast<float*>(spl_smem);
    if (A_KC) {
        f32x4 s = spacc;
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += __shfl_xor(s[c], 1);
        if ((tid & 1) == 0) {
            const int m = m0 + a_row;
            for (int r = 0; r < p.sp_r; ++r) {
                if (p.splitk > 1) p.partial[((int64_t)z * p.M + m) * p.npart + p.N + r] = s[r];
                else p.cx[(int64_t)m * p.cx_ms + r * p.cx_rs] = s[r];
            }
        }
    } else {
        if (tid >= 128) *reinterpret_cast<f32x4*>(smem + (tid - 128) * 4) = spacc;
        __syncthreads();
        if (tid < 128) {
            const f32x4 s = spacc + *reinterpret_cast<const f32x4*>(smem + tid * 4);
            const int m = m0 + tid;
            for (int r = 0; r < p.sp_r; ++r) {
                if (p.splitk > 1) p.partial[((int64_t)z * p.M + m) * p.npart + p.N + r] = s[r];
                else p.cx[(int64_t)m * p.cx_ms + r * p.cx_rs] = s[r];
            }
        }
        __syncthreads();
    }
    gemm_lean_epilogue<2, 2, 0>(p, acc, smem, m0, n0, wm, wn, lane, wave, z, bidx);
}

size_t rn_gemm_split_planes_bytes(int K, int N) { return rn_align((size_t)(K / 8) * N * 16 * 3); }

// Launcher: the (layout, operand kind) combinations of the DCN-v2 step, N = 128 (one column tile: the side product belongs to
// it).  `planes` != NULL: B is split once into planes there (rn_gemm_split_planes_bytes(K, N) bytes) before the product.
// RECNOW_EUNSUPPORTED -> the caller runs the fp32 kernel.
int rn_gemm_launch_split(const GemmK& k, bool a_kc, bool b_kc, int a2k, void* planes, dim3 grid, hipStream_t st) {
    if (grid.y != 1 || k.K % SPL_BK || k.kchunk % SPL_BK || k.sp_r <= 0) return RECNOW_EUNSUPPORTED;
    // element offsets inside a tile are 32-bit
    if ((int64_t)128 * k.lda >= (1ll << 31) || (int64_t)128 * k.ldb >= (1ll << 31)) return RECNOW_EUNSUPPORTED;
    const int64_t pb = (int64_t)(k.K / 8) * k.N * 16;
    if (planes && k.batch == 1 && a_kc) {
        const int64_t units = (int64_t)(k.K / 8) * k.N;
        hipLaunchKernelGGL(k_split_planes, (unsigned)((units + 255) / 256), 256, 0, st, k.B, k.ldb, b_kc ? 1 : 0, k.K, k.N, (char*)planes);
        RN_LAUNCH_CHECK();
        if (a2k == 0) hipLaunchKernelGGL((k_gemm_split<true, 0, 2>), grid, GEMM_THREADS, SPL_LDS, st, k, (const char*)planes, pb);
        else if (a2k == 1) hipLaunchKernelGGL((k_gemm_split<true, 1, 2>), grid, GEMM_THREADS, SPL_LDS, st, k, (const char*)planes, pb);
        else return RECNOW_EUNSUPPORTED;
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
#define X(AKC, BKC, A2)                                                                                               \
    if (a_kc == AKC && b_kc == BKC && a2k == A2) {                                                                    \
        hipLaunchKernelGGL((k_gemm_split<AKC, A2, BKC ? 1 : 0>), grid, GEMM_THREADS, SPL_LDS, st, k, (const char*)nullptr, (int64_t)0);  \
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
