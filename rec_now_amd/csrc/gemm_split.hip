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
#include "gemm_split.hpp"
#include "prof.hpp"
#include <atomic>
#include <type_traits>

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

// Several operands in ONE launch (the packed weights of every cross layer, once per step: dcnmix.hip): blockIdx.y = job
__global__ void __launch_bounds__(256) k_split_planes_multi(const RnSplitJobs jobs) {
    const RnSplitJob j = jobs.job[blockIdx.y];
    const int64_t total = (int64_t)(j.K / 8) * j.N;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int n = (int)(u % j.N), o = (int)(u / j.N);
        float x[8];
        if (j.b_kc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(j.B + (int64_t)n * j.ldb + 8 * o), b = *reinterpret_cast<const f32x4*>(j.B + (int64_t)n * j.ldb + 8 * o + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = j.B[(int64_t)(8 * o + e) * j.ldb + n];
        }
        u32x4 w[3];
        spl_split8(x, w);
        const int64_t ps = total * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(j.planes + s * ps + u * 16) = w[s];
    }
}
int rn_split_planes_multi(const RnSplitJobs& jobs, hipStream_t st) {
    if (jobs.n < 1 || jobs.n > RN_SPLIT_MAX_JOBS) return RECNOW_EINVAL;
    int64_t most = 0;
    for (int i = 0; i < jobs.n; ++i) {
        const int64_t u = (int64_t)(jobs.job[i].K / 8) * jobs.job[i].N;
        if (jobs.job[i].K % 8 || (jobs.job[i].b_kc && (jobs.job[i].ldb % 4 || ((uintptr_t)jobs.job[i].B & 15)))) return RECNOW_EUNSUPPORTED;
        if (u > most) most = u;
    }
    hipLaunchKernelGGL(k_split_planes_multi, dim3((unsigned)((most + 255) / 256), (unsigned)jobs.n), 256, 0, st, jobs);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

int rn_split_planes(const float* B, int64_t ldb, int b_kc, int K, int N, void* planes, hipStream_t st) {
    const int64_t units = (int64_t)(K / 8) * N;
    hipLaunchKernelGGL(k_split_planes, (unsigned)((units + 255) / 256), 256, 0, st, B, ldb, b_kc ? 1 : 0, K, N, (char*)planes);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
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


// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the lean form of the product (k_gemm_s3).  Same arithmetic, tile, LDS image and pipeline depth as k_gemm_split; what the counters
// and the ISA of k_gemm_split blamed is gone from the k-loop (tools/micro/split3/split3_bench.hip is the stand-alone bench the numbers come from):
//   * B is ALWAYS piece planes (k_split_planes: weights, and -- new -- the (B, 128) activation operand of the K = B products, split once per launch
//     instead of by each of the eight row-tile workgroups that read it): the loader copies three 16-byte units per thread and k-tile;
//   * the side weights (<= 2 extra columns) of the whole k-chunk are staged in LDS once, in front of the loop -- k_gemm_split loaded and stored
//     them inside the loop under `tid < 16` (a branch around a load: a full vmcnt(0) at its merge) -- and the side product is 16 scalar
//     FMAs per thread and k-tile on two columns (four packed ones on a padded quad before);
//   * operand loads take the `global_load v, voff, s[base]` form (block-uniform tile base + an opaque 32-bit per-lane byte offset);
//   * the last k-tile runs without the (surplus) staging of a tile nobody reads: no weights of zero in the side product.
// Instructions per MFMA in the k-loop: 4.7 (A alone) / 5.1 (A * A2) against 6.9; one MI355X, GEMM1 shape (65 536 x 1024 x 128 + 2 side columns):
// 124-127 us -> 98-103 us; [k][row] operand with pre-split B (the dU shape) 95 us.  In-kernel clock under this loop 1.70 GHz (stamped build), MFMAs
// alone on the same fragments 65 us at 1.78 GHz: the loop holds the matrix pipe 68 % busy at the clock the power budget leaves for bf16 MFMAs.
// A thread stages one unit (8 consecutive k of one row): [row][k] operands (row = tid >> 1, octet = tid & 1): adjacent lanes cover 64 contiguous
// bytes of a row (two streams of 32 B per lane at a row stride measured 20 % slower once a second operand doubles the bytes); [k][row] operands
// (row = tid & 127, octet = tid >> 7): eight dword loads, one per k row (a wave reads 256 contiguous bytes of each).
#define S3_BX_MAXK 2048                                   // side weights staged per k-chunk: 2 floats per k
#define S3_LDS(kchunk) (SPL_BX_OFF + (kchunk) * 8)
// PAIR (round 6, second session; [row][k] operands, whole groups of four k-tiles): the A loads of k-tiles (2 j, 2 j + 1) are issued TOGETHER.  A wave load of a
// k-contiguous operand covers 32 rows x 64 B -- half of each 128-byte line -- and the other half used to be asked for one k-tile (~1 us) later: with 32 CUs x 2
// workgroups x 2 streams x 3 k-tiles in flight per XCD (6 MB against 4 MB of L2) the line was often gone by then and came from HBM a second time (PMC:
// `k_gemm_s3<true, 1>` 738 MB per launch for 570 MB of operands).  Three register sets instead of two: E holds the even k-tiles, O1 / O2 the odd ones in turn.
template <bool A_KC, int A2K, bool PAIR = false>
__global__ void __launch_bounds__(GEMM_THREADS, 2)
k_gemm_s3(const GemmK p, const char* __restrict__ b_planes, int64_t b_plane_bytes) {
    static_assert(!PAIR || A_KC, "paired loads: k-contiguous A operands");
    extern __shared__ __attribute__((aligned(16))) char spl_smem[];
    int bx = blockIdx.x, z = blockIdx.z;
    if (p.xcd_remap == 1) {       // as k_gemm: the row tiles of one k-slab become consecutive workgroups of one XCD
        const int gx = gridDim.x, lin = bx + gx * z, xcd = lin & 7, i = lin >> 3;
        z = xcd * ((int)gridDim.z >> 3) + i / gx;
        bx = i % gx;
    }
    const int ks = z;                                      // batch == 1 (the side product belongs to one product)
    const int k_begin = ks * p.kchunk, k_end = min(p.K, k_begin + p.kchunk);
    const int m0 = bx * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int nt = (k_end - k_begin) / SPL_BK;             // even, >= 2 (the launcher's check)
    const int a_row = A_KC ? tid >> 1 : tid & 127;
    const int a_h = A_KC ? tid & 1 : __builtin_amdgcn_readfirstlane(tid >> 7);
    const int b_row = tid & 127, b_h = tid >> 7;
    const unsigned ldau = (unsigned)p.lda;
    const unsigned a_goff = A_KC ? (unsigned)a_row * ldau + 8u * a_h : 8u * a_h * ldau + (unsigned)a_row;
    const int a_soff = (a_h * SPL_PLANE_H + a_row) * 16;
    const int b_soff = SPL_OPER + (b_h * SPL_PLANE_H + b_row) * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // side product (the gate logits of a cross layer when this is GEMM1): the 8 terms of a thread's unit are summed in fp32, the running sum over the
    // k-tiles is kept in fp64 (2 v_cvt + 2 v_add_f64 per k-tile).  Why: with fp32 running sums (512 sequential FMAs per thread) the layer-0 logits of
    // bench.py's parity inputs (|logit| up to 39) carried an absolute error of 3.5e-6 rms -- row-wide gate errors up to 5.7e-6, by themselves
    // 1.2e-5 of max|d loss / d x| in the worst row of 65 536 (tools/micro/error_budget_cpu.py); this form leaves 2.5e-7 rms.
    double sp0 = 0.0, sp1 = 0.0;
    constexpr int NSET = PAIR ? 3 : 2;
    float va[NSET][8], ya[A2K != RECNOW_OPMODE_NONE ? NSET : 1][8];
    u32x4 bpl[3], wq[3];

    auto a_base = [&](int t) { return A_KC ? (int64_t)m0 * p.lda + k_begin + t * SPL_BK : (int64_t)(k_begin + t * SPL_BK) * p.lda + m0; };
    auto clampt = [&](int t) { return min(t, nt - 1); };
    unsigned a_bo[A_KC ? 1 : 8];
    a_bo[0] = a_goff * 4u;
    if (!A_KC) {
#pragma unroll
        for (int e = 1; e < 8; ++e) a_bo[A_KC ? 0 : e] = (a_goff + (unsigned)e * ldau) * 4u;
    }
    auto a_issue = [&](float (&v)[8], float (&y)[8], int t) {
        const char* pa = reinterpret_cast<const char*>(p.A + a_base(t));
        const char* pa2 = reinterpret_cast<const char*>(p.A2 + a_base(t));
        if (A_KC) {
            asm volatile("" : "+v"(a_bo[0]));
            const f32x4 a = *reinterpret_cast<const f32x4*>(pa + a_bo[0]), b = *reinterpret_cast<const f32x4*>(pa + a_bo[0] + 16);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if (A2K != RECNOW_OPMODE_NONE) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(pa2 + a_bo[0]), d = *reinterpret_cast<const f32x4*>(pa2 + a_bo[0] + 16);
                y[0] = c.x; y[1] = c.y; y[2] = c.z; y[3] = c.w; y[4] = d.x; y[5] = d.y; y[6] = d.z; y[7] = d.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                asm volatile("" : "+v"(a_bo[A_KC ? 0 : e]));
                v[e] = *reinterpret_cast<const float*>(pa + a_bo[A_KC ? 0 : e]);
                if (A2K != RECNOW_OPMODE_NONE) y[e] = *reinterpret_cast<const float*>(pa2 + a_bo[A_KC ? 0 : e]);
            }
        }
    };
    unsigned b_bo = (unsigned)((b_h * 128 + b_row) * 16);
    auto b_issue = [&](int t) {
        const char* src = b_planes + (int64_t)((k_begin + t * SPL_BK) >> 3) * 128 * 16;
        asm volatile("" : "+v"(b_bo));
#pragma unroll
        for (int s = 0; s < 3; ++s) bpl[s] = *reinterpret_cast<const u32x4*>(src + s * b_plane_bytes + b_bo);
    };
    auto b_store = [&](char* S) {
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + b_soff + s * SPL_PLANE) = bpl[s];
    };
    auto a_combine = [&](float (&v)[8], const float (&y)[8]) {
        if (A2K == RECNOW_OPMODE_MUL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= y[e];
        } else if (A2K != RECNOW_OPMODE_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= rn_act_grad_from_out(y[e], p.a_act);
        }
    };
    const float* bxl = reinterpret_cast<const float*>(spl_smem + SPL_BX_OFF);
    auto a_side = [&](const float (&v)[8], int t) {       // this thread's unit times the two side columns of k-tile t (broadcast reads)
        const float* b = bxl + (t * SPL_BK + 8 * a_h) * 2;
        f32x4 q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = *reinterpret_cast<const f32x4*>(b + 4 * e);
        float t0 = v[0] * q[0].x, t1 = v[0] * q[0].y;
        t0 = fmaf(v[1], q[0].z, t0);
        t1 = fmaf(v[1], q[0].w, t1);
#pragma unroll
        for (int e = 1; e < 4; ++e) {
            t0 = fmaf(v[2 * e], q[e].x, t0);
            t1 = fmaf(v[2 * e], q[e].y, t1);
            t0 = fmaf(v[2 * e + 1], q[e].z, t0);
            t1 = fmaf(v[2 * e + 1], q[e].w, t1);
        }
        sp0 += (double)t0;
        sp1 += (double)t1;
    };
    auto split_pairs = [&](const float (&v)[8], int e0) {
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e) {
            unsigned p1, p2, p3;
            spl_split2(v[2 * e], v[2 * e + 1], p1, p2, p3);
            wq[0][e] = p1; wq[1][e] = p2; wq[2][e] = p3;
        }
    };
    auto a_store = [&](char* S) {
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(S + a_soff + s * SPL_PLANE) = wq[s];
    };

    // the chunk's side weights -> LDS: bxl[k][0 .. 1] (a missing second column: zeros)
    for (int i = tid; i < k_end - k_begin; i += GEMM_THREADS) {
        const float* src = p.bx + (int64_t)(k_begin + i) * p.bx_ks;
        f32x2 w;
        w.x = src[0];
        w.y = p.sp_r > 1 ? src[p.bx_rs] : 0.f;
        *reinterpret_cast<f32x2*>(spl_smem + SPL_BX_OFF + i * 8) = w;
    }
    // prologue: k-tile 0 -> stage 0; A of k-tiles 1 (set 1) and 2 (set 0), B of k-tile 1 requested
    // (PAIR: k-tiles 0 and 1 requested together into sets E = 0 and O1 = 1, then 2 and 3 into E and O2 = 2)
    constexpr int Y1 = A2K != RECNOW_OPMODE_NONE ? 1 : 0, Y2 = A2K != RECNOW_OPMODE_NONE ? 2 : 0;
    a_issue(va[0], ya[0], 0);
    if constexpr (PAIR) a_issue(va[1], ya[Y1], clampt(1));
    b_issue(0);
    if constexpr (!PAIR) a_issue(va[1], ya[Y1], clampt(1));
    __syncthreads();
    a_combine(va[0], ya[0]);
    a_side(va[0], 0);
    split_pairs(va[0], 0);
    split_pairs(va[0], 2);
    a_store(spl_smem);
    b_store(spl_smem);
    a_issue(va[0], ya[0], clampt(2));
    if constexpr (PAIR) a_issue(va[2], ya[Y2], clampt(3));
    b_issue(clampt(1));
    __syncthreads();

    int a_off[2], b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_off[i] = ((lane >> 5) * SPL_PLANE_H + wm * 64 + i * 32 + (lane & 31)) * 16;
        b_off[i] = SPL_OPER + ((lane >> 5) * SPL_PLANE_H + wn * 64 + i * 32 + (lane & 31)) * 16;
    }
#define S3_TERM(SA, SB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SA][i], bf[SB][j], acc[i][j], 0, 0, 0);
    // one k-tile: t = its index; SLOT holds A of k-tile t + 1; STG: stage k-tile t + 1 (false: the last k-tile, compute only)
    // (PAIR: `issue_c` = the set that takes k-tile t + 4 beside SLOT's t + 3, or -1: this k-tile issues no A loads)
    auto ktile = [&](int t, auto slot_c, auto stage_c, auto issue_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        constexpr int YS = A2K != RECNOW_OPMODE_NONE ? SLOT : 0;
        constexpr bool STG = decltype(stage_c)::value;
        constexpr int ISS = decltype(issue_c)::value;
        const char* S = spl_smem + (t & 1) * SPL_STAGE;
        char* Sn = spl_smem + ((t & 1) ^ 1) * SPL_STAGE;
        bf16x8 af[3][2], bf[3][2];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[s][i] = *reinterpret_cast<const bf16x8*>(S + s * SPL_PLANE + a_off[i]);
                bf[s][i] = *reinterpret_cast<const bf16x8*>(S + s * SPL_PLANE + b_off[i]);
            }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(0, 0)
        if (STG) {
            a_combine(va[SLOT], ya[YS]);
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
            if constexpr (!PAIR) {
                a_issue(va[SLOT], ya[YS], clampt(t + 3));
            } else if constexpr (ISS >= 0) {
                a_issue(va[SLOT], ya[YS], clampt(t + 3));
                a_issue(va[ISS], ya[A2K != RECNOW_OPMODE_NONE ? ISS : 0], clampt(t + 4));
            }
            b_store(Sn);
        }
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(0, 2)
        if (STG) b_issue(clampt(t + 2));
        __builtin_amdgcn_sched_barrier(0);
        S3_TERM(2, 0)
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using IN = std::integral_constant<int, -1>;
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;
    int t = 0;
    if constexpr (!PAIR) {
        for (; t + 2 < nt; t += 2) {
            ktile(t, I1(), BT(), IN());
            ktile(t + 1, I0(), BT(), IN());
        }
        ktile(t, I1(), BT(), IN());
        ktile(t + 1, I0(), BF(), IN());
    } else {
        // k-tile t stages k-tile t + 1: an odd one from O1 / O2 (t = 0, 2 mod 4), an even one from E (t odd), which then takes k-tile t + 3 while the
        // odd set that was emptied one k-tile before takes t + 4 -- the two halves of the same 128-byte lines, requested back to back
        for (; t + 4 < nt; t += 4) {
            ktile(t, I1(), BT(), IN());
            ktile(t + 1, I0(), BT(), I1());
            ktile(t + 2, I2(), BT(), IN());
            ktile(t + 3, I0(), BT(), I2());
        }
        ktile(t, I1(), BT(), IN());
        ktile(t + 1, I0(), BT(), IN());
        ktile(t + 2, I2(), BT(), IN());
        ktile(t + 3, I0(), BF(), IN());
    }
#undef S3_TERM

    // side product: the two threads of a row (its two k-octets): adjacent lanes (KC) or 128 threads apart (through LDS)
    float* smem = reinterpret_cast<float*>(spl_smem);
    auto side_out = [&](int m, float s0, float s1) {
        if (p.splitk > 1) {
            float* dst = p.partial + ((int64_t)z * p.M + m) * p.npart + p.N;
            dst[0] = s0;
            if (p.sp_r > 1) dst[1] = s1;
        } else {
            p.cx[(int64_t)m * p.cx_ms] = s0;
            if (p.sp_r > 1) p.cx[(int64_t)m * p.cx_ms + p.cx_rs] = s1;
        }
    };
    if (A_KC) {
        sp0 += __shfl_xor(sp0, 1);
        sp1 += __shfl_xor(sp1, 1);
        if ((tid & 1) == 0) side_out(m0 + a_row, (float)sp0, (float)sp1);
    } else {
        double* const dsm = reinterpret_cast<double*>(spl_smem);
        if (tid >= 128) { dsm[(tid - 128) * 2] = sp0; dsm[(tid - 128) * 2 + 1] = sp1; }
        __syncthreads();
        if (tid < 128) side_out(m0 + tid, (float)(sp0 + dsm[tid * 2]), (float)(sp1 + dsm[tid * 2 + 1]));
        __syncthreads();
    }
    gemm_lean_epilogue<2, 2, 0>(p, acc, smem, m0, 0, wm, wn, lane, wave, z, 0);
}

size_t rn_gemm_split_planes_bytes(int K, int N) { return rn_align((size_t)(K / 8) * N * 16 * 3); }

// Launcher: the (layout, operand kind) combinations of the DCN-v2 step, N = 128 (one column tile: the side product belongs to
// it).  `planes` != NULL: B is split once into planes there (rn_gemm_split_planes_bytes(K, N) bytes) before the product.
// RECNOW_EUNSUPPORTED -> the caller runs the fp32 kernel.
// Lean form (round 6): N = 128, at most two side columns, whole k-chunks of an even number of k-tiles, the chunk's side weights in LDS.
static bool s3_shape(const GemmK& k, int a2k) {
    return k.batch == 1 && k.N == 128 && k.sp_r >= 1 && k.sp_r <= 2 && k.kchunk % (2 * SPL_BK) == 0 && k.K % k.kchunk == 0 && k.kchunk <= S3_BX_MAXK &&
           (a2k == RECNOW_OPMODE_NONE || a2k == RECNOW_OPMODE_MUL) && !k.as_out;
}
static std::atomic<int> g_s3_lds_ready[8];      // dynamic-LDS attribute raised per instantiation (the default limit is 64 KB; a chunk of 2048 k needs 66 KB)
template <bool A_KC, int A2K>
static int s3_launch(const GemmK& k, const char* planes, int64_t pb, dim3 grid, hipStream_t st, int slot) {
    const size_t lds = S3_LDS(k.kchunk);
    // paired A loads (see the kernel): k-contiguous operands, whole groups of four k-tiles, rows that start on a 128-byte line; RECNOW_S3_PAIR=0: A/B switch
    static const bool pair_on = []() { const char* e = getenv("RECNOW_S3_PAIR"); return !e || e[0] != '0'; }();
    if constexpr (A_KC) {
        if (pair_on && k.kchunk % (4 * SPL_BK) == 0 && k.lda % 32 == 0 && ((uintptr_t)k.A & 127) == 0 && (A2K == RECNOW_OPMODE_NONE || ((uintptr_t)k.A2 & 127) == 0)) {
            if (lds > 64 * 1024 && !g_s3_lds_ready[4 + slot].load(std::memory_order_acquire)) {
                RN_HIP(hipFuncSetAttribute((const void*)k_gemm_s3<A_KC, A2K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S3_LDS(S3_BX_MAXK)));
                g_s3_lds_ready[4 + slot].store(1, std::memory_order_release);
            }
            hipLaunchKernelGGL((k_gemm_s3<A_KC, A2K, true>), grid, GEMM_THREADS, lds, st, k, planes, pb);
            RN_LAUNCH_CHECK();
            return RECNOW_OK;
        }
    }
    if (lds > 64 * 1024 && !g_s3_lds_ready[slot].load(std::memory_order_acquire)) {
        RN_HIP(hipFuncSetAttribute((const void*)k_gemm_s3<A_KC, A2K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S3_LDS(S3_BX_MAXK)));
        g_s3_lds_ready[slot].store(1, std::memory_order_release);
    }
    hipLaunchKernelGGL((k_gemm_s3<A_KC, A2K>), grid, GEMM_THREADS, lds, st, k, planes, pb);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// Launcher: the (layout, operand kind) combinations of the DCN-v2 step, N = 128 (one column tile: the side product belongs to
// it).  `planes` != NULL: B is split once into planes there (rn_gemm_split_planes_bytes(K, N) bytes) before the product.
// RECNOW_EUNSUPPORTED -> the caller runs the fp32 kernel.
int rn_gemm_launch_split(const GemmK& k, bool a_kc, bool b_kc, int a2k, void* planes, dim3 grid, hipStream_t st, const void* ready) {
    if (grid.y != 1 || k.K % SPL_BK || k.kchunk % SPL_BK || k.sp_r <= 0) return RECNOW_EUNSUPPORTED;
    // element offsets inside a tile are 32-bit
    if ((int64_t)128 * k.lda >= (1ll << 31) || (int64_t)128 * k.ldb >= (1ll << 31)) return RECNOW_EUNSUPPORTED;
    const int64_t pb = (int64_t)(k.K / 8) * k.N * 16;
    static const bool s3_on = []() { const char* e = getenv("RECNOW_SPLIT_LEAN"); return !e || e[0] != '0'; }();      // A/B switch: 0 = k_gemm_split of rounds 2-5
    // [k][row] operands (the K = B weight-gradient products) with a pre-split activation operand: built and measured in the step -- 103 / 130 us
    // (A / A * A2) + 18 us for the split of the (B, 128) operand against 111 / 126 us for k_gemm_split, which splits B in every workgroup: the
    // pre-pass costs what the leaner loop gains, so these products stay on k_gemm_split (RECNOW_SPLIT_LEAN=2 routes them here: A/B switch)
    static const bool s3_kb = []() { const char* e = getenv("RECNOW_SPLIT_LEAN"); return e && e[0] == '2'; }();
    // ready != NULL: the caller holds the planes of this product's B already (the packed weights of a step are split once, dcnmix.hip)
    if (ready && s3_on && s3_shape(k, a2k) && a_kc)
        return a2k == 0 ? s3_launch<true, 0>(k, (const char*)ready, pb, grid, st, 0) : s3_launch<true, RECNOW_OPMODE_MUL>(k, (const char*)ready, pb, grid, st, 1);
    if (planes && s3_on && s3_shape(k, a2k) && (a_kc || (!b_kc && s3_kb))) {
        const int64_t units = (int64_t)(k.K / 8) * k.N;
        hipLaunchKernelGGL(k_split_planes, (unsigned)((units + 255) / 256), 256, 0, st, k.B, k.ldb, b_kc ? 1 : 0, k.K, k.N, (char*)planes);
        RN_LAUNCH_CHECK();
        if (a_kc) return a2k == 0 ? s3_launch<true, 0>(k, (const char*)planes, pb, grid, st, 0) : s3_launch<true, RECNOW_OPMODE_MUL>(k, (const char*)planes, pb, grid, st, 1);
        return a2k == 0 ? s3_launch<false, 0>(k, (const char*)planes, pb, grid, st, 2) : s3_launch<false, RECNOW_OPMODE_MUL>(k, (const char*)planes, pb, grid, st, 3);
    }
    if (planes && k.batch == 1 && a_kc && k.K <= 4096) {
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
