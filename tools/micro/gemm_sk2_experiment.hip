// EXPERIMENT, NOT PART OF THE LIBRARY (round 3): a role-split variant of k_gemm_shortk that was built, verified
// (parity green through tests/test_step_gpu.py, tests/test_fused_gpu.py with it switched in) and measured SLOWER than the kernel it was
// meant to replace: 228-235 us per launch against 196 us on the six K = 144 products of the c3 step (one box, A/B in one gpurun call).
// Diagnostics on the same box: without its epilogue 160 us (the MFMA role with one barrier per k-tile and one workgroup per CU), without
// its MFMAs 180 us (four memory waves per CU stream the epilogue at 4.9 TB/s) -- the two roles meet at a barrier every k-tile, so every
// late store or load of the memory role stalls the MFMA role: 235 us instead of max(160, 180).  To build it into the library again:
// copy it to rec_now_amd/csrc/, declare rn_gemm_launch_sk2 in gemm_kernel.hpp and call it at the top of rn_gemm_launch_shortk.
// Role-split persistent short-K GEMM (K = 144 = nine 16-deep k-tiles: the N*S+N deep products of DCNMixLayer, reference
// rec_now/layers/dcn_mix_layer.py:141-143 and their backward), exact fp32 on v_mfma_f32_32x32x2_f32.  Same contract, epilogue modes,
// tile order and arithmetic as k_gemm_shortk (gemm_shortk.hip); what differs is WHO does what inside a workgroup.
//
// k_gemm_shortk: 4 waves, each loads operands, writes LDS, runs MFMAs and then its own epilogue (global loads of the epilogue
// operands, LDS staging, global stores) -- the MFMA pipe of a CU idles while its workgroups are in their epilogues and the epilogue's
// memory instructions interleave with the MFMA issue of the co-resident workgroup: the products run at 0.52-0.55 of BOTH rooflines.
// Here a workgroup has 8 waves, one per role and SIMD:
//   waves 0-3 (MFMA role): fragment reads + MFMAs, nothing else; after the ninth k-tile the 128 x 128 accumulator tile is parked in an
//     LDS staging tile and the waves go straight on to the next tile;
//   waves 4-7 (memory role): operand tiles global -> registers (two k-tiles ahead, three rotating register sets) -> LDS, AND the
//     epilogue of the PREVIOUS tile in eight slices spread over the current tile's k-loop: staged accumulators (LDS) x epilogue
//     operands (prefetched two slices ahead) -> global stores.
// One barrier per k-tile keeps the two roles in step (the memory role's slice of work fits inside a k-tile of MFMAs); one workgroup per
// CU (117 KB of LDS: 2 operand buffers + the staging tile).
#include "gemm_kernel.hpp"

#define SK2_BM 128
#define SK2_BN 128
#define SK2_BK 16
#define SK2_NK 9
#define SK2_LK 4             // operand look-ahead of the memory role, in k-tiles
#define SK2_LDC 132          // row stride of the staging tile (floats): rows stay 16-byte aligned, a wave's accumulator column write
                             // (32 consecutive floats per row) and a half-wave's float4 row read are conflict-free

template <bool B_KC, int EP, int DUAL>
__global__ void __launch_bounds__(512, 1)
k_gemm_sk2(const GemmK p, int row_tiles, int col_tiles, int xcd_aware) {
    constexpr int LDA = SK2_BM + 1;                          // A is k-contiguous in memory: transposed on the LDS write, odd stride
    constexpr int LDB = B_KC ? SK2_BN + 1 : SK2_BN;
    constexpr int A_SZ = SK2_BK * LDA, B_SZ = SK2_BK * LDB, BUF = (A_SZ + B_SZ + 3) / 4 * 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [operand buffer 0 | operand buffer 1 | staging tile]
    float* const Cs = smem + 2 * BUF;
    constexpr int SK2_EK = (DUAL == 2 || DUAL == 4) ? 3 : 4;      // epilogue-operand look-ahead in slices (two operand tensors: one less, registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = row_tiles * col_tiles;
    if ((int)blockIdx.x >= ntiles) return;
    auto tile_of = [&](int slot, int& m0, int& n0) {         // as k_gemm_shortk: the column tiles of a row tile on one XCD
        int rt, ct;
        if (xcd_aware) {
            const int xcd = slot & 7, idx = slot >> 3;
            rt = (idx / col_tiles) * 8 + xcd;
            ct = idx % col_tiles;
        } else {
            rt = slot / col_tiles;
            ct = slot % col_tiles;
        }
        m0 = rt * SK2_BM;
        n0 = ct * SK2_BN;
    };
    const int n_my = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup
    const int dbg = p.stagger_ticks;             // diagnostics (RECNOW_SK2_DBG): 1 = no epilogue, 2 = no MFMAs, 3 = no operand traffic (timing only: wrong results)

    if (wave < 4) {
        // ------------------------------------------------------------------------------------------------ MFMA role
        const int wm = wave >> 1, wn = wave & 1;
        const int a_off = (lane >> 5) * LDA + wm * 64 + (lane & 31);
        const int b_off = (lane >> 5) * LDB + wn * 64 + (lane & 31);
        const int col_l = lane & 31, row_l = 4 * (lane >> 5);
        f32x16 acc[2][2];
        auto ktile = [&](int cur, int npairs) {
            const float* as = smem + cur * BUF + a_off;
            const float* bs = smem + cur * BUF + A_SZ + b_off;
            float a0[2], b0[2], a1[2], b1[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a0[i] = FR(as + i * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[j] = FR(bs + j * 32);
#pragma unroll
            for (int kk = 0; kk < SK2_BK; kk += 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a1[i] = FR(as + (kk + 2) * LDA + i * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j) b1[j] = FR(bs + (kk + 2) * LDB + j * 32);
                __builtin_amdgcn_sched_barrier(0);
                if (kk / 2 < npairs) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 4 < SK2_BK) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a0[i] = FR(as + (kk + 4) * LDA + i * 32);
#pragma unroll
                    for (int j = 0; j < 2; ++j) b0[j] = FR(bs + (kk + 4) * LDB + j * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kk / 2 + 1 < npairs) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        __builtin_amdgcn_s_setprio(2);            // MFMA issue ahead of the memory role's instructions on the same SIMD
        __syncthreads();                          // k-tile 0 of the first tile is in buffer 0
        int f = 0;
        for (int it = 0; it <= n_my; ++it) {
            const bool have = it < n_my;          // workgroup-uniform; the last iteration only drains the memory role's epilogue
            if (have) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            }
#pragma unroll
            for (int t = 0; t < SK2_NK; ++t) {
                if (have && dbg != 2) ktile((f + t) & 1, t == SK2_NK - 1 ? p.tail_pairs : SK2_BK / 2);
                if (t == SK2_NK - 1 && have) {
                    // park the tile: the memory role read the previous tile's staging during steps 0 .. 7 (behind the barrier of step 7)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            float* cp = Cs + (wm * 64 + i * 32 + row_l) * SK2_LDC + wn * 64 + j * 32 + col_l;
#pragma unroll
                            for (int r = 0; r < 16; ++r) FW(cp + ((r & 3) + 8 * (r >> 2)) * SK2_LDC) = acc[i][j][r];
                        }
                }
                __syncthreads();
            }
            f ^= 1;                               // nine k-tiles: the buffer parity flips per tile
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- memory role
    const int mt = tid - 256;
    // operand tiles: thread -> two float4 of A and two of B per k-tile
    int a_r[2], a_k[2], b_r[2], b_k[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = mt + 256 * i;
        a_r[i] = idx >> 2; a_k[i] = (idx & 3) * 4;
        if (B_KC) { b_r[i] = idx >> 2; b_k[i] = (idx & 3) * 4; }
        else { b_r[i] = (idx & 31) * 4; b_k[i] = idx >> 5; }         // b_r = column of the float4, b_k = k row
    }
    unsigned a_boff[2], b_boff[2];               // byte offsets inside a tile (k0 = 0), 32-bit (the host checks the range)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_boff[i] = (unsigned)(((int64_t)a_r[i] * p.lda + a_k[i]) * 4);
        b_boff[i] = B_KC ? (unsigned)(((int64_t)b_r[i] * p.ldb + b_k[i]) * 4) : (unsigned)(((int64_t)b_k[i] * p.ldb + b_r[i]) * 4);
    }
    // register sets: one per k-tile index of a tile (compile-time indices in the unrolled step loop; a set is live from its request to its
    // LDS write, SK2_LK steps later, so ~SK2_LK + 1 sets hold data at any time)
    f32x4 ra[SK2_NK][2], rb[SK2_NK][2];
    auto ring_load = [&](int set, int mm, int nn, int kt) {
        const char* __restrict__ pa = reinterpret_cast<const char*>(p.A + (int64_t)mm * p.lda + kt * SK2_BK);
        const char* __restrict__ pb = reinterpret_cast<const char*>(B_KC ? p.B + (int64_t)nn * p.ldb + kt * SK2_BK : p.B + (int64_t)kt * SK2_BK * p.ldb + nn);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            asm volatile("" : "+v"(a_boff[i]));
            ra[set][i] = *reinterpret_cast<const f32x4*>(pa + a_boff[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            asm volatile("" : "+v"(b_boff[i]));
            rb[set][i] = *reinterpret_cast<const f32x4*>(pb + b_boff[i]);
        }
    };
    auto ring_store = [&](int set, int buf) {
        float* As = smem + buf * BUF;
        float* Bs = As + A_SZ;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* s = As + a_k[i] * LDA + a_r[i];
            FW(s) = ra[set][i].x; FW(s + LDA) = ra[set][i].y; FW(s + 2 * LDA) = ra[set][i].z; FW(s + 3 * LDA) = ra[set][i].w;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (B_KC) {
                float* s = Bs + b_k[i] * LDB + b_r[i];
                FW(s) = rb[set][i].x; FW(s + LDB) = rb[set][i].y; FW(s + 2 * LDB) = rb[set][i].z; FW(s + 3 * LDB) = rb[set][i].w;
            } else {
                *reinterpret_cast<f32x4*>(Bs + b_k[i] * LDB + b_r[i]) = rb[set][i];
            }
        }
    };
    // epilogue: chunk c (0 .. 15) of a tile = rows c * 8 + (mt >> 5), columns (mt & 31) * 4 .. + 3; a slice = two chunks
    const int e_row = mt >> 5, e_col = (mt & 31) * 4;
    f32x4 ev[8][2], cv[8][2], fv[8][2], dv[8][2];        // one slot per slice, requested SK2_EK steps ahead
    float rs[8][2];
    auto ep_issue = [&](int slotr, int q, int m0, int n0) {      // request the epilogue operands of slice q of tile (m0, n0)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t row = m0 + (2 * q + h) * 8 + e_row;
            const int64_t col = n0 + e_col;
            if (EP & 1) ev[slotr][h] = *reinterpret_cast<const f32x4*>(p.emul + row * p.lde + col);
            if (EP & 2) cv[slotr][h] = *reinterpret_cast<const f32x4*>(p.C + row * p.ldc + col);
            if (DUAL == 2 || DUAL == 4) fv[slotr][h] = *reinterpret_cast<const f32x4*>(p.E2 + row * p.lde2 + col);
            if (DUAL == 2) dv[slotr][h] = *reinterpret_cast<const f32x4*>(p.C2 + row * p.ldc2 + col);
            if (DUAL == 4) {
                dv[slotr][h] = *reinterpret_cast<const f32x4*>(p.E3 + row * p.lde3 + col);
                rs[slotr][h] = p.rv[row];
            }
        }
    };
    f32x4 hv4 = mk4(0.f, 0.f, 0.f, 0.f);
    auto ep_process = [&](int slotr, int q, int m0, int n0) {    // slice q of the parked tile -> global
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rl = (2 * q + h) * 8 + e_row;
            const int64_t row = m0 + rl;
            const int64_t col = n0 + e_col;
            const f32x4 a = *reinterpret_cast<const f32x4*>(Cs + rl * SK2_LDC + e_col);
            f32x4 v = a;
            if (EP & 1) v = v * ev[slotr][h];
            if (EP & 2) v = v + cv[slotr][h];
            if (DUAL != 3) *reinterpret_cast<f32x4*>(p.C + row * p.ldc + col) = v;
            if (DUAL == 1 || (DUAL == 3 && p.C2 != nullptr)) *reinterpret_cast<f32x4*>(p.C2 + row * p.ldc2 + col) = a;
            if (DUAL == 2) *reinterpret_cast<f32x4*>(p.C2 + row * p.ldc2 + col) = dv[slotr][h] + a * fv[slotr][h];
            if (DUAL == 4) *reinterpret_cast<f32x4*>(p.C2 + row * p.ldc2 + col) = a * fv[slotr][h] + (hv4 * rs[slotr][h]) * dv[slotr][h];
            if (DUAL == 3) {                      // row-dot with the head vector: two partials per column tile (columns 0-63 | 64-127)
                const f32x4 tq = v * hv4;
                float s = (tq.x + tq.y) + (tq.z + tq.w);
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                s += __shfl_xor(s, 8, 64);
                if ((mt & 15) == 0) p.hp[row * p.hp_ld + 2 * (n0 / SK2_BN) + ((mt & 31) >> 4)] = s;
            }
        }
    };

    int m0, n0;
    tile_of(blockIdx.x, m0, n0);
#pragma unroll
    for (int kt = 0; kt < SK2_LK; ++kt) ring_load(kt, m0, n0, kt);
    ring_store(0, 0);
    __syncthreads();
    int f = 0;
    int pm0 = m0, pn0 = n0;                       // the parked (previous) tile
    for (int it = 0; it <= n_my; ++it) {
        const bool have = it < n_my, prev = it > 0;          // workgroup-uniform
        int m0n = m0, n0n = n0;                   // the next tile (a workgroup without one re-reads its own: the loads stay unconditional)
        if (it + 1 < n_my) tile_of(blockIdx.x + (it + 1) * gridDim.x, m0n, n0n);
        if (prev && (DUAL == 3 || DUAL == 4)) hv4 = *reinterpret_cast<const f32x4*>((DUAL == 3 ? p.hv : p.cv) + pn0 + e_col);
#pragma unroll
        for (int t = 0; t < SK2_NK; ++t) {
            // operands SK2_LK k-tiles ahead (a load takes 2-3 us beside the epilogue streams, a k-tile of MFMAs ~1 us, and with one
            // workgroup per CU nothing else covers a late operand): k-tile t + LK of this tile, or of the next one
            if (t + SK2_LK < SK2_NK) ring_load(t + SK2_LK, m0, n0, t + SK2_LK);
            else ring_load(t + SK2_LK - SK2_NK, m0n, n0n, t + SK2_LK - SK2_NK);
            // epilogue operands SK2_EK slices ahead: slices EK .. 7 of the parked tile (steps 0 .. 7 - EK), then slices 0 .. EK - 1 of the
            // tile in flight (steps 9 - EK .. 8), whose epilogue starts with the next iteration
            if ((EP != 0 || DUAL == 2 || DUAL == 4) && dbg != 1) {
                if (t + SK2_EK < 8) { if (prev) ep_issue(t + SK2_EK, t + SK2_EK, pm0, pn0); }
                else if (t >= SK2_NK - SK2_EK) { if (have) ep_issue(t - (SK2_NK - SK2_EK), t - (SK2_NK - SK2_EK), m0, n0); }
            }
            if (t < 8 && prev && dbg != 1) ep_process(t, t, pm0, pn0);
            if (dbg != 3) ring_store((t + 1) % SK2_NK, (f + t + 1) & 1);
            __syncthreads();
        }
        f ^= 1;
        pm0 = m0; pn0 = n0;
        m0 = m0n; n0 = n0n;
    }
}

template <bool B_KC, int EP, int DUAL>
static int launch_sk2(const GemmK& k, hipStream_t st) {
    constexpr int LDA = SK2_BM + 1, LDB = B_KC ? SK2_BN + 1 : SK2_BN;
    constexpr int BUF = (SK2_BK * LDA + SK2_BK * LDB + 3) / 4 * 4;
    constexpr size_t lds = (size_t)(2 * BUF + SK2_BM * SK2_LDC) * sizeof(float);
    const int rt = k.M / SK2_BM, ct = k.N / SK2_BN;
    int cus = 256;
    const int grid = rt * ct < cus ? rt * ct : cus;
    const int xcd = (rt % 8 == 0 && grid % 8 == 0) ? 1 : 0;
    static bool raised = false;
    if (!raised) {
        RN_HIP(hipFuncSetAttribute((const void*)k_gemm_sk2<B_KC, EP, DUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    GemmK kk = k;
    static const int dbg = []() { const char* e = getenv("RECNOW_SK2_DBG"); return e ? atoi(e) : 0; }();
    kk.stagger_ticks = dbg;
    hipLaunchKernelGGL((k_gemm_sk2<B_KC, EP, DUAL>), grid, 512, lds, st, kk, rt, ct, xcd);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// The products of the DCN-v2 step that k_gemm_shortk runs with its ring schedule (K = 144); anything else: RECNOW_EUNSUPPORTED.
int rn_gemm_launch_sk2(const GemmK& k, bool b_kc, int ep, int c2_mode, hipStream_t st) {
    if (k.K != SK2_NK * SK2_BK || k.M % SK2_BM || k.N % SK2_BN) return RECNOW_EUNSUPPORTED;
    if (!b_kc && ep == 1) {
        if (c2_mode == 0) return launch_sk2<false, 1, 0>(k, st);
        if (c2_mode == 1) return launch_sk2<false, 1, 1>(k, st);
        if (c2_mode == 3) return launch_sk2<false, 1, 3>(k, st);
    }
    if (b_kc && ep == 0) {
        if (c2_mode == 2) return launch_sk2<true, 0, 2>(k, st);
        if (c2_mode == 4) return launch_sk2<true, 0, 4>(k, st);
    }
    if (b_kc && ep == 2 && c2_mode == 0) return launch_sk2<true, 2, 0>(k, st);
    return RECNOW_EUNSUPPORTED;
}
