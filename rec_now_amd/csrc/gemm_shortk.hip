// Persistent short-K GEMM:  C[M][N] = (A[M][K] B) [* emul] [+ C],  K <= 512, exact fp32 on v_mfma_f32_32x32x2_f32.
//
// The K = N*S+N (= 130 -> 144) deep products of DCNMixLayer (reference rec_now/layers/dcn_mix_layer.py:141-143 and their
// backward) have 4096 output tiles of 128x128 with only nine 16-deep k-tiles each.  Launched as one workgroup per tile
// (k_gemm) every tile pays an exposed prologue (first operand fetch, ~3 us) and epilogue (4-16 us) around a ~31 us main loop
// that shares the CU's MFMA pipe four ways; per-workgroup timestamps (tools/gemm_trace.py) show the four co-resident
// workgroups of a CU running those phases in lockstep, so the pipe idles a quarter of the time.
// Here the grid is one wave of co-resident workgroups (4 per CU) that each walk a list of tiles:
//   * the next tile's first k-tile is fetched under the current tile's last MFMA step and parked in the free LDS buffer, so
//     a workgroup goes from epilogue straight into MFMAs: there is no prologue after the first tile;
//   * the epilogue requests the operands of sub-tile s+1 before it stages and stores sub-tile s, and never waits on a store;
//   * tiles are handed out XCD-aware: workgroup b runs on XCD b % 8 (round-robin dispatch), and the 8 column tiles of one
//     row tile go to consecutive workgroups of the SAME XCD, so the A row-panel is fetched into one L2 instead of eight.
//   * nine-k-tile products (K = 144, DCN-v2) run the RING schedule (NK = 9 below): operands two k-tiles ahead in three rotating
//     register sets, the k-loop unrolled, every wait an exact count.
#include "gemm_split.hpp"

#define SK_BM 128
#define SK_BN 128
#define SK_BK 16
#define SK_STAGGER_US_DEFAULT 0      // phase stagger of the co-resident workgroups of a CU (see the kernel); 0 = off

// EP: bit 0 = C *= emul, bit 1 = C += C_old.   DUAL: 0 none, 1: C2 = acc, 2: C2 += acc * E2,
// 3 (scoring head folded into the last cross layer): C2 = acc, C = acc * emul is NOT stored, its row-dot with hv leaves as
//   partials hp[m][2 * column tile + wave column] (fixed order, summed by the consumer);
// 4 (first dx write of the backward): C = acc, C2 = acc * E2 + rv[m] * cv[n] * E3[m][n] (C2 is written, never read).
// 5 / 6 (round 5: the input gradient accumulated ONCE, by the last product of the backward pass): C = acc + E2 * E3 [+ E4 * E5] + rv[m] * cv[n] * E6
//   (6: without the bracket); three / five (B, D) tensors read once, C written once.  Their epilogue moves in HALF sub-tiles (16 rows): two
//   register sets of 2 x 5 float4 instead of 4 x 5 per set, the loads of half s + 1 requested before half s is combined and stored.
// Workgroups per CU (= waves per SIMD): the epilogue's operand registers set it.  These products are HBM-bound, three or two
// co-resident workgroups still cover each other's epilogues; spilling the epilogue operands to scratch does not.
// PRE: the whole tile of `emul` (64 registers per lane) is requested BEFORE the tile's k-loop, so it lands under the MFMAs and the
// epilogue never waits for a load (EP == 1 products: y = x * (T2g [W; b]) streams x).
// The epilogue's streamed operands (emul, old C, E2, E3: read once per launch) and its plain outputs move as NON-TEMPORAL accesses (round 4; A/B on one
// box, tools/ab_lib.sh: step 3.526 -> 3.492 ms, 213 -> 208 us per launch); -DRN_SK_PLAIN (tools/build_variant.py) builds the plain-access variant.  The ring's
// operand loads stay plain: the weights are re-read by every workgroup from L2.
#ifndef RN_SK_PLAIN
#define SK_LDS4(ptr) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr))
#define SK_STS4(ptr, v) __builtin_nontemporal_store((v), reinterpret_cast<f32x4*>(ptr))
#else
#define SK_LDS4(ptr) (*reinterpret_cast<const f32x4*>(ptr))
#define SK_STS4(ptr, v) (*reinterpret_cast<f32x4*>(ptr) = (v))
#endif
// SPL (round 6): the split-precision form of the ring schedule -- A (fp32, [row][k]) is split into three bf16 pieces on its way into LDS, B comes as
// piece planes (rn_split_planes of the packed weights, once per launch), a k-tile is six groups of four v_mfma_f32_32x32x16_bf16 (the arithmetic of
// gemm_split.hip: per-product error <= 2^-25, fp32 accumulation); 216 8-pass MFMAs per wave and tile instead of 260 16-pass ones.  The accumulator
// layout is the fp32 kernel's, so every epilogue below is shared.  LDS: two stages of 2 x 3 planes (50 688 B): two workgroups per CU.
__host__ __device__ constexpr int sk_wg_per_cu(int EP, int DUAL, bool PRE, int NK = 0, bool SPL = false) {
    if (SPL) return 2;
    return (DUAL >= 4 || PRE || (NK > 0 && DUAL == 2)) ? 2 : (EP == 3 || DUAL == 2 || DUAL == 3 || NK > 0) ? 3 : 4;      // ring schedule: 32 more operand registers
}

// NK > 0 (the depth in k-tiles as a template parameter, NK % 3 == 0): the RING schedule.  Per-workgroup timestamps (tools/gemm_trace.py)
// showed the k-loop of the NK = 0 schedule latency-bound: operands of k-tile t+1 are requested at the top of k-tile t and needed 32
// MFMAs (0.9 us) later, while a load takes 2-3 us next to the epilogues' HBM streams -- the MFMA pipe of a CU was 58 % busy with two
// workgroups in their k-loops.  Here three register sets rotate (set = k-tile % 3): k-tile t+2 is requested at the top of k-tile t, so a
// load has two k-tiles to land; the sets carry on across tiles (the last two k-tiles of a tile request k-tiles 0 and 1 of the next
// one), every load and LDS write is unconditional (a workgroup without a next tile re-reads its own), and the k-loop is unrolled, so
// every `s_waitcnt vmcnt` is an exact count.  PRE: the tile of `emul` is requested in four groups at k-tiles 0, 2, 4, 6,
// each behind that k-tile's ring loads, so no ring wait stands behind a load younger than two k-tiles.
template <bool B_KC, int EP, int DUAL, bool PRE = false, int NK = 0, bool SPL = false>
__global__ void __launch_bounds__(GEMM_THREADS, sk_wg_per_cu(EP, DUAL, PRE, NK, SPL))
k_gemm_shortk(const GemmK p, int row_tiles, int col_tiles, int xcd_aware, const char* __restrict__ b_planes, int64_t b_plane_bytes) {
    static_assert(!SPL || NK >= 3, "split precision: the ring schedule only");
    static_assert(!PRE || ((EP == 1 || EP == 2) && DUAL != 2 && DUAL != 4), "whole-tile prefetch: emul (EP 1) or the old C (EP 2)");
    static_assert(NK % 3 == 0, "ring schedule: three register sets, the same set holds k-tile 0 of every tile");
    static_assert(!PRE || NK == 0 || NK >= 7, "ring schedule: the four groups of the emul tile go out at k-tiles 0, 2, 4, 6");
    using TA = Tile<SK_BM, SK_BK, true>;
    using TB = Tile<SK_BN, SK_BK, B_KC>;
    constexpr int A_SZ = SK_BK * TA::LD, B_SZ = SK_BK * TB::LD, BUF = SPL ? SPL_STAGE / 4 : A_SZ + B_SZ;      // floats per LDS stage
    static_assert(4 * 16 * 36 <= BUF, "half sub-tile staging of 4 waves fits one operand buffer");
    extern __shared__ __attribute__((aligned(16))) float smem[];       // [A0 | B0 | A1 | B1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int a_off = (lane >> 5) * TA::LD + wm * 64 + (lane & 31);
    const int b_off = (lane >> 5) * TB::LD + wn * 64 + (lane & 31);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    const int rr0 = lane >> 3, cc = (lane & 7) * 4;
    const unsigned e_lane = (unsigned)((wm * 64 + rr0) * p.lde + wn * 64 + cc);     // this lane's element inside a tile of emul / C
    const unsigned c_lane = (unsigned)((wm * 64 + rr0) * p.ldc + wn * 64 + cc);
    const unsigned f_lane = (unsigned)((wm * 64 + rr0) * p.lde2 + wn * 64 + cc);
    const unsigned d_lane = (unsigned)((wm * 64 + rr0) * p.ldc2 + wn * 64 + cc);
    const unsigned g_lane = (unsigned)((wm * 64 + rr0) * p.lde3 + wn * 64 + cc);
    const int nk = p.K / SK_BK;
    const int ntiles = row_tiles * col_tiles;
    constexpr bool EARLY = NK >= 3 && ((EP && !PRE) || DUAL == 2 || DUAL == 4);       // ring schedule with epilogue loads
    constexpr bool ONCE = DUAL == 5 || DUAL == 6;                                     // C = acc + E2 * E3 [+ E4 * E5] + rv (x) cv * E6
    static_assert(!ONCE || (EP == 0 && NK >= 3), "the one-go input gradient: ring schedule, no emul / accumulate");

    // slot (= workgroup index + round * grid) -> tile.  XCD-aware: slots of one XCD enumerate (row tile, column tile) with
    // the column tile fastest; row tiles are dealt round-robin to the 8 XCDs.
    auto tile_of = [&](int slot, int& m0, int& n0) {
        int rt, ct;
        if (xcd_aware) {
            const int xcd = slot & 7, idx = slot >> 3;
            rt = (idx / col_tiles) * 8 + xcd;
            ct = idx % col_tiles;
        } else {
            rt = slot / col_tiles;
            ct = slot % col_tiles;
        }
        m0 = rt * SK_BM;
        n0 = ct * SK_BN;
    };

    TA ta;
    TB tb;
    ta.init(p.lda);
    tb.init(p.ldb);
    int slot = blockIdx.x;
    if (slot >= ntiles) return;
    // Phase stagger.  The co-resident workgroups of a CU run identical work, so without it they move in lockstep: all of them in
    // the k-loop (sharing the MFMA pipe), then all of them in the epilogue (the pipe idle, HBM hit by every CU at once).  Each
    // workgroup takes an arrival number on its CU (XCC id + SE/SH/CU id of HW_ID) and starts that many delay units late; the
    // counter is given back at exit, so the buffer stays zero between launches.  Every wave leaves the bounded wait loop.
    int cu_key = -1;
    if (p.cu_slots != nullptr && p.stagger_ticks > 0) {
        __shared__ int s_arrival;
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));        // HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));       // XCC_ID
            cu_key = (int)(((xcc & 15u) << 8) | ((hw >> 8) & 255u));                                 // CU_ID[11:8], SH_ID[12], SE_ID[15:13]
            s_arrival = atomicAdd(&p.cu_slots[cu_key], 1);
        }
        __syncthreads();
        const int arrival = s_arrival & 7;
        const long long t0 = wall_clock64(), wait = (long long)arrival * p.stagger_ticks;
        while (wall_clock64() - t0 < wait && wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(16);     // <= 200 us, always ends
    }
    int m0, n0;
    tile_of(slot, m0, n0);
    // ring schedule: operand registers of the three sets (set s = k-tile % 3)
    f32x4 ra[NK ? 3 : 1][TA::NV], rb[(NK && !SPL) ? 3 : 1][TB::NV];
    u32x4 rp[SPL ? 3 : 1][3];           // split precision: the three piece units of B
    // split precision: this thread's unit (8 consecutive k of one row) of each operand: A (row = tid >> 1, octet = tid & 1: adjacent lanes cover
    // 64 contiguous bytes of a row), B planes (column = tid & 127, octet = tid >> 7: a wave copies 1 KiB)
    unsigned sa_bo = (unsigned)(((threadIdx.x >> 1) * p.lda + 8 * (threadIdx.x & 1)) * 4);
    unsigned sb_bo = (unsigned)((((threadIdx.x >> 7) * p.N) + (threadIdx.x & 127)) * 16);
    const int sa_soff = ((threadIdx.x & 1) * SPL_PLANE_H + (threadIdx.x >> 1)) * 16;
    const int sb_soff = SPL_OPER + ((threadIdx.x >> 7) * SPL_PLANE_H + (threadIdx.x & 127)) * 16;
    const int sfa_off = ((lane >> 5) * SPL_PLANE_H + wm * 64 + (lane & 31)) * 16;
    const int sfb_off = SPL_OPER + ((lane >> 5) * SPL_PLANE_H + wn * 64 + (lane & 31)) * 16;
    auto ring_load = [&](int set, int mm, int nn, int k0) {
        if constexpr (SPL) {
            const char* __restrict__ pa = reinterpret_cast<const char*>(p.A + (int64_t)mm * p.lda + k0);
            const char* __restrict__ pb = b_planes + ((int64_t)(k0 >> 3) * p.N + nn) * 16;
            asm volatile("" : "+v"(sa_bo));
            ra[set][0] = *reinterpret_cast<const f32x4*>(pa + sa_bo);
            ra[set][1] = *reinterpret_cast<const f32x4*>(pa + sa_bo + 16);
            asm volatile("" : "+v"(sb_bo));
#pragma unroll
            for (int q = 0; q < 3; ++q) rp[SPL ? set : 0][q] = *reinterpret_cast<const u32x4*>(pb + q * b_plane_bytes + sb_bo);
        } else {
        const char* __restrict__ pa = reinterpret_cast<const char*>(p.A + TA::tile_base(p.lda, mm, k0));
        const char* __restrict__ pb = reinterpret_cast<const char*>(p.B + TB::tile_base(p.ldb, nn, k0));
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) {           // SGPR base + opaque 32-bit byte offset: see Tile::issue
            asm volatile("" : "+v"(ta.boff[i]));
            ra[set][i] = *reinterpret_cast<const f32x4*>(pa + ta.boff[i]);
        }
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) {
            asm volatile("" : "+v"(tb.boff[i]));
            rb[SPL ? 0 : set][i] = *reinterpret_cast<const f32x4*>(pb + tb.boff[i]);
        }
        }
    };
    auto ring_store = [&](int set, float* S) {
        if constexpr (SPL) {
            char* Sc = reinterpret_cast<char*>(S);
            const float x[8] = {ra[set][0].x, ra[set][0].y, ra[set][0].z, ra[set][0].w, ra[set][1].x, ra[set][1].y, ra[set][1].z, ra[set][1].w};
            u32x4 w[3];
            spl_split8(x, w);
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(Sc + sa_soff + q * SPL_PLANE) = w[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(Sc + sb_soff + q * SPL_PLANE) = rp[SPL ? set : 0][q];
        } else {
#pragma unroll
        for (int i = 0; i < TA::NV; ++i) { ta.v[i] = ra[set][i]; ta.store_slot(i, S); }
#pragma unroll
        for (int i = 0; i < TB::NV; ++i) { tb.v[i] = rb[SPL ? 0 : set][i]; tb.store_slot(i, S + A_SZ); }
        }
    };
    if (NK) {
        ring_load(0, m0, n0, 0);
        ring_load(1, m0, n0, SK_BK);
        ring_store(0, smem);
    } else {
        ta.template load_fast<false>(p.A, nullptr, 0, 0, p.lda, m0, 0);
        tb.template load_fast<false>(p.B, nullptr, 0, 0, p.ldb, n0, 0);
        ta.store(smem);
        tb.store(smem + A_SZ);
    }
    __syncthreads();
    int f = 0;                                     // buffer holding k-tile 0 of the current tile
    for (; slot < ntiles; slot += gridDim.x) {
        const int nslot = slot + gridDim.x;
        const bool has_next = nslot < ntiles;
        int m0n = NK ? m0 : 0, n0n = NK ? n0 : 0;       // ring: a workgroup without a next tile re-reads its own (loads stay unconditional)
        if (has_next) tile_of(nslot, m0n, n0n);
#ifdef RN_GEMM_TRACE
#define SK_TR(i) do { if (p.trace && threadIdx.x == 0) p.trace[(long long)slot * 8 + (i)] = wall_clock64(); } while (0)
        if (p.trace && threadIdx.x == 0) {
            p.trace[(long long)slot * 8 + 5] = clock64();
            p.trace[(long long)slot * 8 + 4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11)) |
                                               ((long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) << 32);
        }
#else
#define SK_TR(i) do { } while (0)
#endif
        SK_TR(0); SK_TR(1);
        // k-loop: its fragment reads and MFMAs go ahead of the co-resident workgroups' epilogue instructions (issue is arbitrated by
        // priority, then age: MI355X_MICROARCH.md, 'Two waves per SIMD'); measured 227 -> 214 us per launch
        if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        f32x4 evt[PRE ? 4 : 1][4];
        const float* Etp = !PRE ? nullptr : EP == 1 ? p.emul + (int64_t)m0 * p.lde + n0 : p.C + (int64_t)m0 * p.ldc + n0;
        const int64_t ldp = EP == 1 ? p.lde : p.ldc;
        auto pre_load = [&](int s2) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                evt[PRE ? s2 : 0][q] = SK_LDS4(Etp + (int64_t)((s2 >> 1) * 32 + q * 8) * ldp + (s2 & 1) * 32 + (EP == 1 ? e_lane : c_lane));
        };
        // fragment reads and MFMAs of one k-tile in LDS buffer `cur`.  npairs = k-pairs of this k-tile that hold data: all 8, except in
        // the last k-tile of a zero-padded depth (DCN-v2: K = N*S + N = 130 stored as 144 -> 1 pair): the MFMA groups of the padding are
        // skipped behind a scalar branch each (10 % of a tile's MFMAs).  The fragment reads and the schedule of the groups stay as they are.
        // split precision: the staging of the next k-tile (register set `nset` -> the free stage Sn: split of A, six 16-byte writes) is cut into
        // pieces behind the MFMA groups of this one (nset < 0: nothing to stage)
        auto ktile = [&](int cur, int npairs, int nset = -1, float* Sn = nullptr) {
            if constexpr (SPL) {       // (the zero padding of the depth holds zeros in both operands: every k-tile runs whole)
                const char* S = reinterpret_cast<const char*>(smem + cur * BUF);
                char* Sc = reinterpret_cast<char*>(Sn);
                bf16x8 af[3][2], bf[3][2];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        af[q][i] = *reinterpret_cast<const bf16x8*>(S + q * SPL_PLANE + sfa_off + i * 512);
                        bf[q][i] = *reinterpret_cast<const bf16x8*>(S + q * SPL_PLANE + sfb_off + i * 512);
                    }
                u32x4 w[3];
                const int ns = nset < 0 ? 0 : nset;
                auto pairs = [&](int e0) {
                    const float x[4] = {e0 ? ra[ns][1].x : ra[ns][0].x, e0 ? ra[ns][1].y : ra[ns][0].y, e0 ? ra[ns][1].z : ra[ns][0].z, e0 ? ra[ns][1].w : ra[ns][0].w};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        unsigned p1, p2, p3;
                        spl_split2(x[2 * e], x[2 * e + 1], p1, p2, p3);
                        w[0][e0 + e] = p1; w[1][e0 + e] = p2; w[2][e0 + e] = p3;
                    }
                };
                __builtin_amdgcn_sched_barrier(0);
#define SKS_TERM(SA, SB)                                                                                              \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SA][i], bf[SB][j], acc[i][j], 0, 0, 0);
                SKS_TERM(0, 0)
                if (nset >= 0) pairs(0);
                __builtin_amdgcn_sched_barrier(0);
                SKS_TERM(0, 1)
                if (nset >= 0) pairs(2);
                __builtin_amdgcn_sched_barrier(0);
                SKS_TERM(1, 0)
                if (nset >= 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(Sc + sa_soff + q * SPL_PLANE) = w[q];
                }
                __builtin_amdgcn_sched_barrier(0);
                SKS_TERM(1, 1)
                if (nset >= 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(Sc + sb_soff + q * SPL_PLANE) = rp[SPL ? ns : 0][q];
                }
                __builtin_amdgcn_sched_barrier(0);
                SKS_TERM(0, 2) SKS_TERM(2, 0)
#undef SKS_TERM
                return;
            }
            const float* as = smem + cur * BUF + a_off;
            const float* bs = smem + cur * BUF + A_SZ + b_off;
            float a0[2], b0[2], a1[2], b1[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a0[i] = FR(as + i * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[j] = FR(bs + j * 32);
#pragma unroll
            for (int kk = 0; kk < SK_BK; kk += 4) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a1[i] = FR(as + (kk + 2) * TA::LD + i * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j) b1[j] = FR(bs + (kk + 2) * TB::LD + j * 32);
                __builtin_amdgcn_sched_barrier(0);
                if (kk / 2 < npairs) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 4 < SK_BK) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a0[i] = FR(as + (kk + 4) * TA::LD + i * 32);
#pragma unroll
                    for (int j = 0; j < 2; ++j) b0[j] = FR(bs + (kk + 4) * TB::LD + j * 32);
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
        // Epilogue addresses = (tile- and sub-tile-uniform 64-bit base, kept in SGPRs) + (lane offset, 32-bit, tile-invariant).
        f32x4 ev[2][4], cv[2][4], fv[2][4], dv[2][4];
        float rs[2][4];                     // DUAL 4: the rank-one factor's row values of the sub-tile's four row groups
        f32x4 hv4[2];                       // DUAL 3 / 4: the column vector of the two column sub-tiles
        float hs[4];                        // DUAL 3: row-dot sums of the current row sub-tile (rows 16h + 8q + rr0)
        const float* Et = (EP & 1) ? p.emul + (int64_t)m0 * p.lde + n0 : nullptr;
        float* Ct = p.C + (int64_t)m0 * p.ldc + n0;
        const float* Ft = (DUAL == 2 || DUAL == 4) ? p.E2 + (int64_t)m0 * p.lde2 + n0 : nullptr;
        const float* Gt = (DUAL == 4) ? p.E3 + (int64_t)m0 * p.lde3 + n0 : nullptr;
        float* Dt = DUAL ? p.C2 + (int64_t)m0 * p.ldc2 + n0 : nullptr;
        auto issue = [&](int s2, int buf) {
            const int i = s2 >> 1, j = s2 & 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if ((EP & 1) && !PRE) ev[buf][q] = SK_LDS4(Et + (int64_t)(i * 32 + q * 8) * p.lde + j * 32 + e_lane);
                if ((EP & 2) && !PRE) cv[buf][q] = SK_LDS4(Ct + (int64_t)(i * 32 + q * 8) * p.ldc + j * 32 + c_lane);
                if (DUAL == 2 || DUAL == 4) fv[buf][q] = SK_LDS4(Ft + (int64_t)(i * 32 + q * 8) * p.lde2 + j * 32 + f_lane);
                if (DUAL == 2) dv[buf][q] = SK_LDS4(Dt + (int64_t)(i * 32 + q * 8) * p.ldc2 + j * 32 + d_lane);
                if (DUAL == 4) {
                    dv[buf][q] = SK_LDS4(Gt + (int64_t)(i * 32 + q * 8) * p.lde3 + j * 32 + g_lane);
                    rs[buf][q] = p.rv[m0 + wm * 64 + i * 32 + q * 8 + rr0];
                }
            }
        };
        // ONCE: operands of half sub-tile `step` = 2 * s2 + h (rows i * 32 + 16 h + 8 q + rr0, q = 0, 1) into register set `buf`
        f32x4 o5[ONCE ? 2 : 1][5][2];
        float r5[ONCE ? 2 : 1][2];
        const float* F1t = ONCE ? p.E2 + (int64_t)m0 * p.lde2 + n0 : nullptr;
        const float* G1t = ONCE ? p.E3 + (int64_t)m0 * p.lde2 + n0 : nullptr;
        const float* F2t = DUAL == 5 ? p.E4 + (int64_t)m0 * p.lde2 + n0 : nullptr;
        const float* G2t = DUAL == 5 ? p.E5 + (int64_t)m0 * p.lde2 + n0 : nullptr;
        const float* H6t = ONCE ? p.E6 + (int64_t)m0 * p.lde2 + n0 : nullptr;
        auto issue5 = [&](int step, int buf) {
            const int s2 = step >> 1, h = step & 1, i = s2 >> 1, j = s2 & 1;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t off = (int64_t)(i * 32 + 16 * h + q * 8) * p.lde2 + j * 32 + f_lane;
                o5[ONCE ? buf : 0][0][q] = SK_LDS4(F1t + off);
                o5[ONCE ? buf : 0][1][q] = SK_LDS4(G1t + off);
                if (DUAL == 5) {
                    o5[ONCE ? buf : 0][2][q] = SK_LDS4(F2t + off);
                    o5[ONCE ? buf : 0][3][q] = SK_LDS4(G2t + off);
                }
                o5[ONCE ? buf : 0][4][q] = SK_LDS4(H6t + off);
                r5[ONCE ? buf : 0][q] = p.rv[m0 + wm * 64 + i * 32 + 16 * h + q * 8 + rr0];
            }
        };
        if (NK) {
#pragma unroll
            for (int t = 0; t < (NK ? NK : 1); ++t) {
                const int cur = (f + t) & 1;
                // k-tile t+2 of this tile, or k-tile 0 / 1 of the next one, into the set that k-tile t left free
                if (t + 2 < NK) ring_load((t + 2) % 3, m0, n0, (t + 2) * SK_BK);
                else ring_load((t + 2) % 3, m0n, n0n, (t + 2 - NK) * SK_BK);
                if (PRE && (t & 1) == 0 && t < 8) pre_load(t >> 1);
                // epilogue operands of sub-tile 0: requested three k-tiles early (behind this k-tile's ring loads: the ring wait two k-tiles on
                // is the first one that stands behind them), so the epilogue starts on landed data
                if (EARLY && t == NK - 3) issue(0, 0);
                if (ONCE && t == NK - 3) issue5(0, 0);
                if constexpr (SPL) {
                    ktile(cur, SK_BK / 2, (t + 1) % 3, smem + (cur ^ 1) * BUF);
                } else {
                    if (t == NK - 1) ktile(cur, p.tail_pairs);
                    else ktile(cur, SK_BK / 2);
                    ring_store((t + 1) % 3, smem + (cur ^ 1) * BUF);
                }
                __syncthreads();
            }
        } else {
            if (PRE) {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) pre_load(s2);
            }
            for (int t = 0; t < nk; ++t) {
                const int cur = (f + t) & 1;
                const bool more = t + 1 < nk;
                if (more) {
                    ta.template load_fast<false>(p.A, nullptr, 0, 0, p.lda, m0, (t + 1) * SK_BK);
                    tb.template load_fast<false>(p.B, nullptr, 0, 0, p.ldb, n0, (t + 1) * SK_BK);
                } else if (has_next) {                 // next tile's first k-tile flies under this tile's last MFMAs
                    ta.template load_fast<false>(p.A, nullptr, 0, 0, p.lda, m0n, 0);
                    tb.template load_fast<false>(p.B, nullptr, 0, 0, p.ldb, n0n, 0);
                }
                ktile(cur, (t == nk - 1) ? p.tail_pairs : SK_BK / 2);
                if (more || has_next) {
                    ta.store(smem + (cur ^ 1) * BUF);
                    tb.store(smem + (cur ^ 1) * BUF + A_SZ);
                }
                __syncthreads();
            }
        }
        SK_TR(2);
        if (p.prio) __builtin_amdgcn_s_setprio(0);
        // buffer `fr` was consumed by the last k-tile and is free: staging space of the epilogue (16 rows x 36 per wave)
        const int fr = (f + nk - 1) & 1;
        float* stg = smem + fr * BUF + wave * (16 * 36);
        if (DUAL == 3 || DUAL == 4 || ONCE) {
            const float* colv = (DUAL == 3 ? p.hv : p.cv) + n0 + wn * 64 + cc;
            hv4[0] = *reinterpret_cast<const f32x4*>(colv);
            hv4[1] = *reinterpret_cast<const f32x4*>(colv + 32);
        }
        if (((EP && !PRE) || DUAL == 2 || DUAL == 4) && !EARLY) issue(0, 0);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int i = s2 >> 1, j = s2 & 1, buf = s2 & 1;
            if (((EP && !PRE) || DUAL == 2 || DUAL == 4) && s2 + 1 < 4) issue(s2 + 1, buf ^ 1);
            if (DUAL == 3 && j == 0) hs[0] = hs[1] = hs[2] = hs[3] = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {          // accumulator registers 8h..8h+7 hold rows 16h..16h+15 of the sub-tile
                if (ONCE && 2 * s2 + h + 1 < 8) issue5(2 * s2 + h + 1, (h ^ 1));      // (2 s2 + h) & 1 = h: the sets alternate with h
#pragma unroll
                for (int r = 0; r < 8; ++r) stg[((r & 3) + 8 * (r >> 2) + row_l) * 36 + col_l] = acc[i][j][8 * h + r];
                RN_LDS_WAVE_SYNC();
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(stg + (q * 8 + rr0) * 36 + cc);
                    f32x4 v = a;
                    if (ONCE) {
                        // the order of the former three launches: (g_2 * O_1 + (w_head * ds) * O_2) + g_1 * O_0, then + g_0
                        f32x4 t = (hv4[j] * r5[ONCE ? h : 0][q]) * o5[ONCE ? h : 0][4][q];
                        if (DUAL == 5) t = o5[ONCE ? h : 0][2][q] * o5[ONCE ? h : 0][3][q] + t;
                        t = o5[ONCE ? h : 0][0][q] * o5[ONCE ? h : 0][1][q] + t;
                        v = t + a;
                    }
                    if (EP & 1) v = v * (PRE ? evt[PRE ? s2 : 0][2 * h + q] : ev[buf][2 * h + q]);
                    if (EP & 2) v = v + (PRE ? evt[PRE ? s2 : 0][2 * h + q] : cv[buf][2 * h + q]);
                    if (DUAL != 3) SK_STS4(Ct + (int64_t)(i * 32 + 16 * h + q * 8) * p.ldc + j * 32 + c_lane, v);
                    if (DUAL == 1 || (DUAL == 3 && p.C2 != nullptr)) SK_STS4(Dt + (int64_t)(i * 32 + 16 * h + q * 8) * p.ldc2 + j * 32 + d_lane, a);
                    if (DUAL == 2)
                        *reinterpret_cast<f32x4*>(Dt + (int64_t)(i * 32 + 16 * h + q * 8) * p.ldc2 + j * 32 + d_lane) =
                            dv[buf][2 * h + q] + a * fv[buf][2 * h + q];
                    if (DUAL == 3) {
                        const f32x4 t = v * hv4[j];
                        hs[2 * h + q] += (t.x + t.y) + (t.z + t.w);
                    }
                    if (DUAL == 4)
                        *reinterpret_cast<f32x4*>(Dt + (int64_t)(i * 32 + 16 * h + q * 8) * p.ldc2 + j * 32 + d_lane) =
                            a * fv[buf][2 * h + q] + (hv4[j] * rs[buf][2 * h + q]) * dv[buf][2 * h + q];
                }
                RN_LDS_WAVE_SYNC();
            }
            if (DUAL == 3 && j == 1) {            // both column sub-tiles of row sub-tile i are in: join the 8 lanes of a row
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = hs[r];
                    t += __shfl_xor(t, 1, 64);
                    t += __shfl_xor(t, 2, 64);
                    t += __shfl_xor(t, 4, 64);
                    if ((lane & 7) == 0)
                        p.hp[(int64_t)(m0 + wm * 64 + i * 32 + r * 8 + rr0) * p.hp_ld + 2 * (n0 / SK_BN) + wn] = t;
                }
            }
        }
        SK_TR(3);
#ifdef RN_GEMM_TRACE
        if (p.trace && threadIdx.x == 0) p.trace[(long long)slot * 8 + 6] = clock64();
#endif
        __syncthreads();                            // every wave is out of the staging buffer before k-tile 1 overwrites it
        f = (f + nk) & 1;
        m0 = m0n;
        n0 = n0n;
    }
    if (cu_key >= 0) atomicSub(&p.cu_slots[cu_key], 1);          // thread 0 only (cu_key stays -1 elsewhere)
}

template <bool B_KC, int EP, int DUAL, bool PRE = false, int NK = 0, bool SPL = false>
static int launch_sk(const GemmK& k, hipStream_t st, const char* planes = nullptr) {
    using TA = Tile<SK_BM, SK_BK, true>;
    using TB = Tile<SK_BN, SK_BK, B_KC>;
    constexpr size_t lds = SPL ? (size_t)2 * SPL_STAGE : 2 * SK_BK * (size_t)(TA::LD + TB::LD) * sizeof(float);
    const int rt = k.M / SK_BM, ct = k.N / SK_BN;
    const int resident = 256 * sk_wg_per_cu(EP, DUAL, PRE, NK, SPL);
    int grid = rt * ct < resident ? rt * ct : resident;
    const int xcd = (rt % 8 == 0 && grid % 8 == 0) ? 1 : 0;
    GemmK kk = k;
    // stagger unit (microseconds, RECNOW_SK_STAGGER_US; 0 = off) and the per-CU arrival counters (allocated once, zero between launches)
    static const int stagger_us = []() { const char* e = getenv("RECNOW_SK_STAGGER_US"); return e ? atoi(e) : SK_STAGGER_US_DEFAULT; }();
    static int* cu_slots = nullptr;
    kk.cu_slots = nullptr;
    kk.stagger_ticks = 0;
    static const int prio = []() { const char* e = getenv("RECNOW_SK_PRIO"); return e ? atoi(e) : 1; }();     // 0 = off (A/B switch)
    kk.prio = prio;
    if (stagger_us > 0 && grid >= 512) {
        if (!cu_slots) {
            if (hipMalloc((void**)&cu_slots, 4096 * sizeof(int)) != hipSuccess) return RECNOW_EINVAL;
            RN_HIP(hipMemset(cu_slots, 0, 4096 * sizeof(int)));
        }
        kk.cu_slots = cu_slots;
        kk.stagger_ticks = stagger_us * 100;
    }
    const int64_t pb = (int64_t)(k.K / 8) * k.N * 16;
    hipLaunchKernelGGL((k_gemm_shortk<B_KC, EP, DUAL, PRE, NK, SPL>), grid, GEMM_THREADS, lds, st, kk, rt, ct, xcd, planes, pb);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

size_t rn_gemm_shortk_planes_bytes(int K, int N) { return rn_align((size_t)(K / 8) * N * 16 * 3); }

// Split-precision forms of the six products of the DCN-v2 step (K = 144, ring schedule).  `planes`: rn_gemm_shortk_planes_bytes(K, N) bytes of
// workspace; the packed weights are split into them first.  RECNOW_EUNSUPPORTED: the caller runs the fp32 kernel.
int rn_gemm_launch_shortk_split(const GemmK& k, bool b_kc, int ep, int c2_mode, void* planes, hipStream_t st, const void* ready) {
    if (k.K != 9 * SK_BK || (!planes && !ready) || (int64_t)128 * k.lda >= (1ll << 29)) return RECNOW_EUNSUPPORTED;
    const bool fwd = !b_kc && ep == 1 && (c2_mode == 0 || c2_mode == 1 || c2_mode == 3);
    const bool fwd0 = !b_kc && ep == 0 && c2_mode == 0;
    const bool bwd = b_kc && ((ep == 2 && c2_mode == 0) || (ep == 0 && (c2_mode == 0 || c2_mode == 2 || c2_mode == 4 || c2_mode == 5 || c2_mode == 6)));
    if (!fwd && !fwd0 && !bwd) return RECNOW_EUNSUPPORTED;
    if (!ready) {       // (ready: the caller split the packed weights already, once per step)
        int rc = rn_split_planes(k.B, k.ldb, b_kc ? 1 : 0, k.K, k.N, planes, st);
        if (rc) return rc;
    }
    const char* pl = ready ? (const char*)ready : (const char*)planes;
    if (fwd) {      // (no whole-tile prefetch of emul: its 64 registers are the fragments' here)
        if (c2_mode == 1) return launch_sk<false, 1, 1, false, 9, true>(k, st, pl);
        if (c2_mode == 3) return launch_sk<false, 1, 3, false, 9, true>(k, st, pl);
        return launch_sk<false, 1, 0, false, 9, true>(k, st, pl);
    }
    if (fwd0) return launch_sk<false, 0, 0, false, 9, true>(k, st, pl);
    if (ep == 2) return launch_sk<true, 2, 0, false, 9, true>(k, st, pl);
    if (c2_mode == 2) return launch_sk<true, 0, 2, false, 9, true>(k, st, pl);
    if (c2_mode == 4) return launch_sk<true, 0, 4, false, 9, true>(k, st, pl);
    if (c2_mode == 5) return launch_sk<true, 0, 5, false, 9, true>(k, st, pl);
    if (c2_mode == 6) return launch_sk<true, 0, 6, false, 9, true>(k, st, pl);
    return launch_sk<true, 0, 0, false, 9, true>(k, st, pl);
}

// ep: bit 0 = multiply by emul, bit 1 = accumulate into C.  c2_mode: second output (recnow_gemm_desc).  The caller
// guarantees: M, N multiples of 128, K a multiple of 16, A k-contiguous, every operand 16-byte aligned, batch == 1, no
// bias / activation / transposed store.  Second outputs are instantiated for the two products DCN-v2 uses them in.
int rn_gemm_launch_shortk(const GemmK& k, bool b_kc, int ep, int c2_mode, hipStream_t st) {
    static const bool pre = []() { const char* e = getenv("RECNOW_SK_PRE"); return !e || e[0] != '0'; }();     // A/B switch of the whole-tile emul prefetch
    // nine k-tiles (DCN-v2: K = N*S + N = 130 stored as 144): the ring schedule; RECNOW_SK_RING=0 is the A/B switch
    static const int ring = []() { const char* e = getenv("RECNOW_SK_RING"); return e ? atoi(e) : 31; }();
    if (k.K == 9 * SK_BK && (ring & 1) && pre && !b_kc && ep == 1) {
        if (c2_mode == 1) return launch_sk<false, 1, 1, true, 9>(k, st);
        if (c2_mode == 3) return launch_sk<false, 1, 3, true, 9>(k, st);
        if (c2_mode == 0) return launch_sk<false, 1, 0, true, 9>(k, st);
    }
    if (k.K == 9 * SK_BK && (ring & 1) && !pre && !b_kc && ep == 1) {      // RECNOW_SK_PRE=0: the ring without the whole-tile prefetch, 3 workgroups per CU
        if (c2_mode == 1) return launch_sk<false, 1, 1, false, 9>(k, st);
        if (c2_mode == 3) return launch_sk<false, 1, 3, false, 9>(k, st);
        if (c2_mode == 0) return launch_sk<false, 1, 0, false, 9>(k, st);
    }
    if (k.K == 9 * SK_BK && (ring & 1) && !b_kc && ep == 0 && c2_mode == 0) return launch_sk<false, 0, 0, false, 9>(k, st);      // O_l = T2g [W; b] alone (mix_xless)
    if (k.K == 9 * SK_BK && (ring & 4) && b_kc && ep == 2 && c2_mode == 0)
        return pre ? launch_sk<true, 2, 0, true, 9>(k, st) : launch_sk<true, 2, 0, false, 9>(k, st);
    if (k.K == 9 * SK_BK && b_kc && ep == 0) {
        if (c2_mode == 2 && (ring & 2)) return launch_sk<true, 0, 2, false, 9>(k, st);
        if (c2_mode == 4 && (ring & 8)) return launch_sk<true, 0, 4, false, 9>(k, st);
        if (c2_mode == 5) return launch_sk<true, 0, 5, false, 9>(k, st);
        if (c2_mode == 6) return launch_sk<true, 0, 6, false, 9>(k, st);
        if (c2_mode == 0 && (ring & 16)) return launch_sk<true, 0, 0, false, 9>(k, st);      // g_l alone (the input gradient is accumulated once, c2_mode 5 / 6)
    }
    if (c2_mode >= 5) return RECNOW_EUNSUPPORTED;
    if (pre && !b_kc && ep == 1) {
        if (c2_mode == 1) return launch_sk<false, 1, 1, true>(k, st);
        if (c2_mode == 3) return launch_sk<false, 1, 3, true>(k, st);
        if (c2_mode == 0) return launch_sk<false, 1, 0, true>(k, st);
    }
    if (c2_mode == 1) return (!b_kc && ep == 1) ? launch_sk<false, 1, 1>(k, st) : RECNOW_EUNSUPPORTED;
    if (c2_mode == 2) return (b_kc && ep == 0) ? launch_sk<true, 0, 2>(k, st) : RECNOW_EUNSUPPORTED;
    if (c2_mode == 3) return (!b_kc && ep == 1) ? launch_sk<false, 1, 3>(k, st) : RECNOW_EUNSUPPORTED;
    if (c2_mode == 4) return (b_kc && ep == 0) ? launch_sk<true, 0, 4>(k, st) : RECNOW_EUNSUPPORTED;
    if (b_kc) {
        switch (ep) {
            case 0: return launch_sk<true, 0, 0>(k, st);
            case 1: return launch_sk<true, 1, 0>(k, st);
            case 2: return launch_sk<true, 2, 0>(k, st);
            default: return launch_sk<true, 3, 0>(k, st);
        }
    }
    switch (ep) {
        case 0: return launch_sk<false, 0, 0>(k, st);
        case 1: return launch_sk<false, 1, 0>(k, st);
        case 2: return launch_sk<false, 2, 0>(k, st);
        default: return launch_sk<false, 3, 0>(k, st);
    }
}
