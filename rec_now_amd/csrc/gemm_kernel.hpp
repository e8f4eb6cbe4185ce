// Kernel template of the exact-fp32 MFMA GEMM (see gemm.hip for the design notes).  Included by the instantiation
// translation units gemm_inst_*.hip, which are compiled in parallel.
#pragma once
#include "gemm.hpp"
#include <string.h>
#include <type_traits>

#define GEMM_THREADS 256
#ifndef RN_GLDS_SPREAD
#define RN_GLDS_SPREAD 8      // XF & 32: the LDS-DMAs of a k-tile go out behind its first RN_GLDS_SPREAD MFMA groups (of BK / 2)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native 4-wide vector for the staging registers: HIP's float4 is a STRUCT, and whole-struct copies (global -> regs -> LDS)
// are lowered to llvm.memcpy through a private alloca that SROA cannot promote -> the tile would live in scratch.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


struct GemmK {
    const float *A, *A2, *B, *B2, *bias, *emul;
    float *C, *partial;
    int64_t lda, ldb, ldc, lde, sA, sB, sC, sBias, sE;
    int M, N, K, batch;
    int a_mode, a_act, b_mode, b_act, act, act_cols, e_mode, e_act, accumulate, c_trans, splitk, kchunk;
    int a_hq, b_hq;            // OUTER mode: operand(row, col) = second[row][col / hq] * first[row][col % hq]
    int64_t a_ld2, b_ld2;      // row stride of `second` in OUTER mode
    // XF & 1 -- side product: Cx[m][r] = sum_k A'[m][k] * Bx[k][r] for r < sp_r <= 4, computed on the VALU from the A tile
    // that is already in LDS (A' = A after its operand transform).  Lets a product with N = 128 + 2 run as exactly 128
    // MFMA columns instead of a 160-wide tile.
    const float* bx;
    float* cx;
    int64_t bx_ks, bx_rs, cx_ms, cx_rs;
    int sp_r;
    // XF & 2 -- rank-R epilogue update: v[m][n] += sum_r P[m][r] * Q[r][n] before activation / emul (K = 128 + 2 as 128)
    const float *eu_p, *eu_q;
    int64_t eu_pms, eu_qrs, eu_qns;
    int eu_r;
    int npart;                 // row width of a split-K slab: N, or N + 4 when a side product rides along
    int xcd_remap;             // split-K, one column tile, gridDim.z % 8 == 0: row tiles of a k-slab share an XCD (see k_gemm)
    const float* as_in;        // XF & 4: elementwise side output of the A stream, as_out = A * as_in (layout of A)
    float* as_out;
    float* C2;                 // second output of the short-K kernel (c2_mode 1: C2 = acc; 2: C2 += acc * E2;
    const float* E2;           //   3: C2 = acc, C is NOT written, hp = row-dot partials of acc * emul with hv;
    int64_t ldc2, lde2;        //   4: C2 = acc * E2 + rv[m] * cv[n] * E3, C2 is not read)
    const float* E3;           // c2_mode 4: third epilogue tensor (leading dimension lde3)
    int64_t lde3;
    const float *E4, *E5, *E6; // c2_mode 5 / 6: C = acc + E2 * E3 [+ E4 * E5] + rv (x) cv * E6 (all with the leading dimension lde2)
    const float *rv, *cv;      // c2_mode 4: row / column vectors of the rank-one factor
    const float* hv;           // c2_mode 3: column vector of the fused row-dot (the scoring head's kernel)
    float* hp;                 // c2_mode 3: partials hp[m][hp_ld], entry 2 * column tile + wave column
    int hp_ld;
    int prio;                  // short-K kernel: raise the wave priority inside the k-loop (experiment)
    int direct_store;          // lean epilogue: plain outputs straight from the accumulators (RECNOW_GEMM_DIRECT=0: staged through LDS)
    int perm_s;                // split-K reduce: > 0 stores C[row][c] at C[((c / perm_s) * M + row) * perm_s + c % perm_s] (see recnow_gemm_desc.c_perm_s)
    int tail_pairs;            // short-K kernel: k-pairs of the LAST k-tile that hold data (8 = all; fewer: a zero-padded depth)
    int* cu_slots;             // short-K kernel: per-CU arrival counters of the phase stagger (NULL: no stagger)
    int stagger_ticks;         // delay per arrival slot, in 10 ns ticks of the constant 100 MHz clock
    long long* trace;          // RN_GEMM_TRACE builds only: 8 int64 per workgroup (phase timestamps, HW id)
    // XF & 16 -- fused sub-space forward of DCNMixLayer behind a TRANSPOSED GEMM1 (see recnow_gemm_desc.mid_V)
    const float* mid_V;
    float *mid_T1, *mid_T2, *mid_T2g;
    int64_t mid_ld;
    int mid_act_outer;
};

// Order one wave's LDS writes against its own later LDS reads (and reads against later overwrites).  DS instructions of a
// wave execute in issue order, so only the compiler has to be stopped from moving them; waiting on lgkmcnt alone (not a
// fence, which also drains vmcnt) keeps the epilogue's global stores asynchronous.
#define RN_LDS_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ f32x4 mk4(float a, float b, float c, float d) {
    f32x4 r;
    r.x = a; r.y = b; r.z = c; r.w = d;
    return r;
}
__device__ __forceinline__ f32x4 gemm_combine(f32x4 x, f32x4 y, int mode, int act) {
    if (mode == RECNOW_OPMODE_MUL) return x * y;
    return mk4(x.x * rn_act_grad_from_out(y.x, act), x.y * rn_act_grad_from_out(y.y, act),
               x.z * rn_act_grad_from_out(y.z, act), x.w * rn_act_grad_from_out(y.w, act));
}

// ROWS = BM (A) or BN (B).  KC: operand is k-contiguous in memory ([row][k]); else [k][row].
// Two loaders, chosen per k-tile by a BLOCK-UNIFORM condition so that no load ever sits under a per-lane branch
// (a branch around each load makes hipcc drain vmcnt at every merge and serialises the tile's loads):
//   load_fast: whole tile in bounds and 16-B aligned -> unconditional float4 loads, all in flight together;
//   load_safe: edge tiles -> unconditional scalar loads from CLAMPED (always valid) addresses + select to zero.
// LDS access that stays ONE ds_read_b32 / ds_write_b32 with a 16-bit immediate offset.  hipcc pairs plain accesses into ds_read2_b32 /
// ds_write2_b32, whose 8-bit offsets (1020 B) do not span the k-rows of a tile (516 B apart): every pair then needs a VALU add to rebase
// its address, and in the MFMA loop every VALU instruction both takes an issue slot from the MFMAs and sits between a fragment and its
// read (measured: 30 adds per k-tile gone = 4-5 % per launch of the K = 1024 products).
#define FR(ptr) (*(const volatile __attribute__((address_space(3))) float*)(ptr))
#define FW(ptr) (*(volatile __attribute__((address_space(3))) float*)(ptr))

template <int ROWS, int BK, bool KC>
struct Tile {
    static constexpr int NF4 = ROWS * BK / 4;                                   // float4 slots in the tile
    static constexpr int NV = (NF4 + GEMM_THREADS - 1) / GEMM_THREADS;
    static constexpr bool RAGGED = (NF4 % GEMM_THREADS) != 0;                   // last slot only for some threads
    static constexpr int LD = KC ? ROWS + 1 : ROWS;
    f32x4 v[NV];
    f32x4 y[NV];          // sliced schedule only: second operand (MUL / ACTGRAD) or OUTER scale (.x), combined at commit time
    unsigned off[NV];     // element offset of this thread's float4 inside a tile; tile-invariant (set once by init)
    unsigned boff[NV];    // the same in bytes (sliced schedule: `global_load v, voff, s[base]` addressing; the host checks the range)

    // A tile's addresses are (block-uniform tile base) + off[i]: the base lives in SGPRs and advances per k-tile, the
    // offsets are loop-invariant 32-bit VGPRs.  (Per-load 64-bit addresses recomputed every tile cost ~10 VALU each and,
    // under the 256-register cap, got spilled -- the reload's vmcnt(0) then serialised every tile's loads.)
    __device__ __forceinline__ void init(int64_t ld) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            coords(threadIdx.x + i * GEMM_THREADS, r, k);
            off[i] = KC ? (unsigned)(r * ld + k) : (unsigned)(k * ld + r);
            boff[i] = off[i] << 2;
        }
    }
    __device__ __forceinline__ static int64_t tile_base(int64_t ld, int r0, int k0) {
        return KC ? (int64_t)r0 * ld + k0 : (int64_t)k0 * ld + r0;
    }

    __device__ __forceinline__ static void coords(int idx, int& r, int& k) {   // first element of this thread's float4
        if (KC) { k = (idx % (BK / 4)) * 4; r = idx / (BK / 4); }
        else { r = (idx % (ROWS / 4)) * 4; k = idx / (ROWS / 4); }
    }
    __device__ __forceinline__ static bool has(int i) {
        return !RAGGED || i + 1 < NV || (int)threadIdx.x + i * GEMM_THREADS < NF4;
    }

    template <bool SECOND>
    __device__ __forceinline__ void load_fast(const float* __restrict__ p, const float* __restrict__ p2, int mode, int act,
                                              int64_t ld, int r0, int k0) {
        f32x4 y[NV];
        const float* __restrict__ tb = p + tile_base(ld, r0, k0);
        const float* __restrict__ tb2 = SECOND ? p2 + tile_base(ld, r0, k0) : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (has(i)) {
                v[i] = *reinterpret_cast<const f32x4*>(tb + off[i]);
                if (SECOND) y[i] = *reinterpret_cast<const f32x4*>(tb2 + off[i]);
            }
        }
        if (SECOND) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = gemm_combine(v[i], y[i], mode, act);
        }
    }

    // load_fast<true> (MUL) that also writes side_out = first * side_in for the elements it loads (XF & 4)
    __device__ __forceinline__ void load_fast_side(const float* __restrict__ p, const float* __restrict__ p2,
                                                   const float* __restrict__ p3, float* __restrict__ out, int64_t ld, int r0, int k0) {
        f32x4 y[NV], z[NV];
        const int64_t base = tile_base(ld, r0, k0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (has(i)) {
                v[i] = *reinterpret_cast<const f32x4*>(p + base + off[i]);
                y[i] = *reinterpret_cast<const f32x4*>(p2 + base + off[i]);
                z[i] = *reinterpret_cast<const f32x4*>(p3 + base + off[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (has(i)) *reinterpret_cast<f32x4*>(out + base + off[i]) = v[i] * z[i];
            v[i] = v[i] * y[i];
        }
    }

    template <bool SECOND>
    __device__ __forceinline__ void load_safe(const float* __restrict__ p, const float* __restrict__ p2, int mode, int act,
                                              int64_t ld, int r0, int k0, int R, int Kend) {
        float y[NV][4];
        float x[NV][4];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            coords(threadIdx.x + i * GEMM_THREADS, r, k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gr = r0 + r + (KC ? 0 : e), gk = k0 + k + (KC ? e : 0);
                const bool ok = gr < R && gk < Kend;
                const int cr = min(gr, R - 1), ck = min(gk, Kend - 1);          // always a valid address
                const int64_t off = KC ? (int64_t)cr * ld + ck : (int64_t)ck * ld + cr;
                const float a = p[off];
                x[i][e] = ok ? a : 0.f;
                if (SECOND) { const float b = p2[off]; y[i][e] = ok ? b : 0.f; }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = mk4(x[i][0], x[i][1], x[i][2], x[i][3]);
            if (SECOND) v[i] = gemm_combine(v[i], mk4(y[i][0], y[i][1], y[i][2], y[i][3]), mode, act);
        }
    }

    // OUTER mode (CIN's on-the-fly outer product): element (row, col) = p2[row*ld2 + col/hq] * p[row*ld + col%hq], where
    // (row, col) = (tile row, k) for k-contiguous tiles and (k, tile row) otherwise.  hq % 4 == 0 on the fast path, so
    // the four elements of a float4 share one p2 value.
    __device__ __forceinline__ void load_outer_fast(const float* __restrict__ p, const float* __restrict__ p2, int64_t ld,
                                                    int64_t ld2, int hq, int r0, int k0) {
        float s2[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            coords(threadIdx.x + i * GEMM_THREADS, r, k);
            const int row = KC ? r0 + r : k0 + k, col = KC ? k0 + k : r0 + r;
            if (has(i)) {
                v[i] = *reinterpret_cast<const f32x4*>(p + (int64_t)row * ld + (col % hq));
                s2[i] = p2[(int64_t)row * ld2 + (col / hq)];
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = mk4(v[i].x * s2[i], v[i].y * s2[i], v[i].z * s2[i], v[i].w * s2[i]);
    }
    __device__ __forceinline__ void load_outer_safe(const float* __restrict__ p, const float* __restrict__ p2, int64_t ld,
                                                    int64_t ld2, int hq, int r0, int k0, int R, int Kend) {
        float x[NV][4], y[NV][4];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int r, k;
            coords(threadIdx.x + i * GEMM_THREADS, r, k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int gr = r0 + r + (KC ? 0 : e), gk = k0 + k + (KC ? e : 0);
                const bool ok = gr < R && gk < Kend;
                const int cr = min(gr, R - 1), ck = min(gk, Kend - 1);
                const int row = KC ? cr : ck, col = KC ? ck : cr;
                const float a = p[(int64_t)row * ld + (col % hq)];
                const float b = p2[(int64_t)row * ld2 + (col / hq)];
                x[i][e] = ok ? a : 0.f;
                y[i][e] = b;
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = mk4(x[i][0] * y[i][0], x[i][1] * y[i][1], x[i][2] * y[i][2], x[i][3] * y[i][3]);
    }

    // K2: compile-time operand kind (RECNOW_OPMODE_*), or -1 = decided at run time from `mode` (general kernels).
    // The lean kernels fix it at compile time: run-time mode/activation switches inside the tile loads turn the hot loop
    // into a branch jungle with a conservative s_waitcnt vmcnt at every merge (seen in the .s, cost ~25 % wave time).
    template <bool EDGE, int K2>
    __device__ __forceinline__ void load(bool fast, const float* __restrict__ p, const float* __restrict__ p2, int mode_rt,
                                         int act, int64_t ld, int r0, int k0, int R, int Kend, int64_t ld2 = 0, int hq = 1) {
        const int mode = K2 >= 0 ? K2 : mode_rt;
        if (mode == RECNOW_OPMODE_OUTER) {
            if (!EDGE || fast) load_outer_fast(p, p2, ld, ld2, hq, r0, k0);
            else load_outer_safe(p, p2, ld, ld2, hq, r0, k0, R, Kend);
        } else if (mode == RECNOW_OPMODE_NONE) {
            if (!EDGE || fast) load_fast<false>(p, p2, mode, act, ld, r0, k0);
            else load_safe<false>(p, p2, mode, act, ld, r0, k0, R, Kend);
        } else {
            if (!EDGE || fast) load_fast<true>(p, p2, mode, act, ld, r0, k0);
            else load_safe<true>(p, p2, mode, act, ld, r0, k0, R, Kend);
        }
    }

    __device__ __forceinline__ void store_slot(int i, float* __restrict__ S) const {
        int r, k;
        coords(threadIdx.x + i * GEMM_THREADS, r, k);
        if (!has(i)) return;
        if (KC) {
            float* s = S + k * LD + r;
            FW(s) = v[i].x;
            FW(s + LD) = v[i].y;
            FW(s + 2 * LD) = v[i].z;
            FW(s + 3 * LD) = v[i].w;
        } else {
            *reinterpret_cast<f32x4*>(S + k * LD + r) = v[i];
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ S) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) store_slot(i, S);
    }

    // Sliced schedule of the lean kernels (fast tiles only): slot i of the NEXT-BUT-ONE k-tile is requested with issue()
    // and nothing waits for it; commit() -- one k-tile of MFMAs later -- combines it with its second operand and writes it
    // to LDS.  (Combining at load time put an `s_waitcnt vmcnt` right behind the loads: the MUL / ACTGRAD / OUTER kernels
    // sat out the HBM latency of every k-tile in front of their MFMA loop.)
    template <int K2>
    __device__ __forceinline__ void issue(int i, const float* __restrict__ p, const float* __restrict__ p2, int64_t ld,
                                          int r0, int k0, int64_t ld2, int hq) {
        if (!has(i)) return;
        if (K2 == RECNOW_OPMODE_OUTER) {
            int r, k;
            coords(threadIdx.x + i * GEMM_THREADS, r, k);
            const int row = KC ? r0 + r : k0 + k, col = KC ? k0 + k : r0 + r;
            v[i] = *reinterpret_cast<const f32x4*>(p + (int64_t)row * ld + (col % hq));
            y[i].x = p2[(int64_t)row * ld2 + (col / hq)];
        } else {
            // (block-uniform base in SGPRs) + (32-bit byte offset of the thread) = the `global_load v, voff, s[base]` form: no VALU address
            // arithmetic per load.  The empty asm keeps the offset opaque inside the loop body: seen as loop-invariant, its zero-extension
            // is hoisted out, instruction selection then meets a 64-bit VGPR offset and emits a 64-bit VALU add per load.
            const int64_t base = tile_base(ld, r0, k0);
            asm volatile("" : "+v"(boff[i]));
            const unsigned bo = boff[i];
            v[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p + base) + bo);
            if (K2 != RECNOW_OPMODE_NONE) y[i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p2 + base) + bo);
        }
    }
    // XF & 32 (round 5, the A/B VERDICT round 4 asked for): LDS-DMA staging of a plain [k][row] operand.  The float4 slots of such a tile are lane-linear
    // in LDS (slot idx -> float offset 4 * idx: coords() and LD = ROWS), so one global_load_lds_dwordx4 per slot writes the wave's 64 pieces as the
    // contiguous 1 KiB they occupy anyway: the LDS image, the fragment reads and the k order are those of the register-staged kernel (bit-identical results).
    // The destination is wave-uniform (M0), the source per lane: SGPR tile base + the same opaque 32-bit byte offset as issue().
    __device__ __forceinline__ void dma(int i, const float* __restrict__ p, int64_t ld, int r0, int k0, float* __restrict__ S) {
        static_assert(!KC && !RAGGED, "LDS-DMA staging: [k][row] operands, whole slots");
        const int64_t base = tile_base(ld, r0, k0);
        asm volatile("" : "+v"(boff[i]));
        const char* g = reinterpret_cast<const char*>(p + base) + boff[i];
        float* dst = S + (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) * 64 + i * GEMM_THREADS) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    template <int K2>
    __device__ __forceinline__ void commit(int i, float* __restrict__ S, int act) {
        if (K2 == RECNOW_OPMODE_OUTER) v[i] = mk4(v[i].x * y[i].x, v[i].y * y[i].x, v[i].z * y[i].x, v[i].w * y[i].x);
        else if (K2 != RECNOW_OPMODE_NONE) v[i] = gemm_combine(v[i], y[i], K2, act);
        store_slot(i, S);
    }
};

__device__ __forceinline__ bool gemm_aligned(const void* p, int64_t ld, int64_t sb) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0) && (sb % 4 == 0);
}

// Lean epilogue (every tile in bounds, 16-byte aligned): shared by the fp32 kernels below and the split-precision kernel
// (gemm_split.hip) -- both leave the 32x32 MFMA accumulator layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <int TM, int TN, int XF>
__device__ __forceinline__ void gemm_lean_epilogue(const GemmK& p, f32x16 (&acc)[TM][TN], float* smem, int m0, int n0, int wm, int wn,
                                                   int lane, int wave, int z, int bidx) {
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    // Lean epilogue: each 32x32 accumulator sub-tile goes through a wave-private LDS buffer (36-float rows) so that
    // global traffic is float4 per lane (8 lanes = one 128-B row segment) with every epilogue operand of the
    // sub-tile in flight at once -- 4 wide loads/stores instead of 16 + 16 dependent dword round trips.
    float* stg = smem + wave * (32 * 36);          // the k-loop's last barrier has retired every As/Bs read
    const int rr0 = lane >> 3, cc = (lane & 7) * 4;
    float* Cb = p.splitk > 1 ? p.partial + ((int64_t)z * p.M) * p.npart : p.C + (int64_t)bidx * p.sC;
    const int64_t ldc = p.splitk > 1 ? p.npart : p.ldc;
    const bool plain = p.splitk > 1;
    const float* biasb = (!plain && p.bias) ? p.bias + (int64_t)bidx * p.sBias : nullptr;
    const float* Eb = (!plain && p.emul) ? p.emul + (int64_t)bidx * p.sE : nullptr;
    const bool accum = !plain && p.accumulate;
    // Nothing to apply (split-K slabs, and plain products such as the K = 1024 ones of DCN-v2): the accumulators go out as they lie --
    // one dword per lane, a store instruction = two rows x 128 B, 16 per sub-tile, no LDS round trip and nothing to wait for.  The
    // epilogue of these launches runs with the MFMA pipe idle (every workgroup of the one round reaches it at the same time).
    if (p.direct_store && (plain || (!biasb && !Eb && !accum && !p.c_trans && p.act == RECNOW_ACT_LINEAR && (XF & 2) == 0))) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* cp = Cb + (int64_t)(m0 + wm * TM * 32 + i * 32 + row_l) * ldc + n0 + wn * TN * 32 + j * 32 + col_l;
#pragma unroll
                for (int r = 0; r < 16; ++r) cp[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[i][j][r];
            }
        return;
    }
    // rank-R update operands: Q[r][col] depends only on the column sub-tile j, P[row][r] only on the row sub-tile i.
    // Loads are unconditional (r clamped to the last valid one, its weight zeroed): no per-load branches.
    float euq[TN][4][4], eup[4][4];
    if ((XF & 2) && !plain) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rc = min(r, p.eu_r - 1);
                const float on = r < p.eu_r ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    euq[j][r][e] = on * p.eu_q[(int64_t)rc * p.eu_qrs + (int64_t)(n0 + wn * TN * 32 + j * 32 + cc + e) * p.eu_qns];
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if ((XF & 2) && !plain) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    eup[q][r] = p.eu_p[(int64_t)(m0 + wm * TM * 32 + i * 32 + q * 8 + rr0) * p.eu_pms + min(r, p.eu_r - 1)];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row0 = m0 + wm * TM * 32 + i * 32, col0 = n0 + wn * TN * 32 + j * 32 + cc;
            float4 ev[4], cv[4];
            if (Eb) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ev[q] = *reinterpret_cast<const float4*>(Eb + (int64_t)(row0 + q * 8 + rr0) * p.lde + col0);
            }
            const bool ctr = !plain && p.c_trans;      // transposed store C[n][m]: scalar accesses (block-uniform, rare)
            if (accum) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t rw = row0 + q * 8 + rr0;
                    if (!ctr) cv[q] = *reinterpret_cast<const float4*>(Cb + rw * ldc + col0);
                    else cv[q] = make_float4(Cb[(int64_t)col0 * ldc + rw], Cb[(int64_t)(col0 + 1) * ldc + rw],
                                             Cb[(int64_t)(col0 + 2) * ldc + rw], Cb[(int64_t)(col0 + 3) * ldc + rw]);
                }
            }
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (biasb) bv = *reinterpret_cast<const float4*>(biasb + col0);
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + row_l) * 36 + col_l] = acc[i][j][r];
            RN_LDS_WAVE_SYNC();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 a = *reinterpret_cast<const float4*>(stg + (q * 8 + rr0) * 36 + cc);
                float v[4] = {a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w};
                if ((XF & 2) && !plain) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += eup[q][r] * euq[j][r][e];
                }
                if (!plain && p.act != RECNOW_ACT_LINEAR) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col0 + e < p.act_cols) v[e] = rn_act(v[e], p.act);
                }
                if (Eb) {
                    const float ee[4] = {ev[q].x, ev[q].y, ev[q].z, ev[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] *= (p.e_mode == RECNOW_OPMODE_ACTGRAD) ? rn_act_grad_from_out(ee[e], p.e_act) : ee[e];
                }
                if (accum) { v[0] += cv[q].x; v[1] += cv[q].y; v[2] += cv[q].z; v[3] += cv[q].w; }
                const int64_t rw = row0 + q * 8 + rr0;
                if (!ctr) *reinterpret_cast<float4*>(Cb + rw * ldc + col0) = make_float4(v[0], v[1], v[2], v[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Cb[(int64_t)(col0 + e) * ldc + rw] = v[e];
                }
            }
            RN_LDS_WAVE_SYNC();
        }
    }
}

// XF & 16: epilogue of the TRANSPOSED GEMM1 of DCNMixLayer (128 T1 columns x 128 batch rows per workgroup; wave (wm, wn): expert wm
// (S = 64 columns of T1), batch rows wn * 64 .. + 63).  acc[i][j][r] = (x_l U)[row = n0 + wn * 64 + j * 32 + (lane & 31)][s = i * 32 + rmap(r) +
// 4 * (lane >> 5)] of expert wm, rmap(r) = (r & 3) + 8 * (r >> 2): for the second product C = H1 V the register r of a lane IS the A
// fragment of the k pair (s, s + 4) -- lanes 0-31 supply k = s, lanes 32-63 k = s + 4 -- so the S x S product runs straight off the
// accumulators, its B fragments (V, 16 KB per expert, L2-resident) come from global memory.  smem: [0, 512) side-product combine,
// [512, 768) the gate values G[row][2] (written by the caller before the barrier in front of this function), [1024, ..) staging.
__device__ __forceinline__ void gemm_midf_epilogue(const GemmK& p, f32x16 (&acc)[2][2], float* smem, int n0, int wm, int wn, int lane, int wave) {
    const float* Gs = smem + 512;
    float* stg = smem + 1024 + wave * (32 * 36);
    const int h = lane >> 5, l31 = lane & 31;
    const int rr0 = lane >> 3, cc = (lane & 7) * 4;
    // the B fragments of the second product (this lane's 64 values of V_e) are requested first: they land under the activation and the T1 store
    const float* Vn = p.mid_V + (int64_t)wm * 64 * 64 + (int64_t)(4 * h) * 64 + l31;
    float bv[2][16][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s0 = i * 32 + (r & 3) + 8 * (r >> 2);
            bv[i][r][0] = Vn[s0 * 64];
            bv[i][r][1] = Vn[s0 * 64 + 32];
        }
    // 1 + 3. H1 = act_inner(x_l U) and C = H1 V_e, interleaved: the activation of a register (VALU, ~70 cycles with exp2 / rcp) runs in
    // the shadow of the previous register's MFMAs; row block j = 0 first, so that its H2 / T2 / T2g epilogue (4.) can be spread over the
    // MFMA steps of row block j = 1.  acc2[j][tb][r] = C[row = j * 32 + rmap(r) + 4 h][t = tb * 32 + (lane & 31)].
    f32x16 acc2[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][tb][r] = 0.f;
    auto finish = [&](int j, int tb, int r) {      // 4. one register of C: H2 = act_outer(C) -> T2, G_e * H2 -> T2g
        const int rl = wn * 64 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float g = Gs[rl * 2 + wm];
        const int64_t ro = (int64_t)(n0 + rl) * p.mid_ld + wm * 64 + tb * 32 + l31;
        const float h2 = rn_act(acc2[j][tb][r], p.mid_act_outer);
        p.mid_T2[ro] = h2;
        p.mid_T2g[ro] = g * h2;
    };
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = rn_act(acc[i][j][r], p.act);
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc2[j][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[i][j][r], bv[i][r][tb], acc2[j][tb], 0, 0, 0);
                if (j == 1) finish(0, i, r);       // the finished row block's epilogue, one register per MFMA step
            }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) finish(1, tb, r);
    // 2. T1[row][expert * 64 + s] for the backward pass: transposed through a wave-private LDS tile, rows leave as float4
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[l31 * 36 + (r & 3) + 8 * (r >> 2) + 4 * h] = acc[i][j][r];
            RN_LDS_WAVE_SYNC();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = q * 8 + rr0;
                const float4 v = *reinterpret_cast<const float4*>(stg + row * 36 + cc);
                *reinterpret_cast<float4*>(p.mid_T1 + (int64_t)(n0 + wn * 64 + j * 32 + row) * p.mid_ld + wm * 64 + i * 32 + cc) = v;
            }
            RN_LDS_WAVE_SYNC();
        }
}

// EDGE = false: every tile of the launch is in bounds and 16-byte aligned (checked on the host) -> no bounds code at
// all (lean: no spills under the 256-register cap).  EDGE = true: general shapes, clamped loads and predicated stores.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, bool A_KC, bool B_KC, bool EDGE, int A2K, int B2K, int XF = 0>
__global__ void __launch_bounds__(GEMM_THREADS, 2)      // >= 2 waves/SIMD: VGPR+AGPR <= 256, two workgroups per CU
k_gemm(const GemmK p) {
    static_assert(XF == 0 || (!EDGE && (BM == 128 || (BM == 64 && (XF & ~9) == 0))), "side product / rank-R update: lean 128-row kernels (side product: 64 rows too)");
    static_assert((XF & 4) == 0 || (A_KC && A2K == RECNOW_OPMODE_MUL), "A-stream side output: A [M][K] in MUL mode");
    static_assert((XF & 16) == 0 || ((XF & 9) == 9 && !A_KC && B_KC && BM == 128 && BN == 128 && WAVES_M == 2 && A2K == 0 && (B2K == 0 || B2K == RECNOW_OPMODE_MUL)),
                  "fused sub-space forward: transposed GEMM1, two-wide side product from the B tile");
    constexpr bool MIDF = (XF & 16) != 0;
    constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
    using TA = Tile<BM, BK, A_KC>;
    using TB = Tile<BN, BK, B_KC>;
    constexpr int A_SZ = BK * TA::LD, B_SZ = BK * TB::LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                 // two buffers of A_SZ floats, then two of B_SZ
    float* const Bs = smem + 2 * A_SZ;
    float* const Bxs = smem + 2 * A_SZ + 2 * B_SZ;     // XF & 1: two buffers of BK x 4 side-product weights
    // (sp_on / side output use blockIdx.y directly: those launches are never remapped in y, see rn_gemm)
    const bool sp_on = (XF & 1) && (MIDF || blockIdx.y == 0);    // one column-tile computes the side product of a row-tile (MIDF: every workgroup, of its batch rows)
    f32x4 spacc = mk4(0.f, 0.f, 0.f, 0.f);
    // XF & 8 (sliced kernels): the side product has at most two columns (DCN-v2 with two experts): two-wide weights and accumulator --
    // half the FMAs and half the LDS bytes of the weights; the side product costs 15 us of a 180 us launch in its four-wide form
    using SPV = std::conditional_t<(XF & 8) != 0, f32x2, f32x4>;
    SPV spn = SPV(0.f);
    // sliced kernels: `spn` holds ONE k-tile's terms of the side product (fp32 FMAs); the running sum over the k-tiles is fp64 (`spd`, one v_cvt +
    // one v_add_f64 per column and k-tile).  With fp32 running sums the layer-0 gate logits of the north-star step (512 sequential FMAs per thread,
    // |logit| up to 39 on bench.py's parity inputs) were the largest single consumer of the 1e-5 parity budget (tools/micro/error_budget_cpu.py).
    double spd[(XF & 8) != 0 ? 2 : 4] = {};
    float bxr[4] = {0.f, 0.f, 0.f, 0.f};

#ifdef RN_GEMM_TRACE      // per-workgroup phase timestamps for tools/gemm_trace.py (build with RECNOW_TRACE=1)
#define RN_TR(i) do { if (p.trace && threadIdx.x == 0) p.trace[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8ll + (i)] = wall_clock64(); } while (0)
    if (p.trace && threadIdx.x == 0)
        p.trace[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8ll + 4] =
            __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11)) | ((long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) << 32);
    if (p.trace && threadIdx.x == 0) p.trace[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8ll + 5] = clock64();
#else
#define RN_TR(i) do { } while (0)
#endif
    RN_TR(0);
    // Split-K with few row tiles (the K = B weight-gradient products: 8 row tiles x 64 slabs): the row tiles of one k-slab
    // read the same B panel, but consecutive workgroups land on different XCDs (round-robin dispatch), i.e. on different
    // L2s -- the panel was fetched from HBM once per XCD (PMC: +270 MB per launch).  xcd_remap re-labels the workgroups so
    // that the row tiles of a slab are the consecutive workgroups OF ONE XCD.  Pure re-labelling of (blockIdx.x, blockIdx.z).
    int bx = blockIdx.x, by = blockIdx.y, z = blockIdx.z;
    if (p.xcd_remap == 1) {
        const int gx = gridDim.x, lin = bx + gx * z, xcd = lin & 7, i = lin >> 3;
        z = xcd * ((int)gridDim.z >> 3) + i / gx;
        bx = i % gx;
    } else if (p.xcd_remap == 2) {
        // several column tiles per row tile (gridDim.x % 8 == 0): the column tiles of one row tile become consecutive
        // workgroups of one XCD, so the A row panel is fetched into one L2 once instead of once per column tile
        const int gx = gridDim.x, gy = gridDim.y, lin = bx + gx * by, xcd = lin & 7, i = lin >> 3;
        bx = (i / gy) * 8 + xcd;
        by = i % gy;
    }
    const int bidx = z / p.splitk, ks = z % p.splitk;
    const int k_begin = ks * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
    const int m0 = bx * BM, n0 = by * BN;
    const bool a_outer = (A2K >= 0 ? A2K : p.a_mode) == RECNOW_OPMODE_OUTER, b_outer = (B2K >= 0 ? B2K : p.b_mode) == RECNOW_OPMODE_OUTER;
    const float* Ab = p.A + (int64_t)bidx * p.sA;
    const float* A2b = p.A2 ? p.A2 + (a_outer ? 0 : (int64_t)bidx * p.sA) : nullptr;
    const float* Bb = p.B + (int64_t)bidx * p.sB;
    const float* B2b = p.B2 ? p.B2 + (b_outer ? 0 : (int64_t)bidx * p.sB) : nullptr;
    // block-uniform: rows of this tile all in bounds and every float4 16-byte aligned
    const bool a_fast = !EDGE || ((m0 + BM <= p.M) && gemm_aligned(p.A, p.lda, p.sA) &&
                                  (a_outer ? (p.a_hq % 4 == 0) : (!p.A2 || gemm_aligned(p.A2, p.lda, p.sA))));
    const bool b_fast = !EDGE || ((n0 + BN <= p.N) && gemm_aligned(p.B, p.ldb, p.sB) &&
                                  (b_outer ? (p.b_hq % 4 == 0) : (!p.B2 || gemm_aligned(p.B2, p.ldb, p.sB))));

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int a_off = (lane >> 5) * TA::LD + wm * TM * 32 + (lane & 31);
    const int b_off = (lane >> 5) * TB::LD + wn * TN * 32 + (lane & 31);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TA ta;
    TB tb;
    ta.init(p.lda);
    tb.init(p.ldb);
    const int ntile = (k_end - k_begin + BK - 1) / BK;
    // SLICED (lean kernels with compile-time operand kinds): the LDS writes of k-tile t+1 and the loads of k-tile t+2 are cut
    // into per-slot slices that sit BETWEEN the MFMA groups of k-tile t (each slice issues while the last MFMA of its group
    // executes), instead of in a staging section between the barrier and the MFMA loop where -- the two workgroups of a CU
    // run in lockstep -- no wave of the CU had an MFMA to issue (counters: pipe idle 30 % of the kernel).
    constexpr bool SLICED = !EDGE && A2K >= 0 && B2K >= 0 && (XF & 4) == 0;
    constexpr int NSL = TA::NV + TB::NV;
    // XF & 32: both operands go global -> LDS by LDS-DMA (Tile::dma), one k-tile ahead into the buffer the barrier that opened the iteration freed;
    // hipcc counts the DMAs and drains them (vmcnt(0)) in front of the barrier that closes the iteration.  No staging registers, no ds_write.
    constexpr bool GLDS = (XF & 32) != 0;
    static_assert(!GLDS || (SLICED && !A_KC && !B_KC && A2K == RECNOW_OPMODE_NONE && B2K == RECNOW_OPMODE_NONE && !MIDF), "LDS-DMA staging: plain [k][row] operands");
    auto a_issue = [&](int i, int k0) { ta.template issue<A2K>(i, Ab, A2b, p.lda, m0, k0, p.a_ld2, p.a_hq); };
    auto b_issue = [&](int i, int k0) { tb.template issue<B2K>(i, Bb, B2b, p.ldb, n0, k0, p.b_ld2, p.b_hq); };
    if constexpr (GLDS) {
        if (ntile > 0) {
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) ta.dma(i, Ab, p.lda, m0, k_begin, As);
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) tb.dma(i, Bb, p.ldb, n0, k_begin, Bs);
            if ((XF & 1) && threadIdx.x < BK) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bxr[r] = r < p.sp_r ? p.bx[(int64_t)(k_begin + threadIdx.x) * p.bx_ks + r * p.bx_rs] : 0.f;
                *reinterpret_cast<f32x4*>(Bxs + threadIdx.x * 4) = mk4(bxr[0], bxr[1], bxr[2], bxr[3]);
                const int k1 = k_begin + min(1, ntile - 1) * BK;
#pragma unroll
                for (int r = 0; r < 4; ++r) bxr[r] = r < p.sp_r ? p.bx[(int64_t)(k1 + threadIdx.x) * p.bx_ks + r * p.bx_rs] : 0.f;
            }
        }
    } else
    if constexpr (SLICED) {
        if (ntile > 0) {
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) a_issue(i, k_begin);
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) b_issue(i, k_begin);
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) ta.template commit<A2K>(i, As, p.a_act);
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) tb.template commit<B2K>(i, Bs, p.b_act);
            if ((XF & 1) && threadIdx.x < BK) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    bxr[r] = r < p.sp_r ? p.bx[(int64_t)(k_begin + threadIdx.x) * p.bx_ks + r * p.bx_rs] : 0.f;
                *reinterpret_cast<f32x4*>(Bxs + threadIdx.x * 4) = mk4(bxr[0], bxr[1], bxr[2], bxr[3]);
            }
            const int k1 = k_begin + min(1, ntile - 1) * BK;
#pragma unroll
            for (int i = 0; i < TA::NV; ++i) a_issue(i, k1);
#pragma unroll
            for (int i = 0; i < TB::NV; ++i) b_issue(i, k1);
            if ((XF & 1) && threadIdx.x < BK) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bxr[r] = r < p.sp_r ? p.bx[(int64_t)(k1 + threadIdx.x) * p.bx_ks + r * p.bx_rs] : 0.f;
            }
        }
    } else
    if (ntile > 0) {
        const bool kf = !EDGE || (k_begin + BK <= k_end);
        if ((XF & 4) != 0 && blockIdx.y == 0) ta.load_fast_side(Ab, A2b, p.as_in, p.as_out, p.lda, m0, k_begin);      // one column tile writes the side output
        else ta.template load<EDGE, A2K>(a_fast && kf, Ab, A2b, p.a_mode, p.a_act, p.lda, m0, k_begin, p.M, k_end, p.a_ld2, p.a_hq);
        tb.template load<EDGE, B2K>(b_fast && kf, Bb, B2b, p.b_mode, p.b_act, p.ldb, n0, k_begin, p.N, k_end, p.b_ld2, p.b_hq);
        ta.store(As);
        tb.store(Bs);
        if ((XF & 1) && threadIdx.x < BK) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bxr[r] = r < p.sp_r ? p.bx[(int64_t)(k_begin + threadIdx.x) * p.bx_ks + r * p.bx_rs] : 0.f;
            *reinterpret_cast<f32x4*>(Bxs + threadIdx.x * 4) = mk4(bxr[0], bxr[1], bxr[2], bxr[3]);
        }
    }
    // Staging schedule (one register set): the registers hold k-tile t+1 while k-tile t is computed.  They are written to
    // the free LDS buffer right AFTER the barrier that opens iteration t (not before the barrier that closes it, where every
    // wave would sit in "wait vmcnt -> ds_write -> barrier" with the MFMA pipe idle), and the loads of k-tile t+2 are
    // re-issued immediately, so they have a whole iteration of MFMAs to land.
    auto issue_loads = [&](int t1) {
        const int k0 = k_begin + t1 * BK;
        const bool kf = !EDGE || (k0 + BK <= k_end);
        if ((XF & 4) != 0 && blockIdx.y == 0) ta.load_fast_side(Ab, A2b, p.as_in, p.as_out, p.lda, m0, k0);
        else ta.template load<EDGE, A2K>(a_fast && kf, Ab, A2b, p.a_mode, p.a_act, p.lda, m0, k0, p.M, k_end, p.a_ld2, p.a_hq);
        tb.template load<EDGE, B2K>(b_fast && kf, Bb, B2b, p.b_mode, p.b_act, p.ldb, n0, k0, p.N, k_end, p.b_ld2, p.b_hq);
        if ((XF & 1) && threadIdx.x < BK) {
            const int kx = k0 + threadIdx.x;
#pragma unroll
            for (int r = 0; r < 4; ++r) bxr[r] = r < p.sp_r ? p.bx[(int64_t)kx * p.bx_ks + r * p.bx_rs] : 0.f;
        }
    };
    if (!SLICED && ntile > 1) issue_loads(1);
    __syncthreads();
    RN_TR(1);
    if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (p.prio == 5) {
        // asymmetric: the workgroup in the even wave slot of its SIMDs runs above its co-resident twin.  Two equal workgroups that
        // share the MFMA pipe fairly finish every k-tile together, sit in their barriers together and leave the pipe idle
        // together; with a strict order the favoured one runs as if alone and the other fills its gaps.
        if ((__builtin_amdgcn_s_getreg((4) | (0 << 6) | ((4 - 1) << 11)) & 1) == 0) __builtin_amdgcn_s_setprio(2);
    }
    auto ktile_body = [&](const int t, const int cur) __attribute__((always_inline)) {
        if constexpr (SLICED) {
            if ((XF & 1) && threadIdx.x < BK) {   // side-product weights: k-tile t+1 to LDS, t+2 requested (clamped: unconditional)
                *reinterpret_cast<f32x4*>(Bxs + (cur ^ 1) * BK * 4 + threadIdx.x * 4) = mk4(bxr[0], bxr[1], bxr[2], bxr[3]);
                const int kx = k_begin + min(t + 2, ntile - 1) * BK + threadIdx.x;
#pragma unroll
                for (int r = 0; r < 4; ++r) bxr[r] = r < p.sp_r ? p.bx[(int64_t)kx * p.bx_ks + r * p.bx_rs] : 0.f;
            }
        } else
        if (t + 1 < ntile) {                      // k-tile t+1: registers -> the buffer k-tile t-1 was read from
            ta.store(As + (cur ^ 1) * A_SZ);
            tb.store(Bs + (cur ^ 1) * B_SZ);
            if ((XF & 1) && threadIdx.x < BK)
                *reinterpret_cast<f32x4*>(Bxs + (cur ^ 1) * BK * 4 + threadIdx.x * 4) = mk4(bxr[0], bxr[1], bxr[2], bxr[3]);
            if (t + 2 < ntile) issue_loads(t + 2);
        }
        // VALU side product off the A tile in LDS: thread = (row m, half of the tile's k range); its 2*(BK/4) k's are
        // spread over the MFMA loop below (2 per iteration) so the FMAs run in the shadow of in-flight MFMAs.
        // (MIDF: the product is transposed, the rows of the side product are the rows of the B tile)
        constexpr int SP_LD = MIDF ? TB::LD : TA::LD;
        // thread = (row of the side product, part of the tile's k range): 128 rows x 2 halves, or (64-row tiles) 64 rows x 4 quarters
        constexpr int SP_ROWS = MIDF ? BN : BM, SP_KPP = BK * SP_ROWS / GEMM_THREADS, SP_KPI = SP_KPP / (BK / 4);      // k's per part / per MFMA-loop iteration
        static_assert((XF & 1) == 0 || SP_KPI == 1 || SP_KPI == 2, "side product: one or two k per thread and loop iteration");
        const float* asx = (MIDF ? Bs + cur * B_SZ : As + cur * A_SZ) + (threadIdx.x % SP_ROWS) + (threadIdx.x / SP_ROWS) * SP_KPP * SP_LD;
        const float* bxs = Bxs + cur * BK * 4 + (threadIdx.x / SP_ROWS) * SP_KPP * 4;
        const float* as = As + cur * A_SZ + a_off;
        const float* bs = Bs + cur * B_SZ + b_off;
        // ONE loop form for full and tail tiles (two forms make the compiler shuffle every accumulator between them):
        // kv = valid k of this tile; rows beyond it are zero in LDS, so the (at most one) surplus k-step adds zeros.
        const int kv = EDGE ? min(BK, k_end - (k_begin + t * BK)) : BK;
        if (p.prio == 4) __builtin_amdgcn_s_setprio(1);       // experiment: the MFMA loop above the other workgroup's staging section
        float a0[TM], b0[TN], a1[TM], b1[TN];      // explicit fragment double buffer: step kk+2 loads under step kk's MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = FR(as + i * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = FR(bs + j * 32);
        // sched_barrier(0) pins "issue the NEXT step's ds_reads, then this step's MFMAs": without it hipcc sinks the
        // reads below the MFMAs and waits lgkmcnt(0) right in front of their first use (seen in the .s).  An fp32
        // 32x32x2 MFMA group is >= 256 cycles, far longer than the LDS latency, so nothing finer is needed.
        if constexpr (SLICED) {
            // slot s of k-tile t+1 goes registers -> the LDS buffer k-tile t-1 was read from (free since the barrier that
            // opened this iteration), then the same registers request slot s of k-tile t+2 (index clamped to the last tile, so
            // neither half is under a branch: the surplus stores land in a buffer nobody reads, the surplus loads hit L2)
            constexpr int NG = BK / 2;             // MFMA groups per k-tile
            const int k2 = k_begin + min(t + 2, ntile - 1) * BK;
            float* const As_n = As + (cur ^ 1) * A_SZ;
            float* const Bs_n = Bs + (cur ^ 1) * B_SZ;
            const int k1n = k_begin + min(t + 1, ntile - 1) * BK;      // GLDS: the NEXT k-tile (clamped: the surplus DMAs of the last iteration land in the buffer nobody reads)
            auto stage = [&](int g) {
                if constexpr (GLDS) {
#pragma unroll
                    for (int sl = 0; sl < NSL; ++sl) {
                        if (sl * RN_GLDS_SPREAD / NSL != g) continue;
                        if (sl < TA::NV) ta.dma(sl, Ab, p.lda, m0, k1n, As_n);
                        else tb.dma(sl - TA::NV, Bb, p.ldb, n0, k1n, Bs_n);
                    }
                    return;
                }
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    if (sl * NG / NSL != g) continue;
                    if (sl < TA::NV) { ta.template commit<A2K>(sl, As_n, p.a_act); a_issue(sl, k2); }
                    else { tb.template commit<B2K>(sl - TA::NV, Bs_n, p.b_act); b_issue(sl - TA::NV, k2); }
                }
            };
            // side product: the operands of iteration kk are read one iteration ahead and the FMAs are pinned behind the first
            // MFMA group (an empty volatile asm on the accumulator: hipcc otherwise sinks all 32 packed FMAs of the k-tile to the
            // end of the tile, a serial VALU tail in front of every barrier with 80 registers held for it)
            float sa0 = 0.f, sa1 = 0.f;
            SPV sb0 = SPV(0.f), sb1 = SPV(0.f);
            if constexpr ((XF & 1) != 0) {
                sa0 = FR(asx);
                sb0 = *reinterpret_cast<const SPV*>(bxs);
                if constexpr (SP_KPI == 2) {
                    sa1 = FR(asx + SP_LD);
                    sb1 = *reinterpret_cast<const SPV*>(bxs + 4);
                }
            }
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a1[i] = FR(as + (kk + 2) * TA::LD + i * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) b1[j] = FR(bs + (kk + 2) * TB::LD + j * 32);
                float na0 = 0.f, na1 = 0.f;
                SPV nb0 = sb0, nb1 = sb1;
                if constexpr ((XF & 1) != 0) {
                    if (kk + 4 < BK) {
                        const int kq = ((kk >> 2) + 1) * SP_KPI;
                        na0 = FR(asx + kq * SP_LD);
                        nb0 = *reinterpret_cast<const SPV*>(bxs + kq * 4);
                        if constexpr (SP_KPI == 2) {
                            na1 = FR(asx + (kq + 1) * SP_LD);
                            nb1 = *reinterpret_cast<const SPV*>(bxs + (kq + 1) * 4);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
                stage(kk / 2);
                if constexpr ((XF & 1) != 0) {
                    // packed FMAs (two fp32 FMAs per VALU issue), the A value broadcast to both halves by op_sel: hipcc scalarises
                    // `scalar * f32x2`.  Same operations in the same order per component as `spn += sa0 * sb0; spn += sa1 * sb1`.
                    // Being volatile they also stay here, behind the first MFMA group (otherwise all FMAs of a k-tile sink to its end).
                    const f32x2 sa = {sa0, sa1};
#ifdef RN_SP_PLAIN_FMA
                    // A/B build (tools/build_variant.py -DRN_SP_PLAIN_FMA): the same operations as two plain v_fma_f32 per packed one (the guide
                    // prices one v_pk_fma_f32 at +22 cycles against two v_fma_f32 beside MFMAs); measured in profiles/r04_gemm_pmc.csv
                    static_assert(SP_KPI == 2, "the plain-FMA A/B variant covers the 128-row kernels only");
                    if constexpr ((XF & 8) != 0) {
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.x) : "v"(sa0), "v"(sb0.x));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.y) : "v"(sa0), "v"(sb0.y));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.x) : "v"(sa1), "v"(sb1.x));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.y) : "v"(sa1), "v"(sb1.y));
                    } else {
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.x) : "v"(sa0), "v"(sb0.x));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.y) : "v"(sa0), "v"(sb0.y));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.z) : "v"(sa0), "v"(sb0.z));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.w) : "v"(sa0), "v"(sb0.w));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.x) : "v"(sa1), "v"(sb1.x));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.y) : "v"(sa1), "v"(sb1.y));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.z) : "v"(sa1), "v"(sb1.z));
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(spn.w) : "v"(sa1), "v"(sb1.w));
                    }
#else
                    if constexpr ((XF & 8) != 0) {
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(spn) : "v"(sa), "v"(sb0));
                        if constexpr (SP_KPI == 2)
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(spn) : "v"(sa), "v"(sb1));
                    } else {
                        static_assert((XF & 8) != 0 || SP_KPI == 2, "four-wide side product: 128-row kernels only");
                        f32x2 lo = {spn.x, spn.y}, hi = {spn.z, spn.w};
                        const f32x2 b0l = {sb0.x, sb0.y}, b0h = {sb0.z, sb0.w}, b1l = {sb1.x, sb1.y}, b1h = {sb1.z, sb1.w};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(lo) : "v"(sa), "v"(b0l));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(hi) : "v"(sa), "v"(b0h));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(lo) : "v"(sa), "v"(b1l));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(hi) : "v"(sa), "v"(b1h));
                        spn = mk4(lo.x, lo.y, hi.x, hi.y);
                    }
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                const int kn = kk + 4 < BK ? kk + 4 : BK - 2;
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[i] = FR(as + kn * TA::LD + i * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) b0[j] = FR(bs + kn * TB::LD + j * 32);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
                stage(kk / 2 + 1);
                __builtin_amdgcn_sched_barrier(0);
                sa0 = na0; sa1 = na1; sb0 = nb0; sb1 = nb1;
            }
            if constexpr ((XF & 1) != 0) {      // this k-tile's terms join the fp64 running sums
                spd[0] += (double)spn.x;
                spd[1] += (double)spn.y;
                if constexpr ((XF & 8) == 0) {
                    spd[2] += (double)spn.z;
                    spd[3] += (double)spn.w;
                }
                spn = SPV(0.f);
            }
        } else
        for (int kk = 0; kk < kv; kk += 4) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = FR(as + (kk + 2) * TA::LD + i * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[j] = FR(bs + (kk + 2) * TB::LD + j * 32);
            if constexpr ((XF & 1) != 0) {
                const int kq = kk >> 1;        // this iteration's two k's of the thread's half-range
                spacc += asx[kq * SP_LD] * *reinterpret_cast<const f32x4*>(bxs + kq * 4);
                spacc += asx[(kq + 1) * SP_LD] * *reinterpret_cast<const f32x4*>(bxs + (kq + 1) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int kn = min(kk + 4, BK - 2);     // stays inside this buffer on the last iteration (value unused then)
#pragma unroll
            for (int i = 0; i < TM; ++i) a0[i] = FR(as + kn * TA::LD + i * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) b0[j] = FR(bs + kn * TB::LD + j * 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.prio == 4) __builtin_amdgcn_s_setprio(0);
#ifdef RN_GEMM_TRACE      // per-wave barrier arrival / departure of the first 64 workgroups (tools/gemm_trace.py, second table)
        const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        long long* const tr2 = (p.trace && wg_lin < 64 && t < 64 && lane == 0)
                                   ? p.trace + 8ll * gridDim.x * gridDim.y * gridDim.z + ((wg_lin * 4 + wave) * 64 + t) * 2 : nullptr;
        if (tr2) tr2[0] = clock64();
        __syncthreads();
        if (tr2) tr2[1] = clock64();
#else
        __syncthreads();
#endif
    };
    if constexpr (SLICED && (((BM == 128 || BM == 64) && BN == 128 && A2K != RECNOW_OPMODE_OUTER && B2K != RECNOW_OPMODE_OUTER) ||
                             (A2K == RECNOW_OPMODE_NONE && B2K == RECNOW_OPMODE_NONE))) {
        // two copies of the body, the LDS buffer index a compile-time constant in each: every LDS address of the loop is then (a
        // loop-invariant register) + (an immediate offset), no per-k-tile VALU address arithmetic.  (128 x 128 tiles and plain operands only:
        // with the 80-register accumulators of the 128 x 160 tiles plus a second operand, or the OUTER operand's per-slot
        // coordinates, the second copy's registers end in scratch -- 279 vs 213 us per launch, CIN 79 vs 48 ms.)
        for (int t = 0; t < ntile; t += 2) {
            ktile_body(t, 0);
            if (t + 1 >= ntile) break;
            ktile_body(t + 1, 1);
        }
    } else {
        for (int t = 0; t < ntile; ++t) ktile_body(t, t & 1);
    }
    if constexpr ((XF & 1) != 0 && SLICED) {
        if constexpr ((XF & 8) != 0) spacc = mk4((float)spd[0], (float)spd[1], 0.f, 0.f);
        else spacc = mk4((float)spd[0], (float)spd[1], (float)spd[(XF & 8) != 0 ? 0 : 2], (float)spd[(XF & 8) != 0 ? 1 : 3]);
    }
    if constexpr ((XF & 1) != 0) {
        // combine the k-parts of a row through LDS (free now) and write the side columns
        constexpr int CB_ROWS = MIDF ? BN : BM, CB_PARTS = GEMM_THREADS / CB_ROWS;
        if (sp_on && threadIdx.x >= CB_ROWS) *reinterpret_cast<f32x4*>(smem + (threadIdx.x - CB_ROWS) * 4) = spacc;
        __syncthreads();
        if constexpr (MIDF) {
            // the two gate logits of batch row n0 + tid: softmax -> G (LDS, for the epilogue) and the gate columns [128, 144) of the three
            // activation tensors: T1 keeps the logits, T2 and T2g [G | 0] (the zero padding is read by the K = 144 product that follows)
            if (threadIdx.x < 128) {
                const f32x4 o = spacc + *reinterpret_cast<const f32x4*>(smem + threadIdx.x * 4);
                const float mx = fmaxf(o.x, o.y);
                const float e0 = expf(o.x - mx), e1 = expf(o.y - mx), sum = e0 + e1;
                const float g0 = e0 / sum, g1 = e1 / sum;
                smem[512 + threadIdx.x * 2] = g0;
                smem[512 + threadIdx.x * 2 + 1] = g1;
                const int64_t ro = (int64_t)(n0 + threadIdx.x) * p.mid_ld + 128;
                p.mid_T1[ro] = o.x;
                p.mid_T1[ro + 1] = o.y;
                const float4 gz = make_float4(g0, g1, 0.f, 0.f), zz = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(p.mid_T2 + ro) = gz;
                *reinterpret_cast<float4*>(p.mid_T2g + ro) = gz;
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    *reinterpret_cast<float4*>(p.mid_T2 + ro + 4 * q) = zz;
                    *reinterpret_cast<float4*>(p.mid_T2g + ro + 4 * q) = zz;
                }
            }
        } else
        if (sp_on && threadIdx.x < CB_ROWS) {
            f32x4 o = spacc;
#pragma unroll
            for (int q = 1; q < CB_PARTS; ++q) o = o + *reinterpret_cast<const f32x4*>(smem + ((q - 1) * CB_ROWS + threadIdx.x) * 4);
            const float ov[4] = {o.x, o.y, o.z, o.w};
            const int m = m0 + threadIdx.x;
            for (int r = 0; r < p.sp_r; ++r) {
                if (p.splitk > 1) p.partial[((int64_t)z * p.M + m) * p.npart + p.N + r] = ov[r];
                else p.cx[(int64_t)m * p.cx_ms + r * p.cx_rs] = ov[r];
            }
        }
        __syncthreads();
    }

    if (p.prio) __builtin_amdgcn_s_setprio(0);
    RN_TR(2);
    // epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    if constexpr (MIDF) {
        gemm_midf_epilogue(p, acc, smem, n0, wm, wn, lane, wave);
        RN_TR(3);
#ifdef RN_GEMM_TRACE
        if (p.trace && threadIdx.x == 0) p.trace[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8ll + 6] = clock64();
#endif
        return;
    } else
    if constexpr (!EDGE) {
        gemm_lean_epilogue<TM, TN, XF>(p, acc, smem, m0, n0, wm, wn, lane, wave, z, bidx);
        RN_TR(3);
#ifdef RN_GEMM_TRACE
        if (p.trace && threadIdx.x == 0) p.trace[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8ll + 6] = clock64();
#endif
        return;
    } else {
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);          // block-uniform
    if (p.splitk > 1) {
        float* P = p.partial + ((int64_t)z * p.M) * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * TN * 32 + j * 32 + col_l;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                    if (interior || (row < p.M && col < p.N)) P[(int64_t)row * p.N + col] = acc[i][j][r];
                }
            }
        return;
    }
    float* Cb = p.C + (int64_t)bidx * p.sC;
    const float* biasb = p.bias ? p.bias + (int64_t)bidx * p.sBias : nullptr;
    const float* Eb = p.emul ? p.emul + (int64_t)bidx * p.sE : nullptr;
    const int64_t c_rs = p.c_trans ? 1 : p.ldc, c_cs = p.c_trans ? p.ldc : 1;     // C[row*c_rs + col*c_cs]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + col_l;
            const int colc = min(col, p.N - 1);
            const float bv = biasb ? biasb[colc] : 0.f;
            const int rbase = m0 + wm * TM * 32 + i * 32 + row_l;
            float e[16], cprev[16];
            // operands of the epilogue are loaded unconditionally from clamped addresses, 16 at a time
            if (Eb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    e[r] = Eb[(int64_t)row * p.lde + colc];
                }
            }
            if (p.accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    cprev[r] = Cb[(int64_t)row * c_rs + (int64_t)colc * c_cs];
                }
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
            if (p.act != RECNOW_ACT_LINEAR && col < p.act_cols) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = rn_act(v[r], p.act);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (Eb) v[r] *= (p.e_mode == RECNOW_OPMODE_ACTGRAD) ? rn_act_grad_from_out(e[r], p.e_act) : e[r];
                if (p.accumulate) v[r] += cprev[r];
                if (interior || (row < p.M && col < p.N)) Cb[(int64_t)row * c_rs + (int64_t)col * c_cs] = v[r];
            }
        }
    }
}


// launchers defined in gemm_inst_*.hip; each returns RECNOW_EUNSUPPORTED when it has no instantiation for the combo
int rn_gemm_launch_lean128(const GemmK& k, bool a_kc, bool b_kc, int bk, int a2k, int b2k, dim3 grid, hipStream_t st);
int rn_gemm_launch_lean160(const GemmK& k, bool a_kc, bool b_kc, int bk, int a2k, int b2k, dim3 grid, hipStream_t st);
int rn_gemm_launch_lean64(const GemmK& k, bool a_kc, bool b_kc, int a2k, int b2k, dim3 grid, hipStream_t st);
int rn_gemm_launch_edge(const GemmK& k, int bm, int bn, bool a_kc, bool b_kc, dim3 grid, hipStream_t st);

template <int BM, int BN, int WM, int WN, int BK, bool AKC, bool BKC, bool EDGE, int A2K, int B2K, int XF = 0>
static inline void rn_gemm_launch_one(const GemmK& k, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (2 * BK * (size_t)(Tile<BM, BK, AKC>::LD + Tile<BN, BK, BKC>::LD) + ((XF & 1) ? 2 * BK * 4 : 0)) * sizeof(float);
    hipLaunchKernelGGL((k_gemm<BM, BN, WM, WN, BK, AKC, BKC, EDGE, A2K, B2K, XF>), grid, GEMM_THREADS, lds, st, k);
}
int rn_gemm_launch_lean128x(const GemmK& k, bool a_kc, bool b_kc, int bk, int a2k, int b2k, int xf, dim3 grid, hipStream_t st);
int rn_gemm_launch_lean64x(const GemmK& k, bool a_kc, bool b_kc, int a2k, int b2k, int xf, dim3 grid, hipStream_t st);
// split-precision (bf16x3) 128x128 kernel, k-tiles of 16 (gemm_split.hip)
int rn_gemm_launch_split(const GemmK& k, bool a_kc, bool b_kc, int a2k, void* planes, dim3 grid, hipStream_t st, const void* ready = nullptr);
size_t rn_gemm_split_planes_bytes(int K, int N);
int rn_gemm_launch_shortk_split(const GemmK& k, bool b_kc, int ep, int c2_mode, void* planes, hipStream_t st, const void* ready = nullptr);
size_t rn_gemm_shortk_planes_bytes(int K, int N);
// persistent short-K kernel (gemm_shortk.hip): ep = (emul ? 1 : 0) | (accumulate ? 2 : 0)
int rn_gemm_launch_shortk(const GemmK& k, bool b_kc, int ep, int c2_mode, hipStream_t st);

