// Host-only dispatch logic of the exact-fp32 MFMA GEMM (gemm.hip): tile family, 64-row small-M rule, K split and slab workspace.  No HIP in here:
// gemm.hip includes it, and tools/san/dispatch_san.cpp compiles it with plain g++ under -fsanitize=address,undefined to sweep shapes on the CPU box
// (sanitizer build).  The two run-time switches that feed the rule come in through `RnDispatchEnv` so that the driver can set them.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include "../../include/recnow.h"

static inline int rnd_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

struct GemmCfg {
    int BM, BN;
};
struct RnDispatchEnv {
    int bm64_max_tiles;      // RECNOW_GEMM_BM64 (default 256; 0 = 64-row tiles off)
    int bm64_kb_mode;        // RECNOW_GEMM_BM64_KB (-1 default, 0 = only to fill the chip, 1 = always)
    int precision;           // rn_gemm_precision(): 0 fp32, 1 bf16x3 (128-row kernels only)
};

static inline GemmCfg pick_cfg(int N) {
    if (N <= 32) return {256, 32};
    if (N <= 64) return {256, 64};
    if (N > 128 && N <= 160) return {128, 160};
    return {128, 128};
}
// Small-M dispatch (round 4): the 128-column products with the two-wide side product -- every long-K product of the DCN-v2 step -- run on
// 64 x 128 tiles when 128-row tiles would leave at most `bm64_max_tiles` output tiles.  Default 256 = the 32 768-, 16 384- and 8192-row shards of the metric's
// 2-, 4- and 8-GPU rows (32 768 rows: 512 tiles in one round without a K split, 1.89 against 1.92 ms per step): at 16 384 rows 256 tiles x 2 K-slices (16 k-tiles each, half the slab traffic) instead of 128 x 4, measured 1.17 against 1.22 ms
// per step; at 8192 rows 128 x 4 fill all 512 workgroup slots where 64 x 4 filled half (the split is capped at 8 k-tiles per slice): 0.78 against
// 0.79 ms (tools/ab_bm64.sh, profiles/r04_small_m.md).  RECNOW_GEMM_BM64 = 0 switches it off, = N sets the tile bound (A/B).
static inline bool wants_bm64(const recnow_gemm_desc* d, const RnDispatchEnv& env) {
    if (d->N != 128 || d->sp_r < 1 || d->sp_r > 2 || d->eu_r > 0 || d->as_out || d->c2_mode || d->mid_V || d->batch != 1 || d->K < 512) return false;
    if (d->M % 64 || env.precision != 0) return false;      // (the opt-in split-precision kernels are 128-row kernels)
    const long long tiles128 = rnd_cdiv(d->M, 128);
    if (tiles128 > env.bm64_max_tiles) return false;
    // few row tiles and a long K (the K = B weight-gradient products, M = D): 64-row tiles where 128-row tiles cannot fill the 512 workgroup slots
    // under the split's cap of 8 k-tiles per slice (B = 8192: 8 tiles x 32 slices) and up to K = 32 768 (the shards: 1.114 against 1.128 ms per step at
    // 16 384 rows, 1.805 against 1.812 at 32 768); at the metric's own K = 65 536 the 128-row tiles are the faster ones (145 against 148 us per launch)
    if (tiles128 < 64 && env.bm64_kb_mode != 1) {
        long long s = (512 + tiles128 - 1) / tiles128;
        const long long maxs = d->K / (8 * 32);
        if (s > maxs) s = maxs;
        return tiles128 * (s > 0 ? s : 1) < 512 || (env.bm64_kb_mode != 0 && d->K <= 32768);
    }
    return true;
}
static inline GemmCfg pick_cfg(const recnow_gemm_desc* d, const RnDispatchEnv& env) {
    GemmCfg c = pick_cfg(d->N);
    if (c.BM == 128 && c.BN == 128 && wants_bm64(d, env)) c.BM = 64;
    return c;
}

// slots = workgroup slots the K split aims to fill: 512 (256 CUs x 2 resident workgroups) for a product that runs alone; 256 for the K = B
// weight-gradient products that dcnmix_bwd_tile launches as concurrent PAIRS on two streams (together they fill the chip with half the slices:
// 16 k-tiles per workgroup instead of 8 and half the slab traffic; 8192 rows 0.705 -> 0.680 ms per step, 16 384 rows 1.125 -> 1.098)
static inline void pick_split(const recnow_gemm_desc* d, const GemmCfg& c, int* splitk, int* kchunk, int slots = 512) {
    const long long tiles = (long long)rnd_cdiv(d->M, c.BM) * rnd_cdiv(d->N, c.BN) * d->batch;
    int s = 1;
    if (tiles < slots && !d->as_out && !d->c2_mode && !d->mid_V) {      // fused side / second outputs need the whole K in one workgroup
        s = (int)((slots + tiles - 1) / tiles);          // 256 CUs x 2 resident workgroups (256..511 tiles left half the slots empty until round 2)
        const int maxs = d->K / (8 * 32);      // at least 8 k-tiles per slice
        if (s > maxs) s = maxs;
        if (s < 1) s = 1;
    }
    // accuracy, not occupancy: an fp32 accumulator that walks K >= 16384 terms in sequence carries ~sqrt(K) roundings (the K = 32768
    // weight gradients of PLE sat at 0.95 of the 1e-5 parity bound); slabs of <= 8192 terms, summed in fp64 by the reduce, halve that
    if (s == 1 && d->K >= 16384 && !d->as_out && !d->c2_mode && !d->mid_V && d->c_perm_s == 0) s = d->K / 8192;
    int kc = rnd_cdiv(rnd_cdiv(d->K, s), 32) * 32;
    if (kc < 32) kc = 32;
    s = rnd_cdiv(d->K, kc);
    if (s < 1) s = 1;
    *splitk = s;
    *kchunk = kc;
}

// bytes of split-K slabs of a product (0: not split); `align` = the workspace carve granule
static inline size_t rnd_slab_bytes(const recnow_gemm_desc* d, int splitk, size_t align) {
    if (splitk <= 1) return 0;
    const size_t b = (size_t)splitk * d->batch * d->M * (d->N + (d->sp_r > 0 ? 4 : 0)) * sizeof(float);
    return (b + align - 1) / align * align;
}
// what a workspace query reserves: the larger of the two slot targets a launch may run with (fewer slots usually means fewer slices, but a product
// that is not split for occupancy any more may still be split for accuracy: tools/san found M 1000, N 4096, K 32 768 with 4 slices against 2)
static inline size_t rnd_slab_bytes_any(const recnow_gemm_desc* d, const GemmCfg& c, size_t align) {
    int s = 1, kc = 0;
    pick_split(d, c, &s, &kc, 512);
    size_t b = rnd_slab_bytes(d, s, align);
    pick_split(d, c, &s, &kc, 256);
    const size_t b2 = rnd_slab_bytes(d, s, align);
    return b2 > b ? b2 : b;
}
