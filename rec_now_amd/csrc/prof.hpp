// Optional per-launch HIP-event timing of the GEMM kernels (bench.py's roofline figure).  Off by default: no events
// are recorded and nothing is allocated.  Not thread-safe by design (one process per GPU, one launching thread).
#pragma once
#include "common.hpp"

struct RnProfRecord {
    int tag;            // kernel family (RN_TAG_*)
    double flops;       // algorithmic flops of the launch
    double bytes;       // algorithmic HBM bytes of the launch (operands read once + outputs written once)
    hipEvent_t e0, e1;
    bool closed;        // rn_prof_end has recorded e1 (a launch path that returned an error in between leaves a half-open record: skipped by the readers)
};
#define RN_TAG_GEMM_128x128 1
#define RN_TAG_GEMM_128x160 2
#define RN_TAG_GEMM_256x64 3
#define RN_TAG_GEMM_256x32 4
#define RN_TAG_GEMM_SHORTK 5        // k_gemm_shortk (persistent, K <= 512): a kernel of its own in rocprof, a family of its own here
#define RN_TAG_MIX_MID_FWD 6        // k_mix_mid_fwd (DCN-v2 sub-space stage, HBM-bound)
#define RN_TAG_MIX_MID_BWD 7        // k_mix_mid_bwd
#define RN_TAG_GEMM_SPLIT 8         // k_gemm_split (bf16x3 split-precision 128x128 products, opt-in)
#define RN_TAG_GEMM_MIDF 9          // k_gemm<128,128,..,25>: GEMM1 of DCN-v2 computed transposed with the sub-space forward in its epilogue
#define RN_TAG_MIX_TILE_FWD 10     // k_mix_tile_fwd: row-block persistent forward of all cross layers (shard sizes)
#define RN_TAG_MIX_TILE_BWD 11     // k_mix_tile_bwd: the data-gradient chain of the cross layers, row-block persistent
#define RN_TAG_GEMM_64x128 12       // k_gemm<64,128,..>: the small-M dispatch of the long-K products (shards below 256 row tiles of 128)
// Phase tags: recorded only in the every-launch mode (recnow_prof_sample_every(1)), where the intervals of ALL hooked launches give the
// step's account (recnow_prof_intervals: bench.py's exclusive time per kernel family under two streams).  In the sampled mode they would
// shift which launch positions the every-n-th rule picks.
#define RN_TAG_STEP_GROUP 13        // grouping of the batch (keys, radix sort, segments): RECNOW_STEP_GROUP
#define RN_TAG_STEP_LOSS 14         // the loss stage: pair walks, finalize, d loss / d scores (RECNOW_STEP_LOSS)
#define RN_TAG_LAYER_END 15         // weight packs, layer-end slab reductions, head post-processing
#define RN_TAG_MAX 16
#define RN_TAG_FIRST_PHASE 13

bool rn_prof_on();
// returns a slot (or nullptr when profiling is off / the pool is full) and records e0 on st
RnProfRecord* rn_prof_begin(int tag, double flops, double bytes, hipStream_t st);
void rn_prof_end(RnProfRecord* r, hipStream_t st);
