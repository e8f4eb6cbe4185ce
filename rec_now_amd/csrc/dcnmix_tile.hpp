// Row-block persistent forward of the DCN-v2 cross layers for the per-rank shards of the 2/4/8-GPU rows (dcnmix_tile.hip).
#pragma once
#include "common.hpp"

#define RN_TILE_MAX_L 8
// shapes the row-block kernels are instantiated for: two experts of 64, D = 256 / 512 / 1024, whole 32-row blocks
bool rn_mix_tile_supported(int64_t B, int D, int S, int N, int L, int LDT);
// bytes of the tile packs of all layers (fragment-ordered copies of U and W, see dcnmix_tile.hip); 0 for unsupported shapes
size_t rn_mix_tile_pack_bytes(int D, int S, int N, int L, int LDT);

struct RnTileFwd {
    const float* x;                         // (B, D) first input
    const float* U[RN_TILE_MAX_L];          // (N, D, S)
    const float* Kg[RN_TILE_MAX_L];         // (D, N) gate kernels
    const float* V[RN_TILE_MAX_L];          // (N, S, S)
    const float* W[RN_TILE_MAX_L];          // (N, S, D)
    const float* bias[RN_TILE_MAX_L];       // (N, D)
    float* packs;                           // rn_mix_tile_pack_bytes
    float* T1[RN_TILE_MAX_L];               // (B, LDT) saved activations, written
    float* T2[RN_TILE_MAX_L];
    float* T2g[RN_TILE_MAX_L];
    float* O[RN_TILE_MAX_L];                // (B, D) O_l = T2g_l [W; b], or NULL (not kept)
    float* xn[RN_TILE_MAX_L];               // (B, D) x_{l+1} = x * O_l, or NULL (not materialised); xn[L-1] = the layer output y when there is no head
    const float* head_w;                    // (D) scoring head folded into the last layer, or NULL
    const float* head_b;                    // (1) or NULL
    float* scores;                          // (B)
    int64_t B;
    int D, L, act_inner, act_outer;
};
// packs the weights (one launch) and runs every layer of the forward pass in ONE launch
int rn_mix_tile_fwd(const RnTileFwd& p, hipStream_t st);
// the pack launch alone (HOST arrays of L device pointers, as recnow_dcn_mix_fwd takes them)
int rn_mix_tile_pack(const float* const* U, const float* const* Kg, const float* const* V, const float* const* W, const float* const* bias, int D, int L,
                     float* packs, hipStream_t st);

// Data-gradient chain of the backward pass, layers l_hi .. l_lo (descending) in ONE launch: per layer dT2g -> sub-space backward -> g_l,
// with dx accumulated on the way.  The K = B weight-gradient products stay separate launches: they read what this kernel leaves
// (dT1_l, g_l) and the saved activations.
struct RnTileBwd {
    const float* x;                         // (B, D)
    const float* packs;                     // the forward's packs (P3, P4, V^T)
    const float* Kg[RN_TILE_MAX_L];         // (D, N)
    const float* bias[RN_TILE_MAX_L];       // (N, D)
    const float* T1[RN_TILE_MAX_L];         // saved activations
    const float* T2[RN_TILE_MAX_L];
    const float* O[RN_TILE_MAX_L];          // (B, D), read when dx != NULL
    float* dT1[RN_TILE_MAX_L];              // (B, LDT) out: [dA | dlogits | 0]
    float* gout[RN_TILE_MAX_L];             // (B, D) out: g_l = d loss / d x_l for l >= 1 (NULL for l = 0)
    const float* gin;                       // (B, D) gradient w.r.t. the output of layer l_hi, or NULL: the folded head's ds (x) head_w
    const float* ds;                        // (B) d loss / d scores (folded head)
    const float* head_w;                    // (D)
    float* dx;                              // (B, D) or NULL
    float* dvpart;                          // [layer][workgroup][N S S] partial sums of dV
    int64_t B;
    int D, L, l_hi, l_lo, act_inner, act_outer;
};
int rn_mix_tile_bwd_grid(int64_t B);        // workgroups of the launch = dV partials per layer
int rn_mix_tile_bwd(const RnTileBwd& p, hipStream_t st);
