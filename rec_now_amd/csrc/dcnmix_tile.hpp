// Row-block persistent forward of the DCN-v2 cross layers for the per-rank shards of the 2/4/8-GPU rows (dcnmix_tile.hip).
#pragma once
#include "common.hpp"

#define RN_TILE_MAX_L 8
#define TL_NS 128
#define TL_PACK_FLOATS(D) (4 * (D) * TL_NS + 2 * 64 * 64)          // per layer: P1, P2 (forward), P3, P4, V^T (backward)
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
    int packed;                             // != 0: `packs` are already filled for these weights (the step's front kernel packed them beside the grouping)
    char* splanes;                          // split-precision forward (dcnmix_tile_split.hip): rn_mix_tile_split_pack_bytes, or NULL
};
// split-precision forward: piece planes of the weights in fragment order (16-byte units of 8 bf16), per layer [P1s: 3 planes][P2s: 3 planes]
#define TLS_P1_UNITS(D) ((D) * 16)          // per plane: D / 16 k-steps x 4 column blocks x 64 lanes
#define TLS_P2_UNITS(D) ((D) * 18)          // per plane: 9 k-steps x D / 32 d-blocks x 64 lanes
#define TLS_LAYER_BYTES(D) ((size_t)3 * (TLS_P1_UNITS(D) + TLS_P2_UNITS(D)) * 16)
size_t rn_mix_tile_split_pack_bytes(int D, int S, int N, int L, int LDT);
// packs the piece planes (one launch) and runs every layer of the forward pass in ONE launch on the bf16 MFMA (three pieces, six terms)
int rn_mix_tile_fwd_split(const RnTileFwd& p, hipStream_t st);
#ifdef __HIPCC__
// P1_l[kg][col][i] = U_l[n = col / 64][d = 4 kg + i][s = col % 64]         (D / 4 x 128 float4)      forward GEMM1, B fragments
// P2_l[g][h][d][i] = W_l[t = 8 g + 4 h + i][d],  W_l as (N S, D)            (16 x 2 x D float4)       forward output product, A fragments
// P3_l[kg][t][i]   = W_l[t][d = 4 kg + i]                                   (D / 4 x 128 float4)      backward dT2g product, B fragments
// P4_l[g][h][d][i] = U_l[n][d][s],  n S + s = 8 g + 4 h + i                 (16 x 2 x D float4)       backward g_l product, A fragments
// VT_l[n][t][s]    = V_l[n][s][t]                                                                      backward dA product, B fragments
// One thread per 16-byte output piece (the four i of a piece are four rows (P1, P2) or four consecutive elements (P3, P4) of the source), pieces first,
// first + stride, ...: the body of k_tile_pack (dcnmix_tile.hip) and of the pack workgroups of the step's front kernels (scan_sort.hip).
__device__ __forceinline__ void tl_pack_range(const RnTileFwd& p, int64_t first, int64_t stride) {
    const int D = p.D, L = p.L;
    const int64_t per4 = (int64_t)D * TL_NS / 4, lay4 = TL_PACK_FLOATS(D) / 4, total4 = (int64_t)L * lay4;
    rn_f4* __restrict__ out = reinterpret_cast<rn_f4*>(p.packs);
    for (int64_t e = first; e < total4; e += stride) {
        const int l = (int)(e / lay4);
        const int64_t j = e - (int64_t)l * lay4;
        const int which = (int)(j / per4);
        const int64_t k = j - (int64_t)which * per4;
        rn_f4 v;
        if (which == 0) {               // P1[kg][col]: U[n][4 kg + i][s]
            const int col = (int)(k & 127), kg = (int)(k >> 7);
            const float* src = p.U[l] + ((int64_t)(col >> 6) * D + 4 * kg) * 64 + (col & 63);
            v = rn_f4{src[0], src[64], src[128], src[192]};
        } else if (which == 2) {        // P3[kg][t]: W[t][4 kg .. 4 kg + 3]
            const int t = (int)(k & 127), kg = (int)(k >> 7);
            v = *reinterpret_cast<const rn_f4*>(p.W[l] + (int64_t)t * D + 4 * kg);
        } else if (which == 1) {        // P2[g][h][d]: W[4 (2 g + h) + i][d]
            const int d = (int)(k % D), t0 = 4 * (int)(k / D);
            const float* src = p.W[l] + (int64_t)t0 * D + d;
            v = rn_f4{src[0], src[D], src[2 * D], src[3 * D]};
        } else if (which == 3) {        // P4[g][h][d]: U[n][d][s .. s + 3],  n S + s = 4 (2 g + h)
            const int d = (int)(k % D), t0 = 4 * (int)(k / D);
            v = *reinterpret_cast<const rn_f4*>(p.U[l] + ((int64_t)(t0 >> 6) * D + d) * 64 + (t0 & 63));
        } else {                        // VT[n][t][s .. s + 3] = V[n][s + i][t]
            const int s4 = (int)(k & 15) * 4, t = (int)((k >> 4) & 63), n = (int)(k >> 10);
            const float* src = p.V[l] + (n * 64 + s4) * 64 + t;
            v = rn_f4{src[0], src[64], src[128], src[192]};
        }
        out[e] = v;
    }
}
#endif
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
