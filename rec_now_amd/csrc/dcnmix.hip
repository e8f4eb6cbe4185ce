// DCNMixLayer (DCN-v2 mixture of low-rank experts, reference variant WITHOUT residual):
// /root/reference/rec_now/layers/dcn_mix_layer.py:114-151.  Per layer (x_l = layer input, x = first input):
//     A_n  = x_l U_n            (B,S)   :135      H1 = act_inner(A)        :136
//     C_n  = H1_n V_n           (B,S)   :137      H2 = act_outer(C)        :138
//     O_n  = H2_n W_n + b_n     (B,D)   :141-142  Q_n = x * O_n            :143
//     G    = softmax(x_l K)     (B,N)   :146-147  y   = sum_n G_n Q_n      :149
//
// Mapping onto the exact-fp32 MFMA GEMM (gemm.hip), chosen so that x_l / x / y are each streamed once per GEMM:
//   GEMM1:  T1 = x_l [U_0 | .. | U_{N-1} | K]          (B,D)x(D,NS+N): act_inner on the first NS columns, the last N
//           columns are the gate logits (no separate pass over x_l for the gate).
//   GEMM2:  H2_n = act_outer(H1_n V_n)                  batched over n, K = S.
//   gate :  G = softmax(logits);  T2g = [G_n * H2_n | G]   (elementwise, B x (NS+N))
//   GEMM3:  y = x * (T2g [W_0; ..; W_{N-1}; b])        (B,NS+N)x(NS+N,D) with the x-multiply as epilogue; the
//           gate-weighted bias sum_n G_n b_n rides along as N extra K rows.
// Backward mirrors this with the producers of each gradient GEMM fused into operand loads (x*dy) or epilogues
// (act' multiply), split-K slab reduction for the K = B weight gradients, and O recomputed for dx (dy * O).
// Leading dimension of the small activations: LDT = N*S+N rounded up to the GEMM's column-tile width (32/64/128/160/
// multiples of 128) and K of the (N*S+N)-deep products rounded up to 16/32, all zero padded, so every GEMM of the layer
// takes the lean interior kernel (no bounds code) whenever B is a multiple of 256.  Zero columns/rows are inert.
#include <mutex>
#include "gemm.hpp"
#include "dcnmix_mid.hpp"
#include "dcnmix_tile.hpp"
#include "prof.hpp"

static inline int ldt_of(int S, int N) {
    const int kc = N * S + N;
    if (kc <= 32) return 32;
    if (kc <= 64) return 64;
    if (kc <= 128) return 128;
    if (kc <= 160) return 160;
    return (kc + 127) / 128 * 128;
}
static inline int kp_of(int S, int N) {            // padded depth of the K = N*S+N products
    const int kc = N * S + N;
    return kc <= 256 ? (kc + 15) / 16 * 16 : (kc + 31) / 32 * 32;
}

// Wc1[d][n*S+s] = U[n][d][s];  Wc1[d][NS+n] = K[d][n];  pad columns = 0
__global__ void k_pack_w1(const float* __restrict__ U, const float* __restrict__ K, int D, int S, int N, int LDT, float* __restrict__ Wc1) {
    const int64_t total = (int64_t)D * LDT;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i / LDT), c = (int)(i % LDT);
        float v = 0.f;
        if (c < N * S) v = U[((int64_t)(c / S) * D + d) * S + (c % S)];
        else if (c < N * S + N) v = K[(int64_t)d * N + (c - N * S)];
        Wc1[i] = v;
    }
}
// exact path: Wc1[d][n*S+s] = U[n][d][s]   (D x NS, no gate columns)
__global__ void k_pack_u(const float* __restrict__ U, int D, int S, int N, float* __restrict__ Wc1) {
    const int NS = N * S;
    const int64_t total = (int64_t)D * NS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i / NS), c = (int)(i % NS);
        Wc1[i] = U[((int64_t)(c / S) * D + d) * S + (c % S)];
    }
}
__global__ void k_unpack_u(const float* __restrict__ dWc1, int D, int S, int N, float* __restrict__ dU) {
    const int NS = N * S;
    const int64_t total = (int64_t)D * NS;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i / NS), c = (int)(i % NS);
        dU[((int64_t)(c / S) * D + d) * S + (c % S)] = dWc1[i];
    }
}
// dU[n][d][s] = dWc1[d][n*S+s];  dK[d][n] = dWc1[d][NS+n]
__global__ void k_unpack_w1(const float* __restrict__ dWc1, int D, int S, int N, int LDT, float* __restrict__ dU, float* __restrict__ dK) {
    const int64_t total = (int64_t)D * (N * S + N);
    const int NS = N * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i / (NS + N)), c = (int)(i % (NS + N));
        const float v = dWc1[(int64_t)d * LDT + c];
        if (c < NS) dU[((int64_t)(c / S) * D + d) * S + (c % S)] = v;
        else dK[(int64_t)d * N + (c - NS)] = v;
    }
}

// one wave per row: G = softmax(T1[:, NS:NS+N]);  T2[:, NS+n] = G_n;  T2g = [G_n * H2_n | G]
__global__ void __launch_bounds__(256)
k_dcnmix_gate_fwd(const float* __restrict__ T1, float* __restrict__ T2, float* __restrict__ T2g, int64_t B, int S, int N, int LDT) {
    const int lane = threadIdx.x & 63;
    const int NS = N * S;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += (int64_t)gridDim.x * 4) {
        const float lg = lane < N ? T1[b * LDT + NS + lane] : -INFINITY;
        const float mx = wave_max(lg);
        const float e = lane < N ? expf(lg - mx) : 0.f;
        const float g = e / wave_sum(e);
        if (lane < N) {
            T2[b * LDT + NS + lane] = g;
            T2g[b * LDT + NS + lane] = g;
        }
        for (int c0 = 0; c0 < NS; c0 += 64) {          // wave-uniform trip count (shuffle sources must be active)
            const int c = c0 + lane;
            const float gn = __shfl(g, (c < NS ? c : 0) / S, 64);
            if (c < NS) T2g[b * LDT + c] = gn * T2[b * LDT + c];
        }
        for (int c = NS + N + lane; c < LDT; c += 64) { T2[b * LDT + c] = 0.f; T2g[b * LDT + c] = 0.f; }
    }
}

// one wave per row.  in: dT2g (B,LDT), T2 = [H2 | G].  out: dC[:, n*S+s] = G_n * dT2g * act_outer'(H2);
//                    dT1[:, NS+n] = dlogits_n = G_n * (dG_n - sum_m G_m dG_m),  dG_n = sum_s dT2g*H2 + dT2g[:, NS+n]
__global__ void __launch_bounds__(256)
k_dcnmix_gate_bwd(const float* __restrict__ dT2g, const float* __restrict__ T2, float* __restrict__ dC, float* __restrict__ dT1,
                  int64_t B, int S, int N, int LDT, int act_outer) {
    const int lane = threadIdx.x & 63;
    const int NS = N * S;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += (int64_t)gridDim.x * 4) {
        const float g = lane < N ? T2[b * LDT + NS + lane] : 0.f;
        float dg_mine = lane < N ? dT2g[b * LDT + NS + lane] : 0.f;
        for (int n = 0; n < N; ++n) {
            float p = 0.f;
            const float gn = __shfl(g, n, 64);
            for (int s = lane; s < S; s += 64) {
                const int c = n * S + s;
                const float h2 = T2[b * LDT + c], d = dT2g[b * LDT + c];
                p += d * h2;
                dC[b * LDT + c] = gn * d * rn_act_grad_from_out(h2, act_outer);
            }
            p = wave_sum(p);
            if (lane == n) dg_mine += p;
        }
        const float dot = wave_sum(g * dg_mine);
        if (lane < N) dT1[b * LDT + NS + lane] = g * (dg_mine - dot);
        for (int c = NS + N + lane; c < LDT; c += 64) { dT1[b * LDT + c] = 0.f; dC[b * LDT + c] = 0.f; }
        for (int c = NS + lane; c < NS + N; c += 64) dC[b * LDT + c] = 0.f;
    }
}

struct MixDims {
    int64_t B;
    int D, S, N, L, NS, KC, LDT, KP;   // KC = NS + N; LDT/KP = padded width / depth
    bool exact;                        // exact-128 formulation (side products + rank-N epilogue updates)
};
// Exact formulation: every D-sized product has exactly NS MFMA columns / NS-deep K; the N gate columns and the N
// gate-weighted bias rows are VALU side products / rank-N epilogue updates of the lean 128x128 GEMM kernels.
static inline bool mix_exact(int64_t B, int D, int S, int N) {
    return (N * S) % 128 == 0 && D % 128 == 0 && B % 256 == 0 && N <= 4 && kp_of(S, N) <= 512;     // second outputs: K <= 512
}
static inline MixDims mix_dims(int64_t B, int D, int S, int N, int L) {
    MixDims m;
    m.B = B; m.D = D; m.S = S; m.N = N; m.L = L; m.NS = N * S; m.KC = N * S + N; m.LDT = ldt_of(S, N); m.KP = kp_of(S, N);
    m.exact = mix_exact(B, D, S, N);
    if (m.exact) m.LDT = m.KP;      // [NS main | N gate | zero pad to a multiple of 16]: N = NS products + side product, K = KP products
    return m;
}
static inline size_t act_block(const MixDims& m) { return rn_align((size_t)m.B * m.LDT * sizeof(float)); }
static inline size_t xbuf(const MixDims& m) { return rn_align((size_t)m.B * m.D * sizeof(float)); }
#define MIX_PACK_MAX_L 8
// exact path, L <= MIX_PACK_MAX_L: the packed weights of every layer -- Wc1_l = [U_l | K_l | 0] (D x LDT), Wc2_l = [W_l; b_l; 0]
// (LDT x D) and one (LDT x D) block for the fused head's pre-scaled top-layer weights -- live at the end of `saved`: the forward
// packs them ONCE per step (one launch) and the backward reads them there instead of packing again.
static inline size_t mix_pack_one(const MixDims& m) { return rn_align((size_t)m.D * m.LDT * sizeof(float)); }
static inline size_t mix_pack_bytes(const MixDims& m) { return (m.exact && m.L <= MIX_PACK_MAX_L) ? (size_t)(2 * m.L + 1) * mix_pack_one(m) : 0; }
static inline size_t mix_pack_off(const MixDims& m) {       // byte offset of the pack region inside `saved`
    return (size_t)m.L * 3 * act_block(m) + (size_t)(m.L - 1) * xbuf(m) + (m.exact ? (size_t)m.L * xbuf(m) : 0);
}

// Row-block persistent kernels (dcnmix_tile.hip): their fragment-ordered weight packs live behind the packs above.  RECNOW_TILE=0
// switches the route off, =1 takes it for every supported batch (tests); default: batches up to MIX_TILE_MAX_B rows (the per-rank shards of the
// 4- and 8-GPU rows), where the launch-per-product route is a chain of single-round launches (measured: 8192 rows 0.76 -> 0.67-0.69 ms per step,
// 16 384 rows 1.13 -> 1.10-1.13; 32 768 rows 1.87 -> 1.93 and 65 536 rows 3.32 -> 3.85: off there).
#define MIX_TILE_MAX_B 16384
static inline bool mix_tile_shape(const MixDims& m) {
    return m.exact && m.L <= MIX_PACK_MAX_L && m.L <= RN_TILE_MAX_L && rn_mix_tile_supported(m.B, m.D, m.S, m.N, m.L, m.LDT);
}
static inline size_t mix_tile_pack_bytes(const MixDims& m) { return mix_tile_shape(m) ? rn_mix_tile_pack_bytes(m.D, m.S, m.N, m.L, m.LDT) : 0; }
static inline size_t mix_tile_pack_off(const MixDims& m) { return mix_pack_off(m) + mix_pack_bytes(m); }
// Round 6: the split-precision piece planes of the packed weights, behind the tile packs: per layer P1 ([U | K] as the B operand of GEMM1),
// P2 ([W; b] of the product that leaves the layer), P3 (W^T -- or the fused head's W * w_head -- of the dT2g product), P4 ([U | K]^T of the
// product that forms g_{l-1}); written ONCE per step by the forward (rn_split_planes_multi behind k_pack_all) instead of by a split launch in
// front of each of the 12 products that read them.
static inline bool mix_planes_shape(const MixDims& m) { return m.exact && m.L <= MIX_PACK_MAX_L && 4 * m.L <= RN_SPLIT_MAX_JOBS && m.NS == 128 && m.KP == 144; }
static inline size_t mix_plane_long(const MixDims& m) { return rn_gemm_split_planes_bytes(m.D, 128); }
static inline size_t mix_plane_short(const MixDims& m) { return rn_gemm_split_planes_bytes(m.KP, m.D); }
static inline size_t mix_planes_bytes(const MixDims& m) { return mix_planes_shape(m) ? (size_t)m.L * 2 * (mix_plane_long(m) + mix_plane_short(m)) : 0; }
static inline size_t mix_planes_off(const MixDims& m) { return mix_tile_pack_off(m) + mix_tile_pack_bytes(m); }
// Round 6, second session: the fragment-ordered piece planes of the split-precision row-block forward (dcnmix_tile_split.hip), behind the planes above
static inline size_t mix_tile_split_bytes(const MixDims& m) { return mix_tile_shape(m) ? rn_mix_tile_split_pack_bytes(m.D, m.S, m.N, m.L, m.LDT) : 0; }
static inline size_t mix_tile_split_off(const MixDims& m) { return mix_planes_off(m) + mix_planes_bytes(m); }
static inline char* mix_plane(const MixDims& m, const void* saved, int l, int which) {      // which: 0 = P1, 1 = P2, 2 = P3, 3 = P4
    char* base = (char*)saved + mix_planes_off(m) + (size_t)l * 2 * (mix_plane_long(m) + mix_plane_short(m));
    return base + (which == 0 ? 0 : which == 1 ? mix_plane_long(m) : which == 2 ? mix_plane_long(m) + mix_plane_short(m) : 2 * mix_plane_long(m) + mix_plane_short(m));
}
// `maybe`: everything but the precision mode -- what a backward pass uses to decide whether the forward that filled `saved` MAY have run the
// row-block kernels (and left the product-route weight packs out), whatever the precision switch says by now
// Set by the GROUP phase of recnow_dcn_mix_step when its front kernel has written the tile packs of THIS call's weights into `saved`; taken (and cleared)
// by the row-block forward of the same call, which then skips its own pack launch.  Per host thread: a step is enqueued by one thread.
static thread_local const void* tl_step_tile_packed = nullptr;      // the `saved` buffer whose tile packs are current, or NULL
static bool mix_tile_on(const MixDims& m, bool maybe = false) {
    const char* e = getenv("RECNOW_TILE");            // read per call (tests switch the route inside one process)
    const int mode = e ? atoi(e) : -1;
    if (mode == 0 || !mix_tile_shape(m) || (!maybe && rn_gemm_precision() != 0)) return false;
    return mode == 1 || m.B <= MIX_TILE_MAX_B;
}

// Split-precision products (recnow_set_gemm_precision(1)): the FORWARD pass of all cross layers as one row-block launch on the bf16 MFMA
// (k_mix_tile_fwd_s3), the backward pass stays on the launch-per-product route (which then finds the saved activations, O_l and x_{l+1} exactly where
// the product-route forward leaves them).  RECNOW_TILE_SPLIT: 0 off, 1 every supported batch; default: see MIX_TILE_SPLIT_MIN_B.
#define MIX_TILE_SPLIT_MIN_B (1ll << 62)
static bool mix_tile_split_on(const MixDims& m) {
    if (rn_gemm_precision() != 1 || !mix_tile_shape(m)) return false;
    const char* e = getenv("RECNOW_TILE_SPLIT");          // read per call (tests switch the route inside one process)
    const int mode = e ? atoi(e) : -1;
    if (mode == 0) return false;
    return mode == 1 || m.B >= MIX_TILE_SPLIT_MIN_B;
}

// The row-block backward chain walks all its layers in one launch and leaves g_l of EVERY layer for the weight-gradient products behind it: the two
// ping-pong gradient buffers of the workspace hold g_1 and g_2, i.e. at most three layers (deeper stacks take the product-route backward, which
// consumes g_l layer by layer).  RECNOW_TILE_BWD=0: the product-route backward behind the row-block forward (A/B switch, read per call).
static bool mix_tile_bwd_on(const MixDims& m) {
    const char* e = getenv("RECNOW_TILE_BWD");
    return m.L <= 3 && !(e && e[0] == '0');
}

// Which weight packs the forward that filled a `saved` buffer left in it, and which route the top piece of the backward took.
// The route rule (RECNOW_TILE, the precision mode) is read per call, so it can change between a forward and its backward: a product-route forward
// (precision 1, RECNOW_TILE=0) followed by a row-block backward would read tile packs nobody wrote, and the other way round.  The forward records
// what it packed under the address of `saved` (host side: a stamp inside the device buffer could not be read back without a synchronisation); the
// backward packs what is missing for the route it takes, and the lower pieces of a backward cut into layer ranges follow the top piece.  A buffer
// this process has no record of (filled through another copy of the library, or 256 forwards ago) is "unknown": the backward then packs for itself.
enum { MIX_HAS_PRODUCT_PACKS = 1, MIX_HAS_TILE_PACKS = 2, MIX_BWD_TILE = 4, MIX_HAS_SPLIT_PLANES = 8, MIX_PLANES_HEAD = 16, MIX_XLESS = 32 };      // (.._XLESS: the forward did not materialise x_{l+1})      // (.._HEAD: the top layer's P3 holds W * w_head)
struct MixStamp { const void* sv; int bits; };
static std::mutex g_mix_stamp_mu;
static MixStamp g_mix_stamps[256];
static unsigned g_mix_stamp_next = 0;
static void mix_stamp_put(const void* sv, int bits) {
    std::lock_guard<std::mutex> lk(g_mix_stamp_mu);
    for (int i = 0; i < 256; ++i)
        if (g_mix_stamps[i].sv == sv) { g_mix_stamps[i].bits = bits; return; }
    g_mix_stamps[g_mix_stamp_next++ % 256] = MixStamp{sv, bits};
}
static int mix_stamp_get(const void* sv) {
    std::lock_guard<std::mutex> lk(g_mix_stamp_mu);
    for (int i = 0; i < 256; ++i)
        if (g_mix_stamps[i].sv == sv) return g_mix_stamps[i].bits;
    return -1;
}

// saved layout, per layer l: T1, T2, T2g (B x LDT each); then the L-1 intermediate layer outputs x_1..x_{L-1} (B x D);
// exact path: then O_0..O_{L-1} (B x D)
extern "C" size_t recnow_dcn_mix_saved_bytes(int64_t B, int D, int S, int N, int L) {
    if (B <= 0 || D <= 0 || S <= 0 || N <= 0 || L <= 0) return 256;
    const MixDims m = mix_dims(B, D, S, N, L);
    // exact path: O_l = T2g_l [W; b] of every layer is kept next to x_{l+1} = x * O_l (second output of GEMM3), so the
    // backward forms dx = sum_l g_l * O_l inside kernels that stream g_l anyway instead of recomputing the products.
    return (size_t)L * 3 * act_block(m) + (size_t)(L - 1) * xbuf(m) + (m.exact ? (size_t)L * xbuf(m) : 0) + mix_pack_bytes(m) + mix_tile_pack_bytes(m) + mix_planes_bytes(m) + mix_tile_split_bytes(m) + 256;
}

// Round 4: x_{l+1} = x0 * O_l is NOT materialised between the cross layers of the exact path (two experts).  The product that leaves layer l
// writes O_l = T2g_l [W; b] only (it no longer reads x0 nor writes x_{l+1}: 306 MB instead of 842 MB per launch at B = 65 536, which turns this
// HBM-bound launch into an MFMA-bound one), and the two consumers of x_{l+1} -- GEMM1 of layer l + 1 and its weight-gradient product dU -- form
// x0 * O_l in their operand loads (RECNOW_OPMODE_MUL), where the extra stream rides under MFMA-bound k-loops (+7 us per launch measured on the
// products that already do this for x0 * g).  Same values bit for bit: one fp32 multiply either way.  RECNOW_XLESS=0 switches it off (A/B).
static bool mix_xless(const MixDims& m) {
    static const bool on = []() { const char* e = getenv("RECNOW_XLESS"); return !e || e[0] != '0'; }();
    // measured (tools/ab_xless.sh, one box): 8192 rows 0.768 -> 0.758 ms, 16 384 rows 1.176 -> 1.156 ms per step; at 65 536 rows NEUTRAL (3.445 vs 3.450 ms:
    // the short-K launches lose 13 us on average, the nine long-K launches with a second operand stream gain 5 us each) -- so only below the batch at
    // which GEMM1 carries the sub-space forward in its epilogue (512 row blocks)
    // Round 6: with the split-precision products at every batch -- the launch that leaves a layer is HBM-bound there (842 -> 306 MB, 190 -> 108 us), the two
    // consumers of x_{l+1} pay 34 + 16 us for their second operand stream: 65 536 rows 2.843 / 2.858 -> 2.825 / 2.811 ms per step (one box, alternating),
    // 0.5 GB less HBM traffic per step.  RECNOW_XLESS=2 forces it for the exact products too, =1 keeps the round-4 rule (A/B).  The rule is read by the FORWARD;
    // a backward follows what its forward did (the MIX_XLESS stamp of `saved`), whatever the precision switch says by then.
    static const int mode = []() { const char* e = getenv("RECNOW_XLESS"); return e ? atoi(e) : -1; }();
    return on && m.exact && m.N <= 2 && m.L > 1 && (mode == 2 || m.B / 128 < 512 || (mode != 1 && rn_gemm_precision() == 1));
}
static bool mix_xless_saved(const MixDims& m, const void* sv) {      // what the forward that filled `saved` did (unknown buffer: the rule)
    const int have = mix_stamp_get(sv);
    return have >= 0 ? (have & MIX_XLESS) != 0 : mix_xless(m);
}

static size_t mix_gemm_ws(const MixDims& m) {
    size_t best = 0;
    recnow_gemm_desc d = rn_gemm_desc_zero();
    const int shapes[6][3] = {{(int)m.B, m.LDT, m.D}, {(int)m.B, m.D, m.KP}, {m.LDT, m.D, (int)m.B}, {m.D, m.LDT, (int)m.B},
                              {m.S, m.S, (int)m.B}, {(int)m.B, m.S, m.S}};
    for (int i = 0; i < 6; ++i) {
        d.M = shapes[i][0]; d.N = shapes[i][1]; d.K = shapes[i][2]; d.batch = (i >= 4) ? m.N : 1;
        const size_t s = rn_gemm_ws_bytes(&d);
        if (s > best) best = s;
    }
    if (m.exact) {           // exact-path split-K products carry 4 side columns per slab row
        d.M = m.D; d.N = m.NS; d.K = (int)m.B; d.batch = 1; d.sp_r = m.N; d.a_trans = 1;
        size_t s = rn_gemm_ws_bytes(&d);
        if (s > best) best = s;
        d.M = (int)m.B; d.N = m.NS; d.K = m.D; d.a_trans = 0;       // x_l U / (x*g) W^T: room for the split-precision planes of the weights
        s = rn_gemm_ws_bytes(&d);
        if (s > best) best = s;
    }
    return best;
}

extern "C" size_t recnow_dcn_mix_workspace_bytes(int64_t B, int D, int S, int N, int L) {
    if (B <= 0 || D <= 0 || S <= 0 || N <= 0 || L <= 0) return 256;
    const MixDims m = mix_dims(B, D, S, N, L);
    size_t s = 0;
    s += rn_align((size_t)D * m.LDT * sizeof(float));        // Wc1
    s += rn_align((size_t)m.LDT * D * sizeof(float));        // Wc2 (rows >= KC are zero)
    s += rn_align((size_t)D * m.LDT * sizeof(float));        // dWc1
    s += rn_align((size_t)m.LDT * D * sizeof(float));        // dWc2
    s += 3 * act_block(m);                                   // dT2g, dC, dT1
    s += 2 * xbuf(m);                                        // inter-layer gradient ping-pong
    s += 3 * rn_align(mix_gemm_ws(m));                       // split-K slabs: chain stream, dU and dW (the last two are reduced together at the layer's end)
    s += rn_mix_mid_bwd_ws_bytes(B, S, N);                   // per-workgroup dV partials of the fused sub-space backward
    s += (size_t)L * (rn_align((size_t)D * m.LDT * sizeof(float)) + rn_align((size_t)m.LDT * D * sizeof(float)));   // per-layer packs
    s += rn_align(rn_colsum_ws_bytes(B, 1));                 // fused scoring head: d bias = sum of dscores
    if (mix_tile_shape(m)) {                                 // row-block backward: dT1 of every layer, dV partials per layer and workgroup
        s += (size_t)L * act_block(m);
        s += rn_align((size_t)L * rn_mix_tile_bwd_grid(B) * N * S * S * sizeof(float));
        // ... and one split-K slab buffer per weight-gradient product (2 L of them; three are carved above): behind the chain launch all six
        // products are independent, and sharing slab buffers made each wait for an earlier layer's reduction (round 5, kernel trace at 8192 rows)
        if (2 * L > 3) s += (size_t)(2 * L - 3) * rn_align(mix_gemm_ws(m));
    }
    return s + 4096;
}


// One launch packs the weights of ALL layers (exact path): Wc1_l = [U_l | K_l | 0] (D x LDT) and, when Wc2 != NULL,
// Wc2_l = [W_l; b_l; 0] (LDT x D).  Replaces 4 tiny launches per layer in front of every product.
struct MixPackPtrs {
    const float* U[MIX_PACK_MAX_L];
    const float* K[MIX_PACK_MAX_L];
    const float* W[MIX_PACK_MAX_L];
    const float* b[MIX_PACK_MAX_L];
};
__global__ void __launch_bounds__(256)
k_pack_all(MixPackPtrs p, int L, int D, int S, int N, int LDT, float* __restrict__ Wc1, float* __restrict__ Wc2,
           const float* __restrict__ wh, float* __restrict__ Wh) {
    const int NS = N * S, KC = NS + N;
    const int64_t per = (int64_t)D * LDT, packs = (int64_t)L * per * (Wc2 ? 2 : 1), total = packs + (Wh ? (int64_t)KC * D : 0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (i >= packs) {                                // fused scoring head: Wh[k][c] = [W; b]_{L-1}[k][c] * w_head[c]
            const int64_t e = i - packs;
            const int k = (int)(e / D), c = (int)(e % D);
            Wh[e] = (k < NS ? p.W[L - 1][e] : p.b[L - 1][(int64_t)(k - NS) * D + c]) * wh[c];
            continue;
        }
        const int64_t j = i % ((int64_t)L * per);
        const int l = (int)(j / per);
        const int64_t e = j - (int64_t)l * per;
        if (i < (int64_t)L * per) {                      // Wc1_l[d][c]
            const int d = (int)(e / LDT), c = (int)(e % LDT);
            float v = 0.f;
            if (c < NS) v = p.U[l][((int64_t)(c / S) * D + d) * S + (c % S)];
            else if (c < KC) v = p.K[l][(int64_t)d * N + (c - NS)];
            Wc1[j] = v;
        } else {                                         // Wc2_l[k][d]
            const int k = (int)(e / D), d = (int)(e % D);
            float v = 0.f;
            if (k < NS) v = p.W[l][(int64_t)k * D + d];
            else if (k < KC) v = p.b[l][(int64_t)(k - NS) * D + d];
            Wc2[j] = v;
        }
    }
}
static int pack_all(const MixDims& m, const float* const* U_host, const float* const* W_host, const float* const* bias_host,
                    const float* const* gate_host, float* Wc1_all, float* Wc2_all, const float* head_w, float* Wh, hipStream_t st) {
    MixPackPtrs p;
    for (int l = 0; l < m.L; ++l) {
        p.U[l] = U_host[l]; p.K[l] = gate_host[l]; p.W[l] = W_host[l]; p.b[l] = bias_host[l];
    }
    const int64_t total = (int64_t)m.L * m.D * m.LDT * (Wc2_all ? 2 : 1) + (Wh ? (int64_t)m.KC * m.D : 0);
    int g = rn_cdiv(total, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_pack_all, g, 256, 0, st, p, m.L, m.D, m.S, m.N, m.LDT, Wc1_all, Wc2_all, head_w, head_w ? Wh : nullptr);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

static int pack_weights(const MixDims& m, const float* U, const float* V, const float* W, const float* bias, const float* K,
                        float* Wc1, float* Wc2, hipStream_t st) {
    (void)V;
    int g = rn_cdiv((int64_t)m.D * m.LDT, 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_pack_w1, g, 256, 0, st, U, K, m.D, m.S, m.N, m.LDT, Wc1);
    RN_LAUNCH_CHECK();
    // Wc2 = [W (NS x D); bias (N x D); zero rows up to LDT] -- W and bias are already row-major contiguous
    RN_HIP(hipMemsetAsync(Wc2 + (size_t)m.KC * m.D, 0, (size_t)(m.LDT - m.KC) * m.D * sizeof(float), st));
    RN_HIP(hipMemcpyAsync(Wc2, W, (size_t)m.NS * m.D * sizeof(float), hipMemcpyDeviceToDevice, st));
    RN_HIP(hipMemcpyAsync(Wc2 + (size_t)m.NS * m.D, bias, (size_t)m.N * m.D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return RECNOW_OK;
}

static inline int gate_grid(int64_t B) {
    int64_t g = (B + 3) / 4;
    if (g > 4096) g = 4096;
    return (int)(g > 0 ? g : 1);
}


// Sub-space stage: the fused streaming kernels of dcnmix_mid.hip when the shape fits them, else batched GEMMs + gate kernels.
static int mix_mid_fwd(const MixDims& m, const float* T1, const float* V, float* T2, float* T2g, int act_outer, void* gws,
                       size_t gws_bytes, hipStream_t st, const RnSlabs* slabs = nullptr, int act_inner = 0) {
    if (rn_mix_mid_supported(m.S, m.N, m.LDT)) return rn_mix_mid_fwd(T1, V, T2, T2g, m.B, m.S, m.N, m.LDT, act_outer, st, slabs, act_inner);
    if (slabs) return RECNOW_EUNSUPPORTED;
    int rc;
    {   // GEMM2: H2_n = act_outer(H1_n V_n), batched over the N experts
        recnow_gemm_desc d = rn_gemm_desc_zero();
        d.A = T1; d.lda = m.LDT; d.a_batch_stride = m.S; d.a_trans = 0;
        d.B = V; d.ldb = m.S; d.b_batch_stride = (int64_t)m.S * m.S; d.b_trans = 0;
        d.C = T2; d.ldc = m.LDT; d.c_batch_stride = m.S;
        d.M = (int)m.B; d.N = m.S; d.K = m.S; d.batch = m.N;
        d.act = act_outer;
        if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
    }
    hipLaunchKernelGGL(k_dcnmix_gate_fwd, gate_grid(m.B), 256, 0, st, T1, T2, T2g, m.B, m.S, m.N, m.LDT);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
// dT2g -> dT1 = [dA | dlogits | 0] and dV.  dC is scratch of the unfused route only; mid_ws holds the fused route's partials.
static int mix_mid_bwd(const MixDims& m, const float* dT2g, const float* T2, const float* T1, const float* V, float* dC, float* dT1,
                       float* dV, int act_inner, int act_outer, void* mid_ws, size_t mid_ws_bytes, void* gws, size_t gws_bytes,
                       hipStream_t st) {
    if (rn_mix_mid_supported(m.S, m.N, m.LDT))
        return rn_mix_mid_bwd(dT2g, T2, T1, V, dT1, dV, m.B, m.S, m.N, m.LDT, act_inner, act_outer, mid_ws, mid_ws_bytes, st);
    int rc;
    hipLaunchKernelGGL(k_dcnmix_gate_bwd, gate_grid(m.B), 256, 0, st, dT2g, T2, dC, dT1, m.B, m.S, m.N, m.LDT, act_outer);
    RN_LAUNCH_CHECK();
    {   // dV_n = H1_n^T dC_n            (S x S), K = B
        recnow_gemm_desc d = rn_gemm_desc_zero();
        d.A = T1; d.lda = m.LDT; d.a_batch_stride = m.S; d.a_trans = 1;
        d.B = dC; d.ldb = m.LDT; d.b_batch_stride = m.S; d.b_trans = 0;
        d.C = dV; d.ldc = m.S; d.c_batch_stride = (int64_t)m.S * m.S;
        d.M = m.S; d.N = m.S; d.K = (int)m.B; d.batch = m.N;
        if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
    }
    {   // dA_n = (dC_n V_n^T) * act_inner'(H1_n)  -> first NS columns of dT1
        recnow_gemm_desc d = rn_gemm_desc_zero();
        d.A = dC; d.lda = m.LDT; d.a_batch_stride = m.S; d.a_trans = 0;
        d.B = V; d.ldb = m.S; d.b_batch_stride = (int64_t)m.S * m.S; d.b_trans = 1;
        d.C = dT1; d.ldc = m.LDT; d.c_batch_stride = m.S;
        d.M = (int)m.B; d.N = m.S; d.K = m.S; d.batch = m.N;
        d.emul = T1; d.lde = m.LDT; d.e_batch_stride = m.S; d.e_mode = RECNOW_OPMODE_ACTGRAD; d.e_act = act_inner;
        if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
    }
    return RECNOW_OK;
}


// ---- model-level fusion: the cross layers + a Dense(1) scoring head ------------------------------------------
// score[m] = y[m] . w_head + b_head with y = x * O_{L-1} the last layer's output
// (reference rec_now/layers/dcn_mix_layer.py:149-150 -> multi_dense_layer.py:90-92 with units = 1, num_dnn = 1).
// Forward: the last GEMM3's epilogue forms y's tile in registers, leaves its row-dot with w_head as 2 * D / 128 partials per row
// (c2_mode 3) and never stores y.  Backward: the head's gradient dy = dscore (x) w_head is rank one and is never materialised:
//   dT2g = dscore[m] * (x (W * w_head)^T)          -> plain product with pre-scaled weights, rows scaled when dT2g is read
//   M    = x^T (dscore * T2g)                      -> dW = w_head * M^T, dbias likewise, and d w_head[c] = sum_k [W; b][k][c] M[c][k]
//   dx  += dscore[m] * w_head[c] * O_{L-1}[m][c]   -> added in the first kernel that writes dx (c2_mode 4)
struct MixHead {            // forward
    const float* w;         // (D)
    const float* b;         // (1) or NULL
    float* scores;          // (B)
};
struct MixHeadGrad {        // backward
    const float* w;         // (D)
    const float* dscores;   // (B)
    float* dw;              // (D)
    float* db;              // (1) or NULL
};

__global__ void __launch_bounds__(256)
k_head_scores(const float* __restrict__ hp, int np, const float* __restrict__ bias, int64_t B, float* __restrict__ scores) {
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < B; m += (int64_t)gridDim.x * 256) {
        float s = bias ? bias[0] : 0.f;
        for (int p = 0; p < np; ++p) s += hp[m * np + p];          // fixed order
        scores[m] = s;
    }
}
// out[m][:] = rs[m] * in[m][:]   (B x LD, float4)
__global__ void __launch_bounds__(256)
k_row_scale(const float* __restrict__ in, const float* __restrict__ rs, int64_t B, int LD, float* __restrict__ out) {
    const int64_t total = B * (LD / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float sc = rs[i / (LD / 4)];
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        reinterpret_cast<float4*>(out)[i] = v;
    }
}
// in place: Mw (NS x D) and Mb (N x D) hold M^T = (x^T (dscore * T2g))^T;  dwh[c] = sum_k W[k][c] Mw[k][c] + sum_n bias[n][c] Mb[n][c];
// then Mw *= wh[c] (= dW) and Mb *= wh[c] (= dbias).  A workgroup owns 32 columns; its 8 row groups walk the NS + N rows with a
// stride of 8 (coalesced 128-byte row pieces) and their partial sums are added in a fixed order.
__global__ void __launch_bounds__(256)
k_head_post(float* __restrict__ Mw, float* __restrict__ Mb, const float* __restrict__ W, const float* __restrict__ bias,
            const float* __restrict__ wh, int NS, int N, int D, float* __restrict__ dwh, const float* __restrict__ ds_part, int n_part,
            float* __restrict__ dhb) {
    __shared__ float part[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    float s = 0.f;
    if (c < D) {
        const float w = wh[c];
        for (int k = rg; k < NS + N; k += 8) {
            float* mp = k < NS ? Mw + (int64_t)k * D + c : Mb + (int64_t)(k - NS) * D + c;
            const float wk = k < NS ? W[(int64_t)k * D + c] : bias[(int64_t)(k - NS) * D + c];
            const float m = *mp;
            s += wk * m;
            *mp = m * w;
        }
    }
    part[rg][threadIdx.x & 31] = s;
    __syncthreads();
    if (rg == 0 && c < D) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += part[r][threadIdx.x & 31];
        dwh[c] = t;
    }
    // recnow_dcn_mix_step: d head bias = sum of dscores, from the per-workgroup partial sums its loss stage left (fixed order: thread
    // t sums partials t, t + 256, ..; then a fixed tree)
    if (dhb && blockIdx.x == 0) {
        __shared__ float red[16];
        float t = 0.f;
        for (int i = threadIdx.x; i < n_part; i += 256) t += ds_part[i];
        t = block_sum<float>(t, red);
        if (threadIdx.x == 0) dhb[0] = t;
    }
}
// dx[m][c] = ds[m] * wh[c] * O[m][c]   (single cross layer under a fused head: the only term that is not a product)
__global__ void __launch_bounds__(256)
k_head_dx_top(const float* __restrict__ O, const float* __restrict__ ds, const float* __restrict__ wh, int64_t B, int D, float* __restrict__ dx) {
    const int64_t total = B * (D / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / (D / 4);
        const int c4 = (int)(i % (D / 4));
        const float sc = ds[m];
        const float4 o = reinterpret_cast<const float4*>(O)[i], w = reinterpret_cast<const float4*>(wh)[c4];
        reinterpret_cast<float4*>(dx)[i] = make_float4(sc * w.x * o.x, sc * w.y * o.y, sc * w.z * o.z, sc * w.w * o.w);
    }
}
static inline int ew_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    return (int)(g > 0 ? g : 1);
}
// the fused head rides on the exact-128 formulation with the fused sub-space kernels
static inline bool mix_head_ok(const MixDims& m) {
    return m.exact && rn_mix_mid_supported(m.S, m.N, m.LDT) && 2 * (m.D / 128) <= m.LDT && m.L <= MIX_PACK_MAX_L;
}

static int dcnmix_fwd_impl(const float* x, const float* const* U_host, const float* const* V_host,
                                  const float* const* W_host, const float* const* bias_host, const float* const* gate_host,
                                  int64_t B, int D, int S, int N, int L, int act_inner, int act_outer, float* y, void* saved,
                                  size_t saved_bytes, void* ws, size_t ws_bytes, void* stream, int need_dx, const MixHead* head) {
    if (B < 0 || D < 1 || S < 1 || N < 1 || L < 1 || B > 0x7fffffffll) return RECNOW_EINVAL;
    if (N > 64) return RECNOW_EUNSUPPORTED;
    if (B == 0) return RECNOW_OK;
    if (!x || !U_host || !V_host || !W_host || !bias_host || !gate_host || (!y && !head) || !saved || !ws) return RECNOW_EINVAL;
    if (head && (!head->w || !head->scores)) return RECNOW_EINVAL;
    if (saved_bytes < recnow_dcn_mix_saved_bytes(B, D, S, N, L)) return RECNOW_EWORKSPACE;
    if (ws_bytes < recnow_dcn_mix_workspace_bytes(B, D, S, N, L)) return RECNOW_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const MixDims m = mix_dims(B, D, S, N, L);
    if (head && !mix_head_ok(m)) return RECNOW_EUNSUPPORTED;
    RnCarver c(ws, ws_bytes);
    float* Wc1 = c.take<float>((size_t)D * m.LDT);
    float* Wc2 = c.take<float>((size_t)m.LDT * D);
    c.take<float>((size_t)D * m.LDT);
    c.take<float>((size_t)m.LDT * D);
    float* hp = c.take<float>(3 * act_block(m) / sizeof(float));      // forward: free -> the head's row-dot partials (B x 2D/128)
    c.take<float>(2 * xbuf(m) / sizeof(float));
    const bool pack_once = m.exact && L <= MIX_PACK_MAX_L;          // all layers' packed weights in one launch, kept in `saved` for the backward
    char* sv = (char*)saved;
    float* Wc1_all = pack_once ? (float*)(sv + mix_pack_off(m)) : nullptr;
    float* Wc2_all = pack_once ? Wc1_all + (size_t)L * D * m.LDT : nullptr;
    float* Wh_saved = pack_once ? Wc2_all + (size_t)L * m.LDT * D : nullptr;      // fused head: [W; b]_{L-1} * w_head for the backward
    void* gws = c.base + c.off;
    const size_t gws_bytes = ws_bytes - c.off;
    float* xmid = (float*)(sv + (size_t)L * 3 * act_block(m));
    float* omid = xmid + (size_t)(L - 1) * (xbuf(m) / sizeof(float));        // exact path only
    int rc;
    const bool tile_fwd = mix_tile_on(m) && (y || head);
    // the [U | K] / [W; b] / head packs feed the launch-per-product kernels only: with the row-block kernels in both directions nobody reads them
    // (a product-route backward behind a row-block forward -- RECNOW_TILE_BWD=0 -- packs them itself: dcnmix_bwd_exact)
    const bool pack_product = pack_once && !(tile_fwd && mix_tile_bwd_on(m));
    {
        RnProfRecord* pr_pack = (pack_product && rn_prof_on()) ? rn_prof_begin(RN_TAG_LAYER_END, 0.0, 0.0, st) : nullptr;
        if (pack_product && (rc = pack_all(m, U_host, W_host, bias_host, gate_host, Wc1_all, Wc2_all, head ? head->w : nullptr, Wh_saved, st)))
            return rc;
        rn_prof_end(pr_pack, st);
    }
    // split precision: the piece planes of every layer's packed weights, one launch (RECNOW_SPLIT_PLANES_ONCE=0: each product splits its own, A/B switch)
    static const bool planes_once = []() { const char* e = getenv("RECNOW_SPLIT_PLANES_ONCE"); return !e || e[0] != '0'; }();
    const bool planes_on = planes_once && pack_product && rn_gemm_precision() == 1 && mix_planes_shape(m);
    if (planes_on) {
        RnSplitJobs jobs;
        jobs.n = 0;
        for (int l = 0; l < L; ++l) {
            const float* wc1 = Wc1_all + (size_t)l * D * m.LDT;
            const float* wc2 = Wc2_all + (size_t)l * m.LDT * D;
            jobs.job[jobs.n++] = RnSplitJob{wc1, m.LDT, 0, D, 128, mix_plane(m, saved, l, 0)};
            jobs.job[jobs.n++] = RnSplitJob{wc2, D, 0, m.KP, D, mix_plane(m, saved, l, 1)};
            jobs.job[jobs.n++] = RnSplitJob{(head && l == L - 1) ? Wh_saved : W_host[l], D, 1, D, 128, mix_plane(m, saved, l, 2)};
            jobs.job[jobs.n++] = RnSplitJob{wc1, m.LDT, 1, m.KP, D, mix_plane(m, saved, l, 3)};
        }
        RnProfRecord* pr_pl = rn_prof_on() ? rn_prof_begin(RN_TAG_LAYER_END, 0.0, 0.0, st) : nullptr;
        rc = rn_split_planes_multi(jobs, st);
        rn_prof_end(pr_pl, st);
        if (rc) return rc;
    }
    const bool xless = mix_xless(m);
    mix_stamp_put(saved, (pack_product ? MIX_HAS_PRODUCT_PACKS : 0) | (tile_fwd ? MIX_HAS_TILE_PACKS : 0) | (planes_on ? MIX_HAS_SPLIT_PLANES | (head ? MIX_PLANES_HEAD : 0) : 0) |
                             (xless ? MIX_XLESS : 0));
    const float* xl = x;
    const bool tile_split = !tile_fwd && mix_tile_split_on(m) && (y || head);
    if (tile_fwd || tile_split) {       // every layer (+ the scoring head) in one launch of row-block workgroups
        RnTileFwd t;
        memset(&t, 0, sizeof(t));
        t.x = x; t.B = B; t.D = D; t.L = L; t.act_inner = act_inner; t.act_outer = act_outer;
        t.packs = (float*)(sv + mix_tile_pack_off(m));
        for (int l = 0; l < L; ++l) {
            t.U[l] = U_host[l]; t.Kg[l] = gate_host[l]; t.V[l] = V_host[l]; t.W[l] = W_host[l]; t.bias[l] = bias_host[l];
            t.T1[l] = (float*)(sv + (size_t)(3 * l) * act_block(m));
            t.T2[l] = (float*)(sv + (size_t)(3 * l + 1) * act_block(m));
            t.T2g[l] = (float*)(sv + (size_t)(3 * l + 2) * act_block(m));
            t.O[l] = (need_dx || (xless && l < L - 1)) ? omid + (size_t)l * (xbuf(m) / sizeof(float)) : nullptr;
            t.xn[l] = l < L - 1 ? (xless ? nullptr : xmid + (size_t)l * (xbuf(m) / sizeof(float))) : (head ? nullptr : y);
        }
        if (head) { t.head_w = head->w; t.head_b = head->b; t.scores = head->scores; }
        t.packed = (tl_step_tile_packed != nullptr && tl_step_tile_packed == (const void*)saved) ? 1 : 0;
        tl_step_tile_packed = nullptr;
        if (tile_split) {
            t.splanes = sv + mix_tile_split_off(m);
            return rn_mix_tile_fwd_split(t, st);
        }
        return rn_mix_tile_fwd(t, st);
    }
    for (int l = 0; l < L; ++l) {
        float* T1 = (float*)(sv + (size_t)(3 * l) * act_block(m));
        float* T2 = (float*)(sv + (size_t)(3 * l + 1) * act_block(m));
        float* T2g = (float*)(sv + (size_t)(3 * l + 2) * act_block(m));
        float* out = (l == L - 1) ? y : xmid + (size_t)l * (xbuf(m) / sizeof(float));
        if (m.exact) {
            if (pack_once) {
                Wc1 = Wc1_all + (size_t)l * D * m.LDT;
                Wc2 = Wc2_all + (size_t)l * m.LDT * D;
            } else if ((rc = pack_weights(m, U_host[l], V_host[l], W_host[l], bias_host[l], gate_host[l], Wc1, Wc2, st))) {
                return rc;
            }
            {
            RnDeferredReduce red1;
            red1.valid = 0;
            const bool absorb = rn_mix_mid_absorbs_slabs(B, S, N, m.LDT);
            // GEMM1 with the sub-space forward in its epilogue (recnow_gemm_desc.mid_V): two experts of 64, a batch large enough that the
            // product is not split over K (512 batch tiles), exact-fp32 products; T1, T2 and T2g leave the one kernel, `k_mix_mid_fwd` is gone
            static const int midf_on = []() { const char* e = getenv("RECNOW_MIDF"); return e ? atoi(e) : 1; }();      // A/B switch: 0 off, 2 any batch (tests)
            if (midf_on && N == 2 && S == 64 && m.LDT == 144 && B % 128 == 0 && (B / 128 >= 512 || midf_on == 2) && rn_gemm_precision() == 0 &&
                rn_mix_mid_supported(S, N, m.LDT)) {
                recnow_gemm_desc d = rn_gemm_desc_zero();
                d.A = Wc1; d.lda = m.LDT; d.a_trans = 1;             // [U | K]^T: stored [K = D][M = 128 (+ gate columns, unused here)]
                d.B = xl; d.ldb = D; d.b_trans = 1;                  // x_l^T: stored [N = B][K = D]
                if (xless && l > 0) { d.B = x; d.B2 = omid + (size_t)(l - 1) * (xbuf(m) / sizeof(float)); d.b_mode = RECNOW_OPMODE_MUL; }      // x_l = x0 * O_{l-1}
                d.C = T1; d.ldc = m.LDT;                             // (not written: the epilogue stores T1 / T2 / T2g itself)
                d.M = m.NS; d.N = (int)B; d.K = D;
                d.prof_flops = 2.0 * (double)B * D * m.KC;
                d.act = act_inner;
                d.sp_bx = gate_host[l]; d.sp_bx_ks = N; d.sp_bx_rs = 1; d.sp_cx = T1 + m.NS; d.sp_cx_ms = m.LDT; d.sp_cx_rs = 1; d.sp_r = N;
                d.mid_V = V_host[l]; d.mid_T1 = T1; d.mid_T2 = T2; d.mid_T2g = T2g; d.mid_ld = m.LDT; d.mid_act_outer = act_outer;
                rc = rn_gemm(&d, gws, gws_bytes, st);
                if (rc && rc != RECNOW_EUNSUPPORTED) return rc;
            } else {
                rc = RECNOW_EUNSUPPORTED;
            }
            if (rc == RECNOW_EUNSUPPORTED) {
            {   // GEMM1: T1[:, :NS] = act_inner(x_l U);  gate logits T1[:, NS:NS+N] = x_l K as the VALU side product
                recnow_gemm_desc d = rn_gemm_desc_zero();
                d.A = xl; d.lda = D; d.a_trans = 0;
                if (xless && l > 0) { d.A = x; d.A2 = omid + (size_t)(l - 1) * (xbuf(m) / sizeof(float)); d.a_mode = RECNOW_OPMODE_MUL; }      // x_l = x0 * O_{l-1}
                d.B = Wc1; d.ldb = m.LDT; d.b_trans = 0;
                d.C = T1; d.ldc = m.LDT;
                d.M = (int)B; d.N = m.NS; d.K = D;
                d.prof_flops = 2.0 * (double)B * D * m.KC;
                d.act = act_inner;
                d.sp_bx = gate_host[l]; d.sp_bx_ks = N; d.sp_bx_rs = 1; d.sp_cx = T1 + m.NS; d.sp_cx_ms = m.LDT; d.sp_cx_rs = 1; d.sp_r = N;
                // a shard small enough for this product to be split over K: the sub-space kernel sums the slabs (and applies act_inner)
                // on its way in -- no reduction launch
                if (planes_on) rn_gemm_planes_hint(mix_plane(m, saved, l, 0));
                if (absorb) rc = rn_gemm_deferred(&d, gws, gws_bytes, st, &red1);
                else rc = rn_gemm(&d, gws, gws_bytes, st);
                if (rc) return rc;
            }
            RnSlabs sl;
            sl.n = 0;
            if (absorb && red1.valid) {
                rn_deferred_slabs(&red1, &sl.p, &sl.n, &sl.ld, &sl.stride);
                if (sl.n != 2 && sl.n != 4) {      // a split the sub-space kernel has no variant for: reduce as usual
                    if ((rc = rn_layer_end_reduce(&red1, nullptr, nullptr, 0, 0, nullptr, st))) return rc;
                    sl.n = 0;
                }
            }
            if (sl.n) {
                if ((rc = mix_mid_fwd(m, T1, V_host[l], T2, T2g, act_outer, gws, gws_bytes, st, &sl, act_inner))) return rc;
            } else if ((rc = mix_mid_fwd(m, T1, V_host[l], T2, T2g, act_outer, gws, gws_bytes, st))) return rc;
            }      // (not the fused GEMM1)
            }
            {   // GEMM3: out = x * ([G*H2 | G | 0] [W; b; 0]): K zero-padded NS+N -> KP (a 16-deep k-tile more is cheaper
                // than a rank-N epilogue update: 215 vs 233 us measured)
                recnow_gemm_desc d = rn_gemm_desc_zero();
                d.A = T2g; d.lda = m.LDT; d.a_trans = 0;
                d.B = Wc2; d.ldb = D; d.b_trans = 0;
                d.C = out; d.ldc = D;
                d.M = (int)B; d.N = D; d.K = m.KP; d.k_valid = m.KC;
                d.prof_flops = 2.0 * (double)B * D * m.KC;
                d.emul = x; d.lde = D; d.e_mode = RECNOW_OPMODE_MUL;
                if (need_dx) { d.C2 = omid + (size_t)l * (xbuf(m) / sizeof(float)); d.ldc2 = D; d.c2_mode = 1; }     // O_l only feeds dx
                if (xless && l < L - 1) {       // O_l alone: x_{l+1} = x0 * O_l is formed where it is consumed (mix_xless)
                    d.emul = nullptr; d.e_mode = RECNOW_OPMODE_NONE; d.C2 = nullptr; d.c2_mode = 0;
                    d.C = omid + (size_t)l * (xbuf(m) / sizeof(float));
                }
                if (head && l == L - 1) {       // the layer output only feeds the scoring head: row-dot partials instead of y
                    d.c2_mode = 3; d.ldc2 = D;
                    d.C = omid + (size_t)l * (xbuf(m) / sizeof(float));      // not written (c2_mode 3); a valid aligned address for the checks
                    d.hv = head->w; d.hp = hp; d.hp_ld = 2 * (D / 128);
                }
                if (planes_on) rn_gemm_planes_hint(mix_plane(m, saved, l, 1));
                if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
                if (head && l == L - 1) {
                    hipLaunchKernelGGL(k_head_scores, ew_grid(B), 256, 0, st, hp, 2 * (D / 128), head->b, B, head->scores);
                    RN_LAUNCH_CHECK();
                }
            }
            xl = out;
            continue;
        }
        if ((rc = pack_weights(m, U_host[l], V_host[l], W_host[l], bias_host[l], gate_host[l], Wc1, Wc2, st))) return rc;
        {   // GEMM1: T1 = [act_inner(x_l U) | x_l K]
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = xl; d.lda = D; d.a_trans = 0;
            d.B = Wc1; d.ldb = m.LDT; d.b_trans = 0;
            d.C = T1; d.ldc = m.LDT;
            d.M = (int)B; d.N = m.LDT; d.K = D;
            d.act = act_inner; d.act_cols = m.NS;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        }
        if ((rc = mix_mid_fwd(m, T1, V_host[l], T2, T2g, act_outer, gws, gws_bytes, st))) return rc;
        {   // GEMM3: out = x * (T2g [W; b])
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = T2g; d.lda = m.LDT; d.a_trans = 0;
            d.B = Wc2; d.ldb = D; d.b_trans = 0;
            d.C = out; d.ldc = D;
            d.M = (int)B; d.N = D; d.K = m.KP; d.k_valid = m.KC;
            d.emul = x; d.lde = D; d.e_mode = RECNOW_OPMODE_MUL;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        }
        xl = out;
    }
    return RECNOW_OK;
}

extern "C" int recnow_dcn_mix_fwd(const float* x, const float* const* U_host, const float* const* V_host,
                                  const float* const* W_host, const float* const* bias_host, const float* const* gate_host,
                                  int64_t B, int D, int S, int N, int L, int act_inner, int act_outer, float* y, void* saved,
                                  size_t saved_bytes, void* ws, size_t ws_bytes, void* stream, int need_dx) {
    return dcnmix_fwd_impl(x, U_host, V_host, W_host, bias_host, gate_host, B, D, S, N, L, act_inner, act_outer, y, saved, saved_bytes,
                           ws, ws_bytes, stream, need_dx, nullptr);
}

extern "C" int recnow_dcn_mix_score_supported(int64_t B, int D, int S, int N, int L) {
    if (B <= 0 || D < 1 || S < 1 || N < 1 || L < 1 || N > 64) return 0;
    return mix_head_ok(mix_dims(B, D, S, N, L)) ? 1 : 0;
}

int rn_pair_bpr_onepass(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order, const int32_t* seg_id,
                        const int32_t* seg_first, int64_t B, int flags, float factor, int reduce_mean, float* loss, float* dscores_unnorm,
                        int64_t* n_pair, void* ws, size_t ws_bytes, void* stream, const double** part_out, int* nparts_out);      // pairwise.hip
// scan_sort.hip; pack != NULL: the launch also writes the weight packs of the row-block kernels on further workgroups (k_front_small / k_front_mid)
// zero1 / zeroed: a front kernel also clears one 64-bit word (the step's pair counter) and reports it
int rn_group_small_raw(const void* group, int dtype, int64_t B, int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id,
                       int32_t* n_seg, hipStream_t st, const RnTileFwd* pack, unsigned long long* zero1, int* zeroed, int* packed);
int rn_group_mid_raw(const void* group, int dtype, int64_t B, uint8_t* solo, int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id,
                     int32_t* n_seg, void* ws, size_t ws_bytes, hipStream_t st, const RnTileFwd* pack, unsigned long long* zero1, int* zeroed, int* packed);

extern "C" int recnow_dcn_mix_tile_route(int64_t B, int D, int S, int N, int L) {
    if (B <= 0 || D < 1 || S < 1 || N < 1 || L < 1 || N > 64) return 0;
    const MixDims m = mix_dims(B, D, S, N, L);
    return mix_tile_on(m) ? 1 : mix_tile_split_on(m) ? 2 : 0;
}

extern "C" int recnow_dcn_mix_score_fwd(const float* x, const float* const* U_host, const float* const* V_host,
                                        const float* const* W_host, const float* const* bias_host, const float* const* gate_host,
                                        const float* head_w, const float* head_b, int64_t B, int D, int S, int N, int L, int act_inner,
                                        int act_outer, float* scores, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes,
                                        void* stream, int need_dx) {
    MixHead head;
    head.w = head_w; head.b = head_b; head.scores = scores;
    return dcnmix_fwd_impl(x, U_host, V_host, W_host, bias_host, gate_host, B, D, S, N, L, act_inner, act_outer, nullptr, saved,
                           saved_bytes, ws, ws_bytes, stream, need_dx, &head);
}


// ---- exact-128 backward, optionally on two streams -----------------------------------------------------------------
// The data-gradient chain (dT2g -> gate_bwd -> dA -> dxl) is sequential; the weight-gradient products and the dx
// recompute only consume its intermediates.  Every big GEMM here alternates an MFMA-bound main loop with an HBM-bound
// epilogue and all workgroups of one launch run in phase, so a single stream leaves ~40 % of the MFMA pipe idle.  With a
// second stream (st2 != st) the side products run concurrently with the chain and fill those phases.  Ordering is by
// events only; results are identical to the single-stream order (dx is accumulated by the side stream, then joined).
// Events of the two-stream ordering: taken from a process-wide pool that only grows (hipEventCreate / Destroy per call cost up to a
// millisecond of host time on a busy runtime).  An event may be recorded again while an earlier wait on it is still queued: a wait
// refers to the record that preceded it.
struct MixEvents {
    int n = 0;
    hipEvent_t make() {
        // per device (an event belongs to the device that was current when it was created) AND per host thread: two threads driving one
        // device would otherwise record and wait on the same pool slots (a wait enqueued by one thread could pick up the other's record)
        static thread_local hipEvent_t pool[16][64];
        static thread_local int have[16];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || n >= 64) return nullptr;
        if (n >= have[dev]) {
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
            pool[dev][have[dev]++] = ev;
        }
        return pool[dev][n++];
    }
};
#define MIX_SIGNAL(ev, from)                                  \
    do {                                                      \
        if (two) { ev = evs.make(); if (!ev) return RECNOW_EINVAL; RN_HIP(hipEventRecord(ev, from)); } \
    } while (0)
#define MIX_WAIT(ev, on)                                      \
    do {                                                      \
        if (two && ev) RN_HIP(hipStreamWaitEvent(on, ev, 0)); \
    } while (0)

// ---- exact-128 backward through the row-block kernel (dcnmix_tile.hip) at shard sizes --------------------------------------------
// ONE launch walks the data-gradient chain of layers l_hi .. l_lo (dT2g, sub-space backward, g_l, dx); it leaves dT1_l and g_l, and the
// K = B weight-gradient products of every layer follow (top layer first, so that a layer's event -- and its gradient all-reduce -- is
// not held back by the layers below).  The chain launch occupies every CU by itself (one workgroup per CU, all registers), so nothing
// would run beside it: the products start when it ends.
static int dcnmix_bwd_tile(const MixDims& m, const float* x, const float* const* U_host, const float* const* V_host,
                           const float* const* W_host, const float* const* bias_host, const float* const* gate_host,
                           const float* dy, const char* sv, int act_inner, int act_outer, float* dx, float* const* dU_host,
                           float* const* dV_host, float* const* dW_host, float* const* dbias_host, float* const* dgate_host,
                           void* ws, size_t ws_bytes, hipStream_t st, hipStream_t st2, const MixHeadGrad* hd, void* const* layer_events,
                           int l_hi, int l_lo, const float* T2g_ds_ready, const float* ds_part, int ds_nparts) {
    const int64_t B = m.B;
    const int D = m.D, S = m.S, N = m.N, L = m.L;
    const bool top = l_hi == L - 1;
    const bool two = st2 != nullptr && st2 != st;
    if (!two) st2 = st;
    RnCarver c(ws, ws_bytes);
    c.take<float>((size_t)L * D * m.LDT);
    float* dWc1 = c.take<float>((size_t)D * m.LDT);
    c.take<float>(act_block(m) / sizeof(float));
    float* dC = c.take<float>(act_block(m) / sizeof(float));
    c.take<float>(act_block(m) / sizeof(float));
    float* gbuf0 = c.take<float>(xbuf(m) / sizeof(float));
    float* gbuf1 = c.take<float>(xbuf(m) / sizeof(float));
    const size_t gemm_ws = mix_gemm_ws(m);
    void* gws1 = c.take<char>(gemm_ws);
    void* gws2 = c.take<char>(gemm_ws);
    void* gws3 = c.take<char>(gemm_ws);
    c.take<char>(rn_mix_mid_bwd_ws_bytes(B, S, N));
    const size_t cs_ws_bytes = rn_colsum_ws_bytes(B, 1);
    void* cs_ws = c.take<char>(cs_ws_bytes);
    float* dT1_all = c.take<float>((size_t)L * act_block(m) / sizeof(float));
    const int grid = rn_mix_tile_bwd_grid(B);
    float* dvpart = c.take<float>((size_t)L * grid * N * S * S);
    if (!c.ok()) return RECNOW_EWORKSPACE;
    const float* xmid = (const float*)(sv + (size_t)L * 3 * act_block(m));
    const float* omid = xmid + (size_t)(L - 1) * (xbuf(m) / sizeof(float));
    const float* Wc1_all = (const float*)(sv + mix_pack_off(m));
    (void)Wc1_all;
    MixEvents evs;
    int rc;
    const float* T2g_ds = dC;
    if (hd && top) {
        const float* T2g_top = (const float*)(sv + (size_t)(3 * (L - 1) + 2) * act_block(m));
        if (T2g_ds_ready) {
            T2g_ds = T2g_ds_ready;
        } else {
            hipLaunchKernelGGL(k_row_scale, ew_grid(B * (m.LDT / 4)), 256, 0, st, T2g_top, hd->dscores, B, m.LDT, dC);
            RN_LAUNCH_CHECK();
            if (hd->db && (rc = rn_colsum(hd->dscores, nullptr, 0, 0, B, 1, 1, hd->db, 0, cs_ws, cs_ws_bytes, st))) return rc;
        }
    }
    auto gbuf_of = [&](int l) { return (l & 1) ? gbuf0 : gbuf1; };          // g_l = d loss / d x_l, l >= 1 (the buffers of the product route)
    {
        RnTileBwd t;
        memset(&t, 0, sizeof(t));
        t.x = x; t.packs = (const float*)(sv + mix_tile_pack_off(m)); t.B = B; t.D = D; t.L = L; t.l_hi = l_hi; t.l_lo = l_lo;
        t.act_inner = act_inner; t.act_outer = act_outer; t.dx = dx; t.dvpart = dvpart;
        for (int l = 0; l < L; ++l) {
            t.Kg[l] = gate_host[l]; t.bias[l] = bias_host[l];
            t.T1[l] = (const float*)(sv + (size_t)(3 * l) * act_block(m));
            t.T2[l] = (const float*)(sv + (size_t)(3 * l + 1) * act_block(m));
            t.O[l] = omid + (size_t)l * (xbuf(m) / sizeof(float));
            t.dT1[l] = dT1_all + (size_t)l * (act_block(m) / sizeof(float));
            t.gout[l] = l > 0 ? gbuf_of(l) : nullptr;
        }
        if (top) {
            if (hd) { t.ds = hd->dscores; t.head_w = hd->w; }
            else t.gin = dy;
        } else {
            t.gin = gbuf_of(l_hi + 1);
        }
        if ((rc = rn_mix_tile_bwd(t, st))) return rc;
    }
    hipEvent_t e_chain = nullptr;
    MIX_SIGNAL(e_chain, st);
    MIX_WAIT(e_chain, st2);
    int pg = rn_cdiv((int64_t)D * m.NS, 256);
    if (pg > 2048) pg = 2048;
    struct SlotGuard {      // the dW / dU products of a layer run as a concurrent pair: each aims at half the workgroup slots (gemm_dispatch.hpp)
        explicit SlotGuard(bool on) { if (on) rn_gemm_split_slots(256); }
        ~SlotGuard() { rn_gemm_split_slots(0); }
    } slot_guard(two);
    // Round 5.  The six K = B products behind the chain are independent of each other; what serialised them was the sharing of split-K slab
    // buffers (dW of every layer in one buffer: dW_l waited for the reduction of layer l + 1 on the second stream) and the reductions queued
    // between them.  Kernel trace of the 8192-row step: pairs at 378-427 us, then dU_1 ALONE, dW_1 + dU_0 at 477, dW_0 alone at 551, the last
    // reduction at 587-595 -- 217 us for 12.9 GFLOP.  Now every product owns its slabs, ALL products are issued first (dW_l on the second stream,
    // dU_l on the first: three concurrent pairs back to back), and the reductions follow: the top layer's (+ the head's post-processing) on the
    // first stream, the others on the second, each behind the events of its two products.
    void* slab_w[RN_TILE_MAX_L];
    void* slab_u[RN_TILE_MAX_L];
    {
        void* pool[2 * RN_TILE_MAX_L];
        int np = 0;
        pool[np++] = gws1; pool[np++] = gws2; pool[np++] = gws3;
        for (int i = 3; i < 2 * L; ++i) pool[np++] = c.take<char>(gemm_ws);
        if (!c.ok()) return RECNOW_EWORKSPACE;
        for (int l = 0; l < L; ++l) { slab_w[l] = pool[2 * l]; slab_u[l] = pool[2 * l + 1]; }
    }
    RnDeferredReduce red_dw[RN_TILE_MAX_L], red_du[RN_TILE_MAX_L];
    hipEvent_t e_dw[RN_TILE_MAX_L], e_du[RN_TILE_MAX_L];
    for (int l = 0; l < RN_TILE_MAX_L; ++l) { red_dw[l].valid = red_du[l].valid = 0; e_dw[l] = e_du[l] = nullptr; }
    // The reduction of layer l (slab sums of dW_l / dU_l, the dV partials, the head's post-processing, the layer's event) is enqueued ONE PAIR LATE --
    // behind the products of layer l - 1 on its stream -- and alternates between the streams: layer l_hi's on the first (behind dU of the layer
    // below, waiting for its dW), the next on the second, ...  So a pair never waits for a reduction, every layer's event (= its gradient
    // all-reduce under a process group) still fires while the products of the layers below run, and only the last layer's reduction is a tail.
    auto reduce_layer = [&](int l) -> int {
        const bool on_first = ((l_hi - l) & 1) == 0;
        hipStream_t sr = on_first ? st : st2;
        if (on_first) MIX_WAIT(e_dw[l], st);
        else MIX_WAIT(e_du[l], st2);
        RnProfRecord* pr_end = rn_prof_on() ? rn_prof_begin(RN_TAG_LAYER_END, 0.0, 0.0, sr) : nullptr;
        int rr;
        if ((rr = rn_layer_end_reduce(&red_dw[l], &red_du[l], dvpart + (size_t)l * grid * N * S * S, grid, N * S * S, dV_host[l], sr))) return rr;
        if (hd && l == L - 1) {
            hipLaunchKernelGGL(k_head_post, rn_cdiv(D, 32), 256, 0, sr, dW_host[l], dbias_host[l], W_host[l], bias_host[l], hd->w, m.NS, N, D, hd->dw,
                               ds_part, ds_nparts, ds_part ? hd->db : nullptr);
            RN_LAUNCH_CHECK();
        }
        rn_prof_end(pr_end, sr);
        if (layer_events && layer_events[l]) RN_HIP(hipEventRecord((hipEvent_t)layer_events[l], sr));
        return RECNOW_OK;
    };
    for (int l = l_hi; l >= l_lo; --l) {
        const float* T2g = (const float*)(sv + (size_t)(3 * l + 2) * act_block(m));
        const float* xl = (l == 0) ? x : xmid + (size_t)(l - 1) * (xbuf(m) / sizeof(float));
        const float* g = (l == L - 1) ? dy : gbuf_of(l + 1);
        const float* dT1 = dT1_all + (size_t)l * (act_block(m) / sizeof(float));
        {   // dW^T = (x*g)^T T2g[:, :NS] stored transposed straight into dW (NS x D);  dbias[n][d] as the side product
            recnow_gemm_desc d = rn_gemm_desc_zero();
            const bool top_head = hd && l == L - 1;
            d.A = g; d.A2 = x; d.a_mode = RECNOW_OPMODE_MUL; d.lda = D; d.a_trans = 1;
            d.B = T2g; d.ldb = m.LDT; d.b_trans = 0;
            if (top_head) {      // M^T = (x^T (dscore * T2g))^T: the rank-one head gradient reduced to a row scale of the small operand
                d.A = x; d.A2 = nullptr; d.a_mode = RECNOW_OPMODE_NONE;
                d.B = T2g_ds;
            }
            d.C = dW_host[l]; d.ldc = D; d.c_trans = 1;
            d.M = D; d.N = m.NS; d.K = (int)B;
            d.prof_flops = 2.0 * (double)B * D * m.KC;
            d.sp_bx = d.B + m.NS; d.sp_bx_ks = m.LDT; d.sp_bx_rs = 1; d.sp_cx = dbias_host[l]; d.sp_cx_ms = 1; d.sp_cx_rs = D; d.sp_r = N;
            if ((rc = rn_gemm_deferred(&d, slab_w[l], gemm_ws, st2, &red_dw[l]))) return rc;
            MIX_SIGNAL(e_dw[l], st2);
        }
        {   // dWc1 = x_l^T dT1[:, :NS] -> dU;  dgate[d][n] = x_l^T dlogits as the side product
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = xl; d.lda = D; d.a_trans = 1;
            if (mix_xless_saved(m, sv) && l > 0) { d.A = x; d.A2 = omid + (size_t)(l - 1) * (xbuf(m) / sizeof(float)); d.a_mode = RECNOW_OPMODE_MUL; }      // x_l = x0 * O_{l-1} (not stored)
            d.B = dT1; d.ldb = m.LDT; d.b_trans = 0;
            d.C = dWc1; d.ldc = m.NS;
            d.M = D; d.N = m.NS; d.K = (int)B;
            d.prof_flops = 2.0 * (double)B * D * m.KC;
            d.sp_bx = dT1 + m.NS; d.sp_bx_ks = m.LDT; d.sp_bx_rs = 1; d.sp_cx = dgate_host[l]; d.sp_cx_ms = N; d.sp_cx_rs = 1; d.sp_r = N;
            recnow_gemm_desc dq = d;
            dq.C = dU_host[l]; dq.c_perm_s = S;
            rc = rn_gemm_deferred(&dq, slab_u[l], gemm_ws, st, &red_du[l]);
            if (rc == RECNOW_EUNSUPPORTED) {       // no split: the product stores (D, N S) and is unpacked (dWc1 is reused in stream order)
                if ((rc = rn_gemm(&d, slab_u[l], gemm_ws, st))) return rc;
                hipLaunchKernelGGL(k_unpack_u, pg, 256, 0, st, dWc1, D, S, N, dU_host[l]);
                RN_LAUNCH_CHECK();
            } else if (rc) {
                return rc;
            }
            MIX_SIGNAL(e_du[l], st);
        }
        if (l < l_hi && (rc = reduce_layer(l + 1))) return rc;
    }
    if ((rc = reduce_layer(l_lo))) return rc;
    hipEvent_t e_done = nullptr;
    MIX_SIGNAL(e_done, st2);
    MIX_WAIT(e_done, st);
    (void)U_host; (void)V_host;
    return RECNOW_OK;
}

static int dcnmix_bwd_exact(const MixDims& m, const float* x, const float* const* U_host, const float* const* V_host,
                            const float* const* W_host, const float* const* bias_host, const float* const* gate_host,
                            const float* dy, const char* sv, int act_inner, int act_outer, float* dx, float* const* dU_host,
                            float* const* dV_host, float* const* dW_host, float* const* dbias_host, float* const* dgate_host,
                            void* ws, size_t ws_bytes, hipStream_t st, hipStream_t st2, const MixHeadGrad* hd = nullptr,
                            void* const* layer_events = nullptr, int l_hi = -1, int l_lo = 0, const float* T2g_ds_ready = nullptr,
                            const float* ds_part = nullptr, int ds_nparts = 0) {
    // l_hi .. l_lo (descending, default: all layers): the backward of a sub-range of the layers, so that a caller can cut the pass
    // into pieces (one HIP graph per piece, a gradient all-reduce launched in between).  The pieces share `ws`: the gradient that
    // flows from layer l + 1 into layer l sits in the ping-pong buffer the full pass would have used.
    const int64_t B = m.B;
    const int D = m.D, S = m.S, N = m.N, L = m.L;
    if (l_hi < 0) l_hi = L - 1;
    if (l_lo < 0 || l_lo > l_hi || l_hi > L - 1) return RECNOW_EINVAL;
    // The row-block backward chain (RECNOW_TILE_BWD=0 keeps the product-route backward behind the row-block forward: A/B switch, read per call).
    // Measured on one box, ms per step at 8192 / 16 384 rows per GPU: product route 0.760-0.765 / 1.138-1.143, row-block forward alone 0.723-0.726 /
    // 1.135-1.154, forward and backward 0.699-0.708 / 1.128-1.129 (0.67-0.68 / 1.10 with the paired weight-gradient products at 256 slots).
    const bool top = l_hi == L - 1;
    const int have = mix_stamp_get(sv);          // what the forward left in `saved` (-1: a forward this copy of the library did not see)
    bool want_tile = mix_tile_on(m) && mix_tile_bwd_on(m) && (l_hi < L - 1 || hd || dy);
    if (!top && have >= 0) want_tile = (have & MIX_BWD_TILE) != 0 && mix_tile_shape(m);      // a lower piece follows the top piece of its pass
    if (want_tile) {
        if (top && !(have >= 0 && (have & MIX_HAS_TILE_PACKS))) {      // product-route forward (or unknown): the fragment-ordered packs are made here
            int rc0;
            if ((rc0 = rn_mix_tile_pack(U_host, gate_host, V_host, W_host, bias_host, D, L, (float*)(sv + mix_tile_pack_off(m)), st))) return rc0;
        }
        if (top) mix_stamp_put(sv, (have < 0 ? 0 : have) | MIX_HAS_TILE_PACKS | MIX_BWD_TILE);
        return dcnmix_bwd_tile(m, x, U_host, V_host, W_host, bias_host, gate_host, dy, sv, act_inner, act_outer, dx, dU_host, dV_host, dW_host,
                               dbias_host, dgate_host, ws, ws_bytes, st, st2, hd, layer_events, l_hi, l_lo, T2g_ds_ready, ds_part, ds_nparts);
    }
    const bool defer_dv = rn_mix_mid_supported(S, N, m.LDT);      // the fused sub-space kernel leaves its dV partials for the layer-end reduction
    const bool two = st2 != nullptr && st2 != st;
    if (!two) st2 = st;
    // RECNOW_TWO_STREAMS=2 ("paired"): BOTH weight-gradient products of a layer (MFMA-bound) are held back until the sub-space
    // backward is done, so that they run beside the layer's dx product (HBM-bound) instead of beside the MFMA-bound dT2g product
    static const bool paired = []() { const char* e = getenv("RECNOW_TWO_STREAMS"); return e && e[0] == '2'; }();
    RnCarver c(ws, ws_bytes);
    float* Wc1_all = c.take<float>((size_t)L * D * m.LDT);        // per-layer packs (L > MIX_PACK_MAX_L only: otherwise the forward's, in `saved`)
    float* dWc1 = c.take<float>((size_t)D * m.LDT);
    float* dT2g = c.take<float>(act_block(m) / sizeof(float));
    float* dC = c.take<float>(act_block(m) / sizeof(float));
    float* dT1 = c.take<float>(act_block(m) / sizeof(float));
    float* gbuf0 = c.take<float>(xbuf(m) / sizeof(float));
    float* gbuf1 = c.take<float>(xbuf(m) / sizeof(float));
    const size_t gemm_ws = mix_gemm_ws(m);
    void* gws = c.take<char>(gemm_ws);                              // split-K slabs of the chain stream
    void* gws2 = c.take<char>(gemm_ws);                             // ... of the dU product
    void* gws3 = c.take<char>(gemm_ws);                             // ... of the dW product (kept until the layer-end reduction)
    const size_t mid_ws_bytes = rn_mix_mid_bwd_ws_bytes(B, S, N);
    void* mid_ws = c.take<char>(mid_ws_bytes);
    const size_t cs_ws_bytes = rn_colsum_ws_bytes(B, 1);
    void* cs_ws = c.take<char>(cs_ws_bytes);
    if (!c.ok()) return RECNOW_EWORKSPACE;
    const float* xmid = (const float*)(sv + (size_t)L * 3 * act_block(m));
    const float* omid = xmid + (size_t)(L - 1) * (xbuf(m) / sizeof(float));
    MixEvents evs;
    int rc;
    const float* Wh = nullptr;         // fused head: [W * w_head; bias * w_head] of the top layer, packed by the forward
    if (L <= MIX_PACK_MAX_L) {         // [U | K | 0] of every layer: packed ONCE per step by the forward, kept behind the activations in `saved`
        Wc1_all = (float*)(sv + mix_pack_off(m));
        Wh = Wc1_all + (size_t)2 * L * D * m.LDT;
        // ... unless the forward ran the row-block kernels in both directions and left them out (RECNOW_TILE_BWD=0 A/B pairing, or the precision /
        // RECNOW_TILE switched in between): packed here.  Unknown forward: packed whenever the shape may have taken the row-block route.
        if (top && (have >= 0 ? !(have & MIX_HAS_PRODUCT_PACKS) : mix_tile_on(m, true))) {
            float* Wc2_all = Wc1_all + (size_t)L * D * m.LDT;
            if ((rc = pack_all(m, U_host, W_host, bias_host, gate_host, Wc1_all, Wc2_all, hd ? hd->w : nullptr, const_cast<float*>(Wh), st))) return rc;
        }
        if (top && have >= 0) mix_stamp_put(sv, (have | MIX_HAS_PRODUCT_PACKS) & ~MIX_BWD_TILE);
    } else if (top) {
        int pgw = rn_cdiv((int64_t)D * m.LDT, 256);
        if (pgw > 2048) pgw = 2048;
        for (int l = 0; l < L; ++l) {
            hipLaunchKernelGGL(k_pack_w1, pgw, 256, 0, st, U_host[l], gate_host[l], D, S, N, m.LDT, Wc1_all + (size_t)l * D * m.LDT);
            RN_LAUNCH_CHECK();
        }
    }
    // split precision: the forward of this pass left the piece planes of the packed weights in `saved` (and nothing repacked them since)
    const bool planes_bwd = rn_gemm_precision() == 1 && have >= 0 && (have & MIX_HAS_SPLIT_PLANES) && (have & MIX_HAS_PRODUCT_PACKS) && mix_planes_shape(m);
    const float* T2g_ds = dC;          // fused head: dscore * T2g of the top layer (dC is scratch of the unfused sub-space route only)
    if (hd && top) {
        const float* T2g_top = (const float*)(sv + (size_t)(3 * (L - 1) + 2) * act_block(m));
        if (T2g_ds_ready) {            // the caller's loss stage has already formed dscore * T2g_top and d bias = sum of dscores (recnow_dcn_mix_step)
            T2g_ds = T2g_ds_ready;
        } else {
            hipLaunchKernelGGL(k_row_scale, ew_grid(B * (m.LDT / 4)), 256, 0, st, T2g_top, hd->dscores, B, m.LDT, dC);
            RN_LAUNCH_CHECK();
            if (hd->db && (rc = rn_colsum(hd->dscores, nullptr, 0, 0, B, 1, 1, hd->db, 0, cs_ws, cs_ws_bytes, st))) return rc;
        }
        if (dx && L == 1) {            // a single cross layer: its dx product accumulates on top of the head's term
            hipLaunchKernelGGL(k_head_dx_top, ew_grid(B * (D / 4)), 256, 0, st, omid, hd->dscores, hd->w, B, D, dx);
            RN_LAUNCH_CHECK();
        }
    }
    // the input gradient in one go (c2_mode 5 / 6 of the short-K kernel: two experts of 64, K = 144): fused head, 2 or 3 layers.  RECNOW_DX_ONCE=0: the
    // read-modify-write chain of rounds 1-4 (A/B switch, read per call).  A pass cut into layer pieces decides the same way in every piece.
    const char* dx_once_env = getenv("RECNOW_DX_ONCE");
    const bool dx_once = hd && dx && (L == 2 || L == 3) && m.KP == 144 && !(dx_once_env && dx_once_env[0] == '0');
    hipEvent_t e_g = nullptr;        // "g of this layer (and the packs) are ready" -> side stream may start the layer
    MIX_SIGNAL(e_g, st);
    hipEvent_t e_side_prev = nullptr;   // side stream finished the previous (higher) layer: dT1/dC/g buffers reusable
    const float* g = top ? dy : (((l_hi + 1) & 1) ? gbuf0 : gbuf1);      // a lower piece starts from the buffer layer l_hi + 1 wrote
    int pg = rn_cdiv((int64_t)D * m.NS, 256);
    if (pg > 2048) pg = 2048;
    for (int l = l_hi; l >= l_lo; --l) {
        const float* T1 = (const float*)(sv + (size_t)(3 * l) * act_block(m));
        const float* T2 = (const float*)(sv + (size_t)(3 * l + 1) * act_block(m));
        const float* T2g = (const float*)(sv + (size_t)(3 * l + 2) * act_block(m));
        const float* xl = (l == 0) ? x : xmid + (size_t)(l - 1) * (xbuf(m) / sizeof(float));
        float* gprev = (l == 0) ? nullptr : ((l & 1) ? gbuf0 : gbuf1);
        const float* Wc1 = Wc1_all + (size_t)l * D * m.LDT;
        RnDeferredReduce red_dw, red_du, red_t2g;
        red_dw.valid = red_du.valid = red_t2g.valid = 0;
        const bool absorb_bwd = defer_dv && rn_mix_mid_absorbs_slabs(B, S, N, m.LDT);
        // ---------------- side stream, part 1: needs only g_l and saved activations
        auto side_dw = [&]() -> int {
            // dW^T = (x*g)^T T2g[:, :NS] stored transposed straight into dW (NS x D);  dbias[n][d] as the side product
            recnow_gemm_desc d = rn_gemm_desc_zero();
            const bool top_head = hd && l == L - 1;
            d.A = g; d.A2 = x; d.a_mode = RECNOW_OPMODE_MUL; d.lda = D; d.a_trans = 1;
            d.B = T2g; d.ldb = m.LDT; d.b_trans = 0;
            if (top_head) {      // M^T = (x^T (dscore * T2g))^T: the rank-one head gradient reduced to a row scale of the small operand
                d.A = x; d.A2 = nullptr; d.a_mode = RECNOW_OPMODE_NONE;
                d.B = T2g_ds;
            }
            d.C = dW_host[l]; d.ldc = D; d.c_trans = 1;
            d.M = D; d.N = m.NS; d.K = (int)B;
            d.prof_flops = 2.0 * (double)B * D * m.KC;
            d.sp_bx = d.B + m.NS; d.sp_bx_ks = m.LDT; d.sp_bx_rs = 1; d.sp_cx = dbias_host[l]; d.sp_cx_ms = 1; d.sp_cx_rs = D; d.sp_r = N;
            // the slab reduction waits for the layer's end (one launch with dU's and the dV partial sum)
            if ((rc = rn_gemm_deferred(&d, gws3, gemm_ws, st2, &red_dw))) return rc;
            return RECNOW_OK;
        };
        if (!(two && paired)) {
            MIX_WAIT(e_g, st2);
            if ((rc = side_dw())) return rc;
        }
        // ---------------- chain stream
        {   // dT2g[:, :NS] = (x*g) W^T;  gate columns dT2g[:, NS+n] = (x*g) . bias_n as the side product
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = g; d.A2 = x; d.a_mode = RECNOW_OPMODE_MUL; d.lda = D; d.a_trans = 0;
            d.B = W_host[l]; d.ldb = D; d.b_trans = 1;
            d.C = dT2g; d.ldc = m.LDT;
            d.M = (int)B; d.N = m.NS; d.K = D;
            d.prof_flops = 2.0 * (double)B * D * m.KC;
            d.sp_bx = bias_host[l]; d.sp_bx_ks = 1; d.sp_bx_rs = D; d.sp_cx = dT2g + m.NS; d.sp_cx_ms = m.LDT; d.sp_cx_rs = 1; d.sp_r = N;
            if (hd && l == L - 1) {      // x (W * w_head)^T and x (bias * w_head)^T; the rows are scaled by dscore when dT2g is read
                d.A = x; d.A2 = nullptr; d.a_mode = RECNOW_OPMODE_NONE;
                d.B = Wh;
                d.sp_bx = Wh + (size_t)m.NS * D;
            } else if (l == L - 1 && dx) {      // top layer: this product streams g = dy anyway -> dx = dy * O_{L-1} written on the way
                d.as_in = omid + (size_t)l * (xbuf(m) / sizeof(float));
                d.as_out = dx;
            }
            // split over K at small shards: the sub-space kernel sums the slabs on its way in (no reduction launch)
            // (split precision: the planes of W^T / W * w_head that the forward left; the head's variant only behind a forward that had the head)
            if (planes_bwd && (l < L - 1 || ((hd != nullptr) == ((have & MIX_PLANES_HEAD) != 0)))) rn_gemm_planes_hint(mix_plane(m, sv, l, 2));
            if (absorb_bwd) rc = rn_gemm_deferred(&d, gws, gemm_ws, st, &red_t2g);
            else rc = rn_gemm(&d, gws, gemm_ws, st);
            if (rc) return rc;
        }
        RnSlabs sl_t2g;
        const RnSlabs* slp = nullptr;
        if (absorb_bwd && red_t2g.valid) {
            rn_deferred_slabs(&red_t2g, &sl_t2g.p, &sl_t2g.n, &sl_t2g.ld, &sl_t2g.stride);
            if (sl_t2g.n == 2 || sl_t2g.n == 4) slp = &sl_t2g;
            else if ((rc = rn_layer_end_reduce(&red_t2g, nullptr, nullptr, 0, 0, nullptr, st))) return rc;      // no variant: reduce as usual
        }
        MIX_WAIT(e_side_prev, st);          // dT1 (and the g buffer about to be rewritten) are free again
        // gate backward, dA_n = (dC_n V_n^T) * act_inner'(H1_n) and dV_n = H1_n^T dC_n
        if (hd && l == L - 1) {
            if ((rc = rn_mix_mid_bwd(dT2g, T2, T1, V_host[l], dT1, dV_host[l], B, S, N, m.LDT, act_inner, act_outer, mid_ws, mid_ws_bytes, st, hd->dscores,
                                     defer_dv, slp)))
                return rc;
        } else if (defer_dv) {
            if ((rc = rn_mix_mid_bwd(dT2g, T2, T1, V_host[l], dT1, dV_host[l], B, S, N, m.LDT, act_inner, act_outer, mid_ws, mid_ws_bytes, st, nullptr, true,
                                     slp)))
                return rc;
        } else if ((rc = mix_mid_bwd(m, dT2g, T2, T1, V_host[l], dC, dT1, dV_host[l], act_inner, act_outer, mid_ws, mid_ws_bytes, gws, gemm_ws, st)))
            return rc;
        hipEvent_t e_dT1 = nullptr;
        MIX_SIGNAL(e_dT1, st);
        if (l > 0 || dx) {   // gradient w.r.t. x_l (layer 0's is part of dx only): g_{l-1} = [dA | dlogits | 0] [U | K | 0]^T  (K zero-padded to KP).  The same accumulators
            // also update dx += g_{l-1} * O_{l-1} (second output); layer 0's x_l is x itself, its term goes straight into dx.
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = dT1; d.lda = m.LDT; d.a_trans = 0;
            d.B = Wc1; d.ldb = m.LDT; d.b_trans = 1;
            d.C = (l == 0) ? dx : gprev; d.ldc = D;
            d.M = (int)B; d.N = D; d.K = m.KP; d.k_valid = m.KC;
            d.prof_flops = 2.0 * (double)B * D * m.KC;
            d.accumulate = (l == 0) ? 1 : 0;
            if (dx_once) {
                // Round 5: d loss / d x = g_0 + g_1 * O_0 [+ g_2 * O_1] + dscore (x) w_head * O_{L-1} is written ONCE, by layer 0's product (c2_mode 5 / 6):
                // the products of the layers above leave g_l alone (1 pass over B x D each instead of 4), layer 0's reads g_1, O_0, g_2, O_1, O_{L-1}
                // and writes dx (6 instead of 2): 8 passes per step where the read-modify-write chain took 10.  g_1 / g_2 sit in the two ping-pong
                // buffers until then (hence L <= 3).
                if (l == 0) {
                    d.accumulate = 0;
                    d.c2_mode = (L == 3) ? 5 : 6;
                    d.E2 = gbuf0; d.E3 = omid; d.lde2 = d.lde3 = D;                                       // g_1 (layer 1 writes gbuf0), O_0
                    if (L == 3) { d.E4 = gbuf1; d.E5 = omid + (size_t)1 * (xbuf(m) / sizeof(float)); }    // g_2, O_1
                    d.E6 = omid + (size_t)(L - 1) * (xbuf(m) / sizeof(float)); d.rv = hd->dscores; d.cv = hd->w;
                }
            } else {
            if (l > 0 && dx) { d.C2 = dx; d.ldc2 = D; d.E2 = omid + (size_t)(l - 1) * (xbuf(m) / sizeof(float)); d.lde2 = D; d.c2_mode = 2; }
            if (hd && l == L - 1 && l > 0 && dx) {       // first write of dx: g_{l-1} * O_{l-1} + dscore (x) w_head * O_{L-1}
                d.c2_mode = 4;
                d.E3 = omid + (size_t)l * (xbuf(m) / sizeof(float)); d.lde3 = D; d.rv = hd->dscores; d.cv = hd->w;
            }
            }
            if (planes_bwd) rn_gemm_planes_hint(mix_plane(m, sv, l, 3));
            if ((rc = rn_gemm(&d, gws, gemm_ws, st))) return rc;
        }
        if (l > 0) MIX_SIGNAL(e_g, st);
        // ---------------- side stream, part 2: needs dC / dT1 of this layer
        MIX_WAIT(e_dT1, st2);
        if (two && paired && (rc = side_dw())) return rc;           // (e_dT1 is later than e_g of this layer)
        {   // dWc1 = x_l^T dT1[:, :NS] -> dU;  dgate[d][n] = x_l^T dlogits as the side product
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = xl; d.lda = D; d.a_trans = 1;
            if (mix_xless_saved(m, sv) && l > 0) { d.A = x; d.A2 = omid + (size_t)(l - 1) * (xbuf(m) / sizeof(float)); d.a_mode = RECNOW_OPMODE_MUL; }      // x_l = x0 * O_{l-1} (not stored)
            d.B = dT1; d.ldb = m.LDT; d.b_trans = 0;
            d.C = dWc1; d.ldc = m.NS;
            d.M = D; d.N = m.NS; d.K = (int)B;
            d.prof_flops = 2.0 * (double)B * D * m.KC;
            d.sp_bx = dT1 + m.NS; d.sp_bx_ks = m.LDT; d.sp_bx_rs = 1; d.sp_cx = dgate_host[l]; d.sp_cx_ms = N; d.sp_cx_rs = 1; d.sp_r = N;
            // K = B is always split at these sizes: the slab reduce stores dU (N, D, S) directly; without a split, unpack afterwards
            recnow_gemm_desc dq = d;
            dq.C = dU_host[l]; dq.c_perm_s = S;
            rc = rn_gemm_deferred(&dq, gws2, gemm_ws, st2, &red_du);
            if (rc == RECNOW_EUNSUPPORTED) {
                if ((rc = rn_gemm(&d, gws2, gemm_ws, st2))) return rc;
                hipLaunchKernelGGL(k_unpack_u, pg, 256, 0, st2, dWc1, D, S, N, dU_host[l]);
                RN_LAUNCH_CHECK();
            } else if (rc) {
                return rc;
            }
        }
        // the layer's end: slab reductions of dW (+ dbias) and dU (+ dgate) and the dV partial sum in ONE launch
        RnProfRecord* pr_end = rn_prof_on() ? rn_prof_begin(RN_TAG_LAYER_END, 0.0, 0.0, st2) : nullptr;
        if ((rc = rn_layer_end_reduce(&red_dw, &red_du, defer_dv ? (const float*)mid_ws : nullptr, rn_mix_mid_bwd_nparts(B), N * S * S, dV_host[l], st2)))
            return rc;
        if (hd && l == L - 1) {      // dW = w_head * M^T, dbias likewise, d w_head = sum_k [W; b] * M^T (on the reduced M^T)
            hipLaunchKernelGGL(k_head_post, rn_cdiv(D, 32), 256, 0, st2, dW_host[l], dbias_host[l], W_host[l], bias_host[l], hd->w, m.NS, N, D, hd->dw,
                               ds_part, ds_nparts, ds_part ? hd->db : nullptr);
            RN_LAUNCH_CHECK();
        }
        rn_prof_end(pr_end, st2);
        MIX_SIGNAL(e_side_prev, st2);
        // every weight gradient of layer l has been issued (dW, dbias above; dV in the sub-space kernel; dU, dgate just now): a
        // caller-owned event lets the all-reduce of this layer start while the lower layers' backward still runs
        if (layer_events && layer_events[l]) RN_HIP(hipEventRecord((hipEvent_t)layer_events[l], two ? st2 : st));
        g = gprev;
    }
    MIX_WAIT(e_side_prev, st);              // join: everything the side stream produced is ordered before later work on st
    return RECNOW_OK;
}

extern "C" int recnow_dcn_mix_bwd(const float* x, const float* const* U_host, const float* const* V_host,
                                  const float* const* W_host, const float* const* bias_host, const float* const* gate_host,
                                  const float* dy, const void* saved, size_t saved_bytes, int64_t B, int D, int S, int N, int L,
                                  int act_inner, int act_outer, float* dx, float* const* dU_host, float* const* dV_host,
                                  float* const* dW_host, float* const* dbias_host, float* const* dgate_host, void* ws,
                                  size_t ws_bytes, void* stream, void* stream2) {
    if (B < 0 || D < 1 || S < 1 || N < 1 || L < 1 || B > 0x7fffffffll) return RECNOW_EINVAL;
    if (N > 64) return RECNOW_EUNSUPPORTED;
    if (!dU_host || !dV_host || !dW_host || !dbias_host || !dgate_host) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        for (int l = 0; l < L; ++l) {
            RN_HIP(hipMemsetAsync(dU_host[l], 0, (size_t)N * D * S * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dV_host[l], 0, (size_t)N * S * S * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dW_host[l], 0, (size_t)N * S * D * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dbias_host[l], 0, (size_t)N * D * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dgate_host[l], 0, (size_t)D * N * sizeof(float), st));
        }
        return RECNOW_OK;
    }
    if (!x || !U_host || !V_host || !W_host || !bias_host || !gate_host || !dy || !saved || !ws) return RECNOW_EINVAL;     // dx may be NULL
    if (saved_bytes < recnow_dcn_mix_saved_bytes(B, D, S, N, L)) return RECNOW_EWORKSPACE;
    if (ws_bytes < recnow_dcn_mix_workspace_bytes(B, D, S, N, L)) return RECNOW_EWORKSPACE;
    const MixDims m = mix_dims(B, D, S, N, L);
    if (m.exact)
        return dcnmix_bwd_exact(m, x, U_host, V_host, W_host, bias_host, gate_host, dy, (const char*)saved, act_inner, act_outer, dx,
                                dU_host, dV_host, dW_host, dbias_host, dgate_host, ws, ws_bytes, st, (hipStream_t)stream2);
    RnCarver c(ws, ws_bytes);
    float* Wc1 = c.take<float>((size_t)D * m.LDT);
    float* Wc2 = c.take<float>((size_t)m.LDT * D);
    float* dWc1 = c.take<float>((size_t)D * m.LDT);
    float* dWc2 = c.take<float>((size_t)m.LDT * D);
    float* dT2g = c.take<float>(act_block(m) / sizeof(float));
    float* dC = c.take<float>(act_block(m) / sizeof(float));
    float* dT1 = c.take<float>(act_block(m) / sizeof(float));
    float* gbuf0 = c.take<float>(xbuf(m) / sizeof(float));
    float* gbuf1 = c.take<float>(xbuf(m) / sizeof(float));
    const size_t mid_ws_bytes = rn_mix_mid_bwd_ws_bytes(B, S, N);
    void* mid_ws = c.take<char>(mid_ws_bytes);
    if (!c.ok()) return RECNOW_EWORKSPACE;
    void* gws = c.base + c.off;
    const size_t gws_bytes = ws_bytes - c.off;
    const char* sv = (const char*)saved;
    const float* xmid = (const float*)(sv + (size_t)L * 3 * act_block(m));
    int rc;
    const float* g = dy;                     // gradient w.r.t. the current layer's output
    bool dx_started = false;                 // dx accumulates the x (= x0) contributions of every layer
    for (int l = L - 1; l >= 0; --l) {
        const float* T1 = (const float*)(sv + (size_t)(3 * l) * act_block(m));
        const float* T2 = (const float*)(sv + (size_t)(3 * l + 1) * act_block(m));
        const float* T2g = (const float*)(sv + (size_t)(3 * l + 2) * act_block(m));
        const float* xl = (l == 0) ? x : xmid + (size_t)(l - 1) * (xbuf(m) / sizeof(float));
        float* gprev = (l == 0) ? nullptr : ((l & 1) ? gbuf0 : gbuf1);
        if ((rc = pack_weights(m, U_host[l], V_host[l], W_host[l], bias_host[l], gate_host[l], Wc1, Wc2, st))) return rc;
        {   // dT2g = (x * g) Wc2^T
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = g; d.A2 = x; d.a_mode = RECNOW_OPMODE_MUL; d.lda = D; d.a_trans = 0;
            d.B = Wc2; d.ldb = D; d.b_trans = 1;
            d.C = dT2g; d.ldc = m.LDT;
            d.M = (int)B; d.N = m.LDT; d.K = D;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        }
        {   // dWc2 = T2g^T (x * g)          -> dW (NS x D) and dbias (N x D).  Computed as the transposed product
            // (M = D rows, N = NS+N columns: tiles 128x160 are 81% full instead of 51% for M = 130) and stored transposed.
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = g; d.A2 = x; d.a_mode = RECNOW_OPMODE_MUL; d.lda = D; d.a_trans = 1;
            d.B = T2g; d.ldb = m.LDT; d.b_trans = 0;
            d.C = dWc2; d.ldc = D; d.c_trans = 1;
            d.M = D; d.N = m.LDT; d.K = (int)B;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
            RN_HIP(hipMemcpyAsync(dW_host[l], dWc2, (size_t)m.NS * D * sizeof(float), hipMemcpyDeviceToDevice, st));
            RN_HIP(hipMemcpyAsync(dbias_host[l], dWc2 + (size_t)m.NS * D, (size_t)N * D * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        if (dx) {   // dx (+)= g * O,  O = T2g Wc2 recomputed
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = T2g; d.lda = m.LDT; d.a_trans = 0;
            d.B = Wc2; d.ldb = D; d.b_trans = 0;
            d.C = dx; d.ldc = D;
            d.M = (int)B; d.N = D; d.K = m.KP; d.k_valid = m.KC;
            d.emul = g; d.lde = D; d.e_mode = RECNOW_OPMODE_MUL;
            d.accumulate = dx_started ? 1 : 0;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
            dx_started = true;
        }
        if ((rc = mix_mid_bwd(m, dT2g, T2, T1, V_host[l], dC, dT1, dV_host[l], act_inner, act_outer, mid_ws, mid_ws_bytes, gws, gws_bytes, st)))
            return rc;
        {   // dWc1 = x_l^T dT1              -> dU, dgate
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = xl; d.lda = D; d.a_trans = 1;
            d.B = dT1; d.ldb = m.LDT; d.b_trans = 0;
            d.C = dWc1; d.ldc = m.LDT;
            d.M = D; d.N = m.LDT; d.K = (int)B;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
            int gg = rn_cdiv((int64_t)D * m.KC, 256);
            if (gg > 2048) gg = 2048;
            hipLaunchKernelGGL(k_unpack_w1, gg, 256, 0, st, dWc1, D, S, N, m.LDT, dU_host[l], dgate_host[l]);
            RN_LAUNCH_CHECK();
        }
        if (l > 0 || dx) {   // gradient w.r.t. x_l:  dT1 Wc1^T.  Layer 0's x_l is x itself -> accumulate into dx.
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = dT1; d.lda = m.LDT; d.a_trans = 0;
            d.B = Wc1; d.ldb = m.LDT; d.b_trans = 1;
            d.C = (l == 0) ? dx : gprev; d.ldc = D;
            d.M = (int)B; d.N = D; d.K = m.KP; d.k_valid = m.KC;
            d.accumulate = (l == 0) ? 1 : 0;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        }
        g = gprev;
    }
    return RECNOW_OK;
}


// Backward of recnow_dcn_mix_score_fwd.  dscores (B) = d loss / d scores.  layer_events_host: optional HOST array of L hipEvent_t
// (entries may be NULL); event l is recorded once every weight gradient of layer l has been issued, the head's gradients are
// complete at event L-1.
extern "C" int recnow_dcn_mix_score_bwd(const float* x, const float* const* U_host, const float* const* V_host,
                                        const float* const* W_host, const float* const* bias_host, const float* const* gate_host,
                                        const float* head_w, const float* dscores, const void* saved, size_t saved_bytes, int64_t B,
                                        int D, int S, int N, int L, int act_inner, int act_outer, float* dx, float* const* dU_host,
                                        float* const* dV_host, float* const* dW_host, float* const* dbias_host,
                                        float* const* dgate_host, float* dhead_w, float* dhead_b, void* ws, size_t ws_bytes,
                                        void* stream, void* stream2, void* const* layer_events_host) {
    if (B < 0 || D < 1 || S < 1 || N < 1 || L < 1 || B > 0x7fffffffll) return RECNOW_EINVAL;
    if (N > 64) return RECNOW_EUNSUPPORTED;
    if (!dU_host || !dV_host || !dW_host || !dbias_host || !dgate_host || !dhead_w) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        for (int l = 0; l < L; ++l) {
            RN_HIP(hipMemsetAsync(dU_host[l], 0, (size_t)N * D * S * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dV_host[l], 0, (size_t)N * S * S * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dW_host[l], 0, (size_t)N * S * D * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dbias_host[l], 0, (size_t)N * D * sizeof(float), st));
            RN_HIP(hipMemsetAsync(dgate_host[l], 0, (size_t)D * N * sizeof(float), st));
            if (layer_events_host && layer_events_host[l]) RN_HIP(hipEventRecord((hipEvent_t)layer_events_host[l], st));
        }
        RN_HIP(hipMemsetAsync(dhead_w, 0, (size_t)D * sizeof(float), st));
        if (dhead_b) RN_HIP(hipMemsetAsync(dhead_b, 0, sizeof(float), st));
        return RECNOW_OK;
    }
    if (!x || !U_host || !V_host || !W_host || !bias_host || !gate_host || !head_w || !dscores || !saved || !ws) return RECNOW_EINVAL;
    if (saved_bytes < recnow_dcn_mix_saved_bytes(B, D, S, N, L)) return RECNOW_EWORKSPACE;
    if (ws_bytes < recnow_dcn_mix_workspace_bytes(B, D, S, N, L)) return RECNOW_EWORKSPACE;
    const MixDims m = mix_dims(B, D, S, N, L);
    if (!mix_head_ok(m)) return RECNOW_EUNSUPPORTED;
    MixHeadGrad hd;
    hd.w = head_w; hd.dscores = dscores; hd.dw = dhead_w; hd.db = dhead_b;
    return dcnmix_bwd_exact(m, x, U_host, V_host, W_host, bias_host, gate_host, nullptr, (const char*)saved, act_inner, act_outer, dx,
                            dU_host, dV_host, dW_host, dbias_host, dgate_host, ws, ws_bytes, st, (hipStream_t)stream2, &hd,
                            layer_events_host);
}


// ---- the north-star step, one call per phase (include/recnow.h: recnow_dcn_mix_step) -------------------------------------------
#define STEP_ROWS 64          // rows per workgroup of the loss stage's gradient kernel
struct StepWs {
    void* saved; size_t saved_bytes;
    void* mix; size_t mix_bytes;
    uint32_t* words; uint8_t* solo;
    int32_t *order, *seg_id, *seg_first, *super_id, *n_seg;
    void* grp; size_t grp_bytes;
    void* pair; size_t pair_bytes;
    float *dsu, *ds, *T2g_ds, *ds_part;
    int n_words;
    size_t total;
    bool ok;
};
static StepWs step_carve(void* ws, size_t ws_bytes, int64_t B, int D, int S, int N, int L, int group_dtype) {
    StepWs w;
    const MixDims m = mix_dims(B, D, S, N, L);
    RnCarver c(ws, ws_bytes);
    w.n_words = recnow_key_words(group_dtype);
    if (w.n_words < 1) w.n_words = 1;
    w.saved_bytes = recnow_dcn_mix_saved_bytes(B, D, S, N, L);
    w.saved = c.take<char>(w.saved_bytes);
    w.mix_bytes = recnow_dcn_mix_workspace_bytes(B, D, S, N, L);
    w.mix = c.take<char>(w.mix_bytes);
    w.words = c.take<uint32_t>((size_t)w.n_words * B);
    w.solo = c.take<uint8_t>(B);
    w.order = c.take<int32_t>(B);
    w.seg_id = c.take<int32_t>(B);
    w.seg_first = c.take<int32_t>(B + 1);
    w.super_id = c.take<int32_t>(B);
    w.n_seg = c.take<int32_t>(2);
    w.grp_bytes = recnow_group_segments_workspace_bytes(B, w.n_words);
    w.grp = c.take<char>(w.grp_bytes);
    w.pair_bytes = recnow_pairwise_workspace_bytes(B);
    w.pair = c.take<char>(w.pair_bytes);
    w.dsu = c.take<float>(B);
    w.ds = c.take<float>(B);
    w.T2g_ds = c.take<float>(act_block(m) / sizeof(float));
    w.ds_part = c.take<float>(rn_cdiv(B, STEP_ROWS));
    w.total = c.off;
    w.ok = c.ok();
    return w;
}
extern "C" size_t recnow_dcn_mix_step_workspace_bytes(int64_t B, int D, int S, int N, int L, int group_dtype) {
    if (B <= 0 || D <= 0 || S <= 0 || N <= 0 || L <= 0) return 256;
    return step_carve(nullptr, 0, B, D, S, N, L, group_dtype).total + 256;      // a carve of a null base only adds up the sizes
}

// Loss stage, last kernel: ds = d(loss)/d(scores) from the unnormalised pair gradients (x 1 / (P + eps) when the loss is the mean);
// the top layer's dscore * T2g (the row-scaled small operand of the fused head's weight gradient) for the same rows; the
// workgroup's partial sum of ds (d head bias, joined by k_head_post); {loss, (float) P} for the caller's statistics.
__global__ void __launch_bounds__(1024)
k_step_dscore(const float* __restrict__ dsu, const unsigned long long* __restrict__ n_pair, int reduce_mean, float eps, int64_t B, int64_t BP,
              const float* __restrict__ T2g_top, int LDT, float* __restrict__ ds, float* __restrict__ T2g_ds, float* __restrict__ ds_part,
              float* __restrict__ loss, float* __restrict__ stats, const int32_t* __restrict__ n_seg, const double* __restrict__ loss_part, int n_loss_part) {
    // Round 5: the first workgroup also sums the pair kernel's loss partials -- 1024 threads, the code of k_loss_finalize (pairwise.hip): the same
    // terms in the same order, so the loss is bit-identical to the autograd route's -- instead of a launch of its own in front of this one.
    // B rows of the batch, BP >= B rows of storage (recnow_dcn_mix_step_desc.B_pad): the padding rows get d loss / d score = 0 and a zero
    // row of dscore * T2g, so that the backward products, which run over all BP rows, add exactly nothing for them
    __shared__ float sds[STEP_ROWS];
    const float P = (float)(*n_pair);
    // n_seg[0] < 0: the cooperative grouping launch timed out at a grid barrier and left the identity grouping (scan_sort.hip) -- zero
    // pairs would read as "loss 0, all-zero gradients" with RECNOW_OK.  Poisoned instead: loss, statistics and d loss / d scores are NaN.
    const bool bad = n_seg[0] < 0;
    const float sc = bad ? __int_as_float(0x7fc00000) : (reduce_mean ? 1.f / (P + eps) : 1.f);
    const int64_t r0 = (int64_t)blockIdx.x * STEP_ROWS;
    if (blockIdx.x == 0) {                    // block-uniform
        __shared__ double red[16];
        double s = 0.0;
        for (int i = threadIdx.x; i < n_loss_part; i += blockDim.x) s += loss_part[i];     // fixed order per thread, fixed tree
        s = block_sum<double>(s, red);
        if (threadIdx.x == 0) {
            float v = (float)s;
            if (reduce_mean) v = v / (P + eps);
            loss[0] = v;
        }
        __syncthreads();
    }
    if (threadIdx.x < STEP_ROWS) {            // one wave
        const int64_t r = r0 + threadIdx.x;
        const float v = r < B ? dsu[r] * sc : 0.f;
        if (r < BP) ds[r] = v;
        sds[threadIdx.x] = v;
        const float t = wave_sum(v);
        if (threadIdx.x == 0) ds_part[blockIdx.x] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (bad) loss[0] = sc;
        if (stats) { stats[0] = bad ? sc : loss[0]; stats[1] = P; }
    }
    __syncthreads();
    const int q = LDT / 4;
    for (int i = threadIdx.x; i < STEP_ROWS * q; i += 1024) {
        const int rl = i / q, c4 = i % q;
        const int64_t r = r0 + rl;
        if (r < BP) {
            float4 v = reinterpret_cast<const float4*>(T2g_top + r * LDT)[c4];
            const float s = sds[rl];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            reinterpret_cast<float4*>(T2g_ds + r * LDT)[c4] = v;
        }
    }
}

extern "C" int recnow_dcn_mix_step(const recnow_dcn_mix_step_desc* d, int phases, int layer_hi, int layer_lo, void* stream) {
    if (!d || d->B < 0 || d->D < 1 || d->S < 1 || d->N < 1 || d->L < 1 || d->B > 0x7fffffffll) return RECNOW_EINVAL;
    if (phases & ~(RECNOW_STEP_GROUP | RECNOW_STEP_FORWARD | RECNOW_STEP_LOSS | RECNOW_STEP_BACKWARD)) return RECNOW_EINVAL;
    // B rows of the batch; BP rows of storage behind x, dx and scores (B_pad: a ragged per-rank batch on the fast route, rows >= B of x zero).
    // The layers run over BP rows, the grouping and the pair loss over the first B; k_step_dscore gives the padding rows a zero gradient.
    const int64_t B = d->B;
    const int64_t BP = d->B_pad > 0 ? d->B_pad : B;
    if (BP < B || BP > 0x7fffffffll || (BP != B && BP % 256)) return RECNOW_EINVAL;
    if (B == 0 || !recnow_dcn_mix_score_supported(BP, d->D, d->S, d->N, d->L)) return RECNOW_EUNSUPPORTED;
    if (recnow_key_words(d->group_dtype) < 1) return RECNOW_EINVAL;
    if (!d->ws || d->ws_bytes < recnow_dcn_mix_step_workspace_bytes(BP, d->D, d->S, d->N, d->L, d->group_dtype)) return RECNOW_EWORKSPACE;
    const int D = d->D, S = d->S, N = d->N, L = d->L;
    const StepWs w = step_carve(d->ws, d->ws_bytes, BP, D, S, N, L, d->group_dtype);
    if (!w.ok) return RECNOW_EWORKSPACE;
    const MixDims m = mix_dims(BP, D, S, N, L);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((phases & RECNOW_STEP_FORWARD) && !d->scores) return RECNOW_EINVAL;
    int npair_zeroed = 0;                    // the GROUP phase of this call cleared *n_pair for its LOSS phase
    if (phases & RECNOW_STEP_GROUP) {
        if (!d->groups) return RECNOW_EINVAL;
        RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_STEP_GROUP, 0.0, 16.0 * B, st) : nullptr;
        // small shards (<= 8192 rows, float32 / int32 ids): keys, solo flags, sort and segments in ONE launch straight from the id tensor
        // ... and, when the forward pass of this call runs the row-block kernels, the packs of their weights on the launch's other workgroups
        // (RECNOW_STEP_FRONT=0: the pack stays a launch of its own in front of the forward kernel)
        static const bool front = []() { const char* e = getenv("RECNOW_STEP_FRONT"); return !e || e[0] != '0'; }();
        RnTileFwd pk;
        const RnTileFwd* pack = nullptr;
        if (front && (phases & RECNOW_STEP_FORWARD) && mix_tile_on(m) && d->U_host && d->V_host && d->W_host && d->bias_host && d->gate_host) {
            memset(&pk, 0, sizeof(pk));
            pk.D = D; pk.L = L; pk.packs = (float*)((char*)w.saved + mix_tile_pack_off(m));
            for (int l = 0; l < L; ++l) { pk.U[l] = d->U_host[l]; pk.Kg[l] = d->gate_host[l]; pk.V[l] = d->V_host[l]; pk.W[l] = d->W_host[l]; pk.bias[l] = d->bias_host[l]; }
            pack = &pk;
        }
        tl_step_tile_packed = nullptr;
        // (the loss stage of THIS call adds into *n_pair: a front kernel clears it on the way)
        unsigned long long* zero1 = ((phases & RECNOW_STEP_LOSS) && d->n_pair) ? (unsigned long long*)d->n_pair : nullptr;
        int packs_written = 0;          // the grouping launch also wrote the row-block kernels' weight packs (only the front kernels do)
        rc = rn_group_small_raw(d->groups, d->group_dtype, B, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, st, pack, zero1, &npair_zeroed, &packs_written);
        // larger batches: the cooperative launch forms keys and solo flags from the id tensor as well
        if (rc == RECNOW_EUNSUPPORTED)
            rc = rn_group_mid_raw(d->groups, d->group_dtype, B, w.solo, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, w.grp, w.grp_bytes, st, pack, zero1,
                                  &npair_zeroed, &packs_written);
        if (rc == RECNOW_OK && pack && packs_written) tl_step_tile_packed = w.saved;
        if (rc == RECNOW_EUNSUPPORTED) {
            RN_HIP(hipMemsetAsync(w.solo, 0, (size_t)B, st));
            if ((rc = recnow_group_keys(d->groups, d->group_dtype, B, w.words, w.solo, stream))) return rc;
            rc = recnow_group_segments(w.words, w.solo, B, w.n_words, w.n_words, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, w.grp,
                                       w.grp_bytes, stream);
        }
        if (rc) return rc;
        rn_prof_end(pr, st);
    }
    if (phases & RECNOW_STEP_FORWARD) {
        if (!d->scores) return RECNOW_EINVAL;
        rc = recnow_dcn_mix_score_fwd(d->x, d->U_host, d->V_host, d->W_host, d->bias_host, d->gate_host, d->head_w, d->head_b, BP, D, S, N, L,
                                      d->act_inner, d->act_outer, d->scores, w.saved, w.saved_bytes, w.mix, w.mix_bytes, stream, d->dx ? 1 : 0);
        tl_step_tile_packed = nullptr;     // (taken by the row-block forward; cleared here whatever route or error the call took)
        if (rc) return rc;
    }
    if (phases & RECNOW_STEP_LOSS) {
        if (!d->scores || !d->labels || !d->loss || !d->n_pair) return RECNOW_EINVAL;
        // no pack launch: the pair walk fills its LDS stages from the inputs (RN_PAIR_UNPACKED); RECNOW_STEP_PACK=1 keeps the packed form (A/B)
        static const bool packed = []() { const char* e = getenv("RECNOW_STEP_PACK"); return e && e[0] == '1'; }();
        const int flags = RECNOW_PAIR_LABEL_GT | (d->only_use_wrong_order_pair ? RECNOW_PAIR_WRONG_ORDER : 0) |
                          (packed ? 0 : (RN_PAIR_UNPACKED | (npair_zeroed ? RN_PAIR_NPAIR_ZEROED : 0)));
        RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_STEP_LOSS, 0.0, 16.0 * B + 8.0 * BP * m.LDT, st) : nullptr;
        const double* loss_part = nullptr;
        int n_loss_part = 0;
        if ((rc = rn_pair_bpr_onepass(d->scores, d->labels, d->mask, w.order, w.seg_id, w.seg_first, B, flags, d->factor, d->reduce_mean,
                                      d->loss, w.dsu, d->n_pair, w.pair, w.pair_bytes, stream, &loss_part, &n_loss_part)))
            return rc;
        const float* T2g_top = (const float*)((const char*)w.saved + (size_t)(3 * (L - 1) + 2) * act_block(m));
        hipLaunchKernelGGL(k_step_dscore, rn_cdiv(BP, STEP_ROWS), 1024, 0, st, w.dsu, (const unsigned long long*)d->n_pair, d->reduce_mean, 1.0e-10f, B, BP,
                           T2g_top, m.LDT, w.ds, w.T2g_ds, w.ds_part, d->loss, d->stats, w.n_seg, loss_part, n_loss_part);
        RN_LAUNCH_CHECK();
        rn_prof_end(pr, st);
    }
    if (phases & RECNOW_STEP_BACKWARD) {
        if (!d->x || !d->U_host || !d->V_host || !d->W_host || !d->bias_host || !d->gate_host || !d->head_w || !d->dU_host || !d->dV_host ||
            !d->dW_host || !d->dbias_host || !d->dgate_host || !d->dhead_w)
            return RECNOW_EINVAL;
        MixHeadGrad hd;
        hd.w = d->head_w; hd.dscores = w.ds; hd.dw = d->dhead_w; hd.db = d->dhead_b;
        if ((rc = dcnmix_bwd_exact(m, d->x, d->U_host, d->V_host, d->W_host, d->bias_host, d->gate_host, nullptr, (const char*)w.saved,
                                   d->act_inner, d->act_outer, d->dx, d->dU_host, d->dV_host, d->dW_host, d->dbias_host, d->dgate_host, w.mix,
                                   w.mix_bytes, st, (hipStream_t)d->stream2, &hd, d->layer_events_host, layer_hi, layer_lo, w.T2g_ds, w.ds_part, rn_cdiv(BP, STEP_ROWS))))
            return rc;
    }
    return RECNOW_OK;
}
