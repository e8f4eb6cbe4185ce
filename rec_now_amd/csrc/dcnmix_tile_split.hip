// Split-precision form of the row-block persistent FORWARD of DCNMixLayer (+ the folded scoring head): k_mix_tile_fwd (dcnmix_tile.hip) with its two
// large products per layer on v_mfma_f32_32x32x16_bf16 -- every fp32 operand as three bf16 pieces, six MFMA terms, fp32 accumulation: the arithmetic of
// gemm_split.hip (per-product error <= 2^-25, the same 1e-5 parity tests) -- instead of v_mfma_f32_32x32x2_f32.
// /root/reference/rec_now/layers/dcn_mix_layer.py:123-150, every layer in ONE launch; the layer input never leaves the registers.
//
// What changes against the exact kernel (whose structure, phases, LDS tiles and stores are kept):
//   * the block's layer input stays in the registers as fp32 A fragments exactly as in the exact kernel (lane (row = lane & 31, h = lane >> 5) holds
//     x_l[row][32 b + 8 q + 4 h + i] in xa[b][q][i]); k-step (b, u) of GEMM1 takes xa[b][2 u] and xa[b][2 u + 1] -- eight values -- as the eight k-slots
//     8 h + 4 qq + i of one v_mfma_f32_32x32x16_bf16 A fragment (the contraction order of a product is free; the weight planes are packed to match,
//     k_tile_pack_split) and splits them into the three bf16 pieces on the spot: ~45 VALU instructions under the 24 MFMAs of the step.  (First form:
//     x_l as three piece PLANES, 192 registers per lane at D = 1024, split where x_{l+1} is formed -- the kernel then spilled ~460 registers into
//     its product loops.)
//   * the weights stream from L2 as piece planes in fragment order: one 16-byte load per lane, piece and 32 x 16 fragment, 1 KiB contiguous per wave
//     instruction.  1.63 MB per layer and 32-row block (1.05 MB in fp32): at the 66-73 GB/s a CU takes from its XCD's L2 (MI355X_MICROARCH.md,
//     "Indexed rows", shared table) that is ~24 us per layer and block against ~14 us of MFMA -- the products of this kernel are bound by the L2 -> CU
//     stream of the weights, not by the matrix pipe (measured with the phase stamps: GEMM1 12-20 us, output product 16-33 us per layer and block;
//     tools/tile_split_trace.py): 32 rows per workgroup is all the register file holds of x_l.
//   * the output product reads its B fragments (the gated sub-space outputs T2g) from bf16 piece planes of the tile in LDS, split once per layer by
//     the whole workgroup; the bias rows ride in the packed weights as k = 128, 129 against the gates (as in the K = 144 products of the other route).
//   * gate logits: fp32 terms per k-step, fp64 running sums (the gate logits of layer 0 on inputs of O(6) were the largest single consumer of the
//     parity budget with fp32 sums); the gate kernel of the layer in flight sits in LDS.
// The sub-space stage (64 x 64 per expert) stays on the exact fp32 MFMA: 32 of the ~850 MFMAs of a layer.
// MEASURED (one MI355X, B = 65 536, D = 1024, L = 3): 1.25 ms per launch -- the exact kernel's time, and 0.30 ms MORE than the three layers' forward
// launches of the product route in split precision (0.95 ms): the route is therefore OPT-IN (RECNOW_TILE_SPLIT=1, dcnmix.hip), held to the oracle by
// tests/test_tile_gpu.py.  Where the time goes: DESIGN.md 5l.
#include <string.h>
#include <atomic>
#include "dcnmix_tile.hpp"
#include "gemm_split.hpp"
#include "prof.hpp"

#define TS_ROWS 32
#define TS_LDT 144
#define TS_LDP 132
#define TS_LDA 129
#define TS_LDG 132
#define TS_LDGP 152         // bf16 row stride of the T2g piece planes in LDS: 76 dwords -- the 16 lanes of a ds_read_b128 group land on 16 distinct bank quads
#define TS_PLG (TS_ROWS * TS_LDGP * 2)
#define TS_LDS_BYTES(D) ((4 * TS_ROWS * TS_LDP + 4 * TS_ROWS * 2 * 2 + TS_ROWS * TS_LDA + TS_ROWS * 2 + 4 * TS_ROWS + 3 * (D)) * 4 + 3 * TS_PLG)

size_t rn_mix_tile_split_pack_bytes(int D, int S, int N, int L, int LDT) {
    if (!rn_mix_tile_supported(TS_ROWS, D, S, N, L, LDT)) return 0;
    return rn_align((size_t)L * TLS_LAYER_BYTES(D));
}

// Piece planes of the weights in fragment order, per layer [P1s: 3 planes][P2s: 3 planes]:
//   P1s unit (gs, cb, h, col)  = pieces of U_l[n][d(gs, h, j)][s], n S + s = 32 cb + col, d = 32 (gs >> 1) + 8 (2 (gs & 1) + (j >> 2)) + 4 h + (j & 3)    GEMM1, B fragments
//   P2s unit (st, db, h, dr)   = pieces of [W_l; b_l; 0][t = 16 st + 8 h + j][d = 32 db + dr]                                                              output product, A fragments
// One thread per 16-byte unit (its eight source values are eight rows of the source: strided reads of 2 MB once per step).
__global__ void __launch_bounds__(256) k_tile_pack_split(const RnTileFwd p) {
    const int D = p.D;
    const int64_t u1 = TLS_P1_UNITS(D), u2 = TLS_P2_UNITS(D), per = u1 + u2, total = (int64_t)p.L * per;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int l = (int)(e / per);
        const int64_t j = e - (int64_t)l * per;
        char* base = p.splanes + (size_t)l * TLS_LAYER_BYTES(D);
        float x[8];
        char* dst;
        size_t plane;
        if (j < u1) {
            const int col = (int)(j & 31), h = (int)((j >> 5) & 1), cb = (int)((j >> 6) & 3), gs = (int)(j >> 8);
            const int cn = cb * 32 + col, n = cn >> 6, s = cn & 63;
            const float* src = p.U[l] + (int64_t)n * D * 64 + s;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) x[jj] = src[(int64_t)(32 * (gs >> 1) + 8 * (2 * (gs & 1) + (jj >> 2)) + 4 * h + (jj & 3)) * 64];
            dst = base + j * 16;
            plane = (size_t)u1 * 16;
        } else {
            const int64_t j2 = j - u1;
            const int dr = (int)(j2 & 31), h = (int)((j2 >> 5) & 1), db = (int)((j2 >> 6) % (D / 32)), st = (int)((j2 >> 6) / (D / 32));
            const int d = 32 * db + dr;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int t = 16 * st + 8 * h + jj;
                x[jj] = t < 128 ? p.W[l][(int64_t)t * D + d] : t < 130 ? p.bias[l][(int64_t)(t - 128) * D + d] : 0.f;
            }
            dst = base + 3 * (size_t)u1 * 16 + j2 * 16;
            plane = (size_t)u2 * 16;
        }
        u32x4 w[3];
        spl_split8(x, w);
#pragma unroll
        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(dst + s * plane) = w[s];
    }
}

// diagnostic build (tools/build_variant.py tstrace -DRN_TILE_TRACE, tools/tile_split_trace.py): wall-clock stamps (100 MHz) of workgroup 0, wave 0
#ifdef RN_TILE_TRACE
__device__ long long g_ts_trace[64];
#define TS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ts_trace[(i)] = wall_clock64(); } while (0)
extern "C" int recnow_debug_tile_split_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ts_trace), sizeof(long long) * 64); }
#else
#define TS_STAMP(i) do { } while (0)
#endif
typedef float ts_f32x16 __attribute__((ext_vector_type(16)));
#define TS_OPAQUE(v) asm volatile("" : "+v"(v))
#define TS_SB() __builtin_amdgcn_sched_barrier(0)
#define TS_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
#define TS_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// the six terms of one fragment pair: A pieces a0..a2, B pieces b0..b2 (largest first)
#define TS_SIX(a0, a1, a2, b0, b1, b2, c)   \
    do {                                    \
        c = TS_MFMA(a0, b0, c);             \
        c = TS_MFMA(a0, b1, c);             \
        c = TS_MFMA(a1, b0, c);             \
        c = TS_MFMA(a1, b1, c);             \
        c = TS_MFMA(a0, b2, c);             \
        c = TS_MFMA(a2, b0, c);             \
    } while (0)

template <int NB, bool TANH>
__global__ void __launch_bounds__(256, 1) k_mix_tile_fwd_s3(const RnTileFwd p) {
    constexpr int D = 128 * NB;
    constexpr size_t PL1 = (size_t)TLS_P1_UNITS(D) * 16, PL2 = (size_t)TLS_P2_UNITS(D) * 16;
    static_assert(NB % 2 == 0, "d-blocks are walked in pairs");
    extern __shared__ float lds[];
    float* Ps = lds;                                                  // [4][32][LDP] partial T1 tiles of the four waves (GEMM1 -> phase B) ...
    float* T2s = Ps;                                                  // ... then [32][LDG] T2 tile and
    float* G2 = Ps + TS_ROWS * TS_LDG;                                //     [32][LDG] T2g tile, fp32, on their way to memory and to the piece planes
    double* Pg = reinterpret_cast<double*>(Ps + 4 * TS_ROWS * TS_LDP);   // [4][32][2] partial gate logits (fp64)
    float* Hs = reinterpret_cast<float*>(Pg + 4 * TS_ROWS * 2);       // [32][LDA]    H1 tile
    float* Gs = Hs + TS_ROWS * TS_LDA;                                // [32][2]      gates
    float* Sc = Gs + TS_ROWS * 2;                                     // [4][32]      score partials
    float* Hv = Sc + 4 * TS_ROWS;                                     // [D]          the scoring head's vector (zeros without a head)
    float* KgS = Hv + D;                                              // [D][2]       gate kernel of the layer whose input is being formed
    char* G2p = reinterpret_cast<char*>(KgS + 2 * D);                 // [3][32][LDGP] bf16 piece planes of the T2g tile
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 31, h = lane >> 5;
    const int act_inner = TANH ? RECNOW_ACT_TANH : p.act_inner, act_outer = TANH ? RECNOW_ACT_TANH : p.act_outer;
    const int64_t ntiles = p.B / TS_ROWS;
    for (int i = tid; i < D; i += 256) Hv[i] = p.head_w ? p.head_w[i] : 0.f;
    for (int i = tid; i < D / 2; i += 256) reinterpret_cast<rn_f4*>(KgS)[i] = reinterpret_cast<const rn_f4*>(p.Kg[0])[i];
    __syncthreads();
    unsigned vL = (unsigned)lane * 16u;
    const int gs0 = w * NB * 2, db0 = w * NB;
    const float* kgl = KgS + (w * NB * 32 + 4 * h) * 2;               // this lane's first gate-kernel row (+ (32 b + 8 q) * 2)
    // GEMM1 ring: step gsl = one k-step of 16 over all four column blocks: 12 loads (3 pieces x 4 column blocks), 24 MFMAs; PF1 steps ahead.  The first
    // PF1 steps of a layer are requested one phase early: in the last steps of the output product of the layer before (for the last layer: layer 0
    // again, what the workgroup's next block starts with), so that no loop starts with a load latency.
    constexpr int NS1 = 2 * NB, PF1 = 2, NSL1 = PF1 + 1;
    u32x4 wr1[NSL1][12];
    auto ld1 = [&](const char* P1l, int gsl, int slot) {
        const char* base = P1l + (size_t)(gs0 + gsl) * 4096;
        TS_OPAQUE(vL);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) wr1[slot][s * 4 + cb] = *reinterpret_cast<const u32x4*>(base + s * PL1 + cb * 1024 + vL);
    };
#pragma unroll
    for (int s = 0; s < PF1; ++s) ld1(p.splanes, s, s);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TS_ROWS;
        const int64_t xoff = (r0 + c) * D + 4 * h + w * NB * 32;
        // the block's rows in the A-fragment layout, fp32 (as the exact kernel's xa): lane (row, h) holds x_l[row][32 b + 8 q + 4 h + i] in xa[b][q][i];
        // k-step (b, u) of GEMM1 splits xa[b][2 u] and xa[b][2 u + 1] into its three bf16 fragments on the spot (the VALU work rides under the 24 MFMAs
        // of the step).  Kept as planes instead (192 registers) the D = 1024 kernel spilled ~460 registers into its product loops.
        rn_f4 xa[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) xa[b][q] = *reinterpret_cast<const rn_f4*>(p.x + xoff + 32 * b + 8 * q);
        TS_STAMP(0);
        for (int l = 0; l < p.L; ++l) {
            const char* P1l = p.splanes + (size_t)l * TLS_LAYER_BYTES(D);
            const char* P2l = P1l + 3 * PL1;
            const int ln = l + 1 < p.L ? l + 1 : 0;
            // ---- GEMM1 over this wave's quarter of K: partial T1 (4 column blocks) + partial gate logits (fp32 terms per k-step, fp64 running sums)
            {
                ts_f32x16 acc[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
                double dg0 = 0.0, dg1 = 0.0;
#pragma unroll
                for (int gsl = 0; gsl < NS1; ++gsl) {
                    const int b = gsl >> 1, u = gsl & 1, slot = gsl % NSL1;
                    if (gsl + PF1 < NS1) ld1(P1l, gsl + PF1, (gsl + PF1) % NSL1);
                    const rn_f4 va = xa[b][2 * u], vb4 = xa[b][2 * u + 1];
                    const float* kp = kgl + (32 * b + 16 * u) * 2;
                    const rn_f4 k0 = *reinterpret_cast<const rn_f4*>(kp), k1 = *reinterpret_cast<const rn_f4*>(kp + 4),
                                k2 = *reinterpret_cast<const rn_f4*>(kp + 16), k3 = *reinterpret_cast<const rn_f4*>(kp + 20);
                    TS_SB();
                    u32x4 fa[3];
                    {
                        unsigned p1, p2, p3;
                        spl_split2(va.x, va.y, p1, p2, p3); fa[0][0] = p1; fa[1][0] = p2; fa[2][0] = p3;
                        spl_split2(va.z, va.w, p1, p2, p3); fa[0][1] = p1; fa[1][1] = p2; fa[2][1] = p3;
                        spl_split2(vb4.x, vb4.y, p1, p2, p3); fa[0][2] = p1; fa[1][2] = p2; fa[2][2] = p3;
                        spl_split2(vb4.z, vb4.w, p1, p2, p3); fa[0][3] = p1; fa[1][3] = p2; fa[2][3] = p3;
                    }
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        TS_SIX(fa[0], fa[1], fa[2], wr1[slot][cb], wr1[slot][4 + cb], wr1[slot][8 + cb], acc[cb]);
                    {
                        float t0 = va.x * k0.x, t1 = va.x * k0.y;
                        t0 = fmaf(va.y, k0.z, t0); t1 = fmaf(va.y, k0.w, t1);
                        t0 = fmaf(va.z, k1.x, t0); t1 = fmaf(va.z, k1.y, t1);
                        t0 = fmaf(va.w, k1.z, t0); t1 = fmaf(va.w, k1.w, t1);
                        t0 = fmaf(vb4.x, k2.x, t0); t1 = fmaf(vb4.x, k2.y, t1);
                        t0 = fmaf(vb4.y, k2.z, t0); t1 = fmaf(vb4.y, k2.w, t1);
                        t0 = fmaf(vb4.z, k3.x, t0); t1 = fmaf(vb4.z, k3.y, t1);
                        t0 = fmaf(vb4.w, k3.z, t0); t1 = fmaf(vb4.w, k3.w, t1);
                        dg0 += (double)t0;
                        dg1 += (double)t1;
                    }
                    TS_SB();
                }
                TS_STAMP(2 + 6 * l);
                {
                    float* Pw = Ps + w * TS_ROWS * TS_LDP + 4 * h * TS_LDP + c;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) Pw[((r & 3) + 8 * (r >> 2)) * TS_LDP + cb * 32] = acc[cb][r];
                }
                dg0 += __shfl_xor(dg0, 32, 64);
                dg1 += __shfl_xor(dg1, 32, 64);
                if (h == 0) {
                    Pg[(w * TS_ROWS + c) * 2] = dg0;
                    Pg[(w * TS_ROWS + c) * 2 + 1] = dg1;
                }
            }
            // the output product's ring: step s2 = (d-block db, k-step st): 3 loads (the pieces of one 32 x 16 fragment of [W; b]^T), 6 MFMAs; PF2 steps ahead.
            // Its first steps are requested here, two barriers ahead of the loop.
            constexpr int NST2 = NB * 9, PF2 = 6, NSL2 = PF2 + 1;
            u32x4 wr2[NSL2][3];
            auto ld2 = [&](int s2, int slot) {
                const int db = s2 / 9, st = s2 % 9;
                const char* base = P2l + ((size_t)st * (D / 32) + db0 + db) * 1024;
                TS_OPAQUE(vL);
#pragma unroll
                for (int s = 0; s < 3; ++s) wr2[slot][s] = *reinterpret_cast<const u32x4*>(base + s * PL2 + vL);
            };
#pragma unroll
            for (int s = 0; s < PF2; ++s) ld2(s, s);
            // B fragments of the sub-space stage (this wave's output block (n, cb)): requested now, needed two barriers on
            const int en = w >> 1, ecb = w & 1;
            float vb[32];
            {
                const float* __restrict__ Vp = p.V[l] + en * 4096 + h * 64 + ecb * 32 + c;
#pragma unroll
                for (int st = 0; st < 32; ++st) vb[st] = Vp[st * 128];
            }
            __syncthreads();
            TS_STAMP(3 + 6 * l);
            // ---- phase B: T1 = act_inner(sum of the partials) -> global + H1 tile; gate softmax; the next input's gate kernel -> LDS
            {
                const int r = 4 * (tid >> 5) + ((tid >> 3) & 3);
                float* __restrict__ T1g = p.T1[l] + (r0 + r) * TS_LDT;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k4 = ((tid & 7) + 8 * i) * 4;
                    const float* pp = Ps + r * TS_LDP + k4;
                    const rn_f4 a0 = *reinterpret_cast<const rn_f4*>(pp), a1 = *reinterpret_cast<const rn_f4*>(pp + TS_ROWS * TS_LDP),
                                a2 = *reinterpret_cast<const rn_f4*>(pp + 2 * TS_ROWS * TS_LDP), a3 = *reinterpret_cast<const rn_f4*>(pp + 3 * TS_ROWS * TS_LDP);
                    rn_f4 v = (a0 + a1) + (a2 + a3);
                    v.x = rn_act(v.x, act_inner); v.y = rn_act(v.y, act_inner); v.z = rn_act(v.z, act_inner); v.w = rn_act(v.w, act_inner);
                    *reinterpret_cast<rn_f4*>(T1g + k4) = v;
                    float* d = Hs + r * TS_LDA + k4;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
                if (tid < TS_ROWS) {
                    float lg[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        lg[n] = (float)((Pg[tid * 2 + n] + Pg[(TS_ROWS + tid) * 2 + n]) + (Pg[(2 * TS_ROWS + tid) * 2 + n] + Pg[(3 * TS_ROWS + tid) * 2 + n]));
                    p.T1[l][(r0 + tid) * TS_LDT + TL_NS] = lg[0];
                    p.T1[l][(r0 + tid) * TS_LDT + TL_NS + 1] = lg[1];
                    const float mx = lg[0] > lg[1] ? lg[0] : lg[1];
                    const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx), sum = e0 + e1;
                    Gs[tid * 2] = e0 / sum;
                    Gs[tid * 2 + 1] = e1 / sum;
                }
                {   // (read by the epilogue of this layer's output product -- the input of layer l + 1 -- or by the next block's first rows)
                    const rn_f4* __restrict__ src = reinterpret_cast<const rn_f4*>(p.Kg[ln]);
                    for (int i = tid; i < D / 2; i += 256) reinterpret_cast<rn_f4*>(KgS)[i] = src[i];
                }
            }
            __syncthreads();
            TS_STAMP(4 + 6 * l);
            // ---- phase C: H2_n = act_outer(H1_n V_n) (exact fp32 MFMA), T2 = [H2 | G | 0], T2g = [G_n H2_n | G | 0]
            {
                ts_f32x16 a2;
#pragma unroll
                for (int r = 0; r < 16; ++r) a2[r] = 0.f;
                const float* ap = Hs + c * TS_LDA + en * 64 + h;
#pragma unroll
                for (int st = 0; st < 32; ++st) a2 = TS_MFMA32(ap[2 * st], vb[st], a2);
                const int col = en * 64 + ecb * 32 + c;
                float gsel[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) gsel[r] = Gs[((r & 3) + 8 * (r >> 2) + 4 * h) * 2 + en];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float h2 = rn_act(a2[r], act_outer);
                    T2s[rr * TS_LDG + col] = h2;
                    G2[rr * TS_LDG + col] = gsel[r] * h2;
                }
                {   // columns 128 .. 143 of T2 and T2g: [G | 0]; thread = (row, tensor, float4)
                    const int row = tid >> 3, q = tid & 3;
                    rn_f4 g4 = {0.f, 0.f, 0.f, 0.f};
                    if (q == 0) { g4.x = Gs[row * 2]; g4.y = Gs[row * 2 + 1]; }
                    float* const t2u = p.T2[l];
                    float* const t2gu = p.T2g[l];
                    float* dst = ((tid & 4) ? t2gu : t2u) + (r0 + row) * TS_LDT + TL_NS + 4 * q;
                    *reinterpret_cast<rn_f4*>(dst) = g4;
                }
            }
            __syncthreads();
            {   // T2 and T2g rows of the tile: thread = (row, 16-byte piece), eight threads per 128 bytes
                const int row = tid >> 3;
                float* __restrict__ T2r = p.T2[l] + (r0 + row) * TS_LDT;
                float* __restrict__ T2gr = p.T2g[l] + (r0 + row) * TS_LDT;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k4 = ((tid & 7) + 8 * i) * 4;
                    *reinterpret_cast<rn_f4*>(T2r + k4) = *reinterpret_cast<const rn_f4*>(T2s + row * TS_LDG + k4);
                    *reinterpret_cast<rn_f4*>(T2gr + k4) = *reinterpret_cast<const rn_f4*>(G2 + row * TS_LDG + k4);
                }
            }
            {   // the T2g tile [G_n H2_n | G | 0] (32 x 144) -> bf16 piece planes: one 16-byte unit (8 consecutive t of a row) per thread and pass
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int uidx = tid + 256 * it;
                    if (uidx < TS_ROWS * 18) {
                        const int row = uidx / 18, u8 = uidx - row * 18;
                        float v[8];
                        if (u8 < 16) {
                            const rn_f4 a = *reinterpret_cast<const rn_f4*>(G2 + row * TS_LDG + 8 * u8), b = *reinterpret_cast<const rn_f4*>(G2 + row * TS_LDG + 8 * u8 + 4);
                            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = 0.f;
                            if (u8 == 16) { v[0] = Gs[row * 2]; v[1] = Gs[row * 2 + 1]; }
                        }
                        u32x4 wq[3];
                        spl_split8(v, wq);
#pragma unroll
                        for (int s = 0; s < 3; ++s) *reinterpret_cast<u32x4*>(G2p + s * TS_PLG + row * (TS_LDGP * 2) + u8 * 16) = wq[s];
                    }
                }
            }
            __syncthreads();
            TS_STAMP(5 + 6 * l);
            // ---- phase D: O^T = [W; b]^T T2g^T over this wave's d-blocks, one block at a time; x_{l+1} = x * O_l -> the next layer's planes and gate logits
            {
                const bool last = l == p.L - 1;
                float* __restrict__ Og = p.O[l];
                float* __restrict__ Xg = p.xn[l];
                const bool use_head = last && p.head_w != nullptr;
                const float* hvp = Hv + 4 * h + w * NB * 32;
                const char* tbp = G2p + c * (TS_LDGP * 2) + h * 16;
                const char* P1n = p.splanes + (size_t)ln * TLS_LAYER_BYTES(D);
                float sp = 0.f;
                u32x4 tb[2][3];
#pragma unroll
                for (int s = 0; s < 3; ++s) tb[0][s] = *reinterpret_cast<const u32x4*>(tbp + s * TS_PLG);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    rn_f4 x0[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x0[q] = *reinterpret_cast<const rn_f4*>(p.x + xoff + 32 * b + 8 * q);
                    ts_f32x16 o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
                    for (int st = 0; st < 9; ++st) {
                        const int s2 = b * 9 + st, slot = s2 % NSL2, cur = s2 & 1;
                        if (s2 + PF2 < NST2) ld2(s2 + PF2, (s2 + PF2) % NSL2);
                        if (b == NB - 1 && st >= 9 - PF1) ld1(P1n, st - (9 - PF1), st - (9 - PF1));      // GEMM1 of the next layer (or block) starts its ring
                        {
                            const int stn = st + 1 < 9 ? st + 1 : 0;        // (the d-block after this one starts at k-step 0 again)
#pragma unroll
                            for (int s = 0; s < 3; ++s) tb[cur ^ 1][s] = *reinterpret_cast<const u32x4*>(tbp + s * TS_PLG + stn * 32);
                        }
                        TS_SB();
                        TS_SIX(wr2[slot][0], wr2[slot][1], wr2[slot][2], tb[cur][0], tb[cur][1], tb[cur][2], o);
                        TS_SB();
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const rn_f4 ov = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
                        if (Og) *reinterpret_cast<rn_f4*>(Og + xoff + 32 * b + 8 * q) = ov;
                        const rn_f4 xv = x0[q] * ov;
                        xa[b][q] = xv;
                        if (Xg) *reinterpret_cast<rn_f4*>(Xg + xoff + 32 * b + 8 * q) = xv;
                        const rn_f4 t = xv * *reinterpret_cast<const rn_f4*>(hvp + 32 * b + 8 * q);
                        sp += (t.x + t.y) + (t.z + t.w);
                    }
                    TS_SB();
                }
                TS_STAMP(6 + 6 * l);
                if (use_head) {                     // scoring head: join the two k-halves of a row, then the four waves in a fixed order
                    sp += __shfl_xor(sp, 32, 64);
                    if (h == 0) Sc[w * TS_ROWS + c] = sp;
                    __syncthreads();
                    if (tid < TS_ROWS)
                        p.scores[r0 + tid] = (p.head_b ? p.head_b[0] : 0.f) + ((Sc[tid] + Sc[TS_ROWS + tid]) + (Sc[2 * TS_ROWS + tid] + Sc[3 * TS_ROWS + tid]));
                }
            }
        }
    }
}

template <int NB>
static int ts_launch(const RnTileFwd& p, int grid, hipStream_t st) {
    const size_t lds = (size_t)TS_LDS_BYTES(128 * NB);
    const bool tanh2 = p.act_inner == RECNOW_ACT_TANH && p.act_outer == RECNOW_ACT_TANH;
    static std::atomic<bool> raised[2][64];
    int dev = 0;
    RN_HIP(hipGetDevice(&dev));
    const int which = tanh2 ? 1 : 0;
    if (dev < 0 || dev >= 64 || !raised[which][dev].load(std::memory_order_acquire)) {
        hipError_t e = tanh2 ? hipFuncSetAttribute((const void*)k_mix_tile_fwd_s3<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                             : hipFuncSetAttribute((const void*)k_mix_tile_fwd_s3<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) raised[which][dev].store(true, std::memory_order_release);
    }
    if (tanh2) hipLaunchKernelGGL((k_mix_tile_fwd_s3<NB, true>), grid, 256, lds, st, p);
    else hipLaunchKernelGGL((k_mix_tile_fwd_s3<NB, false>), grid, 256, lds, st, p);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// packs the piece planes (one launch) and runs every layer of the forward pass in ONE launch; p.splanes: rn_mix_tile_split_pack_bytes
int rn_mix_tile_fwd_split(const RnTileFwd& p, hipStream_t st) {
    if (!rn_mix_tile_supported(p.B, p.D, 64, 2, p.L, TS_LDT) || !p.splanes || !p.x) return RECNOW_EUNSUPPORTED;
    if (p.head_w && !p.scores) return RECNOW_EINVAL;
    {
        const int64_t total = (int64_t)p.L * (TLS_P1_UNITS(p.D) + TLS_P2_UNITS(p.D));
        int g = rn_cdiv(total, 256);
        if (g > 1024) g = 1024;
        hipLaunchKernelGGL(k_tile_pack_split, g, 256, 0, st, p);
        RN_LAUNCH_CHECK();
    }
    const int64_t tiles = p.B / TS_ROWS;
    const int grid = (int)(tiles < 256 ? tiles : 256);
    RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_MIX_TILE_FWD, (double)p.L * (4.0 * p.B * p.D * 130 + 4.0 * p.B * 2 * 64 * 64),
                                                    (double)p.L * (12.0 * p.B * TS_LDT + 8.0 * p.B * p.D), st)
                                    : nullptr;
    int rc;
    switch (p.D) {
        case 256: rc = ts_launch<2>(p, grid, st); break;
        case 512: rc = ts_launch<4>(p, grid, st); break;
        default: rc = ts_launch<8>(p, grid, st); break;
    }
    rn_prof_end(pr, st);
    return rc;
}
