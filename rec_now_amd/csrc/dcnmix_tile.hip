// Row-block persistent forward of DCNMixLayer (+ the folded scoring head) for SMALL batches: the per-rank shards of the metric's
// 2/4/8-GPU rows (8192 .. 32 768 rows).  /root/reference/rec_now/layers/dcn_mix_layer.py:123-150, every layer, in ONE launch.
//
// Why: at 8192 rows the launch-per-product forward is 9 launches (GEMM1, sub-space stage, output product per layer) that are each ONE
// round of workgroups -- prologue + a few k-tiles + epilogue, 27 / 15 / 28 us for 2.2 GFLOP (16 us at the rate the chip sustains) -- and
// every intermediate goes through memory between them.  Here a workgroup of four waves owns a block of 32 rows and takes it through all
// layers; the layer input never leaves the registers:
//   * wave w holds columns [w D/4, (w+1) D/4) of the block's x_l in the A-FRAGMENT layout of v_mfma_f32_32x32x2_f32: lane (row = lane & 31,
//     h = lane >> 5) holds x_l[row][32 b + 8 q + 4 h + i] in xa[b][q][i].  The contraction order of a product is free, so MFMA step (b, q, i)
//     contracts the k-pair (32 b + 8 q + i, 32 b + 8 q + 4 + i) and the weights are packed (k_tile_pack) so that the B fragments of four
//     consecutive steps are one 16-byte load per lane, 512 contiguous bytes per half wave.
//   * GEMM1  T1[row][s] = x_l [U | K]: every wave contracts ITS quarter of K = D for all 128 columns (4 accumulator blocks); the four
//     partial tiles meet in LDS, are summed in a fixed order, activated, stored (T1 is a saved activation) and staged for
//   * the sub-space stage (one 32 x 32 output block per wave, as k_mix_mid_fwd_fast), whose gated outputs T2g stay in LDS for
//   * the output product computed TRANSPOSED, O^T[d][row] = [W; b]^T T2g^T, wave w owning the d-blocks of its own quarter: the accumulator
//     of a d-block then holds O[row = lane & 31][32 b + 8 q + 4 h + i] in register 4 q + i -- exactly xa[b][q][i] of the next layer once
//     multiplied by x (loaded in the same layout).  O_l (and x_{l+1} when asked for) leave as 16-byte pieces per lane.
//   * the scoring head of the last layer is a row dot in that layout (no y tensor).
// MFMA work per block and layer: 512 + 32 + 520 instructions per wave = 33 us at 2.1 GHz; the weights stream from L2 (every workgroup
// reads the same 1 MB per layer).  Exact fp32 like the GEMM kernels; other summation order (K in four quarters).
#include "dcnmix_tile.hpp"
#include "prof.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TL_ROWS 32
#define TL_NS 128
#define TL_LDT 144
#define TL_LDP 132          // row stride of a partial tile (float4 reads of phase B stay aligned)
#define TL_LDA 129          // H1 tile: odd stride, the per-lane ds_read_b32 of the A fragments (lane = row) is conflict-free
#define TL_LDG 131          // T2g tile [G*H2 | G]: 130 columns, odd stride
#define TL_LDS_FLOATS (4 * TL_ROWS * TL_LDP + 4 * TL_ROWS * 2 + TL_ROWS * TL_LDA + TL_ROWS * 2 + TL_ROWS * TL_LDG + 4 * TL_ROWS)

bool rn_mix_tile_supported(int64_t B, int D, int S, int N, int L, int LDT) {
    return N == 2 && S == 64 && LDT == TL_LDT && (D == 256 || D == 512 || D == 1024) && B > 0 && B % TL_ROWS == 0 && L >= 1 && L <= RN_TILE_MAX_L;
}
// per layer: P1, P2 (forward) and room for the two packs of a backward kernel in the same layouts
size_t rn_mix_tile_pack_bytes(int D, int S, int N, int L, int LDT) {
    if (!rn_mix_tile_supported(TL_ROWS, D, S, N, L, LDT)) return 0;
    return rn_align((size_t)L * 4 * D * TL_NS * sizeof(float));
}

// P1_l[kg][col][i] = U_l[n = col / 64][d = 4 kg + i][s = col % 64]         (D / 4 x 128 float4)
// P2_l[g][h][d][i] = W_l[t = 8 g + 4 h + i][d],  W_l as (N S, D)            (16 x 2 x D float4)
__global__ void __launch_bounds__(256) k_tile_pack(const RnTileFwd p) {
    const int D = p.D, L = p.L;
    const int64_t per = (int64_t)D * TL_NS, total = (int64_t)L * 2 * per;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int l = (int)(e / (2 * per));
        const int64_t j = e - (int64_t)l * 2 * per;
        float* dst = p.packs + (int64_t)l * 4 * per + j;
        if (j < per) {
            const int i = (int)(j & 3), col = (int)((j >> 2) & 127), kg = (int)(j >> 9);
            *dst = p.U[l][((int64_t)(col >> 6) * D + (4 * kg + i)) * 64 + (col & 63)];
        } else {
            const int64_t k = j - per;
            const int i = (int)(k & 3), d = (int)((k >> 2) % D), gh = (int)((k >> 2) / D);
            *dst = p.W[l][(int64_t)(4 * gh + i) * D + d];        // t = 8 g + 4 h + i = 4 (2 g + h) + i
        }
    }
}

// diagnostic build (tools/build_variant.py tiletrace -DRN_TILE_TRACE, tools/tile_trace.py): wall-clock stamps (100 MHz) of workgroup 0, wave 0
#ifdef RN_TILE_TRACE
__device__ long long g_tile_trace[64];
#define TL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tile_trace[(i)] = wall_clock64(); } while (0)
extern "C" int recnow_debug_tile_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_trace), sizeof(long long) * 64); }
#else
#define TL_STAMP(i) do { } while (0)
#endif
#define TL_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define TL_SB() __builtin_amdgcn_sched_barrier(0)

template <int NB, bool TANH>
__global__ void __launch_bounds__(256, 1) k_mix_tile_fwd(const RnTileFwd p) {
    constexpr int D = 128 * NB;
    static_assert(NB % 2 == 0, "d-blocks are walked in pairs");
    extern __shared__ float lds[];
    float* Ps = lds;                                  // [4][32][LDP] partial T1 tiles of the four waves
    float* Pg = Ps + 4 * TL_ROWS * TL_LDP;            // [4][32][2]   partial gate logits
    float* Hs = Pg + 4 * TL_ROWS * 2;                 // [32][LDA]    H1 tile
    float* Gs = Hs + TL_ROWS * TL_LDA;                // [32][2]      gates
    float* G2 = Gs + TL_ROWS * 2;                     // [32][LDG]    T2g tile
    float* Sc = G2 + TL_ROWS * TL_LDG;                // [4][32]      score partials
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int act_inner = TANH ? RECNOW_ACT_TANH : p.act_inner, act_outer = TANH ? RECNOW_ACT_TANH : p.act_outer;
    const int64_t ntiles = p.B / TL_ROWS;
    TL_STAMP(0);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TL_ROWS;
        const int64_t xoff = (r0 + c) * D + 4 * h + w * NB * 32;        // this lane's row, first column of its quarter (+ 32 b + 8 q)
        rn_f4 xa[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) xa[b][q] = *reinterpret_cast<const rn_f4*>(p.x + xoff + 32 * b + 8 * q);
        for (int l = 0; l < p.L; ++l) {
            const rn_f4* __restrict__ P1 = reinterpret_cast<const rn_f4*>(p.packs + (int64_t)l * 4 * D * TL_NS);
            const rn_f4* __restrict__ P2 = P1 + D * 32;
            const rn_f4* __restrict__ Kg4 = reinterpret_cast<const rn_f4*>(p.Kg[l]);
            // ---- GEMM1 over this wave's quarter of K: partial T1 (4 column blocks) + partial gate logits
            f32x16 acc[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
            float g0 = 0.f, g1 = 0.f;
            {
                constexpr int NST = NB * 4;
                rn_f4 wr[3][4], gr[3][2];
                const int kg0 = w * NB * 8 + h;
                auto ld1 = [&](int s, int slot) {
                    const int kg = kg0 + (s >> 2) * 8 + 2 * (s & 3);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) wr[slot][cb] = P1[kg * 128 + cb * 32 + c];
                    gr[slot][0] = Kg4[kg * 2];
                    gr[slot][1] = Kg4[kg * 2 + 1];
                };
                ld1(0, 0);
                ld1(1, 1);
#pragma unroll
                for (int s = 0; s < NST; ++s) {
                    if (s + 2 < NST) ld1(s + 2, (s + 2) % 3);
                    TL_SB();
                    const int b = s >> 2, q = s & 3, slot = s % 3;
                    const rn_f4 a = xa[b][q];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[cb] = TL_MFMA(a[i], wr[slot][cb][i], acc[cb]);
                    g0 += a.x * gr[slot][0].x + a.y * gr[slot][0].z + a.z * gr[slot][1].x + a.w * gr[slot][1].z;
                    g1 += a.x * gr[slot][0].y + a.y * gr[slot][0].w + a.z * gr[slot][1].y + a.w * gr[slot][1].w;
                    TL_SB();
                }
            }
            TL_STAMP(2 + 6 * l);
            // B fragments of the sub-space stage (this wave's output block (n, cb)): requested now, needed two barriers on
            const int en = w >> 1, ecb = w & 1;
            float vb[32];
            {
                const float* __restrict__ Vp = p.V[l] + en * 4096 + h * 64 + ecb * 32 + c;
#pragma unroll
                for (int st = 0; st < 32; ++st) vb[st] = Vp[st * 128];
            }
            {
                float* Pw = Ps + w * TL_ROWS * TL_LDP + 4 * h * TL_LDP + c;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Pw[((r & 3) + 8 * (r >> 2)) * TL_LDP + cb * 32] = acc[cb][r];
                g0 += __shfl_xor(g0, 32, 64);
                g1 += __shfl_xor(g1, 32, 64);
                if (h == 0) {
                    Pg[(w * TL_ROWS + c) * 2] = g0;
                    Pg[(w * TL_ROWS + c) * 2 + 1] = g1;
                }
            }
            __syncthreads();
            TL_STAMP(3 + 6 * l);
            // ---- phase B: T1 = act_inner(sum of the partials) -> global + H1 tile; gate softmax
            {
                const int r = 4 * (tid >> 5) + ((tid >> 3) & 3);
                float* __restrict__ T1g = p.T1[l] + (r0 + r) * TL_LDT;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k4 = ((tid & 7) + 8 * i) * 4;
                    const float* pp = Ps + r * TL_LDP + k4;
                    const rn_f4 a0 = *reinterpret_cast<const rn_f4*>(pp), a1 = *reinterpret_cast<const rn_f4*>(pp + TL_ROWS * TL_LDP),
                                a2 = *reinterpret_cast<const rn_f4*>(pp + 2 * TL_ROWS * TL_LDP), a3 = *reinterpret_cast<const rn_f4*>(pp + 3 * TL_ROWS * TL_LDP);
                    rn_f4 v = (a0 + a1) + (a2 + a3);
                    v.x = rn_act(v.x, act_inner); v.y = rn_act(v.y, act_inner); v.z = rn_act(v.z, act_inner); v.w = rn_act(v.w, act_inner);
                    *reinterpret_cast<rn_f4*>(T1g + k4) = v;
                    float* d = Hs + r * TL_LDA + k4;
                    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
                }
                if (tid < TL_ROWS) {
                    float lg[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        lg[n] = (Pg[tid * 2 + n] + Pg[(TL_ROWS + tid) * 2 + n]) + (Pg[(2 * TL_ROWS + tid) * 2 + n] + Pg[(3 * TL_ROWS + tid) * 2 + n]);
                    p.T1[l][(r0 + tid) * TL_LDT + TL_NS] = lg[0];
                    p.T1[l][(r0 + tid) * TL_LDT + TL_NS + 1] = lg[1];
                    const float mx = lg[0] > lg[1] ? lg[0] : lg[1];
                    const float e0 = expf(lg[0] - mx), e1 = expf(lg[1] - mx), sum = e0 + e1;
                    Gs[tid * 2] = e0 / sum;
                    Gs[tid * 2 + 1] = e1 / sum;
                }
            }
            __syncthreads();
            TL_STAMP(4 + 6 * l);
            // ---- phase C: H2_n = act_outer(H1_n V_n), T2 = [H2 | G | 0], T2g = [G_n H2_n | G | 0]; T2g also into LDS
            {
                f32x16 a2;
#pragma unroll
                for (int r = 0; r < 16; ++r) a2[r] = 0.f;
                const float* ap = Hs + c * TL_LDA + en * 64 + h;
#pragma unroll
                for (int st = 0; st < 32; ++st) a2 = TL_MFMA(ap[2 * st], vb[st], a2);
                const int col = en * 64 + ecb * 32 + c;
                float* __restrict__ T2p = p.T2[l] + r0 * TL_LDT + col;
                float* __restrict__ T2gp = p.T2g[l] + r0 * TL_LDT + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float h2 = rn_act(a2[r], act_outer);
                    const float t = Gs[rr * 2 + en] * h2;
                    T2p[rr * TL_LDT] = h2;
                    T2gp[rr * TL_LDT] = t;
                    G2[rr * TL_LDG + col] = t;
                }
                {   // columns 128 .. 143 of T2 and T2g: [G | 0]; thread = (row, tensor, float4)
                    const int row = tid >> 3, q = tid & 3;
                    rn_f4 g4 = {0.f, 0.f, 0.f, 0.f};
                    if (q == 0) { g4.x = Gs[row * 2]; g4.y = Gs[row * 2 + 1]; }
                    float* dst = ((tid & 4) ? p.T2g[l] : p.T2[l]) + (r0 + row) * TL_LDT + TL_NS + 4 * q;
                    *reinterpret_cast<rn_f4*>(dst) = g4;
                }
                if (tid < 2 * TL_ROWS) G2[(tid >> 1) * TL_LDG + TL_NS + (tid & 1)] = Gs[tid];
            }
            __syncthreads();
            TL_STAMP(5 + 6 * l);
            // ---- phase D: O^T = [W; b]^T T2g^T over this wave's d-blocks, two blocks at a time; x_{l+1} = x * O_l into xa
            {
                float tb[16][4], tg;
                {
                    const float* gp = G2 + c * TL_LDG + 4 * h;
#pragma unroll
                    for (int g = 0; g < 16; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) tb[g][i] = gp[8 * g + i];
                    tg = G2[c * TL_LDG + TL_NS + h];
                }
                const bool last = l == p.L - 1;
                float* __restrict__ Og = p.O[l];
                float* __restrict__ Xg = p.xn[l];
                const float* __restrict__ hv = last ? p.head_w : nullptr;
                const float* __restrict__ bl = p.bias[l] + h * D + w * NB * 32 + c;
                float sp = 0.f;
                constexpr int NST = (NB / 2) * 16;                      // steps: (pair of blocks, g); 8 MFMAs each
                rn_f4 wr[8];                                            // ring of four step slots: 3 steps ahead, two loads per step
                const int d0w = w * NB * 32 + c;
                auto ld2 = [&](int s, int slot) {
                    const int pb = s >> 4, g = s & 15;
                    const int64_t row = (int64_t)(2 * g + h) * D;
                    wr[slot] = P2[row + d0w + (2 * pb) * 32];
                    wr[slot + 1] = P2[row + d0w + (2 * pb + 1) * 32];
                };
#pragma unroll
                for (int s = 0; s < 3; ++s) ld2(s, 2 * s);
#pragma unroll
                for (int pb = 0; pb < NB / 2; ++pb) {
                    rn_f4 x0[2][4];
                    float bz[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) x0[u][q] = *reinterpret_cast<const rn_f4*>(p.x + xoff + 32 * (2 * pb + u) + 8 * q);
                        bz[u] = bl[(2 * pb + u) * 32];
                    }
                    f32x16 o[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[u][r] = 0.f;
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int s = pb * 16 + g;
                        if (s + 3 < NST) ld2(s + 3, (2 * (s + 3)) % 8);
                        TL_SB();
                        const int slot = (2 * s) % 8;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            o[0] = TL_MFMA(wr[slot][i], tb[g][i], o[0]);
                            o[1] = TL_MFMA(wr[slot + 1][i], tb[g][i], o[1]);
                        }
                        TL_SB();
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        o[u] = TL_MFMA(bz[u], tg, o[u]);                // the gate-weighted bias rows (k-pair 128, 129)
                        const int b = 2 * pb + u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const rn_f4 ov = {o[u][4 * q], o[u][4 * q + 1], o[u][4 * q + 2], o[u][4 * q + 3]};
                            if (Og) *reinterpret_cast<rn_f4*>(Og + xoff + 32 * b + 8 * q) = ov;
                            const rn_f4 xv = x0[u][q] * ov;
                            xa[b][q] = xv;
                            if (Xg) *reinterpret_cast<rn_f4*>(Xg + xoff + 32 * b + 8 * q) = xv;
                            if (hv) {
                                const rn_f4 h4 = *reinterpret_cast<const rn_f4*>(hv + 4 * h + w * NB * 32 + 32 * b + 8 * q);
                                const rn_f4 t = xv * h4;
                                sp += (t.x + t.y) + (t.z + t.w);
                            }
                        }
                    }
                }
                if (hv) {                           // scoring head: join the two k-halves of a row, then the four waves in a fixed order
                    sp += __shfl_xor(sp, 32, 64);
                    if (h == 0) Sc[w * TL_ROWS + c] = sp;
                    __syncthreads();
                    if (tid < TL_ROWS)
                        p.scores[r0 + tid] = (p.head_b ? p.head_b[0] : 0.f) + ((Sc[tid] + Sc[TL_ROWS + tid]) + (Sc[2 * TL_ROWS + tid] + Sc[3 * TL_ROWS + tid]));
                }
            }
            TL_STAMP(6 + 6 * l);
        }
    }
}

template <int NB>
static int tile_launch(const RnTileFwd& p, int grid, hipStream_t st) {
    const size_t lds = (size_t)TL_LDS_FLOATS * sizeof(float);
    const bool tanh2 = p.act_inner == RECNOW_ACT_TANH && p.act_outer == RECNOW_ACT_TANH;
    static bool allowed[2] = {false, false};
    if (!allowed[tanh2 ? 1 : 0]) {
        hipError_t e = tanh2 ? hipFuncSetAttribute((const void*)k_mix_tile_fwd<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                             : hipFuncSetAttribute((const void*)k_mix_tile_fwd<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        allowed[tanh2 ? 1 : 0] = true;
    }
    if (tanh2) hipLaunchKernelGGL((k_mix_tile_fwd<NB, true>), grid, 256, lds, st, p);
    else hipLaunchKernelGGL((k_mix_tile_fwd<NB, false>), grid, 256, lds, st, p);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

int rn_mix_tile_fwd(const RnTileFwd& p, hipStream_t st) {
    if (!rn_mix_tile_supported(p.B, p.D, 64, 2, p.L, TL_LDT) || !p.packs || !p.x) return RECNOW_EUNSUPPORTED;
    if (p.head_w && !p.scores) return RECNOW_EINVAL;
    {
        const int64_t total = (int64_t)p.L * 2 * p.D * TL_NS;
        int g = rn_cdiv(total, 256 * 4);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(k_tile_pack, g, 256, 0, st, p);
        RN_LAUNCH_CHECK();
    }
    const int64_t tiles = p.B / TL_ROWS;
    const int grid = (int)(tiles < 256 ? tiles : 256);
    // algorithmic work of the launch: 3 products per layer (2 B D (NS + N) each for the two large ones) and the saved activations + O
    RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_MIX_TILE_FWD, (double)p.L * (4.0 * p.B * p.D * 130 + 4.0 * p.B * 2 * 64 * 64),
                                                    (double)p.L * (12.0 * p.B * TL_LDT + 8.0 * p.B * p.D), st)
                                    : nullptr;
    int rc;
    switch (p.D) {
        case 256: rc = tile_launch<2>(p, grid, st); break;
        case 512: rc = tile_launch<4>(p, grid, st); break;
        default: rc = tile_launch<8>(p, grid, st); break;
    }
    rn_prof_end(pr, st);
    return rc;
}
