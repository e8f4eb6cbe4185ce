// Row-block persistent forward and backward of DCNMixLayer (+ the folded scoring head) for SMALL batches: the per-rank shards of the metric's
// 4- and 8-GPU rows (batches up to 16 384 rows; dcnmix.hip `mix_tile_on`).  /root/reference/rec_now/layers/dcn_mix_layer.py:123-150 and its
// backward, every layer, in ONE launch per direction.  Design notes, measurements and what the stamps / the ISA showed: DESIGN.md 5i.
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
// reads the same 1 MB per layer): operands three steps (48 MFMAs) ahead in register rings, addressed as uniform base + opaque per-lane offset.
// Exact fp32 like the GEMM kernels; other summation order (K in four quarters).  One wave per SIMD and 512 registers per lane: nothing runs
// beside these launches.  Rules this file follows because their absence was measured: no load under a run-time condition,
// no per-lane choice between kernel-argument array elements, tiles leave for memory through LDS as coalesced 16-byte pieces.
#include <string.h>
#include <atomic>
#include "dcnmix_tile.hpp"
#include "prof.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TL_ROWS 32
#define TL_LDT 144
#define TL_LDP 132          // row stride of a partial tile (float4 reads of phase B stay aligned)
#define TL_LDA 129          // H1 tile: odd stride, the per-lane ds_read_b32 of the A fragments (lane = row) is conflict-free
#define TL_LDG 132          // T2g / dA tile: 130 columns used; float4 rows (the tile leaves for memory as coalesced 16-byte pieces)
#define TL_LDS_FLOATS(D) (4 * TL_ROWS * TL_LDP + 4 * TL_ROWS * 2 + TL_ROWS * TL_LDA + TL_ROWS * 2 + 2 * TL_ROWS * TL_LDG + 4 * TL_ROWS + (D))

bool rn_mix_tile_supported(int64_t B, int D, int S, int N, int L, int LDT) {
    return N == 2 && S == 64 && LDT == TL_LDT && (D == 256 || D == 512 || D == 1024) && B > 0 && B % TL_ROWS == 0 && L >= 1 && L <= RN_TILE_MAX_L;
}
size_t rn_mix_tile_pack_bytes(int D, int S, int N, int L, int LDT) {
    if (!rn_mix_tile_supported(TL_ROWS, D, S, N, L, LDT)) return 0;
    return rn_align((size_t)L * TL_PACK_FLOATS(D) * sizeof(float));
}

// (layouts of the packs: tl_pack_range, dcnmix_tile.hpp)
__global__ void __launch_bounds__(256) k_tile_pack(const RnTileFwd p) {
    tl_pack_range(p, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}

// diagnostic build (tools/build_variant.py tiletrace -DRN_TILE_TRACE, tools/tile_trace.py): wall-clock stamps (100 MHz) of workgroup 0, wave 0
#ifdef RN_TILE_TRACE
__device__ long long g_tile_trace[64];
#define TL_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_tile_trace[(i)] = wall_clock64(); } while (0)
extern "C" int recnow_debug_tile_trace(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_trace), sizeof(long long) * 64); }
#else
#define TL_STAMP(i) do { } while (0)
#endif
// uniform base pointer (SGPRs) + per-lane 32-bit byte offset kept opaque, so that the compiler neither folds the step's constant into a
// new 64-bit address per load nor parks one address per step in registers
__device__ __forceinline__ rn_f4 tl_ld4(const float* base, unsigned off) {
    return *reinterpret_cast<const rn_f4*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ float tl_ld1(const float* base, unsigned off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off); }
__device__ __forceinline__ void tl_st4(float* base, unsigned off, rn_f4 v) { *reinterpret_cast<rn_f4*>(reinterpret_cast<char*>(base) + off) = v; }
#define TL_OPAQUE(v) asm volatile("" : "+v"(v))
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
    float* T2s = G2 + TL_ROWS * TL_LDG;               // [32][LDG]    T2 tile (H2), on its way to memory
    float* Sc = T2s + TL_ROWS * TL_LDG;               // [4][32]      score partials
    float* Hv = Sc + 4 * TL_ROWS;                     // [D]          the scoring head's vector (zeros without a head)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int act_inner = TANH ? RECNOW_ACT_TANH : p.act_inner, act_outer = TANH ? RECNOW_ACT_TANH : p.act_outer;
    const int64_t ntiles = p.B / TL_ROWS;
    for (int i = tid; i < D; i += 256) Hv[i] = p.head_w ? p.head_w[i] : 0.f;
    __syncthreads();
    TL_STAMP(0);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TL_ROWS;
        const int64_t xoff = (r0 + c) * D + 4 * h + w * NB * 32;        // this lane's row, first column of its quarter (+ 32 b + 8 q)
        // Operand rings of the two product loops.  Their first steps are requested one phase EARLY -- GEMM1's at the end of the layer before
        // (for the last layer: layer 0 again, what the workgroup's next block starts with), the output product's before the sub-space stage --
        // so that no loop starts with a load latency (every one of them cost ~3 us).  No load sits under a condition.
        constexpr int NST1 = NB * 4, PF1 = 3, NSL1 = PF1 + 1;       // GEMM1: operands PF1 steps (PF1 x 16 MFMAs) ahead in a ring of NSL1 register sets
        rn_f4 wr1[NSL1][4], gr1[NSL1][2];
        unsigned vP = (unsigned)(((w * NB * 8 + h) * 128 + c) * 16), vG = (unsigned)((w * NB * 8 + h) * 32);
        auto ld1 = [&](const float* P1f, const float* Kgf, int s, int slot) {
            const int ks = (s >> 2) * 8 + 2 * (s & 3);                  // static part of kg
            TL_OPAQUE(vP); TL_OPAQUE(vG);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) wr1[slot][cb] = tl_ld4(P1f, vP + (unsigned)(ks * 128 * 16 + cb * 512));
            gr1[slot][0] = tl_ld4(Kgf, vG + (unsigned)(ks * 32));
            gr1[slot][1] = tl_ld4(Kgf, vG + (unsigned)(ks * 32 + 16));
        };
#pragma unroll
        for (int s = 0; s < PF1; ++s) ld1(p.packs, p.Kg[0], s, s);
        rn_f4 xa[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) xa[b][q] = *reinterpret_cast<const rn_f4*>(p.x + xoff + 32 * b + 8 * q);
        for (int l = 0; l < p.L; ++l) {
            const rn_f4* __restrict__ P1 = reinterpret_cast<const rn_f4*>(p.packs + (int64_t)l * TL_PACK_FLOATS(D));
            const rn_f4* __restrict__ P2 = P1 + D * 32;
            const int ln = l + 1 < p.L ? l + 1 : 0;                     // the layer whose GEMM1 comes next in this workgroup
            // ---- GEMM1 over this wave's quarter of K: partial T1 (4 column blocks) + partial gate logits
            f32x16 acc[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
            float g0 = 0.f, g1 = 0.f;
            {
                const float* __restrict__ P1f = reinterpret_cast<const float*>(P1);
                const float* __restrict__ Kgf = p.Kg[l];
#pragma unroll
                for (int s = 0; s < NST1; ++s) {
                    if (s + PF1 < NST1) ld1(P1f, Kgf, s + PF1, (s + PF1) % NSL1);
                    TL_SB();
                    const int b = s >> 2, q = s & 3, slot = s % NSL1;
                    const rn_f4 a = xa[b][q];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[cb] = TL_MFMA(a[i], wr1[slot][cb][i], acc[cb]);
                    g0 += a.x * gr1[slot][0].x + a.y * gr1[slot][0].z + a.z * gr1[slot][1].x + a.w * gr1[slot][1].z;
                    g1 += a.x * gr1[slot][0].y + a.y * gr1[slot][0].w + a.z * gr1[slot][1].y + a.w * gr1[slot][1].w;
                    TL_SB();
                }
            }
            TL_STAMP(2 + 6 * l);
            // the output product's ring: its first three steps are requested here, two barriers ahead of the loop
            constexpr int NST2 = (NB / 2) * 16;                     // steps: (pair of blocks, g); 8 MFMAs each
            rn_f4 wr2[8];                                           // ring of four step slots: 3 steps ahead, two loads per step
            const int d0w = w * NB * 32 + c;
            auto ld2 = [&](int s, int slot) {
                const int pb = s >> 4, g = s & 15;
                const int64_t row = (int64_t)(2 * g + h) * D;
                wr2[slot] = P2[row + d0w + (2 * pb) * 32];
                wr2[slot + 1] = P2[row + d0w + (2 * pb + 1) * 32];
            };
#pragma unroll
            for (int s = 0; s < 3; ++s) ld2(s, 2 * s);
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
#ifdef RN_TILE_TRACE
                if (l == 1) { if (a2[0] == 12345.f) Sc[0] = 1.f; TL_STAMP(20); }       // (the compare makes the stamp wait for the chain)
#endif
                const int col = en * 64 + ecb * 32 + c;
                // the epilogue's LDS reads go out together and the 16 activations are independent chains (one wave per SIMD: a wait per
                // element cost 0.4 us each); T2 / T2g leave through LDS tiles as coalesced 16-byte pieces behind the barrier
                float gsel[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) gsel[r] = Gs[((r & 3) + 8 * (r >> 2) + 4 * h) * 2 + en];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float h2 = rn_act(a2[r], act_outer);
                    T2s[rr * TL_LDG + col] = h2;
                    G2[rr * TL_LDG + col] = gsel[r] * h2;
                }
                if (l == 1) TL_STAMP(21);
                {   // columns 128 .. 143 of T2 and T2g: [G | 0]; thread = (row, tensor, float4)
                    const int row = tid >> 3, q = tid & 3;
                    rn_f4 g4 = {0.f, 0.f, 0.f, 0.f};
                    if (q == 0) { g4.x = Gs[row * 2]; g4.y = Gs[row * 2 + 1]; }
                    float* const t2u = p.T2[l];            // uniform pointers first: a per-lane choice between two kernel-argument array
                    float* const t2gu = p.T2g[l];          // elements became a VECTOR load of the pointer with a full vmcnt(0) behind it
                    float* dst = ((tid & 4) ? t2gu : t2u) + (r0 + row) * TL_LDT + TL_NS + 4 * q;
                    *reinterpret_cast<rn_f4*>(dst) = g4;
                }
                if (tid < 2 * TL_ROWS) G2[(tid >> 1) * TL_LDG + TL_NS + (tid & 1)] = Gs[tid];
            }
            if (l == 1) TL_STAMP(22);
            __syncthreads();
            TL_STAMP(5 + 6 * l);
            {   // T2 and T2g rows of the tile: thread = (row, 16-byte piece), eight threads per 128 bytes
                const int row = tid >> 3;
                float* __restrict__ T2r = p.T2[l] + (r0 + row) * TL_LDT;
                float* __restrict__ T2gr = p.T2g[l] + (r0 + row) * TL_LDT;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k4 = ((tid & 7) + 8 * i) * 4;
                    *reinterpret_cast<rn_f4*>(T2r + k4) = *reinterpret_cast<const rn_f4*>(T2s + row * TL_LDG + k4);
                    *reinterpret_cast<rn_f4*>(T2gr + k4) = *reinterpret_cast<const rn_f4*>(G2 + row * TL_LDG + k4);
                }
            }
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
                const bool use_head = last && p.head_w != nullptr;
                // the head vector is read from LDS: a global load under the `last layer` condition had a wait for EVERYTHING in flight behind it
                const float* hvp = Hv + 4 * h + w * NB * 32;
                const float* __restrict__ bl = p.bias[l] + h * D + w * NB * 32 + c;
                float sp = 0.f;
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
                        if (s + 3 < NST2) ld2(s + 3, (2 * (s + 3)) % 8);
                        if (pb == NB / 2 - 1 && g >= 16 - PF1)          // the last steps: GEMM1 of the next layer (or of the next block) starts its ring
                            ld1(p.packs + (int64_t)ln * TL_PACK_FLOATS(D), p.Kg[ln], g - (16 - PF1), g - (16 - PF1));
                        TL_SB();
                        const int slot = (2 * s) % 8;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            o[0] = TL_MFMA(wr2[slot][i], tb[g][i], o[0]);
                            o[1] = TL_MFMA(wr2[slot + 1][i], tb[g][i], o[1]);
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
                            const rn_f4 t = xv * *reinterpret_cast<const rn_f4*>(hvp + 32 * b + 8 * q);
                            sp += (t.x + t.y) + (t.z + t.w);
                        }
                    }
                }
                if (use_head) {                     // scoring head: join the two k-halves of a row, then the four waves in a fixed order
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
    const size_t lds = (size_t)TL_LDS_FLOATS(128 * NB) * sizeof(float);
    const bool tanh2 = p.act_inner == RECNOW_ACT_TANH && p.act_outer == RECNOW_ACT_TANH;
    // more than 64 KB of dynamic LDS: raised once per DEVICE and kernel (the attribute belongs to the function on the current device: a process
    // that drives a second GPU must raise it there too); std::atomic<bool>: concurrent host threads may race to raise it, never to skip it
    static std::atomic<bool> raised[2][64];
    int dev = 0;
    RN_HIP(hipGetDevice(&dev));
    const int which = tanh2 ? 1 : 0;
    if (dev < 0 || dev >= 64 || !raised[which][dev].load(std::memory_order_acquire)) {
        hipError_t e = tanh2 ? hipFuncSetAttribute((const void*)k_mix_tile_fwd<NB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                             : hipFuncSetAttribute((const void*)k_mix_tile_fwd<NB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 64) raised[which][dev].store(true, std::memory_order_release);
    }
    if (tanh2) hipLaunchKernelGGL((k_mix_tile_fwd<NB, true>), grid, 256, lds, st, p);
    else hipLaunchKernelGGL((k_mix_tile_fwd<NB, false>), grid, 256, lds, st, p);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

static int tile_pack_launch(const RnTileFwd& p, hipStream_t st) {
    const int64_t total = (int64_t)p.L * TL_PACK_FLOATS(p.D);
    int g = rn_cdiv(total, 256 * 4);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_tile_pack, g, 256, 0, st, p);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// the packs alone: a row-block backward behind a forward that did not run the row-block kernels (dcnmix.hip, route stamps)
int rn_mix_tile_pack(const float* const* U, const float* const* Kg, const float* const* V, const float* const* W, const float* const* bias, int D, int L,
                     float* packs, hipStream_t st) {
    if (L < 1 || L > RN_TILE_MAX_L || !packs || !(D == 256 || D == 512 || D == 1024)) return RECNOW_EUNSUPPORTED;
    RnTileFwd p;
    memset(&p, 0, sizeof(p));
    p.D = D; p.L = L; p.packs = packs;
    for (int l = 0; l < L; ++l) { p.U[l] = U[l]; p.Kg[l] = Kg[l]; p.V[l] = V[l]; p.W[l] = W[l]; p.bias[l] = bias[l]; }
    return tile_pack_launch(p, st);
}

int rn_mix_tile_fwd(const RnTileFwd& p, hipStream_t st) {
    if (!rn_mix_tile_supported(p.B, p.D, 64, 2, p.L, TL_LDT) || !p.packs || !p.x) return RECNOW_EUNSUPPORTED;
    if (p.head_w && !p.scores) return RECNOW_EINVAL;
    if (!p.packed) {
        int rcp;
        if ((rcp = tile_pack_launch(p, st))) return rcp;
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

// ---- backward: the data-gradient chain of layers l_hi .. l_lo for a block of 32 rows, one launch -----------------------------------
// Mirror of the forward.  The gradient g_{l+1} w.r.t. the layer's output sits in the A-fragment layout (ga, as xa above):
//   * dT2g = (g_{l+1} * x) [W; b]^T: every wave contracts its quarter of D; x, O_l and the old dx come in block-wise as whole row pieces through
//     wave-private LDS tiles, and dx (+)= g_{l+1} * O_l leaves the same way from the same loop; partial tiles meet in LDS;
//   * sub-space backward on the summed tile (k_mix_mid_bwd_fast's arithmetic: dC, <dT2g_n, H2_n>, gate, dA = (dC V^T) act'(H1), dV += H1^T dC);
//     dT1 = [dA | dlogits | 0] goes to memory (the weight-gradient product dU reads it) and stays in LDS for
//   * g_l^T = [U | K] dT1^T, transposed like the forward's output product: the accumulators ARE the next layer's ga; g_l is stored for the
//     weight-gradient product of layer l - 1 (l >= 1) or added to dx (l = 0).
// dV partial sums: one (N, S, S) block per workgroup and layer (a workgroup with several row blocks adds to its own block), summed by
// rn_layer_end_reduce in a fixed order.
#define TLB_LDS_FLOATS (4 * TL_ROWS * TL_LDP + 4 * TL_ROWS * 2 + 2 * TL_ROWS * TL_LDA + TL_ROWS * TL_LDG + 2 * TL_ROWS * 2)


// The layer gradient ga (128 registers per lane) is kept in ACCUMULATION registers and copied out four at a time where a step needs it: left to
// the allocator it stayed in the 256 architectural registers, and the rows of the next block -- in flight from memory -- were what got parked
// in AGPRs, behind a full s_waitcnt vmcnt(0) right after their loads (one memory round trip per block of the dT2g loop).
__device__ __forceinline__ float tl_a2v(float a) { float v; asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }
__device__ __forceinline__ float tl_v2a(float v) { float a; asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
struct TlAcc4 { float x, y, z, w; };
__device__ __forceinline__ TlAcc4 tl_park(rn_f4 v) { return TlAcc4{tl_v2a(v.x), tl_v2a(v.y), tl_v2a(v.z), tl_v2a(v.w)}; }
__device__ __forceinline__ rn_f4 tl_fetch(const TlAcc4& a) { return rn_f4{tl_a2v(a.x), tl_a2v(a.y), tl_a2v(a.z), tl_a2v(a.w)}; }

template <int NB, bool TANH, bool DX>
__global__ void __launch_bounds__(256, 1) k_mix_tile_bwd(const RnTileBwd p) {
    constexpr int D = 128 * NB;
    static_assert(NB % 2 == 0, "d-blocks are walked in pairs");
    extern __shared__ float lds[];
    float* Ps = lds;                                  // [4][32][LDP] partial dT2g tiles
    float* Pg = Ps + 4 * TL_ROWS * TL_LDP;            // [4][32][2]   partial gate columns of dT2g
    float* Cs = Pg + 4 * TL_ROWS * 2;                 // [32][LDA]    dC tile
    float* Hs = Cs + TL_ROWS * TL_LDA;                // [32][LDA]    H1 tile
    float* E2 = Hs + TL_ROWS * TL_LDA;                // [32][LDG]    dA tile (the main columns of dT1)
    float* Ps2 = E2 + TL_ROWS * TL_LDG;               // [32][2]      <dT2g_n, H2_n>
    float* Ds = Ps2 + TL_ROWS * 2;                    // [32][2]      dlogits
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 31, h = lane >> 5;
    const int en = w >> 1, ecb = w & 1;
    const int act_inner = TANH ? RECNOW_ACT_TANH : p.act_inner, act_outer = TANH ? RECNOW_ACT_TANH : p.act_outer;
    const int64_t ntiles = p.B / TL_ROWS;
    // per-lane byte offsets (tile-invariant; the tile's first row goes into the uniform base pointers)
    unsigned vX = (unsigned)((c * D + 4 * h + w * NB * 32) * 4);              // this lane's row of a (B, D) tensor, first column of its quarter
    unsigned vP = (unsigned)(((w * NB * 8 + h) * 128 + c) * 16);              // P3: float4 (kg, t)
    unsigned vB = (unsigned)((w * NB * 8 + h) * 16);                          // rows d0 = 4 kg of a (N, D) tensor
    unsigned vQ = (unsigned)((h * D + w * NB * 32 + c) * 16);                 // P4: float4 (g, h, d)
    unsigned vK = (unsigned)(((w * NB * 32 + c) * 2 + h) * 4);                // Kg[d][h]
    const int tr = 4 * (tid >> 5) + ((tid >> 3) & 3);                         // phase c: this thread's row of the tile
    unsigned vT = (unsigned)((tr * TL_LDT + (tid & 7) * 4) * 4);
    TL_STAMP(32);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * TL_ROWS;
        const bool first_tile = tile == (int64_t)blockIdx.x;
        const float* __restrict__ xt = p.x + r0 * D;
        TlAcc4 ga[NB][4];
        if (p.gin) {
            const float* __restrict__ gt = p.gin + r0 * D;
#pragma unroll
            for (int b = 0; b < NB; b += 2) {          // eight loads in flight, then parked (one at a time each load was a round trip)
                rn_f4 t[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[u][q] = tl_ld4(gt, vX + (32 * (b + u) + 8 * q) * 4);
                TL_SB();
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ga[b + u][q] = tl_park(t[u][q]);
                TL_SB();
            }
        } else {
            const float dsr = p.ds[r0 + c];
            const float* __restrict__ hvp = p.head_w + 4 * h + w * NB * 32;
#pragma unroll
            for (int b = 0; b < NB; b += 2) {
                rn_f4 t[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t[u][q] = *reinterpret_cast<const rn_f4*>(hvp + 32 * (b + u) + 8 * q);
                TL_SB();
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ga[b + u][q] = tl_park(t[u][q] * dsr);
                TL_SB();
            }
        }
        for (int l = p.l_hi; l >= p.l_lo; --l) {
            const float* __restrict__ PL = p.packs + (int64_t)l * TL_PACK_FLOATS(D);
            const float* __restrict__ P3 = PL + 2 * D * TL_NS;
            const float* __restrict__ P4 = PL + 3 * D * TL_NS;
            const float* __restrict__ VT = PL + 4 * D * TL_NS;
            const float* __restrict__ bl = p.bias[l];
            const float* __restrict__ Ot = DX ? p.O[l] + r0 * D : nullptr;
            float* __restrict__ dxt = DX ? p.dx + r0 * D : nullptr;
            const bool keep_dx = l != p.L - 1;          // the top layer writes dx, the layers below add to it
            // ---- dT2g partial over this wave's quarter of D; dx (+)= g * O_l on the way
            f32x16 acc[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
            float g0 = 0.f, g1 = 0.f;
            {
                // The (B, D) streams of this loop -- x, O_l, the old dx in, the new dx out -- move BLOCK-wise as whole 128-byte row pieces (a wave
                // instruction = 8 rows x 128 B = 8 lines; in the fragment layout it was 32 rows x 32 B = 32 line requests per instruction and
                // the loop was bound by the address unit: 46 us per layer against 16 for the same loop without the streams) and change layout
                // in a wave-private LDS tile: written as rows, read back as this lane's fragment pieces.  The tiles live in the wave's own
                // partial-tile region, free until the loop's end; LDS operations of one wave complete in order, so no barrier is involved.
                constexpr int NST = NB * 4, PF = 2, NSL = PF + 1;       // weights PF steps ahead in a ring of NSL register sets (a third step ahead spills)
                constexpr int SLD = 36, STILE = TL_ROWS * SLD;         // staging tile: 32 rows x 32 columns, row stride 36 (conflict-free both ways)
                static_assert(3 * STILE <= TL_ROWS * TL_LDP, "three staging tiles fit the wave's partial-tile region");
                float* Sx = Ps + w * TL_ROWS * TL_LDP;
                float* So = Sx + STILE;
                float* Sd = So + STILE;
                const int lC = (lane >> 3) * SLD + (lane & 7) * 4;      // row layout: lane = (row % 8, 16-byte piece), instruction j = rows 8 j ..
                const int lA = c * SLD + 4 * h;                         // fragment layout: this lane's row, pieces 8 q + 4 h
                unsigned vC = (unsigned)(((lane >> 3) * D + w * NB * 32 + (lane & 7) * 4) * 4);
                rn_f4 wr[NSL][4], br[NSL][2];
                rn_f4 px[4], po[4], pd[4];                              // block b + 1 on its way in (row layout)
                auto ld1 = [&](int s, int slot) {
                    const int ks = (s >> 2) * 8 + 2 * (s & 3);                  // static part of kg
                    TL_OPAQUE(vP); TL_OPAQUE(vB);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) wr[slot][cb] = tl_ld4(P3, vP + (unsigned)(ks * 128 * 16 + cb * 512));
                    br[slot][0] = tl_ld4(bl, vB + (unsigned)(ks * 16));
                    br[slot][1] = tl_ld4(bl + D, vB + (unsigned)(ks * 16));
                };
                auto ldblk = [&](int b) {
                    TL_OPAQUE(vC);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned so = (unsigned)((8 * j * D + 32 * b) * 4);
                        px[j] = tl_ld4(xt, vC + so);
                        if (DX) {       // no load under a run-time condition: the top layer reads the old dx too and drops it
                            po[j] = tl_ld4(Ot, vC + so);
                            pd[j] = tl_ld4(dxt, vC + so);
                        }
                    }
                };
                ldblk(0);
#pragma unroll
                for (int s = 0; s < PF; ++s) ld1(s, s);
                rn_f4 fx[2], fo[2], fd[2];                              // fragment pieces of step s (slot s & 1) and s + 1
                auto rdfrag = [&](int q, int slot) {
                    fx[slot] = *reinterpret_cast<const rn_f4*>(Sx + lA + 8 * q);
                    if (DX) {
                        fo[slot] = *reinterpret_cast<const rn_f4*>(So + lA + 8 * q);
                        fd[slot] = *reinterpret_cast<const rn_f4*>(Sd + lA + 8 * q);
                    }
                };
#pragma unroll
                for (int s = 0; s < NST; ++s) {
                    const int b = s >> 2, q = s & 3, slot = s % NSL;
                    if (q == 0) {           // block b: rows into the tiles, block b + 1 requested, the first two steps' pieces read back
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            *reinterpret_cast<rn_f4*>(Sx + lC + 8 * j * SLD) = px[j];
                            if (DX) {
                                *reinterpret_cast<rn_f4*>(So + lC + 8 * j * SLD) = po[j];
                                *reinterpret_cast<rn_f4*>(Sd + lC + 8 * j * SLD) = pd[j];
                            }
                        }
                        if (b + 1 < NB) ldblk(b + 1);
                        rdfrag(0, s & 1);
                    }
                    if (s + PF < NST) ld1(s + PF, (s + PF) % NSL);
                    TL_SB();
                    const rn_f4 gv = tl_fetch(ga[b][q]);
                    if (DX) {
                        const rn_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
                        const rn_f4 dv = gv * fo[s & 1] + (keep_dx ? fd[s & 1] : zero4);      // a select: the dropped words may be anything
                        *reinterpret_cast<rn_f4*>(Sd + lA + 8 * q) = dv;
                    }
                    const rn_f4 a = gv * fx[s & 1];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc[cb] = TL_MFMA(a[i], wr[slot][cb][i], acc[cb]);
                    g0 += (a.x * br[slot][0].x + a.y * br[slot][0].y) + (a.z * br[slot][0].z + a.w * br[slot][0].w);
                    g1 += (a.x * br[slot][1].x + a.y * br[slot][1].y) + (a.z * br[slot][1].z + a.w * br[slot][1].w);
                    // the next step's pieces are read BEHIND this step's MFMAs: LDS counts complete in order, so a read issued in front of them
                    // stood between this step's own pieces and its products (a wait of one LDS latency per step, one wave per SIMD)
                    TL_SB();
                    if (q < 3) rdfrag(q + 1, (s + 1) & 1);
                    if (DX && q == 3) {     // the block's new dx: back to rows, out as whole 128-byte pieces
                        TL_OPAQUE(vC);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            tl_st4(dxt, vC + (unsigned)((8 * j * D + 32 * b) * 4), *reinterpret_cast<const rn_f4*>(Sd + lC + 8 * j * SLD));
                    }
                    TL_SB();
                }
            }
            TL_STAMP(34 + 6 * l);
            // operands of the sub-space backward for this thread's part of the tile: requested now, used behind the barrier
            const float* __restrict__ T2t = p.T2[l] + r0 * TL_LDT;
            const float* __restrict__ T1t = p.T1[l] + r0 * TL_LDT;
            rn_f4 t2v[4], t1v[4];
            TL_OPAQUE(vT);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t2v[i] = tl_ld4(T2t, vT + (unsigned)(i * 128));
                t1v[i] = tl_ld4(T1t, vT + (unsigned)(i * 128));
            }
            const float pg0 = T2t[tr * TL_LDT + TL_NS], pg1 = T2t[tr * TL_LDT + TL_NS + 1];
            const float pgg0 = T2t[(tid & 31) * TL_LDT + TL_NS], pgg1 = T2t[(tid & 31) * TL_LDT + TL_NS + 1];      // gate math rows (threads < 32)
            float vtb[32];
            {
                const float* __restrict__ Vp = VT + en * 4096 + h * 64 + ecb * 32 + c;
#pragma unroll
                for (int st = 0; st < 32; ++st) vtb[st] = Vp[st * 128];
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
            TL_STAMP(35 + 6 * l);
            // ---- phase c: dT2g = sum of the partials; dC, H1 tiles, <dT2g_n, H2_n>
            float dgt0 = 0.f, dgt1 = 0.f;
            {
                float ppv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k4 = ((tid & 7) + 8 * i) * 4;
                    const float* pp = Ps + tr * TL_LDP + k4;
                    const rn_f4 a0 = *reinterpret_cast<const rn_f4*>(pp), a1 = *reinterpret_cast<const rn_f4*>(pp + TL_ROWS * TL_LDP),
                                a2 = *reinterpret_cast<const rn_f4*>(pp + 2 * TL_ROWS * TL_LDP), a3 = *reinterpret_cast<const rn_f4*>(pp + 3 * TL_ROWS * TL_LDP);
                    const rn_f4 d = (a0 + a1) + (a2 + a3);
                    const rn_f4 h2 = t2v[i], h1 = t1v[i];
                    const float g = i < 2 ? pg0 : pg1;
                    float* cd = Cs + tr * TL_LDA + k4;
                    cd[0] = g * d.x * rn_act_grad_from_out(h2.x, act_outer);
                    cd[1] = g * d.y * rn_act_grad_from_out(h2.y, act_outer);
                    cd[2] = g * d.z * rn_act_grad_from_out(h2.z, act_outer);
                    cd[3] = g * d.w * rn_act_grad_from_out(h2.w, act_outer);
                    float* hd = Hs + tr * TL_LDA + k4;
                    hd[0] = h1.x; hd[1] = h1.y; hd[2] = h1.z; hd[3] = h1.w;
                    ppv[i] = d.x * h2.x + d.y * h2.y + d.z * h2.z + d.w * h2.w;
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    float pp = ppv[2 * n] + ppv[2 * n + 1];
                    pp += __shfl_xor(pp, 4, 64);
                    pp += __shfl_xor(pp, 2, 64);
                    pp += __shfl_xor(pp, 1, 64);
                    if ((tid & 7) == 0) Ps2[tr * 2 + n] = pp;
                }
                if (tid < TL_ROWS) {
                    dgt0 = (Pg[tid * 2] + Pg[(TL_ROWS + tid) * 2]) + (Pg[(2 * TL_ROWS + tid) * 2] + Pg[(3 * TL_ROWS + tid) * 2]);
                    dgt1 = (Pg[tid * 2 + 1] + Pg[(TL_ROWS + tid) * 2 + 1]) + (Pg[(2 * TL_ROWS + tid) * 2 + 1] + Pg[(3 * TL_ROWS + tid) * 2 + 1]);
                }
            }
            __syncthreads();
            TL_STAMP(36 + 6 * l);
            // ---- phase d: gate backward; dA block -> dT1 + LDS; dV blocks
            float* __restrict__ dT1t = p.dT1[l] + r0 * TL_LDT;
            {
                if (tid < TL_ROWS) {
                    const float dg0 = Ps2[tid * 2] + dgt0, dg1 = Ps2[tid * 2 + 1] + dgt1;
                    const float dot = pgg0 * dg0 + pgg1 * dg1;
                    Ds[tid * 2] = pgg0 * (dg0 - dot);
                    Ds[tid * 2 + 1] = pgg1 * (dg1 - dot);
                }
                f32x16 a2;
#pragma unroll
                for (int r = 0; r < 16; ++r) a2[r] = 0.f;
                const float* ap = Cs + c * TL_LDA + en * 64 + h;
#pragma unroll
                for (int st = 0; st < 32; ++st) a2 = TL_MFMA(ap[2 * st], vtb[st], a2);
                const int col = en * 64 + ecb * 32 + c;
                float hsel[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) hsel[r] = Hs[((r & 3) + 8 * (r >> 2) + 4 * h) * TL_LDA + col];
#pragma unroll
                for (int r = 0; r < 16; ++r)       // dA leaves for memory from the LDS tile, behind the barrier, as coalesced 16-byte pieces
                    E2[((r & 3) + 8 * (r >> 2) + 4 * h) * TL_LDG + col] = a2[r] * rn_act_grad_from_out(hsel[r], act_inner);
                float* __restrict__ dvp = p.dvpart + ((int64_t)l * gridDim.x + blockIdx.x) * (2 * 64 * 64);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int item = w + 4 * j, n = item >> 2, mb = (item >> 1) & 1, cb = item & 1;
                    float* __restrict__ dst = dvp + n * 4096 + (mb * 32 + 4 * h) * 64 + cb * 32 + c;
                    f32x16 av;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {       // the load is unconditional (workspace memory), the first block of a workgroup drops it
                        const float old = dst[((r & 3) + 8 * (r >> 2)) * 64];
                        av[r] = first_tile ? 0.f : old;
                    }
                    const float* hp2 = Hs + h * TL_LDA + n * 64 + mb * 32 + c;
                    const float* cp2 = Cs + h * TL_LDA + n * 64 + cb * 32 + c;
#pragma unroll
                    for (int st = 0; st < TL_ROWS / 2; ++st) av = TL_MFMA(hp2[2 * st * TL_LDA], cp2[2 * st * TL_LDA], av);
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * 64] = av[r];
                }
            }
            __syncthreads();
            TL_STAMP(37 + 6 * l);
            {   // dT1 rows of the tile = [dA | dlogits | 0]: thread = (row, 16-byte piece), eight threads per 128 bytes
                const int row = tid >> 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k4 = ((tid & 7) + 8 * i) * 4;
                    *reinterpret_cast<rn_f4*>(dT1t + row * TL_LDT + k4) = *reinterpret_cast<const rn_f4*>(E2 + row * TL_LDG + k4);
                }
                if ((tid & 4) == 0) {
                    const int q = tid & 3;
                    rn_f4 g4 = {0.f, 0.f, 0.f, 0.f};
                    if (q == 0) { g4.x = Ds[row * 2]; g4.y = Ds[row * 2 + 1]; }
                    *reinterpret_cast<rn_f4*>(dT1t + row * TL_LDT + TL_NS + 4 * q) = g4;
                }
            }
            // ---- g_l^T = [U | K] dT1^T over this wave's d-blocks, two blocks at a time; the accumulators are the next layer's ga
            if (l > 0 || DX) {
                float tb[16][4], tg;
                {
                    const float* gp = E2 + c * TL_LDG + 4 * h;
#pragma unroll
                    for (int g = 0; g < 16; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) tb[g][i] = gp[8 * g + i];
                    tg = Ds[c * 2 + h];
                }
                float* __restrict__ gt = l > 0 ? p.gout[l] + r0 * D : dxt;            // l = 0: g_0 is added to dx
                const bool add_dx = l == 0;
                const float* __restrict__ Kl = p.Kg[l];
                constexpr int NST = (NB / 2) * 16;
                rn_f4 wr[8];                                            // ring of four step slots: 3 steps ahead, two loads per step
                auto ld2 = [&](int s, int slot) {
                    const int pb = s >> 4, g = s & 15;
                    TL_OPAQUE(vQ);
                    wr[slot] = tl_ld4(P4, vQ + (unsigned)((2 * g * D + (2 * pb) * 32) * 16));
                    wr[slot + 1] = tl_ld4(P4, vQ + (unsigned)((2 * g * D + (2 * pb + 1) * 32) * 16));
                };
#pragma unroll
                for (int s = 0; s < 3; ++s) ld2(s, 2 * s);
#pragma unroll
                for (int pb = 0; pb < NB / 2; ++pb) {
                    rn_f4 dxo[2][4];
                    float kz[2];
                    TL_OPAQUE(vK); TL_OPAQUE(vX);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        kz[u] = tl_ld1(Kl, vK + (unsigned)((2 * pb + u) * 32 * 2 * 4));
                        if (DX) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) dxo[u][q] = tl_ld4(dxt, vX + (unsigned)((32 * (2 * pb + u) + 8 * q) * 4));
                        }
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
                        o[u] = TL_MFMA(kz[u], tg, o[u]);                // the gate kernel's columns (k-pair 128, 129: dlogits)
                        const int b = 2 * pb + u;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const rn_f4 gv = {o[u][4 * q], o[u][4 * q + 1], o[u][4 * q + 2], o[u][4 * q + 3]};
                            ga[b][q] = tl_park(gv);
                            TL_OPAQUE(vX);
                            const rn_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
                            tl_st4(gt, vX + (unsigned)((32 * b + 8 * q) * 4), DX ? gv + (add_dx ? dxo[u][q] : zero4) : gv);
                        }
                    }
                }
            }
            TL_STAMP(38 + 6 * l);
        }
    }
}

int rn_mix_tile_bwd_grid(int64_t B) {
    const int64_t tiles = B / TL_ROWS;
    return (int)(tiles < 256 ? tiles : 256);
}

template <int NB>
static int tile_bwd_launch(const RnTileBwd& p, int grid, hipStream_t st) {
    const size_t lds = (size_t)TLB_LDS_FLOATS * sizeof(float);
    const bool tanh2 = p.act_inner == RECNOW_ACT_TANH && p.act_outer == RECNOW_ACT_TANH;
    const bool dx = p.dx != nullptr;
    static std::atomic<bool> raised[4][64];      // per device, as in tile_launch
    const int v = (tanh2 ? 2 : 0) + (dx ? 1 : 0);
    const void* fn = v == 3 ? (const void*)k_mix_tile_bwd<NB, true, true> : v == 2 ? (const void*)k_mix_tile_bwd<NB, true, false>
                   : v == 1 ? (const void*)k_mix_tile_bwd<NB, false, true> : (const void*)k_mix_tile_bwd<NB, false, false>;
    int dev = 0;
    RN_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !raised[v][dev].load(std::memory_order_acquire)) {
        RN_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) raised[v][dev].store(true, std::memory_order_release);
    }
    if (v == 3) hipLaunchKernelGGL((k_mix_tile_bwd<NB, true, true>), grid, 256, lds, st, p);
    else if (v == 2) hipLaunchKernelGGL((k_mix_tile_bwd<NB, true, false>), grid, 256, lds, st, p);
    else if (v == 1) hipLaunchKernelGGL((k_mix_tile_bwd<NB, false, true>), grid, 256, lds, st, p);
    else hipLaunchKernelGGL((k_mix_tile_bwd<NB, false, false>), grid, 256, lds, st, p);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

int rn_mix_tile_bwd(const RnTileBwd& p, hipStream_t st) {
    if (!rn_mix_tile_supported(p.B, p.D, 64, 2, p.L, TL_LDT) || !p.packs || !p.x || !p.dvpart) return RECNOW_EUNSUPPORTED;
    if (p.l_hi < p.l_lo || p.l_lo < 0 || p.l_hi >= p.L) return RECNOW_EINVAL;
    if (!p.gin && (!p.ds || !p.head_w)) return RECNOW_EINVAL;
    const int grid = rn_mix_tile_bwd_grid(p.B);
    const int nl = p.l_hi - p.l_lo + 1;
    RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_MIX_TILE_BWD, (double)nl * (4.0 * p.B * p.D * 130 + 8.0 * p.B * 2 * 64 * 64),
                                                    (double)nl * (12.0 * p.B * TL_LDT + 16.0 * p.B * p.D), st)
                                    : nullptr;
    int rc;
    switch (p.D) {
        case 256: rc = tile_bwd_launch<2>(p, grid, st); break;
        case 512: rc = tile_bwd_launch<4>(p, grid, st); break;
        default: rc = tile_bwd_launch<8>(p, grid, st); break;
    }
    rn_prof_end(pr, st);
    return rc;
}
