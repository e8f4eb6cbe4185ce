// The sub-space stage of DCNMixLayer, /root/reference/rec_now/layers/dcn_mix_layer.py:137-138 (einsum 'bns,nst->bnt' +
// activation_outer) and :146-149 (softmax gate and its product with the expert outputs), fused into one streaming kernel
// per direction.  The S x S products are 1-2 GFLOP per layer at the north-star shape - nothing for the MFMA pipe - while
// the generic route (batched GEMM + gate kernel; gate-backward kernel + two batched GEMMs) costs five launches that each
// stream the (B, N*S) activations through HBM again.  Here a workgroup keeps every expert's S x S matrix in LDS, walks
// 32-row tiles of the activations, and does all of it per tile:
//   forward : T2 = [act_outer(H1_n V_n) | G | 0],  T2g = [G_n * H2_n | G | 0],  G = softmax(logits)   (logits = T1[:, NS:NS+N])
//   backward: dC = G_n * dT2g * act_outer'(H2);  dlogits = G * (dG - <G, dG>),  dG_n = <dT2g_n, H2_n> + dT2g[:, NS+n];
//             dT1 = [(dC_n V_n^T) * act_inner'(H1_n) | dlogits | 0];  dV_n = H1_n^T dC_n accumulated in AGPRs over the
//             workgroup's tiles, written as one partial per workgroup and summed in a fixed order (deterministic).
// dC never goes to HBM.  Products use v_mfma_f32_32x32x2_f32 (exact fp32) on 32x32 output blocks; LDS tiles are row-major
// with an odd row stride so the per-lane ds_read_b32 of both operand shapes is bank-conflict free.
#include "common.hpp"
#include "dcnmix_mid.hpp"
#include "prof.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MID_ROWS 32          // rows per tile
#define MID_NMAX 8           // experts (registers of the gate code)
#define MID_VITEMS 4         // max dV output blocks per wave
#define MID_PRE 4            // float4 chunks per thread prefetched one tile ahead (covers N*S <= 128)

static inline size_t mid_fwd_lds(int S, int N) { return ((size_t)N * S * S + (size_t)MID_ROWS * (N * S + 1) + (size_t)MID_ROWS * N) * sizeof(float); }
static inline size_t mid_bwd_lds(int S, int N) {
    return ((size_t)N * S * (S + 1) + 2 * (size_t)MID_ROWS * (N * S + 1) + 3 * (size_t)MID_ROWS * N) * sizeof(float);      // + Ds and the padded V^T rows of the fast variant
}

bool rn_mix_mid_supported(int S, int N, int LDT) {
    if (S != 32 && S != 64) return false;
    if (N < 1 || N > MID_NMAX) return false;
    if (N * (S / 32) * (S / 32) > 4 * MID_VITEMS) return false;
    if (LDT % 4 != 0) return false;
    return mid_bwd_lds(S, N) <= 140 * 1024;
}

static inline int mid_grid(int64_t B) {
    static const int cap = []() { const char* e = getenv("RECNOW_MID_BWD_GRID"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 512 ? v : 512; }();      // experiments: <= 512 (the workspace is sized for 512 partials)
    const int64_t tiles = (B + MID_ROWS - 1) / MID_ROWS;
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}

// forward: as many workgroups as the LDS footprint lets a CU hold (the kernel is latency-bound per workgroup)
static inline int mid_fwd_grid(int64_t B, size_t lds) {
    const int64_t tiles = (B + MID_ROWS - 1) / MID_ROWS;
    int per_cu = (int)((160 * 1024) / (lds + 1024));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    const int64_t cap = 256 * per_cu;
    return (int)(tiles < cap ? (tiles > 0 ? tiles : 1) : cap);
}

size_t rn_mix_mid_bwd_ws_bytes(int64_t B, int S, int N) { return rn_align((size_t)mid_grid(B) * N * S * S * sizeof(float)); }

template <int S>
__global__ void __launch_bounds__(256, 2)
k_mix_mid_fwd(const float* __restrict__ T1, const float* __restrict__ V, float* __restrict__ T2, float* __restrict__ T2g, int64_t B,
              int N, int LDT, int act_outer) {
    extern __shared__ float lds[];
    const int NS = N * S, LDA = NS + 1;
    float* Vs = lds;                       // [n][k][col]
    float* As = Vs + N * S * S;            // [row][NS] stride LDA
    float* Gs = As + MID_ROWS * LDA;       // [row][n]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid * 4; i < N * S * S; i += 1024) *reinterpret_cast<float4*>(Vs + i) = *reinterpret_cast<const float4*>(V + i);
    const int64_t ntiles = (B + MID_ROWS - 1) / MID_ROWS;
    const int cpr = NS / 4;
    const int nch = MID_ROWS * cpr / 256;          // float4 chunks per thread and tile (NS % 32 == 0)
    // The next tile's rows are fetched into registers before this tile's MFMA phase: HBM latency hides under it.
    float4 pv[MID_PRE];
    float plg[MID_NMAX];
    auto prefetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < MID_PRE; ++i) {
            pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nch) {
                const int c = tid + 256 * i, r = c / cpr, k4 = (c - r * cpr) * 4;
                if (r0 + r < B) pv[i] = *reinterpret_cast<const float4*>(T1 + (r0 + r) * LDT + k4);
            }
        }
        if (tid < MID_ROWS) {
#pragma unroll
            for (int n = 0; n < MID_NMAX; ++n) plg[n] = (n < N && r0 + tid < B) ? T1[(r0 + tid) * LDT + NS + n] : -INFINITY;
        }
    };
    prefetch((int64_t)blockIdx.x * MID_ROWS);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * MID_ROWS;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MID_PRE; ++i)
            if (i < nch) {
                const int c = tid + 256 * i, r = c / cpr, k4 = (c - r * cpr) * 4;
                float* d = As + r * LDA + k4;
                d[0] = pv[i].x; d[1] = pv[i].y; d[2] = pv[i].z; d[3] = pv[i].w;
            }
        for (int i = MID_PRE; i < nch; ++i) {        // wider than the prefetch registers (N*S > 128): fetched in place
            const int c = tid + 256 * i, r = c / cpr, k4 = (c - r * cpr) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + r < B) v = *reinterpret_cast<const float4*>(T1 + (r0 + r) * LDT + k4);
            float* d = As + r * LDA + k4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        if (tid < MID_ROWS) {
            const int64_t row = r0 + tid;
            if (row < B) {
                float lg[MID_NMAX];
                float mx = -INFINITY;
#pragma unroll
                for (int n = 0; n < MID_NMAX; ++n) mx = plg[n] > mx ? plg[n] : mx;
                float sum = 0.f;
#pragma unroll
                for (int n = 0; n < MID_NMAX; ++n) {
                    lg[n] = n < N ? expf(plg[n] - mx) : 0.f;
                    sum += lg[n];
                }
#pragma unroll
                for (int n = 0; n < MID_NMAX; ++n)
                    if (n < N) {
                        const float g = lg[n] / sum;
                        Gs[tid * N + n] = g;
                        T2[row * LDT + NS + n] = g;
                        T2g[row * LDT + NS + n] = g;
                    }
                for (int c = NS + N; c < LDT; ++c) { T2[row * LDT + c] = 0.f; T2g[row * LDT + c] = 0.f; }
            } else {
                for (int n = 0; n < N; ++n) Gs[tid * N + n] = 0.f;
            }
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) prefetch((tile + gridDim.x) * MID_ROWS);
        const int nitems = N * (S / 32);
        for (int item = w; item < nitems; item += 4) {                    // wave-uniform
            const int n = item / (S / 32), cb = item - n * (S / 32);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = As + (lane & 31) * LDA + n * S + (lane >> 5);
            const float* bp = Vs + n * S * S + (lane >> 5) * S + cb * 32 + (lane & 31);
#pragma unroll 8
            for (int st = 0; st < S / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], bp[2 * st * S], acc, 0, 0, 0);
            const int col = n * S + cb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t row = r0 + rr;
                const float h2 = rn_act(acc[r], act_outer);
                if (row < B) {
                    T2[row * LDT + col] = h2;
                    T2g[row * LDT + col] = Gs[rr * N + n] * h2;
                }
            }
        }
    }
}

template <int S, int VI>
__global__ void __launch_bounds__(256, 2)
k_mix_mid_bwd(const float* __restrict__ dT2g, const float* __restrict__ T2, const float* __restrict__ T1, const float* __restrict__ V,
              float* __restrict__ dT1, float* __restrict__ dVpart, int64_t B, int N, int LDT, int act_inner, int act_outer,
              const float* __restrict__ rscale /* optional: dT2g is read as rscale[row] * dT2g[row][:] */) {
    extern __shared__ float lds[];
    const int NS = N * S, LDA = NS + 1;
    float* VTs = lds;                      // [n][t][s] = V[n][s][t]
    float* Cs = VTs + N * S * S;           // dC tile
    float* Hs = Cs + MID_ROWS * LDA;       // H1 tile
    float* Ps = Hs + MID_ROWS * LDA;       // [row][n]  <dT2g_n, H2_n>
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < N * S * S; i += 256) {
        const int n = i / (S * S), rem = i - n * S * S, s = rem / S, t = rem - s * S;
        VTs[n * S * S + t * S + s] = V[i];
    }
    f32x16 accV[VI];
#pragma unroll
    for (int j = 0; j < VI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accV[j][r] = 0.f;
    const int64_t ntiles = (B + MID_ROWS - 1) / MID_ROWS;
    const int cpr = NS / 4;
    constexpr int GL = S / 4;              // lanes holding one (row, expert) segment
    const int nvitems = N * (S / 32) * (S / 32);
    const int nch = MID_ROWS * cpr / 256;          // float4 chunks per thread and tile: block-uniform (NS % 32 == 0)
    float4 pd[MID_PRE], ph[MID_PRE], pa[MID_PRE];
    float pg[MID_PRE];
    float pgg[MID_NMAX], pdg[MID_NMAX];            // gate row of thread tid < 32
    auto fetch = [&](int64_t r0, int i, float4& d, float4& h, float4& a, float& g) {
        const int c = tid + 256 * i, r = c / cpr, k4 = (c - r * cpr) * 4, n = k4 / S;
        const int64_t row = r0 + r;
        d = make_float4(0.f, 0.f, 0.f, 0.f); h = d; a = d; g = 0.f;
        if (row < B) {
            d = *reinterpret_cast<const float4*>(dT2g + row * LDT + k4);
            if (rscale) {
                const float sc = rscale[row];
                d.x *= sc; d.y *= sc; d.z *= sc; d.w *= sc;
            }
            h = *reinterpret_cast<const float4*>(T2 + row * LDT + k4);
            a = *reinterpret_cast<const float4*>(T1 + row * LDT + k4);
            g = T2[row * LDT + NS + n];
        }
    };
    auto stage = [&](int i, const float4& d, const float4& h, const float4& a, float g) {
        const int c = tid + 256 * i, r = c / cpr, k4 = (c - r * cpr) * 4, n = k4 / S;
        float* cd = Cs + r * LDA + k4;
        cd[0] = g * d.x * rn_act_grad_from_out(h.x, act_outer);
        cd[1] = g * d.y * rn_act_grad_from_out(h.y, act_outer);
        cd[2] = g * d.z * rn_act_grad_from_out(h.z, act_outer);
        cd[3] = g * d.w * rn_act_grad_from_out(h.w, act_outer);
        float* hd = Hs + r * LDA + k4;
        hd[0] = a.x; hd[1] = a.y; hd[2] = a.z; hd[3] = a.w;
        float p = d.x * h.x + d.y * h.y + d.z * h.z + d.w * h.w;
#pragma unroll
        for (int o = GL / 2; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);      // all 64 lanes run every chunk
        if ((c & (GL - 1)) == 0) Ps[r * N + n] = p;
    };
    auto prefetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < MID_PRE; ++i)
            if (i < nch) fetch(r0, i, pd[i], ph[i], pa[i], pg[i]);
        if (tid < MID_ROWS) {
#pragma unroll
            for (int n = 0; n < MID_NMAX; ++n) {
                const bool ok = n < N && r0 + tid < B;
                pgg[n] = ok ? T2[(r0 + tid) * LDT + NS + n] : 0.f;
                pdg[n] = ok ? dT2g[(r0 + tid) * LDT + NS + n] * (rscale ? rscale[r0 + tid] : 1.f) : 0.f;
            }
        }
    };
    prefetch((int64_t)blockIdx.x * MID_ROWS);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * MID_ROWS;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MID_PRE; ++i)
            if (i < nch) stage(i, pd[i], ph[i], pa[i], pg[i]);
        for (int i = MID_PRE; i < nch; ++i) {        // wider than the prefetch registers (N*S > 128): fetched in place
            float4 d, h, a;
            float g;
            fetch(r0, i, d, h, a, g);
            stage(i, d, h, a, g);
        }
        __syncthreads();
        if (tid < MID_ROWS) {
            const int64_t row = r0 + tid;
            if (row < B) {
                float dg[MID_NMAX];
                float dot = 0.f;
#pragma unroll
                for (int n = 0; n < MID_NMAX; ++n) {
                    dg[n] = n < N ? Ps[tid * N + n] + pdg[n] : 0.f;
                    dot += pgg[n] * dg[n];
                }
#pragma unroll
                for (int n = 0; n < MID_NMAX; ++n)
                    if (n < N) dT1[row * LDT + NS + n] = pgg[n] * (dg[n] - dot);
                for (int c = NS + N; c < LDT; ++c) dT1[row * LDT + c] = 0.f;
            }
        }
        if (tile + gridDim.x < ntiles) prefetch((tile + gridDim.x) * MID_ROWS);
        // dA_n = (dC_n V_n^T) * act_inner'(H1_n): 32x32 output blocks (n, cb)
        const int nitems = N * (S / 32);
        for (int item = w; item < nitems; item += 4) {
            const int n = item / (S / 32), cb = item - n * (S / 32);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = Cs + (lane & 31) * LDA + n * S + (lane >> 5);
            const float* bp = VTs + n * S * S + (lane >> 5) * S + cb * 32 + (lane & 31);
#pragma unroll 8
            for (int st = 0; st < S / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], bp[2 * st * S], acc, 0, 0, 0);
            const int col = n * S + cb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t row = r0 + rr;
                if (row < B) dT1[row * LDT + col] = acc[r] * rn_act_grad_from_out(Hs[rr * LDA + col], act_inner);
            }
        }
        // dV_n += H1_n^T dC_n over this tile's 32 rows: output blocks (n, mb, cb) stay in registers across tiles
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int item = w + 4 * j;
            if (item < nvitems) {
                const int n = item / ((S / 32) * (S / 32)), rem = item - n * (S / 32) * (S / 32), mb = rem / (S / 32), cb = rem - mb * (S / 32);
                const float* ap = Hs + (lane >> 5) * LDA + n * S + mb * 32 + (lane & 31);
                const float* bp = Cs + (lane >> 5) * LDA + n * S + cb * 32 + (lane & 31);
#pragma unroll 8
                for (int st = 0; st < MID_ROWS / 2; ++st)
                    accV[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st * LDA], bp[2 * st * LDA], accV[j], 0, 0, 0);
            }
        }
    }
    float* P = dVpart + (int64_t)blockIdx.x * N * S * S;
#pragma unroll
    for (int j = 0; j < VI; ++j) {
        const int item = w + 4 * j;
        if (item < nvitems) {
            const int n = item / ((S / 32) * (S / 32)), rem = item - n * (S / 32) * (S / 32), mb = rem / (S / 32), cb = rem - mb * (S / 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                P[n * S * S + (mb * 32 + rr) * S + cb * 32 + (lane & 31)] = accV[j][r];
            }
        }
    }
}

// ---- fast variants: every tile full (B % 32 == 0), N a template parameter with N * S = 128 (four 32 x 32 output blocks: one
// per wave), LDT = N * S + 16 -----------------------------------------------------------------------------------------------
// Same arithmetic, same order of operations as the general kernels above (results equal up to fma contraction), restructured around what their
// ISA showed: (1) every `if (row < B)` / `if (i < nch)` around a load became a branch with an `s_waitcnt vmcnt(0)` at its merge, so
// the prefetch loads of a tile were issued one HBM round trip after the other (~10 us per 32-row tile); (2) vmcnt counts loads and
// stores in order, and a wait for the prefetched tile is `vmcnt(number of stores issued since)` only if that number is the same on
// every path -- so no store sits under a lane or wave condition here: the gate columns (computed by 32 lanes) go through LDS and
// are stored by all 256 threads, one float4 each, and every wave owns exactly one output block.
// Chunk mapping: thread = (row 4 (tid >> 5) + ((tid >> 3) & 3), float4 chunks (tid & 7) + 8 i, i < 4): eight lanes read one 128-byte line
// of a row, and the transposing LDS writes of a 32-lane group go to rows r .. r + 3 x chunks 0 .. 7, i.e. 32 distinct banks (row
// stride 129 floats); with 32 lanes on ONE row (chunks 0 .. 31) lanes l and l + 8 shared a bank: 4-way conflicts on every staging
// write (counters: 36 % / 60 % of the LDS cycles of the forward / backward kernel were conflicts).
template <int S, int N, int AO, int NSL>
__device__ __forceinline__ void mid_fwd_fast_body(const float* __restrict__ T1, const float* __restrict__ V, float* __restrict__ T2,
                                                  float* __restrict__ T2g, int64_t B, int LDT, int act_outer_rt, const RnSlabs sl, int act_inner) {
    const int act_outer = AO >= 0 ? AO : act_outer_rt;       // compile-time activation: no per-element switch in the store loop
    constexpr bool SL = NSL > 0;
    float* const T1out = const_cast<float*>(T1);             // SL: T1 is written (the backward pass reads it), not read
    int64_t pr0 = 0;                                         // SL: first row of the prefetched tile
    extern __shared__ float lds[];
    constexpr int NS = N * S, LDA = NS + 1, CPR = NS / 4, NCH = MID_ROWS * CPR / 256;
    static_assert(NS == 128 && NCH == 4 && N * (S / 32) == 4, "N * S = 128");
    float* Vs = lds;                       // [n][k][col]
    float* As = Vs + N * S * S;            // [row][NS] stride LDA
    float* Gs = As + MID_ROWS * LDA;       // [row][n]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid * 4; i < N * S * S; i += 1024) *reinterpret_cast<float4*>(Vs + i) = *reinterpret_cast<const float4*>(V + i);
    const int64_t ntiles = B / MID_ROWS;
    float4 pv[NCH];
    float plg[N];
    auto prefetch = [&](int64_t r0) {
        if constexpr (SL) {
            // the product that feeds this kernel was split over K into NSL slabs: all NSL x NCH loads of the tile go out together and
            // are summed in slab order (fp32; NSL <= 4 terms of a K = 1024 product); the activation of the product's epilogue is
            // applied when the tile is staged
            pr0 = r0;
            float4 t[NSL][NCH];
            float tg[NSL][N];
#pragma unroll
            for (int s = 0; s < NSL; ++s) {
                const float* __restrict__ P = sl.p + (int64_t)s * sl.stride;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int r = 4 * (tid >> 5) + ((tid >> 3) & 3), k4 = ((tid & 7) + 8 * i) * 4;
                    t[s][i] = *reinterpret_cast<const float4*>(P + (r0 + r) * sl.ld + k4);
                }
#pragma unroll
                for (int n = 0; n < N; ++n) tg[s][n] = P[(r0 + (tid & 31)) * sl.ld + NS + n];
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                pv[i] = t[0][i];
#pragma unroll
                for (int s = 1; s < NSL; ++s) { pv[i].x += t[s][i].x; pv[i].y += t[s][i].y; pv[i].z += t[s][i].z; pv[i].w += t[s][i].w; }
            }
#pragma unroll
            for (int n = 0; n < N; ++n) {
                plg[n] = tg[0][n];
#pragma unroll
                for (int s = 1; s < NSL; ++s) plg[n] += tg[s][n];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = 4 * (tid >> 5) + ((tid >> 3) & 3), k4 = ((tid & 7) + 8 * i) * 4;      // see the note on the chunk mapping above
            pv[i] = *reinterpret_cast<const float4*>(T1 + (r0 + r) * LDT + k4);
        }
#pragma unroll
        for (int n = 0; n < N; ++n) plg[n] = T1[(r0 + (tid & 31)) * LDT + NS + n];      // every thread (no branch); threads < 32 use it
    };
    auto stage = [&]() {                       // prefetched tile -> LDS; softmax of its logits -> LDS
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = 4 * (tid >> 5) + ((tid >> 3) & 3), k4 = ((tid & 7) + 8 * i) * 4;      // see the note on the chunk mapping above
            if constexpr (SL) {                // H1 = act_inner(x_l U) leaves for the backward pass from here; logits by every thread of the row
                pv[i] = make_float4(rn_act(pv[i].x, act_inner), rn_act(pv[i].y, act_inner), rn_act(pv[i].z, act_inner), rn_act(pv[i].w, act_inner));
                *reinterpret_cast<float4*>(T1out + (pr0 + r) * LDT + k4) = pv[i];
            }
            float* d = As + r * LDA + k4;
            d[0] = pv[i].x; d[1] = pv[i].y; d[2] = pv[i].z; d[3] = pv[i].w;
        }
        if constexpr (SL) {
#pragma unroll
            for (int n = 0; n < N; ++n) T1out[(pr0 + (tid & 31)) * LDT + NS + n] = plg[n];
        }
        if (tid < MID_ROWS) {
            float mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < N; ++n) mx = plg[n] > mx ? plg[n] : mx;
            float lg[N];
            float sum = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                lg[n] = expf(plg[n] - mx);
                sum += lg[n];
            }
#pragma unroll
            for (int n = 0; n < N; ++n) Gs[tid * N + n] = lg[n] / sum;
        }
    };
    // Loop shape: [stage tile] B [request next tile; gate columns + output block stored] B [stage next tile] ...: the wait for the
    // requested tile sits in the stage at the bottom, behind a fixed number of stores (the stage after the last tile stages the
    // re-read last tile, which nobody uses).
    prefetch((int64_t)blockIdx.x * MID_ROWS);
    stage();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t r0 = tile * MID_ROWS;
        __syncthreads();
        {
            const int64_t nt = tile + gridDim.x < ntiles ? tile + gridDim.x : tile;
            prefetch(nt * MID_ROWS);
        }
        {   // columns NS .. NS + 15 of T2 and T2g: [G | 0]; thread = (row, tensor, float4)
            const int row = tid >> 3, q = tid & 3;
            float g4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) g4[e] = 4 * q + e < N ? Gs[row * N + ((4 * q + e) < N ? 4 * q + e : 0)] : 0.f;
            float* dst = ((tid & 4) ? T2g : T2) + (r0 + row) * LDT + NS + 4 * q;
            *reinterpret_cast<float4*>(dst) = make_float4(g4[0], g4[1], g4[2], g4[3]);
        }
        {
            const int n = w / (S / 32), cb = w - n * (S / 32);              // one output block per wave
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = As + (lane & 31) * LDA + n * S + (lane >> 5);
            const float* bp = Vs + n * S * S + (lane >> 5) * S + cb * 32 + (lane & 31);
#pragma unroll 8
            for (int st = 0; st < S / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], bp[2 * st * S], acc, 0, 0, 0);
            const int col = n * S + cb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t row = r0 + rr;
                const float h2 = rn_act(acc[r], act_outer);
                T2[row * LDT + col] = h2;
                T2g[row * LDT + col] = Gs[rr * N + n] * h2;
            }
        }
        __syncthreads();
        stage();
    }
}

template <int S, int N, int SL = 0>
__global__ void __launch_bounds__(256, 2)
k_mix_mid_fwd_fast(const float* __restrict__ T1, const float* __restrict__ V, float* __restrict__ T2, float* __restrict__ T2g, int64_t B,
                   int LDT, int act_outer, const RnSlabs sl, int act_inner) {
    if (act_outer == RECNOW_ACT_TANH) mid_fwd_fast_body<S, N, RECNOW_ACT_TANH, SL>(T1, V, T2, T2g, B, LDT, act_outer, sl, act_inner);
    else mid_fwd_fast_body<S, N, -1, SL>(T1, V, T2, T2g, B, LDT, act_outer, sl, act_inner);
}

// A/B (tools/build_variant.py -DRN_MID_NT): the streamed tiles of the fast backward kernel (dT2g, T2, T1 in: read once; dT1 out) as non-temporal accesses.
// Measured SLOWER (round 4, tools/ab_lib.sh: 48.5 -> 53.4 us per launch -- dT2g was written by the launch just before and is served from the cache): off.
#ifdef RN_MID_NT
#define MID_LD4(ptr) mid_nt_load4(ptr)
#define MID_ST1(ptr, v) __builtin_nontemporal_store((v), (ptr))
__device__ __forceinline__ float4 mid_nt_load4(const float* p) {
    typedef float mid_f4 __attribute__((ext_vector_type(4)));
    const mid_f4 q = __builtin_nontemporal_load(reinterpret_cast<const mid_f4*>(p));
    return make_float4(q.x, q.y, q.z, q.w);
}
#else
#define MID_LD4(ptr) (*reinterpret_cast<const float4*>(ptr))
#define MID_ST1(ptr, v) (*(ptr) = (v))
#endif
// -DRN_MID_TRACE (tools/build_variant.py midtrace -DRN_MID_TRACE; tools/mid_trace.py): 100 MHz wall-clock stamps of the phases of every tile of
// workgroups 0 and gridDim.x / 2 of the LAST launch, read back through recnow_debug_mid_trace.  Not in product builds (the macros are empty).
// Round 5, one box, microseconds per 32-row tile of workgroup 0 (a tile every 6.2-7.8 us): gate math 0.1-0.6 | request of the next tile (24 loads per
// thread) 0.7-1.4 | dA chain (32 dependent MFMAs + 64 LDS operand reads) 1.4-2.2 | 16 dT1 stores per lane 0.5-1.6 | dV chains (2 x 16 MFMAs) 1.3-1.5 |
// barrier 0.1 | gate columns 0.2-0.5 | staging the next tile into LDS (32 ds_write_b32 per thread, waits for its loads) 0.6-1.5: the phases of a
// workgroup run in series and each is bound by the ISSUE of many small instructions, not by a roofline.  Interleaving the dA and dV chains (three
// independent accumulators in one fully unrolled loop) measured SLOWER (46 -> 52.6 us per launch: the merged loop's LDS reads and register pressure cost
// more than the dependency bubbles it removes) and is not kept.
#ifdef RN_MID_TRACE
__device__ long long g_mid_trace[2 * 16 * 10];
#define MT_STAMP(k, i) do { if (threadIdx.x == 0 && (k) < 16 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) \
        g_mid_trace[((blockIdx.x ? 1 : 0) * 16 + (k)) * 10 + (i)] = wall_clock64(); } while (0)
#define MT_STAMP_ACC(k, i, v) do { asm volatile("" :: "v"(v)); MT_STAMP(k, i); } while (0)      // (the stamp waits for the accumulator it names)
extern "C" int recnow_debug_mid_trace(long long* out_host) {
    return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_mid_trace), sizeof(long long) * 2 * 16 * 10);
}
#else
#define MT_STAMP(k, i) do { } while (0)
#define MT_STAMP_ACC(k, i, v) do { } while (0)
#endif
template <int S, int N, bool RS, int AI, int AO, int NSL>
__device__ __forceinline__ void mid_bwd_fast_body(const float* __restrict__ dT2g, const float* __restrict__ T2, const float* __restrict__ T1,
                                                  const float* __restrict__ V, float* __restrict__ dT1, float* __restrict__ dVpart, int64_t B,
                                                  int LDT, int act_inner_rt, int act_outer_rt, const float* __restrict__ rscale, const RnSlabs sl) {
    const int act_inner = AI >= 0 ? AI : act_inner_rt, act_outer = AO >= 0 ? AO : act_outer_rt;
    constexpr bool SL = NSL > 0;
    extern __shared__ float lds[];
    constexpr int NS = N * S, LDA = NS + 1, CPR = NS / 4, NCH = MID_ROWS * CPR / 256, GL = S / 4;
    constexpr int nvitems = N * (S / 32) * (S / 32), VI = (nvitems + 3) / 4;
    static_assert(NS == 128 && NCH == 4 && N * (S / 32) == 4, "N * S = 128");
    constexpr int LV = S + 1;              // row stride of the transposed matrices: the transposing writes (lane = t) walk a stride of LV
                                           // floats, i.e. 32 banks; with a stride of S = 64 every lane hit ONE bank (32-way conflict,
                                           // ~8 us of the kernel per CU for the 2 x 8192 elements, counters)
    float* VTs = lds;                      // [n][t][s] = V[n][s][t], rows of LV floats
    float* Cs = VTs + N * S * LV;          // dC tile
    float* Hs = Cs + MID_ROWS * LDA;       // H1 tile
    float* Ps = Hs + MID_ROWS * LDA;       // [row][n]  <dT2g_n, H2_n>
    float* Ds = Ps + MID_ROWS * N;         // [row][n]  dlogits of the tile (stored by all threads after the next barrier)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < N * S * S; i += 256) {
        const int n = i / (S * S), rem = i - n * S * S, s = rem / S, t = rem - s * S;
        VTs[(n * S + t) * LV + s] = V[i];
    }
    f32x16 accV[VI];
#pragma unroll
    for (int j = 0; j < VI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accV[j][r] = 0.f;
    const int64_t ntiles = B / MID_ROWS;
    float4 pd[NCH], ph[NCH], pa[NCH];
    float pg[NCH], psc[NCH];
    float pgg[N], pdg[N], psg = 1.f;           // gate row (tid & 31): used by threads < 32
    auto prefetch = [&](int64_t r0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = 4 * (tid >> 5) + ((tid >> 3) & 3), k4 = ((tid & 7) + 8 * i) * 4, n = k4 / S;
            const int64_t row = r0 + r;
            if constexpr (!SL) pd[i] = MID_LD4(dT2g + row * LDT + k4);
            ph[i] = MID_LD4(T2 + row * LDT + k4);
            pa[i] = MID_LD4(T1 + row * LDT + k4);
            pg[i] = T2[row * LDT + NS + n];
            psc[i] = RS ? rscale[row] : 1.f;
        }
        const int64_t grow = r0 + (tid & 31);
#pragma unroll
        for (int n = 0; n < N; ++n) {
            pgg[n] = T2[grow * LDT + NS + n];
            if constexpr (!SL) pdg[n] = dT2g[grow * LDT + NS + n];
        }
        if (RS) psg = rscale[grow];
        if constexpr (SL) {       // the dT2g product was split over K into NSL slabs: loaded together, summed in slab order (fp32, <= 4 terms)
            float4 t[NSL][NCH];
            float tg[NSL][N];
#pragma unroll
            for (int sb = 0; sb < NSL; ++sb) {
                const float* __restrict__ P = sl.p + (int64_t)sb * sl.stride;
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int r = 4 * (tid >> 5) + ((tid >> 3) & 3), k4 = ((tid & 7) + 8 * i) * 4;
                    t[sb][i] = *reinterpret_cast<const float4*>(P + (r0 + r) * sl.ld + k4);
                }
#pragma unroll
                for (int n = 0; n < N; ++n) tg[sb][n] = P[grow * sl.ld + NS + n];
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                pd[i] = t[0][i];
#pragma unroll
                for (int sb = 1; sb < NSL; ++sb) { pd[i].x += t[sb][i].x; pd[i].y += t[sb][i].y; pd[i].z += t[sb][i].z; pd[i].w += t[sb][i].w; }
            }
#pragma unroll
            for (int n = 0; n < N; ++n) {
                pdg[n] = tg[0][n];
#pragma unroll
                for (int sb = 1; sb < NSL; ++sb) pdg[n] += tg[sb][n];
            }
        }
    };
    auto stage = [&]() {
        float ppv[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = 4 * (tid >> 5) + ((tid >> 3) & 3), k4 = ((tid & 7) + 8 * i) * 4;
            float4 d = pd[i];
            if (RS) { d.x *= psc[i]; d.y *= psc[i]; d.z *= psc[i]; d.w *= psc[i]; }
            const float4 h = ph[i], a = pa[i];
            const float g = pg[i];
            float* cd = Cs + r * LDA + k4;
            cd[0] = g * d.x * rn_act_grad_from_out(h.x, act_outer);
            cd[1] = g * d.y * rn_act_grad_from_out(h.y, act_outer);
            cd[2] = g * d.z * rn_act_grad_from_out(h.z, act_outer);
            cd[3] = g * d.w * rn_act_grad_from_out(h.w, act_outer);
            float* hd = Hs + r * LDA + k4;
            hd[0] = a.x; hd[1] = a.y; hd[2] = a.z; hd[3] = a.w;
            ppv[i] = d.x * h.x + d.y * h.y + d.z * h.z + d.w * h.w;
        }
        // <dT2g_n, H2_n> of a row: the S / 4 chunks of a segment are slots i of the row's eight lanes (S = 64: two slots, S = 32: one)
        constexpr int SPS = GL / 8;            // slots per segment
#pragma unroll
        for (int i = 0; i < NCH; i += SPS) {
            float pp = ppv[i];
#pragma unroll
            for (int j = 1; j < SPS; ++j) pp += ppv[i + j];
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) pp += __shfl_xor(pp, o, 64);
            if ((tid & 7) == 0) Ps[(4 * (tid >> 5) + ((tid >> 3) & 3)) * N + (8 * i * 4) / S] = pp;
        }
    };
    // Loop shape: [stage tile] B [gate math -> LDS; request next tile; dA block + stores; dV blocks] B [gate columns stored by all
    // threads; stage next tile] ...: the stage at the bottom belongs to the next iteration (after the last tile it stages the
    // re-read last tile again, which nobody uses).
    prefetch((int64_t)blockIdx.x * MID_ROWS);
    __syncthreads();                       // VTs complete
    if (sl.stagger > 0 && (int)blockIdx.x >= (int)gridDim.x / 2) {      // experiment: de-phase the two workgroups of a CU (every wave leaves the bounded loop)
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (long long)sl.stagger && wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);
    }
    stage();
    MT_STAMP(0, 0);
    int mt_k = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++mt_k) {
        const int64_t r0 = tile * MID_ROWS;
        __syncthreads();
        MT_STAMP(mt_k, 1);
        if (tid < MID_ROWS) {
            float dg[N];
            float dot = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) {
                dg[n] = Ps[tid * N + n] + pdg[n] * psg;
                dot += pgg[n] * dg[n];
            }
#pragma unroll
            for (int n = 0; n < N; ++n) Ds[tid * N + n] = pgg[n] * (dg[n] - dot);
        }
        MT_STAMP(mt_k, 2);
        {
            const int64_t nt = tile + gridDim.x < ntiles ? tile + gridDim.x : tile;
            prefetch(nt * MID_ROWS);
        }
        MT_STAMP(mt_k, 3);
        {   // dA_n = (dC_n V_n^T) * act_inner'(H1_n): one 32x32 output block (n, cb) per wave
            const int n = w / (S / 32), cb = w - n * (S / 32);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = Cs + (lane & 31) * LDA + n * S + (lane >> 5);
            const float* bp = VTs + (n * S + (lane >> 5)) * LV + cb * 32 + (lane & 31);
#pragma unroll 8
            for (int st = 0; st < S / 2; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st], bp[2 * st * LV], acc, 0, 0, 0);
            MT_STAMP_ACC(mt_k, 4, acc[0]);
            const int col = n * S + cb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                MID_ST1(dT1 + (r0 + rr) * LDT + col, acc[r] * rn_act_grad_from_out(Hs[rr * LDA + col], act_inner));
            }
        }
        MT_STAMP(mt_k, 5);
        // dV_n += H1_n^T dC_n over this tile's 32 rows: output blocks (n, mb, cb) stay in registers across tiles
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int item = w + 4 * j;
            if (item < nvitems) {
                const int n = item / ((S / 32) * (S / 32)), rem = item - n * (S / 32) * (S / 32), mb = rem / (S / 32), cb = rem - mb * (S / 32);
                const float* ap = Hs + (lane >> 5) * LDA + n * S + mb * 32 + (lane & 31);
                const float* bp = Cs + (lane >> 5) * LDA + n * S + cb * 32 + (lane & 31);
#pragma unroll 8
                for (int st = 0; st < MID_ROWS / 2; ++st)
                    accV[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * st * LDA], bp[2 * st * LDA], accV[j], 0, 0, 0);
            }
        }
        MT_STAMP_ACC(mt_k, 6, accV[0][0]);
        __syncthreads();
        MT_STAMP(mt_k, 7);
        {   // columns NS .. NS + 15 of dT1: [dlogits | 0]; thread = (row, float4), two threads write each float4 (same values)
            const int row = tid >> 3, q = tid & 3;
            float g4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) g4[e] = 4 * q + e < N ? Ds[row * N + ((4 * q + e) < N ? 4 * q + e : 0)] : 0.f;
            *reinterpret_cast<float4*>(dT1 + (r0 + row) * LDT + NS + 4 * q) = make_float4(g4[0], g4[1], g4[2], g4[3]);
        }
        MT_STAMP(mt_k, 8);
        stage();
        MT_STAMP(mt_k, 9);
    }
    float* P = dVpart + (int64_t)blockIdx.x * N * S * S;
#pragma unroll
    for (int j = 0; j < VI; ++j) {
        const int item = w + 4 * j;
        if (item < nvitems) {
            const int n = item / ((S / 32) * (S / 32)), rem = item - n * (S / 32) * (S / 32), mb = rem / (S / 32), cb = rem - mb * (S / 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                P[n * S * S + (mb * 32 + rr) * S + cb * 32 + (lane & 31)] = accV[j][r];
            }
        }
    }
}

template <int S, int N, bool RS, int SL = 0>
__global__ void __launch_bounds__(256, 2)
k_mix_mid_bwd_fast(const float* __restrict__ dT2g, const float* __restrict__ T2, const float* __restrict__ T1, const float* __restrict__ V,
                   float* __restrict__ dT1, float* __restrict__ dVpart, int64_t B, int LDT, int act_inner, int act_outer,
                   const float* __restrict__ rscale, const RnSlabs sl) {
    if (act_inner == RECNOW_ACT_TANH && act_outer == RECNOW_ACT_TANH)
        mid_bwd_fast_body<S, N, RS, RECNOW_ACT_TANH, RECNOW_ACT_TANH, SL>(dT2g, T2, T1, V, dT1, dVpart, B, LDT, act_inner, act_outer, rscale, sl);
    else mid_bwd_fast_body<S, N, RS, -1, -1, SL>(dT2g, T2, T1, V, dT1, dVpart, B, LDT, act_inner, act_outer, rscale, sl);
}

// dV[i] = sum over workgroup partials, fixed order: 64 elements x 16 strided part groups per block, then a 16-term LDS sum.
__global__ void __launch_bounds__(1024) k_mix_dv_reduce(const float* __restrict__ part, int nparts, int total, float* __restrict__ dV) {
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    if (i < total) {
        int g = q;
        for (; g + 16 * 7 < nparts; g += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += part[(int64_t)(g + 16 * u) * total + i];
        }
        for (; g < nparts; g += 16) s[0] += part[(int64_t)g * total + i];
    }
    red[q][e] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (q == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += red[u][e];
        dV[i] = t;
    }
}

template <typename K>
static int mid_allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) RN_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return RECNOW_OK;
}

static const bool g_mid_fast = []() { const char* e = getenv("RECNOW_MID_FAST"); return !e || e[0] != '0'; }();      // A/B switch
bool rn_mix_mid_absorbs_slabs(int64_t B, int S, int N, int LDT) {
    static const bool on = []() { const char* e = getenv("RECNOW_MID_SLABS"); return !e || e[0] != '0'; }();            // A/B switch
    return on && g_mid_fast && B % MID_ROWS == 0 && ((S == 64 && N == 2) || (S == 32 && N == 4)) && LDT == S * N + 16;
}
int rn_mix_mid_fwd(const float* T1, const float* V, float* T2, float* T2g, int64_t B, int S, int N, int LDT, int act_outer, hipStream_t st,
                   const RnSlabs* slabs, int act_inner) {
    if (!rn_mix_mid_supported(S, N, LDT)) return RECNOW_EUNSUPPORTED;
    if (slabs && (!rn_mix_mid_absorbs_slabs(B, S, N, LDT) || (slabs->n != 2 && slabs->n != 4))) return RECNOW_EUNSUPPORTED;
    RnSlabs sl;
    sl.p = nullptr; sl.n = 0; sl.ld = 0; sl.stride = 0; sl.stagger = 0;
    if (slabs) sl = *slabs;
    const size_t lds = mid_fwd_lds(S, N);
    int rc;
    // measurement hook: read T1, write T2 and T2g (12 * B * LDT bytes)
    RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_MIX_MID_FWD, 4.0 * B * N * S * S, 12.0 * B * LDT, st) : nullptr;
    const bool mid_fast = g_mid_fast;
#define MID_FWD_FAST(SS, NN)                                                                                                  \
    if (mid_fast && B % MID_ROWS == 0 && S == SS && N == NN && LDT == SS * NN + 16) {                                         \
        if (slabs && sl.n == 4) {                                                                                             \
            if ((rc = mid_allow_lds(k_mix_mid_fwd_fast<SS, NN, 4>, lds))) return rc;                                          \
            hipLaunchKernelGGL((k_mix_mid_fwd_fast<SS, NN, 4>), mid_fwd_grid(B, lds), 256, lds, st, T1, V, T2, T2g, B, LDT, act_outer, sl, act_inner); \
        } else if (slabs) {                                                                                                   \
            if ((rc = mid_allow_lds(k_mix_mid_fwd_fast<SS, NN, 2>, lds))) return rc;                                          \
            hipLaunchKernelGGL((k_mix_mid_fwd_fast<SS, NN, 2>), mid_fwd_grid(B, lds), 256, lds, st, T1, V, T2, T2g, B, LDT, act_outer, sl, act_inner); \
        } else {                                                                                                              \
            if ((rc = mid_allow_lds(k_mix_mid_fwd_fast<SS, NN, 0>, lds))) return rc;                                          \
            hipLaunchKernelGGL((k_mix_mid_fwd_fast<SS, NN, 0>), mid_fwd_grid(B, lds), 256, lds, st, T1, V, T2, T2g, B, LDT, act_outer, sl, act_inner); \
        }                                                                                                                     \
        rn_prof_end(pr, st);                                                                                                  \
        RN_LAUNCH_CHECK();                                                                                                    \
        return RECNOW_OK;                                                                                                     \
    }
    MID_FWD_FAST(64, 2)
    MID_FWD_FAST(32, 4)
#undef MID_FWD_FAST
    if (S == 32) {
        if ((rc = mid_allow_lds(k_mix_mid_fwd<32>, lds))) return rc;
        hipLaunchKernelGGL(k_mix_mid_fwd<32>, mid_fwd_grid(B, lds), 256, lds, st, T1, V, T2, T2g, B, N, LDT, act_outer);
    } else {
        if ((rc = mid_allow_lds(k_mix_mid_fwd<64>, lds))) return rc;
        hipLaunchKernelGGL(k_mix_mid_fwd<64>, mid_fwd_grid(B, lds), 256, lds, st, T1, V, T2, T2g, B, N, LDT, act_outer);
    }
    rn_prof_end(pr, st);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

int rn_mix_mid_bwd(const float* dT2g, const float* T2, const float* T1, const float* V, float* dT1, float* dV, int64_t B, int S, int N,
                   int LDT, int act_inner, int act_outer, void* ws, size_t ws_bytes, hipStream_t st, const float* rscale, bool defer_dv,
                   const RnSlabs* slabs) {
    if (!rn_mix_mid_supported(S, N, LDT)) return RECNOW_EUNSUPPORTED;
    if (slabs && (!rn_mix_mid_absorbs_slabs(B, S, N, LDT) || (slabs->n != 2 && slabs->n != 4))) return RECNOW_EUNSUPPORTED;
    RnSlabs sl;
    sl.p = nullptr; sl.n = 0; sl.ld = 0; sl.stride = 0; sl.stagger = 0;
    if (slabs) sl = *slabs;
    static const int stagger = []() { const char* e = getenv("RECNOW_MID_STAGGER"); return e ? atoi(e) : 0; }();      // experiment, 10 ns ticks
    sl.stagger = stagger;
    if (ws_bytes < rn_mix_mid_bwd_ws_bytes(B, S, N)) return RECNOW_EWORKSPACE;
    const size_t lds = mid_bwd_lds(S, N);
    const int grid = mid_grid(B);
    float* part = (float*)ws;
    int rc;
    const int vi = (N * (S / 32) * (S / 32) + 3) / 4;          // dV output blocks per wave
    // measurement hook: read dT2g, T2, T1, write dT1 (16 * B * LDT bytes)
    RnProfRecord* pr = rn_prof_on() ? rn_prof_begin(RN_TAG_MIX_MID_BWD, 8.0 * B * N * S * S, 16.0 * B * LDT, st) : nullptr;
#define MID_BWD(SS, VV)                                                                                                       \
    do {                                                                                                                      \
        if ((rc = mid_allow_lds(k_mix_mid_bwd<SS, VV>, lds))) return rc;                                                      \
        hipLaunchKernelGGL((k_mix_mid_bwd<SS, VV>), grid, 256, lds, st, dT2g, T2, T1, V, dT1, part, B, N, LDT, act_inner, act_outer, rscale); \
    } while (0)
    const bool mid_fast = g_mid_fast;
    bool done = false;
#define MID_BWD_FAST(SS, NN)                                                                                                  \
    if (!done && mid_fast && B % MID_ROWS == 0 && S == SS && N == NN && LDT == SS * NN + 16) {                                                    \
        const int slv = slabs ? sl.n : 0;                                                                                     \
        if (rscale && slv == 4) { if ((rc = mid_allow_lds(k_mix_mid_bwd_fast<SS, NN, true, 4>, lds))) return rc;               \
            hipLaunchKernelGGL((k_mix_mid_bwd_fast<SS, NN, true, 4>), grid, 256, lds, st, dT2g, T2, T1, V, dT1, part, B, LDT, act_inner, act_outer, rscale, sl); } \
        else if (rscale && slv == 2) { if ((rc = mid_allow_lds(k_mix_mid_bwd_fast<SS, NN, true, 2>, lds))) return rc;          \
            hipLaunchKernelGGL((k_mix_mid_bwd_fast<SS, NN, true, 2>), grid, 256, lds, st, dT2g, T2, T1, V, dT1, part, B, LDT, act_inner, act_outer, rscale, sl); } \
        else if (rscale) { if ((rc = mid_allow_lds(k_mix_mid_bwd_fast<SS, NN, true, 0>, lds))) return rc;                      \
            hipLaunchKernelGGL((k_mix_mid_bwd_fast<SS, NN, true, 0>), grid, 256, lds, st, dT2g, T2, T1, V, dT1, part, B, LDT, act_inner, act_outer, rscale, sl); } \
        else if (slv == 4) { if ((rc = mid_allow_lds(k_mix_mid_bwd_fast<SS, NN, false, 4>, lds))) return rc;                   \
            hipLaunchKernelGGL((k_mix_mid_bwd_fast<SS, NN, false, 4>), grid, 256, lds, st, dT2g, T2, T1, V, dT1, part, B, LDT, act_inner, act_outer, rscale, sl); } \
        else if (slv == 2) { if ((rc = mid_allow_lds(k_mix_mid_bwd_fast<SS, NN, false, 2>, lds))) return rc;                   \
            hipLaunchKernelGGL((k_mix_mid_bwd_fast<SS, NN, false, 2>), grid, 256, lds, st, dT2g, T2, T1, V, dT1, part, B, LDT, act_inner, act_outer, rscale, sl); } \
        else { if ((rc = mid_allow_lds(k_mix_mid_bwd_fast<SS, NN, false, 0>, lds))) return rc;                                 \
            hipLaunchKernelGGL((k_mix_mid_bwd_fast<SS, NN, false, 0>), grid, 256, lds, st, dT2g, T2, T1, V, dT1, part, B, LDT, act_inner, act_outer, rscale, sl); } \
        done = true;                                                                                                          \
    }
    MID_BWD_FAST(64, 2)
    MID_BWD_FAST(32, 4)
#undef MID_BWD_FAST
    if (done) {
    } else if (S == 32) {
        if (vi <= 1) MID_BWD(32, 1);
        else MID_BWD(32, 2);
    } else {
        if (vi <= 1) MID_BWD(64, 1);
        else if (vi <= 2) MID_BWD(64, 2);
        else MID_BWD(64, 4);
    }
#undef MID_BWD
    rn_prof_end(pr, st);
    RN_LAUNCH_CHECK();
    if (defer_dv) return RECNOW_OK;
    const int total = N * S * S;
    hipLaunchKernelGGL(k_mix_dv_reduce, rn_cdiv(total, 64), 1024, 0, st, part, grid, total, dV);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
int rn_mix_mid_bwd_nparts(int64_t B) { return mid_grid(B); }
