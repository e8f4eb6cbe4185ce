// Neighbours of the hot path, streaming kernels over rows (InnerPNN's contractions run on
// the fp32 matrix cores; everything else is HBM-bound):
//   InnerPNNLayer   /root/reference/rec_now/layers/inner_pnn_layer.py:25-53
//   SENETLayer      /root/reference/rec_now/layers/senet_layer.py:93-119 (squeeze / excite-scale; the two Dense layers
//                   in between run on the GEMM of multi_dense)
//   attention_by_dot_product   /root/reference/rec_now/rec_block/attention.py:12-38
//   focal_crossentropy_loss    /root/reference/rec_now/rec_block/focal_loss.py:12-66
// Same conventions as fm.hip: `fields` is a DEVICE array of F device pointers to contiguous (B, D_f) fp32 tensors (the
// reference's list-of-tensors input), no float atomics, every reduction in a fixed order.
#include "common.hpp"

// ---------------------------------------------------------------------------------------------------------------------
// InnerPNN: out[b][p(r,c)] = <x_r[b], x_c[b]>, r < c, p = r*F - r(r+1)/2 + (c - r - 1)           (:41-52)
// General shapes (F > 64 or D not in {4, 8, 12, 16}; the usual shapes take the MFMA kernels k_ipnn_*_gram further down):
// one wave per row.  The row's F x D block sits in LDS (row stride D+1: conflict-free per-lane reads); lane <-> column
// field c keeps x_c in registers, x_r is an LDS broadcast.  Stores of one r are consecutive in p: coalesced.
// VALU-bound: F*D FMAs per lane and row (half of them below the diagonal) against 4*F*(F-1)/2 output bytes.
// ---------------------------------------------------------------------------------------------------------------------
template <int DMAX>
__global__ void __launch_bounds__(256)
k_ipnn_fwd(const float* const* __restrict__ fields, int F, int64_t B, int D, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int LD = D + 1;
    // two images of the row's F x D block: xb (row stride DMAX, 16-byte aligned: x_r is read as float4 BROADCASTS, 4x fewer
    // LDS instructions than scalar reads, which is what bounds this kernel) and xs (stride D+1: conflict-free per-lane reads)
    float* xb = lds + (size_t)w * F * (DMAX + LD);
    float* xs = xb + F * DMAX;
    const int64_t P = (int64_t)F * (F - 1) / 2;
    for (int64_t b = (int64_t)blockIdx.x * nw + w; b < B; b += (int64_t)gridDim.x * nw) {
        for (int i = lane; i < F * DMAX; i += 64) {
            const int f = i / DMAX, d = i - f * DMAX;
            const float v = d < D ? fields[f][b * D + d] : 0.f;
            xb[i] = v;
            if (d < D) xs[f * LD + d] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes, then its reads
        float* ob = out + b * P;
        for (int c0 = 0; c0 < F; c0 += 64) {
            const int c = c0 + lane;
            float xc[DMAX];
#pragma unroll
            for (int d = 0; d < DMAX; ++d) xc[d] = (c < F && d < D) ? xs[c * LD + d] : 0.f;
            const int rmax = min(F - 1, c0 + 63);                 // rows r < c for some lane of this chunk
            for (int r = 0; r < rmax; ++r) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < DMAX; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xb + r * DMAX + d);
                    s += v.x * xc[d] + v.y * xc[d + 1] + v.z * xc[d + 2] + v.w * xc[d + 3];
                }
                if (c > r && c < F) ob[(int64_t)r * F - (int64_t)r * (r + 1) / 2 + (c - r - 1)] = s;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next row overwrites the tile
    }
}

// dx_f[b][:] = sum_{g != f} dout[b][p(min(f,g), max(f,g))] * x_g[b][:]
template <int DMAX>
__global__ void __launch_bounds__(256)
k_ipnn_bwd(const float* const* __restrict__ fields, float* const* __restrict__ dfields, int F, int64_t B, int D,
           const float* __restrict__ dout) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int P = F * (F - 1) / 2;
    float* xb = lds + (size_t)w * (F * DMAX + P);        // x rows, stride DMAX (float4 broadcasts)
    float* gs = xb + F * DMAX;
    for (int64_t b = (int64_t)blockIdx.x * nw + w; b < B; b += (int64_t)gridDim.x * nw) {
        for (int i = lane; i < F * DMAX; i += 64) {
            const int f = i / DMAX, d = i - f * DMAX;
            xb[i] = d < D ? fields[f][b * D + d] : 0.f;
        }
        for (int i = lane; i < P; i += 64) gs[i] = dout[b * P + i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int f0 = 0; f0 < F; f0 += 64) {
            const int f = f0 + lane;
            float acc[DMAX];
#pragma unroll
            for (int d = 0; d < DMAX; ++d) acc[d] = 0.f;
            for (int g = 0; g < F; ++g) {
                const int r = min(f, g), c = max(f, g);
                const float wgt = (f < F && g != f) ? gs[r * F - r * (r + 1) / 2 + (c - r - 1)] : 0.f;
#pragma unroll
                for (int d = 0; d < DMAX; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xb + g * DMAX + d);
                    acc[d] += wgt * v.x; acc[d + 1] += wgt * v.y; acc[d + 2] += wgt * v.z; acc[d + 3] += wgt * v.w;
                }
            }
            if (f < F) {
                float* o = dfields[f] + b * D;
                if ((D & 3) == 0 && ((uintptr_t)dfields[f] & 15) == 0) {
#pragma unroll
                    for (int d = 0; d < DMAX; d += 4)
                        if (d < D) *reinterpret_cast<float4*>(o + d) = make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]);
                } else {
#pragma unroll
                    for (int d = 0; d < DMAX; ++d)
                        if (d < D) o[d] = acc[d];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next row overwrites the tile
    }
}

// Forward on the matrix cores: the pair products of one row b are the strict upper triangle of the Gram matrix
// G = X X^T of its F x D field matrix (F <= 64, D in {4, 8, 12, 16}; one row b per wave per step).
//  * loads: D/4 consecutive lanes read the D/4 float4 pieces of one field's row, so each field row is one coalesced
//    request (one lane per field row -- 64 scattered 16-byte pieces per instruction -- ran the loads at 1.35 TB/s); the
//    next row's pieces are requested before this row is consumed;
//  * the tile goes through LDS once to reach the MFMA operand image.  With v_mfma_f32_32x32x2_f32 (exact fp32) lane l
//    supplies row l%32, k-slot l/32; the contraction order is free as long as A and B agree, so k-slot h of step kk is
//    element h*D/2 + kk: lane (j, h) reads the contiguous half row [h*D/2, (h+1)*D/2) of field 32*I + j, and the A image of
//    row block I IS the B image of column block I -- 2 * D/2 VGPRs feed the three needed tiles (0,0), (0,1), (1,1);
//  * C map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  The row's P results are contiguous in
//    `out`; they are collected in LDS (over the dead input tile) and written as whole 16-byte pieces -- ragged per-pair-row
//    dword stores made the L2 fetch the output lines (0.66 ms instead of 0.60 with everything else equal).
typedef float ipnn_acc16 __attribute__((ext_vector_type(16)));
typedef float ipnn_acc4 __attribute__((ext_vector_type(4)));
// LDS hand-over inside ONE wave (the tiles are wave-private): outstanding LDS operations retired, no vmcnt drain
#define IPNN_WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int D>
__global__ void __launch_bounds__(256)
k_ipnn_fwd_gram(const float* const* __restrict__ fields, int F, int64_t B, int64_t wstride, float* __restrict__ out) {
    constexpr int CPF = D / 4;                                    // float4 pieces per field row
    constexpr int NLD = (64 * CPF + 63) / 64;                     // load instructions per row of b (F <= 64)
    constexpr int XS = D == 16 ? 20 : D == 8 ? 12 : D;            // LDS row stride of the input tile: half-row reads spread over the banks
    constexpr int HD = D / 2;
    extern __shared__ __attribute__((aligned(16))) float ipnn_lds[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int64_t P = (int64_t)F * (F - 1) / 2;
    float* tile = ipnn_lds + (threadIdx.x >> 6) * wstride;        // this wave's input tile, then its P results
    const int64_t nw = (int64_t)gridDim.x * 4;
    const bool two = F > 32;
    // which piece of which field this lane fetches in load step t (same for every row b)
    rn_gcf src[NLD];              // global address space: see common.hpp (no flat loads beside the LDS traffic)
    int dst[NLD];
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
        const int idx = t * 64 + lane, f = idx / CPF, c = idx - f * CPF;
        src[t] = f < F ? (rn_gcf)fields[f] + 4 * c : (rn_gcf)nullptr;
        dst[t] = f * XS + 4 * c;
    }
    ipnn_acc4 nx[NLD];                       // native vectors: arrays of HIP's float4 struct are copied with memcpy and stay in scratch
    int64_t bb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bb < B) {
#pragma unroll
        for (int t = 0; t < NLD; ++t)
            if (src[t]) nx[t] = *reinterpret_cast<const RN_GLOBAL ipnn_acc4*>(src[t] + bb * D);
    }
    for (; bb < B; bb += nw) {
        const int64_t b = __builtin_amdgcn_readfirstlane((int)bb);          // row of this wave, uniform for the compiler too (B < 2^31)
#pragma unroll
        for (int t = 0; t < NLD; ++t)
            if (src[t]) *reinterpret_cast<ipnn_acc4*>(tile + dst[t]) = nx[t];
        if (bb + nw < B) {
#pragma unroll
            for (int t = 0; t < NLD; ++t)
                if (src[t]) nx[t] = *reinterpret_cast<const RN_GLOBAL ipnn_acc4*>(src[t] + (bb + nw) * D);
        }
        IPNN_WAVE_SYNC();                                                   // the tile is private to this wave
        float a0[HD], a1[HD];
#pragma unroll
        for (int k = 0; k < HD; k += 2) {
            const float2 u = *reinterpret_cast<const float2*>(tile + j * XS + h * HD + k);
            a0[k] = u.x; a0[k + 1] = u.y;
            const float2 w = *reinterpret_cast<const float2*>(tile + (32 + j) * XS + h * HD + k);      // garbage rows (f >= F) only reach pairs that are never stored
            a1[k] = w.x; a1[k + 1] = w.y;
        }
        IPNN_WAVE_SYNC();                                                   // operands are in registers: the tile may be overwritten
        ipnn_acc16 g00, g01, g11;
#pragma unroll
        for (int v = 0; v < 16; ++v) { g00[v] = 0.f; g01[v] = 0.f; g11[v] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < HD; ++kk) g00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], a0[kk], g00, 0, 0, 0);
        if (two) {
#pragma unroll
            for (int kk = 0; kk < HD; ++kk) {
                g01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], a1[kk], g01, 0, 0, 0);
                g11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], a1[kk], g11, 0, 0, 0);
            }
        }
        int jo = j, ho = h;                          // opaque per row: keeps the 48 result offsets from being hoisted into
        asm volatile("" : "+v"(jo), "+v"(ho));        // ~80 live VGPRs (occupancy); they cost a few VALU each to recompute
        // pair (r, c), r < c < F, lives at r*F - r*(r+1)/2 + (c - r - 1)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = (v & 3) + 8 * (v >> 2) + 4 * ho;
            const int base = r * F - r * (r + 1) / 2 - r - 1;
            if (j > r && j < F) tile[base + jo] = g00[v];
            if (two && 32 + j < F) tile[base + 32 + jo] = g01[v];
        }
        if (two) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                const int base = r * F - r * (r + 1) / 2 - r - 1;
                if (32 + j > r && 32 + j < F) tile[base + 32 + jo] = g11[v];
            }
        }
        IPNN_WAVE_SYNC();
        float* og = out + b * P;
        if (((P & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
            for (int64_t i = lane * 4; i < P; i += 256) *reinterpret_cast<float4*>(og + i) = *reinterpret_cast<const float4*>(tile + i);
        } else {
            for (int64_t i = lane; i < P; i += 64) og[i] = tile[i];
        }
        IPNN_WAVE_SYNC();                                                   // results read back before the next tile lands
    }
}

// Backward on the matrix cores: dX = (U + U^T) X for the row's F x D field matrix X, U the strict upper triangle
// holding the row's P incoming gradients.  v_mfma_f32_16x16x4_f32: A[i][k] from lane (i = l%16, k = l/16), B[k][d] from
// lane (d = l%16, k = l/16), C rows 4*(l/16) + reg, col l%16 -- D <= 16 is the N dimension, no wasted half tile.
//  * the P gradients stay in their packed order in LDS: U[r][c] = lin[T(r) + c] with T(r) = r*F - r*(r+1)/2 - r - 1.
//    Term U X reads A = U[16I+i][4kk+k'] (per-lane T(16I+i), the k step is an immediate); term U^T X reads
//    A = U[4kk+k'][16I+i] (T(4kk+k') recomputed per k step).  Blocks entirely below the diagonal are skipped at compile
//    time (kk < 4I for U, kk > 4I+3 for U^T): 80 MFMAs of 16x16x4 per row at F = 64; the rest is masked by c > r, c < F.
//  * X goes through LDS as in the forward (coalesced float4 pieces, row stride 16); rows >= F of the tile are zeroed
//    once and never written, so k slots past F contribute exact zeros;
//  * dX leaves through the dead X tile, so every field row is again one coalesced 16*D/4-byte store.
// Both inputs of the next row are requested before this row is consumed.  F <= 64, D in {4, 8, 12, 16}.
// both inputs of row `row` of the backward kernel below -> registers
template <int D>
__device__ __forceinline__ void ipnn_bwd_request(const rn_gcf (&src)[D / 4], ipnn_acc4 (&nx)[D / 4], ipnn_acc4 (&nd)[8], const float* dout,
                                                 int64_t row, int64_t P, int lane, int vec) {
#pragma unroll
    for (int t = 0; t < D / 4; ++t)
        if (src[t]) nx[t] = *reinterpret_cast<const RN_GLOBAL ipnn_acc4*>(src[t] + row * D);
    if (vec) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q * 256 + lane * 4 < P) nd[q] = *reinterpret_cast<const ipnn_acc4*>(dout + row * P + q * 256 + lane * 4);
    }
}
// "these values exist now": keeps the LDS reads of one k step ahead of their masks (see the kernel)
template <int NB>
__device__ __forceinline__ void ipnn_keep(float (&u)[NB], float (&l)[NB]) {
    if constexpr (NB == 1) asm volatile("" : "+v"(u[0]), "+v"(l[0]));
    else if constexpr (NB == 2) asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(l[0]), "+v"(l[1]));
    else if constexpr (NB == 3) asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]));
    else asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]));
}
template <int D, int NB, bool FULL>               // NB = ceil(F / 16): 16-row blocks of dX, and 4*NB k steps; FULL: F == 16*NB
__global__ void __launch_bounds__(256)
k_ipnn_bwd_gram(const float* const* __restrict__ fields, float* const* __restrict__ dfields, int F, int64_t B, int64_t wstride,
                const float* __restrict__ dout, int vec) {
    constexpr int CPF = D / 4;                                    // float4 pieces per field row = load steps per row of b
    constexpr int NQ = 8;                                         // float4 pieces of the packed gradients per lane (P <= 2016)
    extern __shared__ __attribute__((aligned(16))) float ipnn_lds[];
    const int lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int64_t P = (int64_t)F * (F - 1) / 2;
    float* xt = ipnn_lds + (threadIdx.x >> 6) * wstride;          // [64][16] X tile, later dX
    float* lin = xt + 64 * 16;                                    // packed gradients of the row
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int i = lane; i < 64 * 16; i += 64) xt[i] = 0.f;
    rn_gcf src[CPF];
    rn_gf dsrc[CPF];
    int dst[CPF];
#pragma unroll
    for (int t = 0; t < CPF; ++t) {
        const int idx = t * 64 + lane, f = idx / CPF, c = idx - f * CPF;
        src[t] = f < F ? (rn_gcf)fields[f] + 4 * c : (rn_gcf)nullptr;
        dsrc[t] = f < F ? (rn_gf)dfields[f] + 4 * c : (rn_gf)nullptr;
        dst[t] = f * 16 + 4 * c;
    }
    ipnn_acc4 nx[CPF], nd[NQ];             // native vectors: arrays of HIP's float4 struct are copied with memcpy and stay in scratch
    int64_t bb = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bb < B) ipnn_bwd_request<D>(src, nx, nd, dout, bb, P, lane, vec);
    // T(r) of this lane's rows, clamped to a valid row so that masked-out reads stay inside the wave's LDS
    int t1[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        const int r = min(16 * I + i16, F - 1);
        t1[I] = r * F - r * (r + 1) / 2 - r - 1;
    }
    for (; bb < B; bb += nw) {
        const int64_t b = __builtin_amdgcn_readfirstlane((int)bb);          // row of this wave, uniform for the compiler too (B < 2^31)
#pragma unroll
        for (int t = 0; t < CPF; ++t)
            if (src[t]) *reinterpret_cast<ipnn_acc4*>(xt + dst[t]) = nx[t];
        if (vec) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q * 256 + lane * 4 < P) *reinterpret_cast<ipnn_acc4*>(lin + q * 256 + lane * 4) = nd[q];
        } else {
            for (int64_t i = lane; i < P; i += 64) lin[i] = dout[b * P + i];
        }
        if (bb + nw < B) ipnn_bwd_request<D>(src, nx, nd, dout, bb + nw, P, lane, vec);
        IPNN_WAVE_SYNC();
        float bx[4 * NB];
#pragma unroll
        for (int kk = 0; kk < 4 * NB; ++kk) bx[kk] = xt[(4 * kk + kq) * 16 + i16];
        ipnn_acc4 acc[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) acc[I] = ipnn_acc4{0.f, 0.f, 0.f, 0.f};
        int io = i16, ko = kq;                       // opaque per row: the masks and offsets below are row-invariant, and hoisting
        asm volatile("" : "+v"(io), "+v"(ko));        // ~160 of them out of the row loop spills SGPRs and VGPRs
#pragma unroll
        for (int kk = 0; kk < 4 * NB; ++kk) {                               // k slots >= F: zero X rows, masked A
            const int ck = 4 * kk + ko;                                     // this lane's k slot: a column of U, a row of U^T
            const int rk = min(ck, F - 1);
            const int t2 = rk * F - rk * (rk + 1) / 2 - rk - 1;
            // all A values of the k step first, unconditionally (the addresses are clamped into the wave's LDS), then one
            // opaque statement: left alone, the compiler sinks every read under its mask -- a branch, a wait and an
            // accumulator shuffle per MFMA
            float au[NB], al[NB];
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                au[I] = kk >= 4 * I ? lin[t1[I] + ck] : 0.f;                // U X:   A[i][k] = U[16I+i][ck], blocks on/above the diagonal
                al[I] = kk <= 4 * I + 3 ? lin[t2 + 16 * I + io] : 0.f;      // U^T X: A[i][k] = U[ck][16I+i], blocks on/below it
            }
            ipnn_keep<NB>(au, al);
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                const int ri = 16 * I + io;
                // FULL (F == 16 * NB: no ragged edge): only the four k steps whose block straddles the diagonal need a mask
                const bool straddle = kk >= 4 * I && kk <= 4 * I + 3;
                if (kk >= 4 * I) {
                    const float a = (FULL && !straddle) ? au[I] : ((ck > ri && ck < F) ? au[I] : 0.f);
                    acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bx[kk], acc[I], 0, 0, 0);
                }
                if (kk <= 4 * I + 3) {
                    const float a = (FULL && !straddle) ? al[I] : ((ri > ck && ri < F) ? al[I] : 0.f);
                    acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bx[kk], acc[I], 0, 0, 0);
                }
            }
        }
        IPNN_WAVE_SYNC();                                                   // every lane has its B operands: the X tile is dead
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int f = 16 * I + 4 * kq + v;
                if (f < F) xt[f * 16 + i16] = acc[I][v];                    // rows >= F stay zero for the next row's k slots
            }
        IPNN_WAVE_SYNC();
#pragma unroll
        for (int t = 0; t < CPF; ++t)
            if (dsrc[t]) *reinterpret_cast<RN_GLOBAL ipnn_acc4*>(dsrc[t] + b * D) = *reinterpret_cast<const ipnn_acc4*>(xt + dst[t]);
        IPNN_WAVE_SYNC();                                                   // dX read back before the next X tile lands
    }
}

static int ipnn_cfg(int F, int D, bool bwd, int* waves, size_t* lds) {
    const int dmax = D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64;
    const size_t per_wave = ((size_t)F * dmax + (bwd ? (size_t)F * (F - 1) / 2 : (size_t)F * (D + 1))) * sizeof(float);
    int w = 4;
    while (w > 1 && per_wave * w > 64 * 1024) w >>= 1;
    if (per_wave * w > 64 * 1024) return RECNOW_EUNSUPPORTED;
    *waves = w;
    *lds = per_wave * w;
    return RECNOW_OK;
}

extern "C" int recnow_inner_pnn_fwd(const float* const* fields, int F, int64_t B, int D, float* out, void* stream) {
    if (F < 1 || B < 0 || D < 1) return RECNOW_EINVAL;
    if (B == 0 || F == 1) return RECNOW_OK;
    if (!fields || !out) return RECNOW_EINVAL;
    if (D > 64) return RECNOW_EUNSUPPORTED;
    int waves;
    size_t lds;
    hipStream_t st = (hipStream_t)stream;
    if (F <= 64 && (D == 4 || D == 8 || D == 12 || D == 16) && B <= 0x7fffffffll) {
        int64_t gs = (B + 3) / 4;
        if (gs > 16384) gs = 16384;
        // per wave: the input tile (64 rows, padded stride <= 20 floats), later overwritten by the row's P results
        const int64_t P = (int64_t)F * (F - 1) / 2;
        int64_t wstride = 64 * 20;
        if (((P + 3) & ~3ll) > wstride) wstride = (P + 3) & ~3ll;
        const size_t slds = 4 * (size_t)wstride * sizeof(float);
        if (D == 16) hipLaunchKernelGGL(k_ipnn_fwd_gram<16>, (int)gs, 256, slds, st, fields, F, B, wstride, out);
        else if (D == 12) hipLaunchKernelGGL(k_ipnn_fwd_gram<12>, (int)gs, 256, slds, st, fields, F, B, wstride, out);
        else if (D == 8) hipLaunchKernelGGL(k_ipnn_fwd_gram<8>, (int)gs, 256, slds, st, fields, F, B, wstride, out);
        else hipLaunchKernelGGL(k_ipnn_fwd_gram<4>, (int)gs, 256, slds, st, fields, F, B, wstride, out);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    int rc = ipnn_cfg(F, D, false, &waves, &lds);
    if (rc) return rc;
    int64_t g = (B + waves - 1) / waves;
    if (g > 4096) g = 4096;
    if (D <= 8) hipLaunchKernelGGL(k_ipnn_fwd<8>, (int)g, waves * 64, lds, st, fields, F, B, D, out);
    else if (D <= 16) hipLaunchKernelGGL(k_ipnn_fwd<16>, (int)g, waves * 64, lds, st, fields, F, B, D, out);
    else if (D <= 32) hipLaunchKernelGGL(k_ipnn_fwd<32>, (int)g, waves * 64, lds, st, fields, F, B, D, out);
    else hipLaunchKernelGGL(k_ipnn_fwd<64>, (int)g, waves * 64, lds, st, fields, F, B, D, out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_inner_pnn_bwd(const float* const* fields, float* const* dfields, int F, int64_t B, int D, const float* dout,
                                    void* stream) {
    if (F < 1 || B < 0 || D < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dfields || (F > 1 && !dout)) return RECNOW_EINVAL;
    if (D > 64) return RECNOW_EUNSUPPORTED;
    int waves;
    size_t lds;
    hipStream_t st = (hipStream_t)stream;
    if (F >= 2 && F <= 64 && (D == 4 || D == 8 || D == 12 || D == 16) && B <= 0x7fffffffll) {
        int64_t gs = (B + 3) / 4;
        if (gs > 16384) gs = 16384;
        const int64_t P = (int64_t)F * (F - 1) / 2;
        // per wave: the [64][16] X tile, then the packed gradients (+64: masked-out reads of the last rows stay inside)
        const int64_t wstride = 64 * 16 + ((P + 64 + 3) & ~3ll);
        const size_t slds = 4 * (size_t)wstride * sizeof(float);
        const int vec = (P & 3) == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0;       // rows of dout are float4-addressable
#define IPNN_BWD_LAUNCH(DD, NBB)                                                                                         \
    do {                                                                                                                 \
        if (F == 16 * NBB) hipLaunchKernelGGL((k_ipnn_bwd_gram<DD, NBB, true>), (int)gs, 256, slds, st, fields, dfields, F, B, wstride, dout, vec); \
        else hipLaunchKernelGGL((k_ipnn_bwd_gram<DD, NBB, false>), (int)gs, 256, slds, st, fields, dfields, F, B, wstride, dout, vec);             \
    } while (0)
#define IPNN_BWD_LAUNCH_D(DD)                                                                                            \
    do {                                                                                                                 \
        if (F <= 16) IPNN_BWD_LAUNCH(DD, 1);                                                                             \
        else if (F <= 32) IPNN_BWD_LAUNCH(DD, 2);                                                                        \
        else if (F <= 48) IPNN_BWD_LAUNCH(DD, 3);                                                                        \
        else IPNN_BWD_LAUNCH(DD, 4);                                                                                     \
    } while (0)
        if (D == 16) IPNN_BWD_LAUNCH_D(16);
        else if (D == 12) IPNN_BWD_LAUNCH_D(12);
        else if (D == 8) IPNN_BWD_LAUNCH_D(8);
        else IPNN_BWD_LAUNCH_D(4);
#undef IPNN_BWD_LAUNCH_D
#undef IPNN_BWD_LAUNCH
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    int rc = ipnn_cfg(F, D, true, &waves, &lds);
    if (rc) return rc;
    int64_t g = (B + waves - 1) / waves;
    if (g > 4096) g = 4096;
    if (D <= 8) hipLaunchKernelGGL(k_ipnn_bwd<8>, (int)g, waves * 64, lds, st, fields, dfields, F, B, D, dout);
    else if (D <= 16) hipLaunchKernelGGL(k_ipnn_bwd<16>, (int)g, waves * 64, lds, st, fields, dfields, F, B, D, dout);
    else if (D <= 32) hipLaunchKernelGGL(k_ipnn_bwd<32>, (int)g, waves * 64, lds, st, fields, dfields, F, B, D, dout);
    else hipLaunchKernelGGL(k_ipnn_bwd<64>, (int)g, waves * 64, lds, st, fields, dfields, F, B, D, dout);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// SENET.  Fields may have different widths: dims[f], offs[f] (column of field f in the concatenation) are DEVICE int32
// arrays.
//   squeeze:   sq[b][f]  = mean_d x_f[b][d]                                   (:104-107)
//   scale:     out[b][offs[f]+d] = x_f[b][d] * w[b][f]                        (:112-117)
//   backward:  dw[b][f] = sum_d dout[b][offs[f]+d] * x_f[b][d];   dx_f[b][d] = dout * w[b][f] + dsq[b][f] / D_f
// ---------------------------------------------------------------------------------------------------------------------
// A workgroup owns R consecutive rows.  Every field's R x D_f block is CONTIGUOUS in its (B, D_f) tensor and every R-row
// slab of the concatenated (B, total) tensors is contiguous too, so all global traffic is long coalesced runs; the
// re-arrangement between the two layouts goes through an LDS tile [R][total].
//   squeeze: tile <- fields;  sq[b][f] = mean over the tile columns of f
//   scale:   tile <- fields * w[b][f];  out slab <- tile
//   dw:      tile <- dout slab;  dw[b][f] = <tile columns of f, x_f[b]>
//   dx:      tile <- dout slab;  dx_f block <- tile * w + dsq / D_f
#define SENET_MODE_SQUEEZE 0
#define SENET_MODE_SCALE 1
#define SENET_MODE_DW 2
#define SENET_MODE_DX 3
template <int MODE>
__global__ void __launch_bounds__(256)
k_senet_tile(const float* const* __restrict__ fields, float* const* __restrict__ dfields, const int* __restrict__ dims,
             const int* __restrict__ offs, int F, int total, int64_t B, int R, const float* __restrict__ w,
             const float* __restrict__ in, const float* __restrict__ dsq, float* __restrict__ out) {
    extern __shared__ float tile[];                  // [R][total + 1]
    const int LDT = total + 1;
    const int tid = threadIdx.x;
    for (int64_t b0 = (int64_t)blockIdx.x * R; b0 < B; b0 += (int64_t)gridDim.x * R) {
        const int rows = (int)min((int64_t)R, B - b0);
        __syncthreads();
        if (MODE == SENET_MODE_DW || MODE == SENET_MODE_DX) {          // dout slab -> tile
            const float* src = in + b0 * total;
            for (int i = tid; i < rows * total; i += 256) tile[(i / total) * LDT + (i % total)] = src[i];
            __syncthreads();
        }
        for (int f = 0; f < F; ++f) {
            const int D = dims[f], o = offs[f];
            const int n = rows * D;
            if (MODE == SENET_MODE_SQUEEZE) {
                const float* x = fields[f] + b0 * D;
                for (int i = tid; i < n; i += 256) tile[(i / D) * LDT + o + (i % D)] = x[i];
            } else if (MODE == SENET_MODE_SCALE) {
                const float* x = fields[f] + b0 * D;
                for (int i = tid; i < n; i += 256) {
                    const int r = i / D;
                    tile[r * LDT + o + (i % D)] = x[i] * w[(b0 + r) * F + f];
                }
            } else if (MODE == SENET_MODE_DW) {                          // tile <- tile * x (products), summed below
                const float* x = fields[f] + b0 * D;
                for (int i = tid; i < n; i += 256) tile[(i / D) * LDT + o + (i % D)] *= x[i];
            } else {                                                     // DX: straight to the field gradient block
                float* dx = dfields[f] + b0 * D;
                for (int i = tid; i < n; i += 256) {
                    const int r = i / D;
                    dx[i] = tile[r * LDT + o + (i % D)] * w[(b0 + r) * F + f] + dsq[(b0 + r) * F + f] / (float)D;
                }
            }
        }
        if (MODE == SENET_MODE_DX) continue;
        __syncthreads();
        if (MODE == SENET_MODE_SCALE) {                                  // tile -> out slab
            float* dst = out + b0 * total;
            for (int i = tid; i < rows * total; i += 256) dst[i] = tile[(i / total) * LDT + (i % total)];
        } else {                                                         // per (row, field) sums over the field's columns
            for (int i = tid; i < rows * F; i += 256) {
                const int r = i / F, f = i - r * F;
                const int D = dims[f], o = offs[f];
                float s = 0.f;
                for (int d = 0; d < D; ++d) s += tile[r * LDT + o + d];
                out[(b0 + r) * F + f] = MODE == SENET_MODE_SQUEEZE ? s / (float)D : s;
            }
        }
    }
}
// Fast path for the usual case of equal field widths D with D % 4 == 0 and D/4 a power of two: thread = (group of
// SENET_RU consecutive rows, float4 q of the concatenated row).  Every access is a float4; the concatenated tensors are
// touched in whole-row runs, and the SENET_RU rows of a field -- SENET_RU*4*D contiguous bytes of its tensor -- are read
// (written) by the same wave, so field lines are not split between workgroups on different XCDs (one row per thread left
// every 128-byte field line half-used by two L2s).  Nothing is staged and the SENET_RU loads of a thread are independent.
// (row, field) sums reduce over the D/4 adjacent lanes of the field with shuffles (groups are lane-aligned because total/4
// is a multiple of D/4).
#define SENET_RU 4
typedef float senet_f4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256)
k_senet_uniform(const float* const* __restrict__ fields, float* const* __restrict__ dfields, int F, int D, int64_t B,
                const float* __restrict__ w, const float* __restrict__ in, const float* __restrict__ dsq, float* __restrict__ out) {
    const int Q = D / 4, QT = F * Q;                 // float4s per field row / per concatenated row
    const int64_t n = (B + SENET_RU - 1) / SENET_RU * QT;       // (row group, q) items
    const int64_t nround = (n + 255) / 256 * 256;    // whole waves stay in the loop: the shuffles need every lane
    const int64_t FD = (int64_t)F * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nround; i += (int64_t)gridDim.x * 256) {
        const bool ok = i < n;
        const int64_t grp = ok ? i / QT : 0;
        const int q = ok ? (int)(i - grp * QT) : 0;
        const int f = q / Q, d = (q - f * Q) * 4;
        const int64_t b0 = grp * SENET_RU;
        const rn_gcf xf = (rn_gcf)fields[f] + d;
        senet_f4 v[SENET_RU], g[SENET_RU];
        bool live[SENET_RU];
#pragma unroll
        for (int j = 0; j < SENET_RU; ++j) {
            live[j] = ok && b0 + j < B;
            v[j] = senet_f4{0.f, 0.f, 0.f, 0.f};
            g[j] = v[j];
            if (MODE != SENET_MODE_DX && live[j]) v[j] = *reinterpret_cast<const RN_GLOBAL senet_f4*>(xf + (b0 + j) * D);
            if ((MODE == SENET_MODE_DX || MODE == SENET_MODE_DW) && live[j]) g[j] = *reinterpret_cast<const senet_f4*>(in + (b0 + j) * FD + q * 4);
        }
#pragma unroll
        for (int j = 0; j < SENET_RU; ++j) {
            const int64_t b = b0 + j;
            if (MODE == SENET_MODE_SCALE) {
                if (live[j]) *reinterpret_cast<senet_f4*>(out + b * FD + q * 4) = v[j] * w[b * F + f];
            } else if (MODE == SENET_MODE_DX) {
                if (live[j]) *reinterpret_cast<RN_GLOBAL senet_f4*>((rn_gf)dfields[f] + b * D + d) = g[j] * w[b * F + f] + dsq[b * F + f] / (float)D;
            } else {
                float s = MODE == SENET_MODE_DW ? (v[j].x * g[j].x + v[j].y * g[j].y) + (v[j].z * g[j].z + v[j].w * g[j].w)
                                                : (v[j].x + v[j].y) + (v[j].z + v[j].w);
                for (int o = Q / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (live[j] && d == 0) out[b * F + f] = MODE == SENET_MODE_SQUEEZE ? s / (float)D : s;
            }
        }
    }
}
// uniform_d: the caller's promise that every field is D wide and every field / concatenated tensor is 16-byte aligned
static inline int senet_uniform_ok(int D, int F, int total) {
    if (D < 4 || (D & 3) || D * F != total) return 0;
    const int q = D / 4;
    return (q & (q - 1)) == 0 && q <= 64 ? D : 0;
}
static int senet_launch(int mode, const float* const* fields, float* const* dfields, const int32_t* dims, const int32_t* offs, int F,
                        int total, int64_t B, const float* w, const float* in, const float* dsq, float* out, int uniform_d, hipStream_t st) {
    if (uniform_d > 0) {
        const int64_t n = (B + SENET_RU - 1) / SENET_RU * (total / 4);       // (row group, float4 of the concatenated row) items
        int64_t g = (n + 255) / 256;
        if (g > 16384) g = 16384;
        const int D = uniform_d;
        switch (mode) {
            case SENET_MODE_SQUEEZE: hipLaunchKernelGGL(k_senet_uniform<SENET_MODE_SQUEEZE>, (int)g, 256, 0, st, fields, dfields, F, D, B, w, in, dsq, out); break;
            case SENET_MODE_SCALE: hipLaunchKernelGGL(k_senet_uniform<SENET_MODE_SCALE>, (int)g, 256, 0, st, fields, dfields, F, D, B, w, in, dsq, out); break;
            case SENET_MODE_DW: hipLaunchKernelGGL(k_senet_uniform<SENET_MODE_DW>, (int)g, 256, 0, st, fields, dfields, F, D, B, w, in, dsq, out); break;
            default: hipLaunchKernelGGL(k_senet_uniform<SENET_MODE_DX>, (int)g, 256, 0, st, fields, dfields, F, D, B, w, in, dsq, out); break;
        }
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    int R = (int)((48 * 1024) / ((size_t)(total + 1) * sizeof(float)));
    if (R < 1) return RECNOW_EUNSUPPORTED;            // one row of the concatenation must fit in LDS (total <= 12287)
    if (R > 32) R = 32;
    const size_t lds = (size_t)R * (total + 1) * sizeof(float);
    int64_t g = (B + R - 1) / R;
    if (g > 4096) g = 4096;
    switch (mode) {
        case SENET_MODE_SQUEEZE: hipLaunchKernelGGL(k_senet_tile<SENET_MODE_SQUEEZE>, (int)g, 256, lds, st, fields, dfields, dims, offs, F, total, B, R, w, in, dsq, out); break;
        case SENET_MODE_SCALE: hipLaunchKernelGGL(k_senet_tile<SENET_MODE_SCALE>, (int)g, 256, lds, st, fields, dfields, dims, offs, F, total, B, R, w, in, dsq, out); break;
        case SENET_MODE_DW: hipLaunchKernelGGL(k_senet_tile<SENET_MODE_DW>, (int)g, 256, lds, st, fields, dfields, dims, offs, F, total, B, R, w, in, dsq, out); break;
        default: hipLaunchKernelGGL(k_senet_tile<SENET_MODE_DX>, (int)g, 256, lds, st, fields, dfields, dims, offs, F, total, B, R, w, in, dsq, out); break;
    }
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_senet_squeeze(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                                    float* sq, int uniform_d, void* stream) {
    if (F < 1 || B < 0 || total < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dims || !offs || !sq) return RECNOW_EINVAL;
    return senet_launch(SENET_MODE_SQUEEZE, fields, nullptr, dims, offs, F, total, B, nullptr, nullptr, nullptr, sq, senet_uniform_ok(uniform_d, F, total), (hipStream_t)stream);
}
extern "C" int recnow_senet_scale_fwd(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                                      const float* w, float* out, int uniform_d, void* stream) {
    if (F < 1 || B < 0 || total < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dims || !offs || !w || !out) return RECNOW_EINVAL;
    return senet_launch(SENET_MODE_SCALE, fields, nullptr, dims, offs, F, total, B, w, nullptr, nullptr, out, senet_uniform_ok(uniform_d, F, total), (hipStream_t)stream);
}
extern "C" int recnow_senet_scale_bwd_w(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total,
                                        int64_t B, const float* dout, float* dw, int uniform_d, void* stream) {
    if (F < 1 || B < 0 || total < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dims || !offs || !dout || !dw) return RECNOW_EINVAL;
    return senet_launch(SENET_MODE_DW, fields, nullptr, dims, offs, F, total, B, nullptr, dout, nullptr, dw, senet_uniform_ok(uniform_d, F, total), (hipStream_t)stream);
}
extern "C" int recnow_senet_scale_bwd_x(float* const* dfields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                                        const float* w, const float* dout, const float* dsq, int uniform_d, void* stream) {
    if (F < 1 || B < 0 || total < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!dfields || !dims || !offs || !w || !dout || !dsq) return RECNOW_EINVAL;
    return senet_launch(SENET_MODE_DX, nullptr, dfields, dims, offs, F, total, B, w, dout, dsq, nullptr, senet_uniform_ok(uniform_d, F, total), (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// attention_by_dot_product (attention.py:12-38): one wave per row b, lane <-> embedding column d (chunks of 64).
//   s_l = <u[b][l], doc[b]>  [max(., 0) if filter_neg];  mat[b][d] = sum_l u[b][l][d] * s_l;  sum[b] = sum_l s_l
// backward (s recomputed):  ds_l = <dmat[b], u[b][l]> + dsum[b]  [* (raw s_l > 0)]
//   du[b][l][d] = dmat[b][d] * s_l + ds_l * doc[b][d];   ddoc[b][d] = sum_l ds_l * u[b][l][d]
// HBM-bound: 4*B*L*D bytes forward, 8*B*L*D backward.
// ---------------------------------------------------------------------------------------------------------------------
#define ATTN_NCH 4      // 64-wide chunks of D kept in registers when one wave serves one row (64 < D <= 256)
// GS lanes serve one row (GS = 16 / 32 / 64 for D <= 16 / 32 / more): a wave streams 64/GS rows at once, the per-row dot
// products reduce inside the lane group with log2(GS) shuffles.
template <int GS>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = GS / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int GS>
__global__ void __launch_bounds__(256)
k_attn_dot_fwd(const float* __restrict__ user, const float* __restrict__ doc, int64_t B, int L, int D, int filter_neg,
               float* __restrict__ mat, float* __restrict__ ssum) {
    constexpr int RPW = 64 / GS, NCH = GS == 64 ? ATTN_NCH : 1;
    const int lane = threadIdx.x & 63, gl = lane % GS, gr = lane / GS;
    const int64_t nrg = (B + RPW - 1) / RPW;                    // row groups; every lane of a wave runs the same trip counts
    for (int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); rg < nrg; rg += (int64_t)gridDim.x * 4) {
        const int64_t b = rg * RPW + gr;
        const bool ok = b < B;
        const float* u = user + (ok ? b : 0) * L * D;
        float dc[NCH], acc[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int d = j * 64 + gl;
            dc[j] = (ok && d < D) ? doc[b * D + d] : 0.f;
            acc[j] = 0.f;
        }
        float tot = 0.f;
        for (int l = 0; l < L; ++l) {
            float uv[NCH], p = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int d = j * 64 + gl;
                uv[j] = (ok && d < D) ? u[(int64_t)l * D + d] : 0.f;
                p += uv[j] * dc[j];
            }
            float sc = group_sum<GS>(p);
            if (filter_neg) sc = fmaxf(sc, 0.f);
            tot += sc;
#pragma unroll
            for (int j = 0; j < NCH; ++j) acc[j] += uv[j] * sc;
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int d = j * 64 + gl;
            if (ok && d < D) mat[b * D + d] = acc[j];
        }
        if (ok && gl == 0) ssum[b] = tot;
    }
}
template <int GS>
__global__ void __launch_bounds__(256)
k_attn_dot_bwd(const float* __restrict__ user, const float* __restrict__ doc, const float* __restrict__ dmat,
               const float* __restrict__ dsum, int64_t B, int L, int D, int filter_neg, float* __restrict__ duser,
               float* __restrict__ ddoc) {
    constexpr int RPW = 64 / GS, NCH = GS == 64 ? ATTN_NCH : 1;
    const int lane = threadIdx.x & 63, gl = lane % GS, gr = lane / GS;
    const int64_t nrg = (B + RPW - 1) / RPW;
    for (int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); rg < nrg; rg += (int64_t)gridDim.x * 4) {
        const int64_t b = rg * RPW + gr;
        const bool ok = b < B;
        const float* u = user + (ok ? b : 0) * L * D;
        float* du = duser + (ok ? b : 0) * L * D;
        const float gs = (ok && dsum) ? dsum[b] : 0.f;
        float dc[NCH], gm[NCH], acc[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int d = j * 64 + gl;
            dc[j] = (ok && d < D) ? doc[b * D + d] : 0.f;
            gm[j] = (ok && d < D && dmat) ? dmat[b * D + d] : 0.f;
            acc[j] = 0.f;
        }
        for (int l = 0; l < L; ++l) {
            float uv[NCH], p = 0.f, q = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int d = j * 64 + gl;
                uv[j] = (ok && d < D) ? u[(int64_t)l * D + d] : 0.f;
                p += uv[j] * dc[j];
                q += uv[j] * gm[j];
            }
            const float sraw = group_sum<GS>(p);
            float ds = group_sum<GS>(q) + gs;
            float sc = sraw;
            if (filter_neg) {
                sc = fmaxf(sraw, 0.f);
                if (!(sraw > 0.f)) ds = 0.f;
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int d = j * 64 + gl;
                if (ok && d < D) du[(int64_t)l * D + d] = gm[j] * sc + ds * dc[j];
                acc[j] += ds * uv[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int d = j * 64 + gl;
            if (ok && d < D) ddoc[b * D + d] = acc[j];
        }
    }
}
static inline int attn_grid(int64_t B, int gs) {
    const int64_t nrg = (B + (64 / gs) - 1) / (64 / gs);
    int64_t g = (nrg + 3) / 4;
    if (g > 8192) g = 8192;
    return (int)(g > 0 ? g : 1);
}
// float4 variants for D % 4 == 0, CPL = D/4 in {1, 2, 4, 8, 16}: 16 lanes per batch row (4 rows per wave) walk the row's
// L*D contiguous floats as float4 pieces -- 256 contiguous bytes per row and load instruction, ATTN_V4 independent loads in
// flight per lane -- so piece k = lane + 16*i belongs to position l = k / CPL and columns 4*(k % CPL)..+3, and k % CPL is the
// same for every piece of a lane.  A score needs a reduce over only the CPL lanes of one position (log2 CPL shuffles, not
// log2 D), the column sums over positions are reduced once at the end.  (One dword per lane and position, with a full
// shuffle tree per position, ran at 2.9 TB/s.)
#define ATTN_V4 4
typedef float attn_f4 __attribute__((ext_vector_type(4)));
template <int CPL, bool BWD>
__global__ void __launch_bounds__(256)
k_attn_dot_v4(const float* __restrict__ user, const float* __restrict__ doc, const float* __restrict__ dmat,
              const float* __restrict__ dsum, int64_t B, int L, int filter_neg, float* __restrict__ out_vec /* mat | ddoc */,
              float* __restrict__ ssum, float* __restrict__ duser) {
    constexpr int D = 4 * CPL;
    const int lane = threadIdx.x & 63, gl = lane & 15, gr = lane >> 4;
    const int c = gl % CPL;
    const int64_t nrg = (B + 3) / 4;
    const int nv = L * CPL;                                     // float4 pieces per row
    for (int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); rg < nrg; rg += (int64_t)gridDim.x * 4) {
        const int64_t b = rg * 4 + gr;
        const bool ok = b < B;
        const attn_f4* u4 = reinterpret_cast<const attn_f4*>(user + (ok ? b : 0) * (int64_t)L * D);
        attn_f4* du4 = BWD ? reinterpret_cast<attn_f4*>(duser + (ok ? b : 0) * (int64_t)L * D) : nullptr;
        const attn_f4 zero = {0.f, 0.f, 0.f, 0.f};
        const attn_f4 dcv = ok ? *reinterpret_cast<const attn_f4*>(doc + b * D + 4 * c) : zero;
        const attn_f4 gmv = (BWD && ok && dmat) ? *reinterpret_cast<const attn_f4*>(dmat + b * D + 4 * c) : zero;
        const float gs = (BWD && ok && dsum) ? dsum[b] : 0.f;
        attn_f4 acc = zero;
        float tot = 0.f;
        for (int k0 = 0; k0 < nv; k0 += 16 * ATTN_V4) {
            attn_f4 uv[ATTN_V4];
#pragma unroll
            for (int i = 0; i < ATTN_V4; ++i) {
                const int k = k0 + 16 * i + gl;
                // (non-temporal streams, round 4: 0.297 -> 0.261 ms fwd+bwd at B 131 072, L 50, D 16; the same change made SENET 11 % and InnerPNN 5 % SLOWER
                // -- their backward passes re-read what the forward streamed -- and was not kept there)
                uv[i] = (ok && k < nv) ? RN_LD_STREAM(u4 + k) : zero;
            }
#pragma unroll
            for (int i = 0; i < ATTN_V4; ++i) {
                const int k = k0 + 16 * i + gl;
                float p = (uv[i].x * dcv.x + uv[i].y * dcv.y) + (uv[i].z * dcv.z + uv[i].w * dcv.w);
                float q = BWD ? (uv[i].x * gmv.x + uv[i].y * gmv.y) + (uv[i].z * gmv.z + uv[i].w * gmv.w) : 0.f;
#pragma unroll
                for (int o = CPL / 2; o > 0; o >>= 1) {
                    p += __shfl_xor(p, o, 64);
                    if (BWD) q += __shfl_xor(q, o, 64);
                }
                float sc = p;
                if (filter_neg) sc = fmaxf(p, 0.f);
                if (!BWD) {
                    if (c == 0) tot += sc;                      // pieces past the row are zero: they add nothing
                    acc += uv[i] * sc;
                } else {
                    float ds = q + gs;
                    if (filter_neg && !(p > 0.f)) ds = 0.f;
                    if (ok && k < nv) RN_ST_STREAM(du4 + k, gmv * sc + dcv * ds);
                    acc += uv[i] * ds;
                }
            }
        }
        // columns 4c..4c+3 were accumulated by the LPR lanes c, c + CPL, ...: add them up (and the score total of the c == 0 lanes)
#pragma unroll
        for (int o = CPL; o < 16; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
            if (!BWD) tot += __shfl_xor(tot, o, 64);
        }
        if (ok && gl < CPL) *reinterpret_cast<attn_f4*>(out_vec + b * D + 4 * c) = acc;
        if (!BWD && ok && gl == 0) ssum[b] = tot;
    }
}
static inline int attn_cpl(int D, const void* a, const void* b2, const void* c2, const void* d2) {
    if (D % 4 || ((((uintptr_t)a | (uintptr_t)b2 | (uintptr_t)c2 | (uintptr_t)d2) & 15) != 0)) return 0;
    const int cpl = D / 4;
    return (cpl == 1 || cpl == 2 || cpl == 4 || cpl == 8 || cpl == 16) ? cpl : 0;
}
#define ATTN_V4_LAUNCH(BWD, ...)                                                                                          \
    switch (cpl) {                                                                                                       \
        case 1: hipLaunchKernelGGL((k_attn_dot_v4<1, BWD>), attn_grid(B, 16), 256, 0, st, __VA_ARGS__); break;           \
        case 2: hipLaunchKernelGGL((k_attn_dot_v4<2, BWD>), attn_grid(B, 16), 256, 0, st, __VA_ARGS__); break;           \
        case 4: hipLaunchKernelGGL((k_attn_dot_v4<4, BWD>), attn_grid(B, 16), 256, 0, st, __VA_ARGS__); break;           \
        case 8: hipLaunchKernelGGL((k_attn_dot_v4<8, BWD>), attn_grid(B, 16), 256, 0, st, __VA_ARGS__); break;           \
        default: hipLaunchKernelGGL((k_attn_dot_v4<16, BWD>), attn_grid(B, 16), 256, 0, st, __VA_ARGS__); break;         \
    }
extern "C" int recnow_attention_dot_fwd(const float* user, const float* doc, int64_t B, int L, int D, int filter_neg, float* mat,
                                        float* score_sum, void* stream) {
    if (B < 0 || L < 0 || D < 1) return RECNOW_EINVAL;
    if (D > 64 * ATTN_NCH) return RECNOW_EUNSUPPORTED;
    if (B == 0) return RECNOW_OK;
    if ((L > 0 && !user) || !doc || !mat || !score_sum) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (const int cpl = attn_cpl(D, user, doc, mat, nullptr)) {
        ATTN_V4_LAUNCH(false, user, doc, (const float*)nullptr, (const float*)nullptr, B, L, filter_neg, mat, score_sum, (float*)nullptr);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (D <= 16) hipLaunchKernelGGL(k_attn_dot_fwd<16>, attn_grid(B, 16), 256, 0, st, user, doc, B, L, D, filter_neg, mat, score_sum);
    else if (D <= 32) hipLaunchKernelGGL(k_attn_dot_fwd<32>, attn_grid(B, 32), 256, 0, st, user, doc, B, L, D, filter_neg, mat, score_sum);
    else hipLaunchKernelGGL(k_attn_dot_fwd<64>, attn_grid(B, 64), 256, 0, st, user, doc, B, L, D, filter_neg, mat, score_sum);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_attention_dot_bwd(const float* user, const float* doc, const float* dmat, const float* dsum, int64_t B, int L,
                                        int D, int filter_neg, float* duser, float* ddoc, void* stream) {
    if (B < 0 || L < 0 || D < 1) return RECNOW_EINVAL;
    if (D > 64 * ATTN_NCH) return RECNOW_EUNSUPPORTED;
    if (B == 0) return RECNOW_OK;
    if ((L > 0 && (!user || !duser)) || !doc || !ddoc) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (const int cpl = attn_cpl(D, user, doc, dmat, (const void*)((uintptr_t)duser | (uintptr_t)ddoc))) {
        ATTN_V4_LAUNCH(true, user, doc, dmat, dsum, B, L, filter_neg, ddoc, (float*)nullptr, duser);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    if (D <= 16) hipLaunchKernelGGL(k_attn_dot_bwd<16>, attn_grid(B, 16), 256, 0, st, user, doc, dmat, dsum, B, L, D, filter_neg, duser, ddoc);
    else if (D <= 32) hipLaunchKernelGGL(k_attn_dot_bwd<32>, attn_grid(B, 32), 256, 0, st, user, doc, dmat, dsum, B, L, D, filter_neg, duser, ddoc);
    else hipLaunchKernelGGL(k_attn_dot_bwd<64>, attn_grid(B, 64), 256, 0, st, user, doc, dmat, dsum, B, L, D, filter_neg, duser, ddoc);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// focal_crossentropy_loss (focal_loss.py:48-66), z = label, x = logit, p = sigmoid(x):
//   ce = max(x,0) - x z + log1p(exp(-|x|));  af = z a + (1-z)(1-a)  [alpha on];  sim = z p + (1-z)(1-p);
//   mod = (1 - sim)^gamma  [gamma on];   loss = af * mod * ce   (mean over B in a fixed order: double block partials)
//   d loss/dx = af * (mod * (p - z) + [!stop_weight_gradient] ce * gamma (1-sim)^(gamma-1) * -(2z-1) p (1-p))
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void focal_terms(float z, float x, float alpha, float gamma, float& ce, float& af, float& mod, float& p,
                                            float& one_m_sim) {
    ce = fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
    af = alpha > 0.f ? z * alpha + (1.f - z) * (1.f - alpha) : 1.f;
    p = rn_sigmoid(x);
    one_m_sim = 1.f - (z * p + (1.f - z) * (1.f - p));
    mod = gamma == 2.f ? one_m_sim * one_m_sim : gamma == 1.f ? one_m_sim : gamma > 0.f ? powf(one_m_sim, gamma) : 1.f;   // gamma = 2 is the default
}
__global__ void __launch_bounds__(256)
k_focal_fwd(const float* __restrict__ labels, const float* __restrict__ logits, int64_t B, float alpha, float gamma,
            float* __restrict__ elem, double* __restrict__ part) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
        float ce, af, mod, p, oms;
        focal_terms(labels[i], logits[i], alpha, gamma, ce, af, mod, p, oms);
        const float v = mod * (af * ce);
        if (elem) elem[i] = v;
        s += (double)v;
    }
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0 && part) part[blockIdx.x] = s;
}
// fixed order: thread t adds partials t, t+256, ..., then the 256 sums are added in thread order (one thread walking all
// partials was a chain of ~1000 dependent loads: 55 us, a third of the whole loss)
__global__ void __launch_bounds__(256)
k_focal_mean(const double* __restrict__ part, int n, int64_t B, float* __restrict__ mean) {
    __shared__ double sums[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    sums[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 256; ++i) t += sums[i];
        *mean = (float)(t / (double)B);
    }
}
__global__ void __launch_bounds__(256)
k_focal_bwd(const float* __restrict__ labels, const float* __restrict__ logits, int64_t B, float alpha, float gamma, int stop_w,
            const float* __restrict__ gelem, const float* __restrict__ gscalar, float scale, float* __restrict__ dlogits) {
    const float gsc = gscalar ? *gscalar * scale : scale;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
        const float z = labels[i];
        float ce, af, mod, p, oms;
        focal_terms(z, logits[i], alpha, gamma, ce, af, mod, p, oms);
        float d = mod * (p - z);
        if (gamma > 0.f && !stop_w) {
            // d mod/dx = gamma (1-sim)^(gamma-1) * d(1-sim)/dx,  d(1-sim)/dx = -(2z-1) p (1-p)
            const float pw = gamma == 2.f ? oms : gamma == 1.f ? 1.f : (oms > 0.f || gamma >= 1.f) ? powf(oms, gamma - 1.f) : 0.f;
            d += ce * gamma * pw * (-(2.f * z - 1.f) * p * (1.f - p));
        }
        dlogits[i] = af * d * (gelem ? gelem[i] * gsc : gsc);
    }
}
static inline int focal_grid(int64_t B) {
    int64_t g = (B + 255) / 256;
    if (g > 1024) g = 1024;
    return (int)(g > 0 ? g : 1);
}
extern "C" size_t recnow_focal_loss_workspace_bytes(int64_t B) { return (size_t)focal_grid(B) * sizeof(double) + 256; }
extern "C" int recnow_focal_loss_fwd(const float* labels, const float* logits, int64_t B, float alpha, float gamma, float* loss_elem,
                                     float* loss_mean, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || (alpha > 0.f && alpha >= 1.f) || gamma < 0.f) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (loss_mean) RN_HIP(hipMemsetAsync(loss_mean, 0xff, sizeof(float), st));      // mean of nothing: NaN, as tf.reduce_mean
        return RECNOW_OK;
    }
    if (!labels || !logits || (!loss_elem && !loss_mean)) return RECNOW_EINVAL;
    if (loss_mean && (!ws || ws_bytes < recnow_focal_loss_workspace_bytes(B))) return RECNOW_EWORKSPACE;
    const int g = focal_grid(B);
    hipLaunchKernelGGL(k_focal_fwd, g, 256, 0, st, labels, logits, B, alpha, gamma, loss_elem, loss_mean ? (double*)ws : nullptr);
    RN_LAUNCH_CHECK();
    if (loss_mean) {
        hipLaunchKernelGGL(k_focal_mean, 1, 256, 0, st, (const double*)ws, g, B, loss_mean);
        RN_LAUNCH_CHECK();
    }
    return RECNOW_OK;
}
extern "C" int recnow_focal_loss_bwd(const float* labels, const float* logits, int64_t B, float alpha, float gamma,
                                     int stop_weight_gradient, const float* gelem, const float* gscalar, float scale, float* dlogits,
                                     void* stream) {
    if (B < 0 || gamma < 0.f) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!labels || !logits || !dlogits) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_focal_bwd, focal_grid(B), 256, 0, (hipStream_t)stream, labels, logits, B, alpha, gamma, stop_weight_gradient,
                       gelem, gscalar, scale, dlogits);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
