// DCNLayer (DCN-v1 cross network, reference variant WITHOUT residual), /root/reference/rec_now/layers/dcn_layer.py:79-103:
//     x_{l+1} = act( x0 * (x_l . w_l) + b_l ),   l = 0 .. L-1
// HBM-bound.  All L layers are fused: a row of x0 lives in the registers of TPR threads (one wave, or one 256-thread
// workgroup for wide rows), every layer is a dot-reduce + elementwise update on registers, so forward moves 8*B*D
// bytes (read x0, write y) and backward 16*B*D (read x0, dy; write dx) + tiny weight-gradient slabs.
// Backward needs only the L scalars c_l = x_l . w_l per row (csave, optional output of the forward): x_l is then elementwise
// in x0.  Without csave it recomputes x_l from x0 for each layer (O(L^2) dot-reduces on registers, L <= 4).
// Weight/bias gradients accumulate in registers over the rows a thread group owns, are combined per workgroup through
// LDS and finally summed over workgroups by the deterministic column-sum (no float atomics).
#include "gemm.hpp"

#define DCN_MAX_L 4

template <int TPR>
__device__ __forceinline__ float row_sum(float v, float* red /* [4]: one slot per wave of the workgroup */) {
    v = wave_sum(v);
    if (TPR == 64) return v;
    // TPR / 64 waves share one row (256 threads: 4 waves = 1 row, 128 threads: 2 rows of 2 waves)
    constexpr int NWR = TPR / 64;
    const int w = threadIdx.x >> 6, first = (w / NWR) * NWR;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NWR; ++i) s += red[first + i];
    return s;
}

template <int TPR, int VEC, int NV>
struct RowRegs {
    float v[NV][VEC];
};

template <int TPR, int VEC, int NV>
__device__ __forceinline__ void row_load(float (&r)[NV][VEC], const float* __restrict__ p, int D, int t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = (i * TPR + t) * VEC;
        if (VEC == 4) {
            float4 q = (d < D) ? *reinterpret_cast<const float4*>(p + d) : make_float4(0.f, 0.f, 0.f, 0.f);
            r[i][0] = q.x; r[i][1 % VEC] = q.y; r[i][2 % VEC] = q.z; r[i][3 % VEC] = q.w;
        } else {
            r[i][0] = (d < D) ? p[d] : 0.f;
        }
    }
}
template <int TPR, int VEC, int NV>
__device__ __forceinline__ void row_store(const float (&r)[NV][VEC], float* __restrict__ p, int D, int t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int d = (i * TPR + t) * VEC;
        if (d < D) {
            if (VEC == 4) *reinterpret_cast<float4*>(p + d) = make_float4(r[i][0], r[i][1 % VEC], r[i][2 % VEC], r[i][3 % VEC]);
            else p[d] = r[i][0];
        }
    }
}
template <int VEC, int NV>
__device__ __forceinline__ float row_dot(const float (&a)[NV][VEC], const float (&b)[NV][VEC]) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += a[i][e] * b[i][e];
    return s;
}

template <int TPR, int VEC, int NV>
__global__ void __launch_bounds__(256)
k_dcn_fwd(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases, int64_t B, int D, int L,
          int act, float* __restrict__ y, float* __restrict__ csave) {
    __shared__ float red[4];
    constexpr int RPB = 256 / TPR;
    const int t = threadIdx.x % TPR, rsub = threadIdx.x / TPR;
    const int64_t stride = (int64_t)gridDim.x * RPB;
    if (NV == 1 && L <= DCN_MAX_L) {
        // One register image per thread and row (D <= TPR * VEC): the L kernel and bias rows are row-invariant and stay in
        // registers for the whole kernel (read per row and layer they were 2 L dependent L2 round trips between the
        // reductions), and the next row of x is requested before the current row's L reductions start.
        float wr[DCN_MAX_L][NV][VEC], br[DCN_MAX_L][NV][VEC];
#pragma unroll
        for (int l = 0; l < DCN_MAX_L; ++l) {
            row_load<TPR, VEC, NV>(wr[l], kernels + (int64_t)(l < L ? l : 0) * D, l < L ? D : 0, t);
            row_load<TPR, VEC, NV>(br[l], biases ? biases + (int64_t)(l < L ? l : 0) * D : kernels, (biases && l < L) ? D : 0, t);
        }
        float xn[NV][VEC];
        {
            const int64_t row = (int64_t)blockIdx.x * RPB + rsub;
            row_load<TPR, VEC, NV>(xn, x + (row < B ? row : 0) * D, row < B ? D : 0, t);
        }
        for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < B; row0 += stride) {
            const int64_t row = row0 + rsub;
            const bool ok = row < B;
            float x0[NV][VEC], xl[NV][VEC];
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) xl[i][e] = x0[i][e] = xn[i][e];
            if (row0 + stride < B) {                                  // block-uniform
                const int64_t nrow = row0 + stride + rsub;
                row_load<TPR, VEC, NV>(xn, x + (nrow < B ? nrow : 0) * D, nrow < B ? D : 0, t);
            }
#pragma unroll
            for (int l = 0; l < DCN_MAX_L; ++l) {
                if (l < L) {
                    const float c = row_sum<TPR>(row_dot<VEC, NV>(xl, wr[l]), red);
                    if (csave && ok && t == 0) csave[row * L + l] = c;        // the layer's scalar x_l . w_l: all the backward needs of x_l
#pragma unroll
                    for (int i = 0; i < NV; ++i)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) xl[i][e] = rn_act(x0[i][e] * c + br[l][i][e], act);
                }
            }
            if (ok) row_store<TPR, VEC, NV>(xl, y + row * D, D, t);
        }
        return;
    }
    float xn[NV][VEC];                 // the next row of x, requested before the current row's L reductions start
    {
        const int64_t row = (int64_t)blockIdx.x * RPB + rsub;
        row_load<TPR, VEC, NV>(xn, x + (row < B ? row : 0) * D, row < B ? D : 0, t);
    }
    for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < B; row0 += stride) {
        const int64_t row = row0 + rsub;
        const bool ok = row < B;
        float x0[NV][VEC], xl[NV][VEC], w[NV][VEC], bb[NV][VEC];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) xl[i][e] = x0[i][e] = xn[i][e];
        if (row0 + stride < B) {                                      // block-uniform
            const int64_t nrow = row0 + stride + rsub;
            row_load<TPR, VEC, NV>(xn, x + (nrow < B ? nrow : 0) * D, nrow < B ? D : 0, t);
        }
        for (int l = 0; l < L; ++l) {
            row_load<TPR, VEC, NV>(w, kernels + (int64_t)l * D, D, t);
            const float c = row_sum<TPR>(row_dot<VEC, NV>(xl, w), red);
            if (csave && ok && t == 0) csave[row * L + l] = c;        // the layer's scalar x_l . w_l: all the backward needs of x_l
            if (biases) row_load<TPR, VEC, NV>(bb, biases + (int64_t)l * D, D, t);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) xl[i][e] = rn_act(x0[i][e] * c + (biases ? bb[i][e] : 0.f), act);
        }
        if (ok) row_store<TPR, VEC, NV>(xl, y + row * D, D, t);
    }
}

// part: [gridDim.x][2*L][D]  (dkernel rows 0..L-1, dbias rows L..2L-1) per workgroup
template <int TPR, int VEC, int NV>
__global__ void __launch_bounds__(256)
k_dcn_bwd(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases,
          const float* __restrict__ dy, int64_t B, int D, int L, int act, float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float comb[];   // [256/TPR][D] when TPR == 64 (cross-wave combine)
    constexpr int RPB = 256 / TPR;
    const int t = threadIdx.x % TPR, rsub = threadIdx.x / TPR;
    float dw[DCN_MAX_L][NV][VEC], db[DCN_MAX_L][NV][VEC];
#pragma unroll
    for (int l = 0; l < DCN_MAX_L; ++l)
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) dw[l][i][e] = db[l][i][e] = 0.f;

    for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < B; row0 += (int64_t)gridDim.x * RPB) {
        const int64_t row = row0 + rsub;
        const bool ok = row < B;
        float x0[NV][VEC], g[NV][VEC], dx0[NV][VEC], xl[NV][VEC], w[NV][VEC], bb[NV][VEC];
        row_load<TPR, VEC, NV>(x0, x + (ok ? row : 0) * D, ok ? D : 0, t);
        row_load<TPR, VEC, NV>(g, dy + (ok ? row : 0) * D, ok ? D : 0, t);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) dx0[i][e] = 0.f;
#pragma unroll
        for (int li = 0; li < DCN_MAX_L; ++li) {
            const int l = L - 1 - li;           // walk layers L-1 .. 0 with a compile-time accumulator index
            if (l < 0) continue;
            // recompute x_l from x0
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) xl[i][e] = x0[i][e];
            for (int m = 0; m < l; ++m) {
                row_load<TPR, VEC, NV>(w, kernels + (int64_t)m * D, D, t);
                const float cm = row_sum<TPR>(row_dot<VEC, NV>(xl, w), red);
                if (biases) row_load<TPR, VEC, NV>(bb, biases + (int64_t)m * D, D, t);
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) xl[i][e] = rn_act(x0[i][e] * cm + (biases ? bb[i][e] : 0.f), act);
            }
            row_load<TPR, VEC, NV>(w, kernels + (int64_t)l * D, D, t);
            const float c = row_sum<TPR>(row_dot<VEC, NV>(xl, w), red);
            if (biases) row_load<TPR, VEC, NV>(bb, biases + (int64_t)l * D, D, t);
            float dz[NV][VEC];
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float out = rn_act(x0[i][e] * c + (biases ? bb[i][e] : 0.f), act);
                    dz[i][e] = g[i][e] * rn_act_grad_from_out(out, act);
                }
            const float dc = row_sum<TPR>(row_dot<VEC, NV>(dz, x0), red);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    db[li][i][e] += dz[i][e];
                    dx0[i][e] += dz[i][e] * c;
                    dw[li][i][e] += xl[i][e] * dc;
                    g[i][e] = dc * w[i][e];         // gradient w.r.t. x_l
                }
        }
        // x_0 is x0 itself: the remaining g is the gradient through the first layer's input
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) dx0[i][e] += g[i][e];
        if (ok) row_store<TPR, VEC, NV>(dx0, dx + row * D, D, t);
    }

    // per-workgroup combine, then one slab per workgroup
    float* slab = part + (int64_t)blockIdx.x * 2 * L * D;
#pragma unroll
    for (int li = 0; li < DCN_MAX_L; ++li) {
        const int l = L - 1 - li;
        if (l < 0) continue;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float* dst = slab + (int64_t)(which * L + l) * D;
            if (TPR == 256) {
                row_store<TPR, VEC, NV>(which ? db[li] : dw[li], dst, D, t);
            } else {
                __syncthreads();
                row_store<TPR, VEC, NV>(which ? db[li] : dw[li], comb + rsub * D, D, t);
                __syncthreads();
                for (int d = threadIdx.x; d < D; d += 256) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < RPB; ++r) s += comb[r * D + d];
                    dst[d] = s;
                }
            }
        }
    }
}


// Backward with the forward's per-row scalars c_l = x_l . w_l (csave, B x L floats): x_l = act(x0 * c_{l-1} + b_{l-1}) is then
// elementwise in x0, so a layer costs ONE dot-reduce (dc = <dz, x0>) instead of l + 2, and with a wave per row the reduce is
// shuffles only (no LDS, no barrier).  Same slab layout as k_dcn_bwd.
// WL: the L kernel rows and L bias rows are staged in LDS once per workgroup (behind the combine rows of `comb`) and read
// from there in the row loop.  Read from global memory they are six dependent L2 round trips per row between the three
// reductions (0.37 -> 0.32 ms at B = 65536, D = 1024, L = 3).
template <int TPR, int VEC, int NV, bool WL>
__device__ __forceinline__ void dcn_bwd_saved_body(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases,
                const float* __restrict__ dy, const float* __restrict__ csave, int64_t B, int D, int L, int act,
                float* __restrict__ dx, float* __restrict__ part) {
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float comb[];
    constexpr int RPB = 256 / TPR;
    const int t = threadIdx.x % TPR, rsub = threadIdx.x / TPR;
    float* wl = comb + (TPR < 256 ? RPB * D : 0);        // [L][D] kernels, then [L][D] biases (WL only)
    if (WL) {
        for (int i = threadIdx.x; i < L * D; i += 256) {
            wl[i] = kernels[i];
            wl[L * D + i] = biases ? biases[i] : 0.f;
        }
        __syncthreads();
    }
    const float* kp = WL ? wl : kernels;
    const float* bp = WL ? wl + L * D : biases;
    const bool has_b = WL || biases;
    float dw[DCN_MAX_L][NV][VEC], db[DCN_MAX_L][NV][VEC];
#pragma unroll
    for (int l = 0; l < DCN_MAX_L; ++l)
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) dw[l][i][e] = db[l][i][e] = 0.f;

    // the next row's x, dy and scalars are requested before the current row's L reductions start (NV <= 2: they fit the
    // register budget; wider register images load in place as before)
    constexpr bool PRE = NV <= 2;
    const int64_t stride = (int64_t)gridDim.x * RPB;
    float xn[NV][VEC], gn[NV][VEC], csn[DCN_MAX_L];
    if (PRE) {
        const int64_t row = (int64_t)blockIdx.x * RPB + rsub;
        const bool ok = row < B;
        row_load<TPR, VEC, NV>(xn, x + (ok ? row : 0) * D, ok ? D : 0, t);
        row_load<TPR, VEC, NV>(gn, dy + (ok ? row : 0) * D, ok ? D : 0, t);
#pragma unroll
        for (int l = 0; l < DCN_MAX_L; ++l) csn[l] = (ok && l < L) ? csave[row * L + l] : 0.f;
    }
    for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < B; row0 += stride) {
        const int64_t row = row0 + rsub;
        const bool ok = row < B;
        float x0[NV][VEC], g[NV][VEC], dx0[NV][VEC], tmp[NV][VEC];
        float cs[DCN_MAX_L];
        if (PRE) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) { x0[i][e] = xn[i][e]; g[i][e] = gn[i][e]; }
#pragma unroll
            for (int l = 0; l < DCN_MAX_L; ++l) cs[l] = csn[l];
            if (row0 + stride < B) {                                  // block-uniform
                const int64_t nrow = row0 + stride + rsub;
                const bool nok = nrow < B;
                row_load<TPR, VEC, NV>(xn, x + (nok ? nrow : 0) * D, nok ? D : 0, t);
                row_load<TPR, VEC, NV>(gn, dy + (nok ? nrow : 0) * D, nok ? D : 0, t);
#pragma unroll
                for (int l = 0; l < DCN_MAX_L; ++l) csn[l] = (nok && l < L) ? csave[nrow * L + l] : 0.f;
            }
        } else {
            row_load<TPR, VEC, NV>(x0, x + (ok ? row : 0) * D, ok ? D : 0, t);
            row_load<TPR, VEC, NV>(g, dy + (ok ? row : 0) * D, ok ? D : 0, t);
#pragma unroll
            for (int l = 0; l < DCN_MAX_L; ++l) cs[l] = (ok && l < L) ? csave[row * L + l] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) dx0[i][e] = 0.f;
#pragma unroll
        for (int li = 0; li < DCN_MAX_L; ++li) {
            const int l = L - 1 - li;           // layers L-1 .. 0, compile-time accumulator index li
            if (l < 0) continue;
            float c = 0.f, cprev = 0.f;
#pragma unroll
            for (int q = 0; q < DCN_MAX_L; ++q) {
                if (q == l) c = cs[q];
                if (q == l - 1) cprev = cs[q];
            }
            // dz = g * act'(out_l),  out_l = act(x0 * c_l + b_l)   (kept in g's registers)
            if (has_b) row_load<TPR, VEC, NV>(tmp, bp + (int64_t)l * D, D, t);
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float out = rn_act(x0[i][e] * c + (has_b ? tmp[i][e] : 0.f), act);
                    g[i][e] *= rn_act_grad_from_out(out, act);
                    p += g[i][e] * x0[i][e];
                }
            const float dc = row_sum<TPR>(p, red);
            if (has_b && l > 0) row_load<TPR, VEC, NV>(tmp, bp + (int64_t)(l - 1) * D, D, t);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xl = l == 0 ? x0[i][e] : rn_act(x0[i][e] * cprev + (has_b ? tmp[i][e] : 0.f), act);
                    db[li][i][e] += g[i][e];
                    dx0[i][e] += g[i][e] * c;
                    dw[li][i][e] += xl * dc;
                }
            row_load<TPR, VEC, NV>(tmp, kp + (int64_t)l * D, D, t);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < VEC; ++e) g[i][e] = dc * tmp[i][e];       // gradient w.r.t. x_l
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) dx0[i][e] += g[i][e];             // x_0 is x0 itself
        if (ok) row_store<TPR, VEC, NV>(dx0, dx + row * D, D, t);
    }

    float* slab = part + (int64_t)blockIdx.x * 2 * L * D;
#pragma unroll
    for (int li = 0; li < DCN_MAX_L; ++li) {
        const int l = L - 1 - li;
        if (l < 0) continue;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float* dst = slab + (int64_t)(which * L + l) * D;
            if (TPR == 256) {
                row_store<TPR, VEC, NV>(which ? db[li] : dw[li], dst, D, t);
            } else {
                __syncthreads();
                row_store<TPR, VEC, NV>(which ? db[li] : dw[li], comb + rsub * D, D, t);
                __syncthreads();
                for (int d = threadIdx.x; d < D; d += 256) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < RPB; ++r) s += comb[r * D + d];
                    dst[d] = s;
                }
            }
        }
    }
}

template <int TPR, int VEC, int NV>
__global__ void __launch_bounds__(256)
k_dcn_bwd_saved(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases,
                const float* __restrict__ dy, const float* __restrict__ csave, int64_t B, int D, int L, int act,
                float* __restrict__ dx, float* __restrict__ part) {
    dcn_bwd_saved_body<TPR, VEC, NV, false>(x, kernels, biases, dy, csave, B, D, L, act, dx, part);
}
template <int TPR, int VEC, int NV>
__global__ void __launch_bounds__(256)
k_dcn_bwd_saved_wl(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases,
                   const float* __restrict__ dy, const float* __restrict__ csave, int64_t B, int D, int L, int act,
                   float* __restrict__ dx, float* __restrict__ part) {
    dcn_bwd_saved_body<TPR, VEC, NV, true>(x, kernels, biases, dy, csave, B, D, L, act, dx, part);
}

struct DcnCfg {
    int tpr, vec, nv;
};
static bool dcn_pick(int D, const void* a, const void* b, const void* c, DcnCfg* cfg) {
    const bool aligned = (D % 4 == 0) && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0);
    if (aligned) {
        if (D <= 256) { *cfg = {64, 4, 1}; return true; }
        if (D <= 1024) { *cfg = {64, 4, 4}; return true; }
        if (D <= 4096) { *cfg = {256, 4, 4}; return true; }
        return false;
    }
    if (D <= 256) { *cfg = {256, 1, 1}; return true; }
    if (D <= 1024) { *cfg = {256, 1, 4}; return true; }
    return false;
}

// backward keeps 2*L accumulator rows per thread: with one wave per 1024-wide row that is 270 VGPRs (1 wave/SIMD on a
// latency-bound loop, 1.1 TB/s measured); one 256-thread workgroup per row needs ~70 (8 waves/SIMD).
static bool dcn_pick_bwd(int D, const void* a, const void* b, const void* c, DcnCfg* cfg) {
    if (!dcn_pick(D, a, b, c, cfg)) return false;
    if (cfg->tpr == 64 && cfg->vec == 4 && cfg->nv == 4) *cfg = {256, 4, 1};
    return true;
}

static inline int dcn_grid(int64_t B, int tpr) {
    const int64_t rpb = 256 / tpr;
    int64_t g = (B + rpb - 1) / rpb;
    if (g > 2048) g = 2048;            // 256 CUs x 8 workgroups; rows are grid-strided
    return (int)(g > 0 ? g : 1);
}

extern "C" size_t recnow_dcn_workspace_bytes(int64_t B, int D, int L) {
    if (B <= 0 || D <= 0 || L <= 0) return 256;
    const size_t slabs = rn_align((size_t)2048 * 2 * L * D * sizeof(float));
    return slabs + rn_colsum_ws_bytes(2048, (int64_t)2 * L * D) + rn_align((size_t)2 * L * D * sizeof(float));
}

#define DCN_DISPATCH(KERNEL, SHMEM, ...)                                                                              \
    do {                                                                                                              \
        if (cfg.tpr == 64 && cfg.vec == 4 && cfg.nv == 1) hipLaunchKernelGGL((KERNEL<64, 4, 1>), G, 256, SHMEM, st, __VA_ARGS__);      \
        else if (cfg.tpr == 64 && cfg.vec == 4 && cfg.nv == 4) hipLaunchKernelGGL((KERNEL<64, 4, 4>), G, 256, SHMEM, st, __VA_ARGS__); \
        else if (cfg.tpr == 128 && cfg.vec == 4 && cfg.nv == 2) hipLaunchKernelGGL((KERNEL<128, 4, 2>), G, 256, SHMEM, st, __VA_ARGS__); \
        else if (cfg.tpr == 256 && cfg.vec == 4 && cfg.nv == 1) hipLaunchKernelGGL((KERNEL<256, 4, 1>), G, 256, SHMEM, st, __VA_ARGS__); \
        else if (cfg.tpr == 256 && cfg.vec == 4) hipLaunchKernelGGL((KERNEL<256, 4, 4>), G, 256, SHMEM, st, __VA_ARGS__);              \
        else if (cfg.tpr == 256 && cfg.vec == 1 && cfg.nv == 1) hipLaunchKernelGGL((KERNEL<256, 1, 1>), G, 256, SHMEM, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<256, 1, 4>), G, 256, SHMEM, st, __VA_ARGS__);                                                  \
    } while (0)

extern "C" int recnow_dcn_fwd(const float* x, const float* kernels, const float* biases, int64_t B, int D, int L, int act, float* y,
                              float* csave, void* stream) {
    if (B < 0 || D < 1 || L < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!x || !kernels || !y) return RECNOW_EINVAL;
    DcnCfg cfg;
    if (!dcn_pick(D, x, y, kernels, &cfg) || (biases && (((uintptr_t)biases & 15) != 0) && cfg.vec == 4)) return RECNOW_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int G = dcn_grid(B, cfg.tpr);
    DCN_DISPATCH(k_dcn_fwd, 0, x, kernels, biases, B, D, L, act, y, csave);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_dcn_bwd(const float* x, const float* kernels, const float* biases, const float* dy, const float* csave, int64_t B,
                              int D, int L, int act, float* dx, float* dkernels, float* dbiases, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || D < 1 || L < 1) return RECNOW_EINVAL;
    if (L > DCN_MAX_L) return RECNOW_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (dkernels) RN_HIP(hipMemsetAsync(dkernels, 0, (size_t)L * D * sizeof(float), st));
        if (dbiases) RN_HIP(hipMemsetAsync(dbiases, 0, (size_t)L * D * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!x || !kernels || !dy || !dx || !dkernels || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_dcn_workspace_bytes(B, D, L)) return RECNOW_EWORKSPACE;
    DcnCfg cfg;
    // with the saved scalars a wave per row fits the register file (no recompute state); without them one workgroup per row
    if (!(csave ? dcn_pick(D, x, dy, dx, &cfg) : dcn_pick_bwd(D, x, dy, dx, &cfg)) ||
        (cfg.vec == 4 && ((((uintptr_t)kernels | (uintptr_t)biases) & 15) != 0)))
        return RECNOW_EUNSUPPORTED;
    if (csave && cfg.tpr == 64 && cfg.vec == 4 && cfg.nv == 4) cfg = {128, 4, 2};     // 2 waves per wide row: half the registers per lane
    const int G = dcn_grid(B, cfg.tpr);
    RnCarver c(ws, ws_bytes);
    float* part = c.take<float>((size_t)2048 * 2 * L * D);
    float* sums = c.take<float>((size_t)2 * L * D);
    void* cs_ws = c.base + c.off;
    const size_t cs_bytes = ws_bytes - c.off;
    const size_t shmem = cfg.tpr < 256 ? (size_t)(256 / cfg.tpr) * D * sizeof(float) : 0;
    const size_t wl_bytes = (size_t)2 * L * D * sizeof(float);            // kernels + biases of every layer, staged in LDS when they fit
    if (csave && shmem + wl_bytes <= 48 * 1024) DCN_DISPATCH(k_dcn_bwd_saved_wl, shmem + wl_bytes, x, kernels, biases, dy, csave, B, D, L, act, dx, part);
    else if (csave) DCN_DISPATCH(k_dcn_bwd_saved, shmem, x, kernels, biases, dy, csave, B, D, L, act, dx, part);
    else DCN_DISPATCH(k_dcn_bwd, shmem, x, kernels, biases, dy, B, D, L, act, dx, part);
    RN_LAUNCH_CHECK();
    int rc = rn_colsum(part, nullptr, 0, 0, G, (int64_t)2 * L * D, (int64_t)2 * L * D, sums, 0, cs_ws, cs_bytes, st);
    if (rc) return rc;
    RN_HIP(hipMemcpyAsync(dkernels, sums, (size_t)L * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (dbiases) RN_HIP(hipMemcpyAsync(dbiases, sums + (size_t)L * D, (size_t)L * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return RECNOW_OK;
}
