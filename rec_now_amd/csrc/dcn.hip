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


// ---- fast variants: D = TPR * 4 * NV exactly, 16-byte aligned, B a multiple of the rows per workgroup, L and the activation
// compile-time ------------------------------------------------------------------------------------------------------------------
// Same math as k_dcn_fwd / dcn_bwd_saved_body.  What their ISA showed: every `d < D ? load : 0`, `if (ok) store`, `if (l < L)`,
// `if (biases)` and the per-element activation switch is a branch; vmcnt counts loads and stores in order, so the wait for the
// prefetched next row is `vmcnt(stores issued since)` only when that number is the same on every path -- with the branches it
// became `vmcnt(0)`: every row waited for its own stores to be acknowledged before the next row's arithmetic started.  Here nothing
// that touches memory sits under a condition: kernels and biases of all layers live in LDS (a missing bias = zeros), the next
// row is always requested (the last request re-reads the own row), the per-row scalars are stored by every lane of the row.
template <int TPR, int NV>
__device__ __forceinline__ void row_load_full(float (&r)[NV][4], const float* __restrict__ p, int t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 q = *reinterpret_cast<const float4*>(p + (i * TPR + t) * 4);
        r[i][0] = q.x; r[i][1] = q.y; r[i][2] = q.z; r[i][3] = q.w;
    }
}
template <int TPR, int NV>
__device__ __forceinline__ void row_store_full(const float (&r)[NV][4], float* __restrict__ p, int t) {
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(p + (i * TPR + t) * 4) = make_float4(r[i][0], r[i][1], r[i][2], r[i][3]);
}
// The streamed tensors of the fast kernels (x, dy in; y, dx out: every byte touched once) move with NON-TEMPORAL loads and stores (round 4, A/B on one
// box, tools/micro/stream_bench.py: forward 111-113 -> 100.5-100.9 us = 4.8 -> 5.3 TB/s, backward 162-167 -> 146-153 us); -DRN_STREAM_PLAIN
// (tools/build_variant.py) builds the plain-access variant.
template <int TPR, int NV>
__device__ __forceinline__ void row_load_stream(float (&r)[NV][4], const float* __restrict__ p, int t) {
#ifndef RN_STREAM_PLAIN
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const rn_f4 q = __builtin_nontemporal_load(reinterpret_cast<const rn_f4*>(p + (i * TPR + t) * 4));
        r[i][0] = q.x; r[i][1] = q.y; r[i][2] = q.z; r[i][3] = q.w;
    }
#else
    row_load_full<TPR, NV>(r, p, t);
#endif
}
template <int TPR, int NV>
__device__ __forceinline__ void row_store_stream(const float (&r)[NV][4], float* __restrict__ p, int t) {
#ifndef RN_STREAM_PLAIN
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        rn_f4 q;
        q.x = r[i][0]; q.y = r[i][1]; q.z = r[i][2]; q.w = r[i][3];
        __builtin_nontemporal_store(q, reinterpret_cast<rn_f4*>(p + (i * TPR + t) * 4));
    }
#else
    row_store_full<TPR, NV>(r, p, t);
#endif
}

template <int NV, int L, int ACT, bool CS>
__global__ void __launch_bounds__(256)
k_dcn_fwd_fast(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases, int64_t B, int act_rt,
               float* __restrict__ y, float* __restrict__ csave) {
    constexpr int TPR = 64, D = TPR * 4 * NV;          // a wave per row: reductions are shuffles only
    const int act = ACT >= 0 ? ACT : act_rt;
    extern __shared__ __attribute__((aligned(16))) float wl[];      // [L][D] kernels, [L][D] biases
    for (int i = threadIdx.x; i < L * D; i += 256) {
        wl[i] = kernels[i];
        wl[L * D + i] = biases ? biases[i] : 0.f;
    }
    __syncthreads();
    const int t = threadIdx.x & 63, rsub = threadIdx.x >> 6;
    const int64_t stride = (int64_t)gridDim.x * 4;
    float xn[NV][4];
    row_load_stream<TPR, NV>(xn, x + ((int64_t)blockIdx.x * 4 + rsub) * D, t);
    for (int64_t row0 = (int64_t)blockIdx.x * 4; row0 < B; row0 += stride) {
        const int64_t row = row0 + rsub;
        float x0[NV][4], xl[NV][4];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) xl[i][e] = x0[i][e] = xn[i][e];
        row_load_stream<TPR, NV>(xn, x + (row0 + stride < B ? row + stride : row) * D, t);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            float w[NV][4], bb[NV][4];
            row_load_full<TPR, NV>(w, wl + l * D, t);
            row_load_full<TPR, NV>(bb, wl + (L + l) * D, t);
            const float c = wave_sum(row_dot<4, NV>(xl, w));
            if (CS) csave[row * L + l] = c;                  // every lane of the row (same value, same address: one transaction)
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) xl[i][e] = rn_act(x0[i][e] * c + bb[i][e], act);
        }
        row_store_stream<TPR, NV>(xl, y + row * D, t);
    }
}

// TPR = 64 (a wave per row, D <= 512) or 128 (two waves per row, D = 1024: half the accumulator registers per lane)
template <int TPR, int NV, int L, int ACT>
__global__ void __launch_bounds__(256)
k_dcn_bwd_fast(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases, const float* __restrict__ dy,
               const float* __restrict__ csave, int64_t B, int act_rt, float* __restrict__ dx, float* __restrict__ part) {
    constexpr int D = TPR * 4 * NV, RPB = 256 / TPR;
    const int act = ACT >= 0 ? ACT : act_rt;
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float comb[];    // [RPB][D] combine rows, then [L][D] kernels, [L][D] biases
    float* wl = comb + RPB * D;
    for (int i = threadIdx.x; i < L * D; i += 256) {
        wl[i] = kernels[i];
        wl[L * D + i] = biases ? biases[i] : 0.f;
    }
    __syncthreads();
    const int t = threadIdx.x % TPR, rsub = threadIdx.x / TPR;
    float dw[L][NV][4], db[L][NV][4];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) dw[l][i][e] = db[l][i][e] = 0.f;
    const int64_t stride = (int64_t)gridDim.x * RPB;
    float xn[NV][4], gn[NV][4], csn[L];
    {
        const int64_t row = (int64_t)blockIdx.x * RPB + rsub;
        row_load_stream<TPR, NV>(xn, x + row * D, t);
        row_load_stream<TPR, NV>(gn, dy + row * D, t);
#pragma unroll
        for (int l = 0; l < L; ++l) csn[l] = csave[row * L + l];
    }
    for (int64_t row0 = (int64_t)blockIdx.x * RPB; row0 < B; row0 += stride) {
        const int64_t row = row0 + rsub;
        float x0[NV][4], g[NV][4], dx0[NV][4], tmp[NV][4];
        float cs[L];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) { x0[i][e] = xn[i][e]; g[i][e] = gn[i][e]; dx0[i][e] = 0.f; }
#pragma unroll
        for (int l = 0; l < L; ++l) cs[l] = csn[l];
        {
            const int64_t nrow = row0 + stride < B ? row + stride : row;
            row_load_stream<TPR, NV>(xn, x + nrow * D, t);
            row_load_stream<TPR, NV>(gn, dy + nrow * D, t);
#pragma unroll
            for (int l = 0; l < L; ++l) csn[l] = csave[nrow * L + l];
        }
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            const float c = cs[l];
            // dz = g * act'(out_l),  out_l = act(x0 * c_l + b_l)   (kept in g's registers)
            row_load_full<TPR, NV>(tmp, wl + (L + l) * D, t);
            float p = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float out = rn_act(x0[i][e] * c + tmp[i][e], act);
                    g[i][e] *= rn_act_grad_from_out(out, act);
                    p += g[i][e] * x0[i][e];
                }
            const float dc = row_sum<TPR>(p, red);
            if (l > 0) row_load_full<TPR, NV>(tmp, wl + (L + l - 1) * D, t);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xl = l == 0 ? x0[i][e] : rn_act(x0[i][e] * cs[l > 0 ? l - 1 : 0] + tmp[i][e], act);
                    db[l][i][e] += g[i][e];
                    dx0[i][e] += g[i][e] * c;
                    dw[l][i][e] += xl * dc;
                }
            row_load_full<TPR, NV>(tmp, wl + l * D, t);
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) g[i][e] = dc * tmp[i][e];       // gradient w.r.t. x_l
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) dx0[i][e] += g[i][e];             // x_0 is x0 itself
        row_store_stream<TPR, NV>(dx0, dx + row * D, t);
    }
    // per-workgroup combine (rows of the workgroup summed in a fixed order), then one slab per workgroup: as dcn_bwd_saved_body
    float* slab = part + (int64_t)blockIdx.x * 2 * L * D;
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float* dst = slab + (int64_t)(which * L + l) * D;
            __syncthreads();
            row_store_full<TPR, NV>(which ? db[l] : dw[l], comb + rsub * D, t);
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

// ---- general shapes: any degree_of_cross, any D, any alignment ------------------------------------------------------------
// The fused kernels above keep a row (and 2 L accumulator rows in the backward) in registers: L <= 4, D <= 4096.  Beyond
// that the same math runs as three streaming kernels built on the per-row scalars c_l = x_l . w_l (reference
// rec_now/layers/dcn_layer.py:91-100 has no limit on degree_of_cross or input_dim):
//   * x_l = act(x0 * c_{l-1} + b_{l-1}) is elementwise in x0, and the gradient that reaches layer l < L-1 is rank one per row,
//     g_l = dc_{l+1} * w_{l+1}  (dc_l = <dz_l, x0>, dz_l = g_l * act'(out_l));
//   * rows kernel (one workgroup per row, the row is re-read from L1/L2 once per layer): the L scalars dc_l, then
//     dx0 = sum_l dz_l * c_l + dc_0 * w_0 in one more pass;
//   * columns kernel (thread per column, row slabs, one layer per blockIdx.z): dw_l = sum_rows x_l * dc_l, db_l = sum_rows dz_l,
//     reduced over the slabs by the deterministic column sum.
__device__ __forceinline__ float dcn_xl(float x0, float cprev, float bprev, int l, int act) {
    return l == 0 ? x0 : rn_act(x0 * cprev + bprev, act);
}

// y == NULL: only the scalars are produced (backward without csave)
__global__ void __launch_bounds__(256)
k_dcn_fwd_general(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases, int64_t B, int D,
                  int L, int act, float* __restrict__ y, float* __restrict__ csave) {
    __shared__ float red[16];
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        const float* xr = x + row * D;
        float cprev = 0.f;
        for (int l = 0; l < L; ++l) {
            const float* w = kernels + (int64_t)l * D;
            const float* bp = (biases && l > 0) ? biases + (int64_t)(l - 1) * D : nullptr;
            float p = 0.f;
            for (int d = threadIdx.x; d < D; d += 256) p += dcn_xl(xr[d], cprev, bp ? bp[d] : 0.f, l, act) * w[d];
            const float c = block_sum(p, red);
            if (csave && threadIdx.x == 0) csave[row * L + l] = c;
            cprev = c;
        }
        if (y) {
            const float* bp = biases ? biases + (int64_t)(L - 1) * D : nullptr;
            for (int d = threadIdx.x; d < D; d += 256) y[row * D + d] = rn_act(xr[d] * cprev + (bp ? bp[d] : 0.f), act);
        }
    }
}

// dz_l at one element: g * act'(out_l), out_l = act(x0 * c_l + b_l), g = dy (top layer) or dc_{l+1} * w_{l+1}
__device__ __forceinline__ float dcn_dz(float x0, float dyv, float c, float b, float dcn, float wn, bool top, int act) {
    const float out = rn_act(x0 * c + b, act);
    return (top ? dyv : dcn * wn) * rn_act_grad_from_out(out, act);
}

__global__ void __launch_bounds__(256)
k_dcn_bwd_rows_general(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases,
                       const float* __restrict__ dy, const float* __restrict__ csave, int64_t B, int D, int L, int act,
                       float* __restrict__ dx, float* __restrict__ dcs) {
    __shared__ float red[16];
    extern __shared__ float sc[];          // [L] c_l, then [L] dc_l of the current row
    float* cs = sc;
    float* ds = sc + L;
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        const float* xr = x + row * D;
        const float* gr = dy + row * D;
        __syncthreads();
        for (int l = threadIdx.x; l < L; l += 256) cs[l] = csave[row * L + l];
        __syncthreads();
        float dcn = 0.f;
        for (int l = L - 1; l >= 0; --l) {
            const bool top = l == L - 1;
            const float c = cs[l];
            const float* b = biases ? biases + (int64_t)l * D : nullptr;
            const float* wn = top ? kernels : kernels + (int64_t)(l + 1) * D;
            float p = 0.f;
            for (int d = threadIdx.x; d < D; d += 256) {
                const float x0 = xr[d];
                p += dcn_dz(x0, top ? gr[d] : 0.f, c, b ? b[d] : 0.f, dcn, wn[d], top, act) * x0;
            }
            const float dc = block_sum(p, red);
            if (threadIdx.x == 0) { ds[l] = dc; dcs[row * L + l] = dc; }
            dcn = dc;
        }
        __syncthreads();
        for (int d = threadIdx.x; d < D; d += 256) {
            const float x0 = xr[d], dyv = gr[d];
            float acc = ds[0] * kernels[d];          // x_0 is x0 itself: the gradient through the first layer's input
            for (int l = 0; l < L; ++l) {
                const bool top = l == L - 1;
                const float wn = top ? 0.f : kernels[(int64_t)(l + 1) * D + d];
                acc += dcn_dz(x0, dyv, cs[l], biases ? biases[(int64_t)l * D + d] : 0.f, top ? 0.f : ds[l + 1], wn, top, act) * cs[l];
            }
            dx[row * D + d] = acc;
        }
    }
}

// part[slab][2L][D]: rows 0..L-1 dkernel, L..2L-1 dbias.  grid (ceil(D/256), slabs, L)
__global__ void __launch_bounds__(256)
k_dcn_bwd_cols_general(const float* __restrict__ x, const float* __restrict__ kernels, const float* __restrict__ biases,
                       const float* __restrict__ dy, const float* __restrict__ csave, const float* __restrict__ dcs, int64_t B, int D,
                       int L, int act, float* __restrict__ part) {
    const int d = blockIdx.x * 256 + threadIdx.x, l = blockIdx.z;
    const int64_t per = (B + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(B, r0 + per);
    if (d >= D) return;
    const bool top = l == L - 1;
    const float b = biases ? biases[(int64_t)l * D + d] : 0.f;
    const float bprev = (biases && l > 0) ? biases[(int64_t)(l - 1) * D + d] : 0.f;
    const float wn = top ? 0.f : kernels[(int64_t)(l + 1) * D + d];
    float aw = 0.f, ab = 0.f;
    for (int64_t row = r0; row < r1; ++row) {
        const float x0 = x[row * D + d];
        const float c = csave[row * L + l], cprev = l > 0 ? csave[row * L + l - 1] : 0.f;
        const float dc = dcs[row * L + l], dcn = top ? 0.f : dcs[row * L + l + 1];
        ab += dcn_dz(x0, top ? dy[row * D + d] : 0.f, c, b, dcn, wn, top, act);
        aw += dcn_xl(x0, cprev, bprev, l, act) * dc;
    }
    float* slab = part + (int64_t)blockIdx.y * 2 * L * D;
    slab[(int64_t)l * D + d] = aw;
    slab[(int64_t)(L + l) * D + d] = ab;
}

#define DCN_GENERAL_SLABS 64

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

static int dcn_occupancy(const void* kernel, size_t shm) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, shm) != hipSuccess || n < 1) n = 1;
    return n;
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
    // + the per-row scalars c_l / dc_l of the general path (any L, any D)
    return slabs + rn_colsum_ws_bytes(2048, (int64_t)2 * L * D) + rn_align((size_t)2 * L * D * sizeof(float)) +
           2 * rn_align((size_t)B * L * sizeof(float));
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
    hipStream_t st = (hipStream_t)stream;
    if (!dcn_pick(D, x, y, kernels, &cfg) || (biases && (((uintptr_t)biases & 15) != 0) && cfg.vec == 4)) {
        hipLaunchKernelGGL(k_dcn_fwd_general, (int)(B < 4096 ? B : 4096), 256, 0, st, x, kernels, biases, B, D, L, act, y, csave);
        RN_LAUNCH_CHECK();
        return RECNOW_OK;
    }
    static const bool dcn_fast = []() { const char* e = getenv("RECNOW_DCN_FAST"); return !e || e[0] != '0'; }();      // A/B switch
    if (dcn_fast && cfg.vec == 4 && cfg.tpr == 64 && D == 256 * cfg.nv && B % 4 == 0 && L <= DCN_MAX_L && (size_t)2 * L * D * sizeof(float) <= 48 * 1024) {
        const int G = dcn_grid(B, 64);
        const size_t shm = (size_t)2 * L * D * sizeof(float);
#define FWD_FAST(NV_, L_)                                                                                                          \
        if (cfg.nv == NV_ && L == L_) {                                                                                            \
            if (act == RECNOW_ACT_LINEAR) {                                                                                        \
                if (csave) hipLaunchKernelGGL((k_dcn_fwd_fast<NV_, L_, RECNOW_ACT_LINEAR, true>), G, 256, shm, st, x, kernels, biases, B, act, y, csave);  \
                else hipLaunchKernelGGL((k_dcn_fwd_fast<NV_, L_, RECNOW_ACT_LINEAR, false>), G, 256, shm, st, x, kernels, biases, B, act, y, csave);      \
            } else {                                                                                                               \
                if (csave) hipLaunchKernelGGL((k_dcn_fwd_fast<NV_, L_, -1, true>), G, 256, shm, st, x, kernels, biases, B, act, y, csave);                 \
                else hipLaunchKernelGGL((k_dcn_fwd_fast<NV_, L_, -1, false>), G, 256, shm, st, x, kernels, biases, B, act, y, csave);                     \
            }                                                                                                                      \
            RN_LAUNCH_CHECK();                                                                                                     \
            return RECNOW_OK;                                                                                                      \
        }
        FWD_FAST(1, 1) FWD_FAST(1, 2) FWD_FAST(1, 3) FWD_FAST(1, 4)
        FWD_FAST(4, 1) FWD_FAST(4, 2) FWD_FAST(4, 3) FWD_FAST(4, 4)
#undef FWD_FAST
    }
    const int G = dcn_grid(B, cfg.tpr);
    DCN_DISPATCH(k_dcn_fwd, 0, x, kernels, biases, B, D, L, act, y, csave);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_dcn_bwd(const float* x, const float* kernels, const float* biases, const float* dy, const float* csave, int64_t B,
                              int D, int L, int act, float* dx, float* dkernels, float* dbiases, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || D < 1 || L < 1) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (dkernels) RN_HIP(hipMemsetAsync(dkernels, 0, (size_t)L * D * sizeof(float), st));
        if (dbiases) RN_HIP(hipMemsetAsync(dbiases, 0, (size_t)L * D * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!x || !kernels || !dy || !dx || !dkernels || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_dcn_workspace_bytes(B, D, L)) return RECNOW_EWORKSPACE;
    DcnCfg cfg;
    RnCarver c(ws, ws_bytes);
    float* part = c.take<float>((size_t)2048 * 2 * L * D);
    float* sums = c.take<float>((size_t)2 * L * D);
    float* dcs = c.take<float>((size_t)B * L);
    float* cs_own = c.take<float>((size_t)B * L);
    void* cs_ws = c.base + c.off;
    const size_t cs_bytes = ws_bytes - c.off;
    // with the saved scalars a wave per row fits the register file (no recompute state); without them one workgroup per row
    if (L > DCN_MAX_L || !(csave ? dcn_pick(D, x, dy, dx, &cfg) : dcn_pick_bwd(D, x, dy, dx, &cfg)) ||
        (cfg.vec == 4 && ((((uintptr_t)kernels | (uintptr_t)biases) & 15) != 0))) {
        // general path: any L, D, alignment (see k_dcn_bwd_rows_general)
        const int GR = (int)(B < 4096 ? B : 4096);
        if (!csave) {
            hipLaunchKernelGGL(k_dcn_fwd_general, GR, 256, 0, st, x, kernels, biases, B, D, L, act, (float*)nullptr, cs_own);
            RN_LAUNCH_CHECK();
            csave = cs_own;
        }
        if ((size_t)2 * L * sizeof(float) > 60 * 1024) return RECNOW_EUNSUPPORTED;           // L > 7680 cross layers
        hipLaunchKernelGGL(k_dcn_bwd_rows_general, GR, 256, (size_t)2 * L * sizeof(float), st, x, kernels, biases, dy, csave, B, D, L, act, dx, dcs);
        RN_LAUNCH_CHECK();
        int slabs = (int)((B + 255) / 256);
        if (slabs > DCN_GENERAL_SLABS) slabs = DCN_GENERAL_SLABS;
        if (L > 65535) return RECNOW_EUNSUPPORTED;
        hipLaunchKernelGGL(k_dcn_bwd_cols_general, dim3((D + 255) / 256, slabs, L), 256, 0, st, x, kernels, biases, dy, csave, dcs, B, D, L, act, part);
        RN_LAUNCH_CHECK();
        int rcg = rn_colsum(part, nullptr, 0, 0, slabs, (int64_t)2 * L * D, (int64_t)2 * L * D, sums, 0, cs_ws, cs_bytes, st);
        if (rcg) return rcg;
        RN_HIP(hipMemcpyAsync(dkernels, sums, (size_t)L * D * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (dbiases) RN_HIP(hipMemcpyAsync(dbiases, sums + (size_t)L * D, (size_t)L * D * sizeof(float), hipMemcpyDeviceToDevice, st));
        return RECNOW_OK;
    }
    // D = 1024: a wave per row up to three cross layers (row sums are shuffles, no barrier in the row loop: the two-waves-per-row form
    // waits on a workgroup barrier per layer and row: 176 vs 166 us at B = 65 536, L = 3, both on one resident wave of workgroups); deeper: 2 waves per row, half the
    // accumulator registers per lane.  RECNOW_DCN_WAVE_ROW=0 is the A/B switch.
    static const bool wave_row = []() { const char* e = getenv("RECNOW_DCN_WAVE_ROW"); return !e || e[0] != '0'; }();
    if (csave && cfg.tpr == 64 && cfg.vec == 4 && cfg.nv == 4 && !(wave_row && L <= 3)) cfg = {128, 4, 2};
    static const bool dcn_fast = []() { const char* e = getenv("RECNOW_DCN_FAST"); return !e || e[0] != '0'; }();      // A/B switch
    static const bool dcn_resident = []() { const char* e = getenv("RECNOW_DCN_RESIDENT"); return !e || e[0] != '0'; }();      // A/B switch
    bool fast_done = false;
    int G = dcn_grid(B, cfg.tpr);
    if (dcn_fast && csave && cfg.vec == 4 && D == cfg.tpr * 4 * cfg.nv && B % (256 / cfg.tpr) == 0 && (cfg.tpr == 64 || cfg.tpr == 128)) {
        const size_t shm = ((size_t)(256 / cfg.tpr) * D + (size_t)2 * L * D) * sizeof(float);
#define BWD_FAST(TPR_, NV_, L_)                                                                                                    \
        if (!fast_done && shm <= 48 * 1024 && cfg.tpr == TPR_ && cfg.nv == NV_ && L == L_) {                                       \
            /* one resident wave of workgroups (rows are grid-strided): every workgroup beyond it costs a weight fill, a combine  \
               and a 2*L*D slab that the column sum reads back (2048 slabs = 50 MB at D = 1024, L = 3) */                        \
            if (act == RECNOW_ACT_LINEAR) {                                                                                        \
                static const int occ = dcn_occupancy((const void*)k_dcn_bwd_fast<TPR_, NV_, L_, RECNOW_ACT_LINEAR>, shm);          \
                if (dcn_resident && G > 256 * occ) G = 256 * occ;                                                                  \
                hipLaunchKernelGGL((k_dcn_bwd_fast<TPR_, NV_, L_, RECNOW_ACT_LINEAR>), G, 256, shm, st, x, kernels, biases, dy, csave, B, act, dx, part); \
            } else {                                                                                                               \
                static const int occ = dcn_occupancy((const void*)k_dcn_bwd_fast<TPR_, NV_, L_, -1>, shm);                         \
                if (dcn_resident && G > 256 * occ) G = 256 * occ;                                                                  \
                hipLaunchKernelGGL((k_dcn_bwd_fast<TPR_, NV_, L_, -1>), G, 256, shm, st, x, kernels, biases, dy, csave, B, act, dx, part);  \
            }                                                                                                                      \
            fast_done = true;                                                                                                      \
        }
        BWD_FAST(64, 1, 1) BWD_FAST(64, 1, 2) BWD_FAST(64, 1, 3) BWD_FAST(64, 1, 4)
        BWD_FAST(128, 2, 1) BWD_FAST(128, 2, 2) BWD_FAST(128, 2, 3) BWD_FAST(128, 2, 4)
        BWD_FAST(64, 4, 1) BWD_FAST(64, 4, 2) BWD_FAST(64, 4, 3)
#undef BWD_FAST
    }
    const size_t shmem = cfg.tpr < 256 ? (size_t)(256 / cfg.tpr) * D * sizeof(float) : 0;
    const size_t wl_bytes = (size_t)2 * L * D * sizeof(float);            // kernels + biases of every layer, staged in LDS when they fit
    if (fast_done) {
    } else if (csave && shmem + wl_bytes <= 48 * 1024) DCN_DISPATCH(k_dcn_bwd_saved_wl, shmem + wl_bytes, x, kernels, biases, dy, csave, B, D, L, act, dx, part);
    else if (csave) DCN_DISPATCH(k_dcn_bwd_saved, shmem, x, kernels, biases, dy, csave, B, D, L, act, dx, part);
    else DCN_DISPATCH(k_dcn_bwd, shmem, x, kernels, biases, dy, B, D, L, act, dx, part);
    RN_LAUNCH_CHECK();
    int rc = rn_colsum(part, nullptr, 0, 0, G, (int64_t)2 * L * D, (int64_t)2 * L * D, sums, 0, cs_ws, cs_bytes, st);
    if (rc) return rc;
    RN_HIP(hipMemcpyAsync(dkernels, sums, (size_t)L * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (dbiases) RN_HIP(hipMemcpyAsync(dbiases, sums + (size_t)L * D, (size_t)L * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return RECNOW_OK;
}


// ---- one cross layer with its own x_l input (the unfused route: user-supplied activation callables between the layers) ----
//     z = act(x0 * (x_l . w) + b)          reference rec_now/layers/dcn_layer.py:91-99, one iteration of the loop
// forward: c (B) is kept for the backward.  backward: dzp = dz * act'(z); dc = <dzp, x0>; dx0 = dzp * c; dx_l = dc * w;
// dw = sum_rows x_l * dc; db = sum_rows dzp.
__global__ void __launch_bounds__(256)
k_dcn_step_fwd(const float* __restrict__ x0, const float* __restrict__ xl, const float* __restrict__ w, const float* __restrict__ b,
               int64_t B, int D, int act, float* __restrict__ z, float* __restrict__ c_out) {
    __shared__ float red[16];
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        float p = 0.f;
        for (int d = threadIdx.x; d < D; d += 256) p += xl[row * D + d] * w[d];
        const float c = block_sum(p, red);
        if (threadIdx.x == 0) c_out[row] = c;
        for (int d = threadIdx.x; d < D; d += 256) z[row * D + d] = rn_act(x0[row * D + d] * c + (b ? b[d] : 0.f), act);
    }
}
__global__ void __launch_bounds__(256)
k_dcn_step_bwd_rows(const float* __restrict__ x0, const float* __restrict__ w, const float* __restrict__ z, const float* __restrict__ c,
                    const float* __restrict__ dz, int64_t B, int D, int act, float* __restrict__ dx0, float* __restrict__ dxl,
                    float* __restrict__ dcs) {
    __shared__ float red[16];
    for (int64_t row = blockIdx.x; row < B; row += gridDim.x) {
        float p = 0.f;
        for (int d = threadIdx.x; d < D; d += 256) {
            const int64_t i = row * D + d;
            p += dz[i] * (act ? rn_act_grad_from_out(z[i], act) : 1.f) * x0[i];
        }
        const float dc = block_sum(p, red);
        if (threadIdx.x == 0) dcs[row] = dc;
        const float cr = c[row];
        for (int d = threadIdx.x; d < D; d += 256) {
            const int64_t i = row * D + d;
            dx0[i] = dz[i] * (act ? rn_act_grad_from_out(z[i], act) : 1.f) * cr;
            dxl[i] = dc * w[d];
        }
    }
}
// part[slab][2][D]
__global__ void __launch_bounds__(256)
k_dcn_step_bwd_cols(const float* __restrict__ xl, const float* __restrict__ z, const float* __restrict__ dz, const float* __restrict__ dcs,
                    int64_t B, int D, int act, float* __restrict__ part) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (B + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(B, r0 + per);
    if (d >= D) return;
    float aw = 0.f, ab = 0.f;
    for (int64_t row = r0; row < r1; ++row) {
        const int64_t i = row * D + d;
        ab += dz[i] * (act ? rn_act_grad_from_out(z[i], act) : 1.f);
        aw += xl[i] * dcs[row];
    }
    part[((int64_t)blockIdx.y * 2) * D + d] = aw;
    part[((int64_t)blockIdx.y * 2 + 1) * D + d] = ab;
}

extern "C" size_t recnow_dcn_step_workspace_bytes(int64_t B, int D) {
    if (B <= 0 || D <= 0) return 256;
    return rn_align((size_t)DCN_GENERAL_SLABS * 2 * D * sizeof(float)) + rn_align((size_t)2 * D * sizeof(float)) +
           rn_align((size_t)B * sizeof(float)) + rn_colsum_ws_bytes(DCN_GENERAL_SLABS, (int64_t)2 * D);
}

extern "C" int recnow_dcn_step_fwd(const float* x0, const float* xl, const float* w, const float* b, int64_t B, int D, int act, float* z,
                                   float* c_out, void* stream) {
    if (B < 0 || D < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!x0 || !xl || !w || !z || !c_out) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_dcn_step_fwd, (int)(B < 4096 ? B : 4096), 256, 0, (hipStream_t)stream, x0, xl, w, b, B, D, act, z, c_out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

extern "C" int recnow_dcn_step_bwd(const float* x0, const float* xl, const float* w, const float* z, const float* c, const float* dz,
                                   int64_t B, int D, int act, float* dx0, float* dxl, float* dw, float* db, void* ws, size_t ws_bytes,
                                   void* stream) {
    if (B < 0 || D < 1) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        if (dw) RN_HIP(hipMemsetAsync(dw, 0, (size_t)D * sizeof(float), st));
        if (db) RN_HIP(hipMemsetAsync(db, 0, (size_t)D * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!x0 || !xl || !w || !c || !dz || !dx0 || !dxl || !dw || !ws || (act != RECNOW_ACT_LINEAR && !z)) return RECNOW_EINVAL;
    if (ws_bytes < recnow_dcn_step_workspace_bytes(B, D)) return RECNOW_EWORKSPACE;
    RnCarver cv(ws, ws_bytes);
    float* part = cv.take<float>((size_t)DCN_GENERAL_SLABS * 2 * D);
    float* sums = cv.take<float>((size_t)2 * D);
    float* dcs = cv.take<float>((size_t)B);
    void* cs_ws = cv.base + cv.off;
    const size_t cs_bytes = ws_bytes - cv.off;
    hipLaunchKernelGGL(k_dcn_step_bwd_rows, (int)(B < 4096 ? B : 4096), 256, 0, st, x0, w, z, c, dz, B, D, act, dx0, dxl, dcs);
    RN_LAUNCH_CHECK();
    int slabs = (int)((B + 255) / 256);
    if (slabs > DCN_GENERAL_SLABS) slabs = DCN_GENERAL_SLABS;
    hipLaunchKernelGGL(k_dcn_step_bwd_cols, dim3((D + 255) / 256, slabs), 256, 0, st, xl, z, dz, dcs, B, D, act, part);
    RN_LAUNCH_CHECK();
    int rc = rn_colsum(part, nullptr, 0, 0, slabs, (int64_t)2 * D, (int64_t)2 * D, sums, 0, cs_ws, cs_bytes, st);
    if (rc) return rc;
    RN_HIP(hipMemcpyAsync(dw, sums, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (db) RN_HIP(hipMemcpyAsync(db, sums + D, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return RECNOW_OK;
}
