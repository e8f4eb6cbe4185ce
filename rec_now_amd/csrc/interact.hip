// Neighbours of the hot path, all HBM- or VALU-bound streaming kernels over rows:
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
// One wave per row.  The row's F x D block sits in LDS (row stride D+1: conflict-free per-lane reads); lane <-> column
// field c keeps x_c in registers, x_r is an LDS broadcast.  Stores of one r are consecutive in p: coalesced.
// VALU-bound: F*D FMAs per lane and row (half of them below the diagonal) against 4*F*(F-1)/2 output bytes.
// ---------------------------------------------------------------------------------------------------------------------
template <int DMAX>
__global__ void __launch_bounds__(256)
k_ipnn_fwd(const float* const* __restrict__ fields, int F, int64_t B, int D, float* __restrict__ out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int LD = D + 1;
    float* xs = lds + (size_t)w * F * LD;
    const int64_t P = (int64_t)F * (F - 1) / 2;
    for (int64_t b = (int64_t)blockIdx.x * nw + w; b < B; b += (int64_t)gridDim.x * nw) {
        for (int i = lane; i < F * D; i += 64) {
            const int f = i / D, d = i - f * D;
            xs[f * LD + d] = fields[f][b * D + d];
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
                for (int d = 0; d < DMAX; ++d)
                    if (d < D) s += xs[r * LD + d] * xc[d];
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
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int LD = D + 1;
    const int P = F * (F - 1) / 2;
    float* xs = lds + (size_t)w * (F * LD + P);
    float* gs = xs + F * LD;
    for (int64_t b = (int64_t)blockIdx.x * nw + w; b < B; b += (int64_t)gridDim.x * nw) {
        for (int i = lane; i < F * D; i += 64) {
            const int f = i / D, d = i - f * D;
            xs[f * LD + d] = fields[f][b * D + d];
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
                for (int d = 0; d < DMAX; ++d)
                    if (d < D) acc[d] += wgt * xs[g * LD + d];
            }
            if (f < F) {
                float* o = dfields[f] + b * D;
#pragma unroll
                for (int d = 0; d < DMAX; ++d)
                    if (d < D) o[d] = acc[d];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // reads done before the next row overwrites the tile
    }
}

static int ipnn_cfg(int F, int D, bool bwd, int* waves, size_t* lds) {
    const size_t per_wave = ((size_t)F * (D + 1) + (bwd ? (size_t)F * (F - 1) / 2 : 0)) * sizeof(float);
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
    int rc = ipnn_cfg(F, D, false, &waves, &lds);
    if (rc) return rc;
    int64_t g = (B + waves - 1) / waves;
    if (g > 4096) g = 4096;
    hipStream_t st = (hipStream_t)stream;
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
    int rc = ipnn_cfg(F, D, true, &waves, &lds);
    if (rc) return rc;
    int64_t g = (B + waves - 1) / waves;
    if (g > 4096) g = 4096;
    hipStream_t st = (hipStream_t)stream;
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
// Thread = (row b, field f) with f FASTEST: a wave covers 64 consecutive fields of one row, so the concatenated (B,total)
// tensors (out, dout) are touched as one contiguous run per wave, and each lane streams its own field row x_f[b][0..D_f) as
// float4s (every 64-byte line it touches is used completely).
__device__ __forceinline__ bool senet_vec(const float* p, int D) { return (D & 3) == 0 && ((uintptr_t)p & 15) == 0; }

__global__ void __launch_bounds__(256)
k_senet_squeeze(const float* const* __restrict__ fields, const int* __restrict__ dims, int F, int64_t B, float* __restrict__ sq) {
    const int64_t n = B * F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const int D = dims[f];
        const float* x = fields[f] + b * D;
        float s = 0.f;
        if (senet_vec(fields[f], D)) {
            for (int d = 0; d < D; d += 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + d);
                s += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int d = 0; d < D; ++d) s += x[d];
        }
        sq[i] = s / (float)D;
    }
}
__global__ void __launch_bounds__(256)
k_senet_scale(const float* const* __restrict__ fields, const int* __restrict__ dims, const int* __restrict__ offs, int F, int total,
              int64_t B, const float* __restrict__ w, float* __restrict__ out) {
    const int64_t n = B * F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const int D = dims[f];
        const float* x = fields[f] + b * D;
        float* o = out + b * total + offs[f];
        const float wf = w[i];
        if (senet_vec(fields[f], D) && senet_vec(o, D) && (total & 3) == 0) {
            for (int d = 0; d < D; d += 4) {
                float4 v = *reinterpret_cast<const float4*>(x + d);
                v.x *= wf; v.y *= wf; v.z *= wf; v.w *= wf;
                *reinterpret_cast<float4*>(o + d) = v;
            }
        } else {
            for (int d = 0; d < D; ++d) o[d] = x[d] * wf;
        }
    }
}
__global__ void __launch_bounds__(256)
k_senet_dw(const float* const* __restrict__ fields, const int* __restrict__ dims, const int* __restrict__ offs, int F, int total,
           int64_t B, const float* __restrict__ dout, float* __restrict__ dw) {
    const int64_t n = B * F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const int D = dims[f];
        const float* x = fields[f] + b * D;
        const float* g = dout + b * total + offs[f];
        float s = 0.f;
        if (senet_vec(fields[f], D) && senet_vec(g, D) && (total & 3) == 0) {
            for (int d = 0; d < D; d += 4) {
                const float4 v = *reinterpret_cast<const float4*>(x + d), q = *reinterpret_cast<const float4*>(g + d);
                s += (v.x * q.x + v.y * q.y) + (v.z * q.z + v.w * q.w);
            }
        } else {
            for (int d = 0; d < D; ++d) s += g[d] * x[d];
        }
        dw[i] = s;
    }
}
__global__ void __launch_bounds__(256)
k_senet_dx(float* const* __restrict__ dfields, const int* __restrict__ dims, const int* __restrict__ offs, int F, int total, int64_t B,
           const float* __restrict__ w, const float* __restrict__ dout, const float* __restrict__ dsq) {
    const int64_t n = B * F;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const int D = dims[f];
        float* dx = dfields[f] + b * D;
        const float* g = dout + b * total + offs[f];
        const float wf = w[i], q = dsq[i] / (float)D;
        if (senet_vec(dfields[f], D) && senet_vec(g, D) && (total & 3) == 0) {
            for (int d = 0; d < D; d += 4) {
                float4 v = *reinterpret_cast<const float4*>(g + d);
                v.x = v.x * wf + q; v.y = v.y * wf + q; v.z = v.z * wf + q; v.w = v.w * wf + q;
                *reinterpret_cast<float4*>(dx + d) = v;
            }
        } else {
            for (int d = 0; d < D; ++d) dx[d] = g[d] * wf + q;
        }
    }
}
static inline int senet_grid(int64_t B, int F) {
    int64_t g = (B * F + 255) / 256;
    if (g > 8192) g = 8192;
    return (int)(g > 0 ? g : 1);
}
extern "C" int recnow_senet_squeeze(const float* const* fields, const int32_t* dims, int F, int64_t B, float* sq, void* stream) {
    if (F < 1 || F > 65535 || B < 0) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dims || !sq) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_senet_squeeze, senet_grid(B, F), 256, 0, (hipStream_t)stream, fields, dims, F, B, sq);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_senet_scale_fwd(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                                      const float* w, float* out, void* stream) {
    if (F < 1 || F > 65535 || B < 0 || total < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dims || !offs || !w || !out) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_senet_scale, senet_grid(B, F), 256, 0, (hipStream_t)stream, fields, dims, offs, F, total, B, w, out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_senet_scale_bwd_w(const float* const* fields, const int32_t* dims, const int32_t* offs, int F, int total,
                                        int64_t B, const float* dout, float* dw, void* stream) {
    if (F < 1 || F > 65535 || B < 0 || total < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dims || !offs || !dout || !dw) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_senet_dw, senet_grid(B, F), 256, 0, (hipStream_t)stream, fields, dims, offs, F, total, B, dout, dw);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_senet_scale_bwd_x(float* const* dfields, const int32_t* dims, const int32_t* offs, int F, int total, int64_t B,
                                        const float* w, const float* dout, const float* dsq, void* stream) {
    if (F < 1 || F > 65535 || B < 0 || total < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!dfields || !dims || !offs || !w || !dout || !dsq) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_senet_dx, senet_grid(B, F), 256, 0, (hipStream_t)stream, dfields, dims, offs, F, total, B, w, dout, dsq);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
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
extern "C" int recnow_attention_dot_fwd(const float* user, const float* doc, int64_t B, int L, int D, int filter_neg, float* mat,
                                        float* score_sum, void* stream) {
    if (B < 0 || L < 0 || D < 1) return RECNOW_EINVAL;
    if (D > 64 * ATTN_NCH) return RECNOW_EUNSUPPORTED;
    if (B == 0) return RECNOW_OK;
    if ((L > 0 && !user) || !doc || !mat || !score_sum) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
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
    mod = gamma > 0.f ? powf(one_m_sim, gamma) : 1.f;
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
__global__ void k_focal_mean(const double* __restrict__ part, int n, int64_t B, float* __restrict__ mean) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        *mean = (float)(s / (double)B);
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
            const float pw = (oms > 0.f || gamma >= 1.f) ? powf(oms, gamma - 1.f) : 0.f;
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
        hipLaunchKernelGGL(k_focal_mean, 1, 64, 0, st, (const double*)ws, g, B, loss_mean);
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
