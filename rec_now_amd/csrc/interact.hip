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
// Fast path for the usual case of equal field widths D with D % 4 == 0 and D/4 a power of two: thread = (row b, float4 q of
// the concatenated row).  Every access is a float4; the concatenated tensors are touched fully coalesced, the field tensors
// in 4*D-byte runs; nothing is staged and all loads of a thread are independent.  (row, field) sums reduce over the D/4
// adjacent lanes of the field with shuffles (groups are lane-aligned because total/4 is a multiple of D/4).
template <int MODE>
__global__ void __launch_bounds__(256)
k_senet_uniform(const float* const* __restrict__ fields, float* const* __restrict__ dfields, int F, int D, int64_t B,
                const float* __restrict__ w, const float* __restrict__ in, const float* __restrict__ dsq, float* __restrict__ out) {
    const int Q = D / 4, QT = F * Q;                 // float4s per field row / per concatenated row
    const int64_t n = B * QT;
    const int64_t nround = (n + 255) / 256 * 256;    // whole waves stay in the loop: the shuffles need every lane
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nround; i += (int64_t)gridDim.x * 256) {
        const bool ok = i < n;
        const int64_t b = ok ? i / QT : 0;
        const int q = ok ? (int)(i - b * QT) : 0;
        const int f = q / Q, d = (q - f * Q) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE != SENET_MODE_DX && ok) v = *reinterpret_cast<const float4*>(fields[f] + b * D + d);
        if (MODE == SENET_MODE_SCALE) {
            if (ok) {
                const float wf = w[b * F + f];
                *reinterpret_cast<float4*>(out + b * (int64_t)(F * D) + q * 4) = make_float4(v.x * wf, v.y * wf, v.z * wf, v.w * wf);
            }
        } else if (MODE == SENET_MODE_DX) {
            if (ok) {
                const float4 g = *reinterpret_cast<const float4*>(in + b * (int64_t)(F * D) + q * 4);
                const float wf = w[b * F + f], qq = dsq[b * F + f] / (float)D;
                *reinterpret_cast<float4*>(dfields[f] + b * D + d) = make_float4(g.x * wf + qq, g.y * wf + qq, g.z * wf + qq, g.w * wf + qq);
            }
        } else {
            float s;
            if (MODE == SENET_MODE_DW) {
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) g = *reinterpret_cast<const float4*>(in + b * (int64_t)(F * D) + q * 4);
                s = (v.x * g.x + v.y * g.y) + (v.z * g.z + v.w * g.w);
            } else {
                s = (v.x + v.y) + (v.z + v.w);
            }
            for (int o = Q / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (ok && d == 0) out[b * F + f] = MODE == SENET_MODE_SQUEEZE ? s / (float)D : s;
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
        const int64_t n = B * (total / 4);
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
