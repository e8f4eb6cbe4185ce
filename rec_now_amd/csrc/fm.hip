// FM second-order interaction, /root/reference/rec_now/layers/fm_layer.py:24-42:
//   y[b] = 0.5 * sum_d [ (sum_f x[f][b][d])^2 - sum_f x[f][b][d]^2 ]          dx[f][b][d] = g[b] * (S[b][d] - x[f][b][d])
//
// HBM-bound: algorithmic traffic 4*B*F*D (fwd) + 8*B*F*D (bwd) bytes.  The reference takes a LIST of F (B,D) tensors;
// the kernel takes a device array of F base pointers, so no stacking copy is made.  Every field tensor is a flat
// contiguous array of B*D floats; a thread owns one 16-byte chunk q of that flat index space for all F fields, so each
// load instruction of a wave is 1 KiB contiguous (coalesced) and the loop over fields keeps 8 loads in flight.
// S (the per-row field sum, B*D floats = 1/F of the input) is saved by forward so backward reads x exactly once.
#include "common.hpp"

#define FM_UNROLL 8
// field streams of the wide kernels are non-temporal (round 4, A/B on one box: forward 89.7 -> 78.9-81.2 us = 6.0 -> 6.6-6.8 TB/s, backward 206.5 -> 194.6-195.1 us);
// -DRN_STREAM_PLAIN (tools/build_variant.py) builds the plain-access variant
#ifndef RN_STREAM_PLAIN
#define FM_LD(p, i) __builtin_nontemporal_load((p) + (i))
#define FM_ST(p, i, v) __builtin_nontemporal_store((v), (p) + (i))
#else
#define FM_LD(p, i) ((p)[i])
#define FM_ST(p, i, v) ((p)[i] = (v))
#endif

template <bool SAVE_S>
__global__ void __launch_bounds__(256)
k_fm_fwd_vec4(const float* const* __restrict__ fields, int F, int64_t nchunk /* B*D/4 */, int lanes_per_row /* D/4, pow2 <= 64 */,
              float* __restrict__ y, float* __restrict__ S) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // grid-stride in units that keep all lanes of a row inside one wave: blockDim (256) is a multiple of lanes_per_row
    for (int64_t q0 = (int64_t)blockIdx.x * blockDim.x; q0 < nchunk; q0 += stride) {
        const int64_t q = q0 + threadIdx.x;
        const bool ok = q < nchunk;
        // Lanes past the end read chunk 0 (no branch around a load); nchunk is a multiple of lanes_per_row, so such a lane's
        // whole row is past the end and nothing of it is stored.
        const int64_t qc = ok ? q : 0;
        rn_f4 s = {0.f, 0.f, 0.f, 0.f}, sq = {0.f, 0.f, 0.f, 0.f};
        int f = 0;
        for (; f + FM_UNROLL <= F; f += FM_UNROLL) {
            rn_gcf4 p[FM_UNROLL];
            rn_f4 v[FM_UNROLL];
#pragma unroll
            for (int u = 0; u < FM_UNROLL; ++u) p[u] = (rn_gcf4)fields[f + u];       // 8 scalar fetches, one wait
#pragma unroll
            for (int u = 0; u < FM_UNROLL; ++u) v[u] = p[u][qc];                     // 8 global loads in flight
#pragma unroll
            for (int u = 0; u < FM_UNROLL; ++u) {
                s += v[u];
                sq += v[u] * v[u];
            }
        }
        for (; f < F; ++f) {
            const rn_f4 v = ((rn_gcf4)fields[f])[qc];
            s += v;
            sq += v * v;
        }
        if (SAVE_S && ok) reinterpret_cast<rn_f4*>(S)[q] = s;
        float r = (s.x * s.x - sq.x) + (s.y * s.y - sq.y) + (s.z * s.z - sq.z) + (s.w * s.w - sq.w);
        for (int o = lanes_per_row >> 1; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        if (ok && (q % lanes_per_row) == 0) y[q / lanes_per_row] = 0.5f * r;
    }
}

// The same with CPL chunks per thread: a wave reads CPL x 1 KiB = 4 KiB CONTIGUOUS of every field (CPL coalesced instructions), FU
// fields at a time -- FU * CPL loads in flight per lane.  One KiB per wave and field left the 64 field streams at 3.9 TB/s.
template <bool SAVE_S, int CPL, int FU>
__global__ void __launch_bounds__(256)
k_fm_fwd_wide(const float* const* __restrict__ fields, int F, int64_t nchunk, int lanes_per_row, float* __restrict__ y, float* __restrict__ S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per_blk = 256 * CPL;
    for (int64_t q0 = (int64_t)blockIdx.x * per_blk; q0 < nchunk; q0 += (int64_t)gridDim.x * per_blk) {
        int64_t q[CPL], qc[CPL];
        rn_f4 s[CPL], sq[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            q[c] = q0 + (int64_t)(wave * CPL + c) * 64 + lane;
            qc[c] = q[c] < nchunk ? q[c] : 0;          // lanes past the end read chunk 0 (no branch around a load); their rows are not stored
            s[c] = sq[c] = rn_f4{0.f, 0.f, 0.f, 0.f};
        }
        int f = 0;
        for (; f + FU <= F; f += FU) {
            rn_gcf4 p[FU];
            rn_f4 v[FU][CPL];
#pragma unroll
            for (int u = 0; u < FU; ++u) p[u] = (rn_gcf4)fields[f + u];
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int c = 0; c < CPL; ++c) v[u][c] = FM_LD(p[u], qc[c]);
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    s[c] += v[u][c];
                    sq[c] += v[u][c] * v[u][c];
                }
        }
        for (; f < F; ++f) {
            const rn_gcf4 p = (rn_gcf4)fields[f];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const rn_f4 v = p[qc[c]];
                s[c] += v;
                sq[c] += v * v;
            }
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const bool ok = q[c] < nchunk;
            if (SAVE_S && ok) reinterpret_cast<rn_f4*>(S)[q[c]] = s[c];
            float r = (s[c].x * s[c].x - sq[c].x) + (s[c].y * s[c].y - sq[c].y) + (s[c].z * s[c].z - sq[c].z) + (s[c].w * s[c].w - sq[c].w);
            for (int o = lanes_per_row >> 1; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
            if (ok && (q[c] % lanes_per_row) == 0) y[q[c] / lanes_per_row] = 0.5f * r;
        }
    }
}

// generic shapes (D not a multiple of 4 or D/4 not a power of two <= 64): one thread per row
template <bool SAVE_S>
__global__ void __launch_bounds__(256)
k_fm_fwd_generic(const float* const* __restrict__ fields, int F, int64_t B, int D, float* __restrict__ y, float* __restrict__ S) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float r = 0.f;
    for (int d = 0; d < D; ++d) {
        float s = 0.f, sq = 0.f;
        for (int f = 0; f < F; ++f) {
            const float v = fields[f][b * D + d];
            s += v;
            sq += v * v;
        }
        if (SAVE_S) S[b * D + d] = s;
        r += s * s - sq;
    }
    y[b] = 0.5f * r;
}

__global__ void __launch_bounds__(256)
k_fm_bwd_vec4(const float* const* __restrict__ fields, float* const* __restrict__ dfields, int F, int64_t nchunk,
              int lanes_per_row, const float* __restrict__ S, const float* __restrict__ gy) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nchunk; q += stride) {
        const float g = gy[q / lanes_per_row];
        const rn_f4 s = reinterpret_cast<const rn_f4*>(S)[q];
        int f = 0;
        // software pipeline over groups of FM_UNROLL fields: the loads of group n+1 are issued before the stores of group n,
        // so a wave always has loads in flight (program order "8 loads, wait, 8 stores" exposed the load latency per group)
        rn_f4 v[FM_UNROLL], nv[FM_UNROLL];
        if (FM_UNROLL <= F) {
            rn_gcf4 p[FM_UNROLL];
#pragma unroll
            for (int u = 0; u < FM_UNROLL; ++u) p[u] = (rn_gcf4)fields[u];
#pragma unroll
            for (int u = 0; u < FM_UNROLL; ++u) v[u] = p[u][q];
        }
        for (; f + FM_UNROLL <= F; f += FM_UNROLL) {
            const bool more = f + 2 * FM_UNROLL <= F;            // block-uniform
            rn_gf4 dp[FM_UNROLL];
#pragma unroll
            for (int u = 0; u < FM_UNROLL; ++u) dp[u] = (rn_gf4)dfields[f + u];
            if (more) {
                rn_gcf4 p[FM_UNROLL];
#pragma unroll
                for (int u = 0; u < FM_UNROLL; ++u) p[u] = (rn_gcf4)fields[f + FM_UNROLL + u];
#pragma unroll
                for (int u = 0; u < FM_UNROLL; ++u) nv[u] = p[u][q];
            }
#pragma unroll
            for (int u = 0; u < FM_UNROLL; ++u) dp[u][q] = g * (s - v[u]);
            if (more) {
#pragma unroll
                for (int u = 0; u < FM_UNROLL; ++u) v[u] = nv[u];
            }
        }
        for (; f < F; ++f) {
            const rn_f4 v = ((rn_gcf4)fields[f])[q];
            ((rn_gf4)dfields[f])[q] = g * (s - v);
        }
    }
}

// CPL chunks per thread, as k_fm_fwd_wide: 4 KiB contiguous per wave and field for the loads and for the stores; the loads of the next
// FU fields are issued before the stores of the current ones.
template <int CPL, int FU>
__global__ void __launch_bounds__(256)
k_fm_bwd_wide(const float* const* __restrict__ fields, float* const* __restrict__ dfields, int F, int64_t nchunk, int lanes_per_row,
              const float* __restrict__ S, const float* __restrict__ gy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per_blk = 256 * CPL;
    // the host launches this kernel only when nchunk is a multiple of 256 * CPL: every chunk index below is in range
    for (int64_t q0 = (int64_t)blockIdx.x * per_blk; q0 < nchunk; q0 += (int64_t)gridDim.x * per_blk) {
        int64_t q[CPL];
        rn_f4 s[CPL];
        float g[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            q[c] = q0 + (int64_t)(wave * CPL + c) * 64 + lane;
            g[c] = gy[q[c] / lanes_per_row];
            s[c] = reinterpret_cast<const rn_f4*>(S)[q[c]];
        }
        rn_f4 v[FU][CPL], nv[FU][CPL];
        int f = 0;
        if (FU <= F) {
            rn_gcf4 p[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) p[u] = (rn_gcf4)fields[u];
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int c = 0; c < CPL; ++c) v[u][c] = FM_LD(p[u], q[c]);
        }
        for (; f + FU <= F; f += FU) {
            const bool more = f + 2 * FU <= F;            // block-uniform
            rn_gf4 dp[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) dp[u] = (rn_gf4)dfields[f + u];
            if (more) {
                rn_gcf4 p[FU];
#pragma unroll
                for (int u = 0; u < FU; ++u) p[u] = (rn_gcf4)fields[f + FU + u];
#pragma unroll
                for (int u = 0; u < FU; ++u)
#pragma unroll
                    for (int c = 0; c < CPL; ++c) nv[u][c] = FM_LD(p[u], q[c]);
            }
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int c = 0; c < CPL; ++c) FM_ST(dp[u], q[c], g[c] * (s[c] - v[u][c]));
            if (more) {
#pragma unroll
                for (int u = 0; u < FU; ++u)
#pragma unroll
                    for (int c = 0; c < CPL; ++c) v[u][c] = nv[u][c];
            }
        }
        for (; f < F; ++f) {
            const rn_gcf4 p = (rn_gcf4)fields[f];
            const rn_gf4 dp = (rn_gf4)dfields[f];
#pragma unroll
            for (int c = 0; c < CPL; ++c) dp[q[c]] = g[c] * (s[c] - p[q[c]]);
        }
    }
}

__global__ void __launch_bounds__(256)
k_fm_bwd_generic(const float* const* __restrict__ fields, float* const* __restrict__ dfields, int F, int64_t B, int D,
                 const float* __restrict__ S, const float* __restrict__ gy) {
    const int64_t n = B * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = gy[i / D], s = S[i];
        for (int f = 0; f < F; ++f) dfields[f][i] = g * (s - fields[f][i]);
    }
}

static inline bool fm_vec_ok(int D) {
    if (D % 4) return false;
    const int l = D / 4;
    return l >= 1 && l <= 64 && (l & (l - 1)) == 0;
}

static inline int fm_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    const int64_t cap = 256 * 8;          // 256 CUs x 8 blocks of 4 waves
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

// fields: device array of F pointers, each to a contiguous (B,D) fp32 tensor (16-byte aligned when D % 4 == 0).
// y: [B];  S: [B*D] saved for backward (may be NULL: forward only).
extern "C" int recnow_fm_fwd(const float* const* fields, int F, int64_t B, int D, float* y, float* S, void* stream) {
    if (F < 1 || B < 0 || D < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !y) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (fm_vec_ok(D)) {
        const int64_t nchunk = B * D / 4;
        static const int wide = []() { const char* e = getenv("RECNOW_FM_WIDE"); return e ? atoi(e) : 1; }();      // A/B switch: 0 = one chunk per thread
        if (wide && nchunk >= 256 * 4 * 512) {          // enough chunks for 512 workgroups of 4 per thread
            const int g = fm_grid((nchunk + 3) / 4);
            if (S) hipLaunchKernelGGL((k_fm_fwd_wide<true, 4, 4>), g, 256, 0, st, fields, F, nchunk, D / 4, y, S);
            else hipLaunchKernelGGL((k_fm_fwd_wide<false, 4, 4>), g, 256, 0, st, fields, F, nchunk, D / 4, y, S);
        } else if (S) hipLaunchKernelGGL(k_fm_fwd_vec4<true>, fm_grid(nchunk), 256, 0, st, fields, F, nchunk, D / 4, y, S);
        else hipLaunchKernelGGL(k_fm_fwd_vec4<false>, fm_grid(nchunk), 256, 0, st, fields, F, nchunk, D / 4, y, S);
    } else {
        if (S) hipLaunchKernelGGL(k_fm_fwd_generic<true>, rn_cdiv(B, 256), 256, 0, st, fields, F, B, D, y, S);
        else hipLaunchKernelGGL(k_fm_fwd_generic<false>, rn_cdiv(B, 256), 256, 0, st, fields, F, B, D, y, S);
    }
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// dfields: device array of F pointers to (B,D) gradient buffers; gy: [B] upstream gradient of y.
extern "C" int recnow_fm_bwd(const float* const* fields, float* const* dfields, int F, int64_t B, int D, const float* S,
                             const float* gy, void* stream) {
    if (F < 1 || B < 0 || D < 1) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!fields || !dfields || !S || !gy) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (fm_vec_ok(D)) {
        const int64_t nchunk = B * D / 4;
        static const int wide = []() { const char* e = getenv("RECNOW_FM_WIDE"); return e ? atoi(e) : 1; }();
        if (wide && nchunk >= 256 * 4 * 512 && nchunk % (256 * 4) == 0)
            hipLaunchKernelGGL((k_fm_bwd_wide<4, 4>), fm_grid(nchunk / 4), 256, 0, st, fields, dfields, F, nchunk, D / 4, S, gy);
        else hipLaunchKernelGGL(k_fm_bwd_vec4, fm_grid(nchunk), 256, 0, st, fields, dfields, F, nchunk, D / 4, S, gy);
    } else {
        hipLaunchKernelGGL(k_fm_bwd_generic, fm_grid(B * D), 256, 0, st, fields, dfields, F, B, D, S, gy);
    }
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
