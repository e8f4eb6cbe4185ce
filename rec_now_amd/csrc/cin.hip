// CINLayer (xDeepFM Compressed Interaction Network), /root/reference/rec_now/layers/cin_layer.py:72-122:
//     X_k[b,d,c] = sum_{f,h} W_k[c, f*H_{k-1} + h] * x0[b,d,f] * X_{k-1}[b,d,h]            (:103-109)
// The reference materialises the outer product (B,D,F,H) with an einsum and multiplies it by W.  Here every (b,d)
// pair is a row m of an implicit GEMM  X_k (M x H_k) = Z (M x F*H_{k-1}) * W_k^T  whose A operand
// Z[m][(f,h)] = x0t[m][f] * X_{k-1}[m][h] is generated inside the operand load of the exact-fp32 MFMA GEMM
// (RECNOW_OPMODE_OUTER) -- Z never exists in HBM.  Bound: fp32 MFMA (2*D*F*sum_k H_{k-1}*H_k flop per sample forward).
// Backward (derived from the forward lines; TF autodiff in the reference) is three more such GEMMs per layer:
//     dW_k      = dX_k^T Z                                  (K = M rows, split-K slabs)
//     dX_{k-1}  = (dX_k (x) x0t) * W_k  viewed [(c,f)][h]
//     dx0t     += (dX_k (x) X_{k-1}) * W_k viewed [(c,h)][f]   (W_k re-laid out once per call)
// Layouts: x0t [M = B*D][F] (the reference's (B,D,F) transpose, :96-97); X_k [M][H_k].
#include "gemm.hpp"
#include "cin_bwd.hpp"

// x0t[(b*D + d)*F + f] = emb[b*(F*D) + f*D + d]
__global__ void k_cin_in(const float* __restrict__ emb, int64_t B, int D, int F, float* __restrict__ x0t) {
    const int64_t total = B * D * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const int64_t m = i / F;
        const int d = (int)(m % D);
        const int64_t b = m / D;
        x0t[i] = emb[b * ((int64_t)F * D) + (int64_t)f * D + d];
    }
}
__global__ void k_cin_in_bwd(const float* __restrict__ dx0t, int64_t B, int D, int F, float* __restrict__ demb) {
    const int64_t total = B * D * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes demb: [b][f][d]
        const int d = (int)(i % D);
        const int64_t t = i / D;
        const int f = (int)(t % F);
        const int64_t b = t / F;
        demb[i] = dx0t[(b * D + d) * F + f];
    }
}
// y[m] (+)= sum_c X[m][c]      one wave per row
__global__ void __launch_bounds__(256)
k_rowsum(const float* __restrict__ X, int64_t M, int H, float* __restrict__ y, int accumulate) {
    const int lane = threadIdx.x & 63;
    for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < H; c += 64) s += X[m * H + c];
        s = wave_sum(s);
        if (lane == 0) y[m] = accumulate ? y[m] + s : s;
    }
}
// dX[m][c] = dy[m]
__global__ void k_fill_rowbcast(const float* __restrict__ dy, int64_t M, int H, float* __restrict__ dX) {
    const int64_t total = M * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) dX[i] = dy[i / H];
}
// concat mode (:119-121): out[b][(coff + c)*D + d] = X[(b*D + d)][c]
__global__ void k_cin_concat(const float* __restrict__ X, int64_t B, int D, int H, int coff, int ctot, float* __restrict__ out) {
    const int64_t total = B * D * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes out's slice: [b][c][d]
        const int d = (int)(i % D);
        const int64_t t = i / D;
        const int c = (int)(t % H);
        const int64_t b = t / H;
        out[b * ((int64_t)ctot * D) + (int64_t)(coff + c) * D + d] = X[(b * D + d) * H + c];
    }
}
__global__ void k_cin_concat_bwd(const float* __restrict__ dout, int64_t B, int D, int H, int coff, int ctot, float* __restrict__ dX) {
    const int64_t total = B * D * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes dX: [m = b*D + d][c]
        const int c = (int)(i % H);
        const int64_t m = i / H;
        const int d = (int)(m % D);
        const int64_t b = m / D;
        dX[i] = dout[b * ((int64_t)ctot * D) + (int64_t)(coff + c) * D + d];
    }
}
// Wt[c][h][f] = W[c][f*H + h]
__global__ void k_cin_wt(const float* __restrict__ W, int Hk, int F, int H, float* __restrict__ Wt) {
    const int64_t total = (int64_t)Hk * F * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const int64_t t = i / F;
        const int h = (int)(t % H);
        const int c = (int)(t / H);
        Wt[i] = W[(int64_t)c * F * H + (int64_t)f * H + h];
    }
}

static inline int ew_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    return (int)(g > 0 ? g : 1);
}

struct CinDims {
    int64_t B, M;
    int D, F, L, Hmax, ctot;
    const int* H;     // hidden sizes, host
};
static bool cin_dims(int64_t B, int D, int F, const int* hidden_host, int L, int output_input, CinDims* c) {
    if (B < 0 || D < 1 || F < 1 || L < 1 || !hidden_host) return false;
    c->B = B; c->M = B * D; c->D = D; c->F = F; c->L = L; c->H = hidden_host;
    c->Hmax = F;
    c->ctot = output_input ? F : 0;
    for (int k = 0; k < L; ++k) {
        if (hidden_host[k] < 1) return false;
        if (hidden_host[k] > c->Hmax) c->Hmax = hidden_host[k];
        c->ctot += hidden_host[k];
    }
    return c->M <= 0x7fffffffll;
}
static inline int hprev(const CinDims& c, int k /* 0-based layer */) { return k == 0 ? c.F : c.H[k - 1]; }

// saved: x0t (M x F), then X_1 .. X_L (M x H_k)
extern "C" size_t recnow_cin_saved_bytes(int64_t B, int D, int F, const int* hidden_host, int L) {
    CinDims c;
    if (!cin_dims(B, D, F, hidden_host, L, 0, &c)) return 0;
    size_t s = rn_align((size_t)c.M * F * sizeof(float));
    for (int k = 0; k < L; ++k) s += rn_align((size_t)c.M * hidden_host[k] * sizeof(float));
    return s + 256;
}
static size_t cin_gemm_ws(const CinDims& c) {
    size_t best = 0;
    recnow_gemm_desc d = rn_gemm_desc_zero();
    for (int k = 0; k < c.L; ++k) {
        const int Hk = c.H[k], Hp = hprev(c, k);
        const long long shapes[4][3] = {{c.M, Hk, (long long)c.F * Hp}, {Hk, (long long)c.F * Hp, c.M}, {c.M, Hp, (long long)Hk * c.F},
                                        {c.M, c.F, (long long)Hk * Hp}};
        for (int i = 0; i < 4; ++i) {
            d.M = (int)shapes[i][0]; d.N = (int)shapes[i][1]; d.K = (int)shapes[i][2];
            const size_t s = rn_gemm_ws_bytes(&d);
            if (s > best) best = s;
        }
    }
    return best;
}
extern "C" size_t recnow_cin_workspace_bytes(int64_t B, int D, int F, const int* hidden_host, int L) {
    CinDims c;
    if (!cin_dims(B, D, F, hidden_host, L, 0, &c)) return 0;
    size_t s = 0;
    s += rn_align((size_t)c.Hmax * F * c.Hmax * sizeof(float));      // Wt
    s += 2 * rn_align((size_t)c.M * c.Hmax * sizeof(float));         // dX ping-pong
    s += rn_align((size_t)c.M * F * sizeof(float));                  // dx0t
    s += cin_gemm_ws(c);
    return s + 256;
}

// emb: (B, F*D) fp32 (the reference's concat of the F field embeddings, :88-91); weights_host[k]: (H_k, F*H_{k-1});
// out: (B, D) when sum_channel else (B, ctot*D), ctot = [F +] sum_k H_k.
extern "C" int recnow_cin_fwd(const float* emb, const float* const* weights_host, int64_t B, int D, int F, const int* hidden_host,
                              int L, int output_input, int sum_channel, float* out, void* saved, size_t saved_bytes, void* ws,
                              size_t ws_bytes, void* stream) {
    CinDims c;
    if (!cin_dims(B, D, F, hidden_host, L, output_input, &c)) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!emb || !weights_host || !out || !saved || !ws) return RECNOW_EINVAL;
    if (saved_bytes < recnow_cin_saved_bytes(B, D, F, hidden_host, L) || ws_bytes < recnow_cin_workspace_bytes(B, D, F, hidden_host, L))
        return RECNOW_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    RnCarver sv(saved, saved_bytes);
    float* x0t = sv.take<float>((size_t)c.M * F);
    RnCarver wc(ws, ws_bytes);
    wc.take<float>((size_t)c.Hmax * F * c.Hmax);
    wc.take<float>((size_t)c.M * c.Hmax);
    wc.take<float>((size_t)c.M * c.Hmax);
    wc.take<float>((size_t)c.M * F);
    void* gws = wc.base + wc.off;
    const size_t gws_bytes = ws_bytes - wc.off;

    hipLaunchKernelGGL(k_cin_in, ew_grid(c.M * F), 256, 0, st, emb, B, D, F, x0t);
    RN_LAUNCH_CHECK();
    int coff = 0;
    if (output_input) {
        if (sum_channel) hipLaunchKernelGGL(k_rowsum, ew_grid(c.M * 64), 256, 0, st, x0t, c.M, F, out, 0);
        else hipLaunchKernelGGL(k_cin_concat, ew_grid(c.M * F), 256, 0, st, x0t, B, D, F, 0, c.ctot, out);
        coff = F;
    }
    const float* Xp = x0t;
    int rc;
    for (int k = 0; k < L; ++k) {
        const int Hk = c.H[k], Hp = hprev(c, k);
        float* Xk = sv.take<float>((size_t)c.M * Hk);
        recnow_gemm_desc d = rn_gemm_desc_zero();
        d.A = Xp; d.lda = Hp; d.A2 = x0t; d.a_ld2 = F; d.a_hq = Hp; d.a_mode = RECNOW_OPMODE_OUTER; d.a_trans = 0;
        d.B = weights_host[k]; d.ldb = (int64_t)F * Hp; d.b_trans = 1;
        d.C = Xk; d.ldc = Hk;
        d.M = (int)c.M; d.N = Hk; d.K = F * Hp;
        if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        if (sum_channel) hipLaunchKernelGGL(k_rowsum, ew_grid(c.M * 64), 256, 0, st, Xk, c.M, Hk, out, (output_input || k > 0) ? 1 : 0);
        else hipLaunchKernelGGL(k_cin_concat, ew_grid(c.M * Hk), 256, 0, st, Xk, B, D, Hk, coff, c.ctot, out);
        RN_LAUNCH_CHECK();
        coff += Hk;
        Xp = Xk;
    }
    return RECNOW_OK;
}

// dout: gradient of `out`; demb: (B, F*D); dweights_host[k]: (H_k, F*H_{k-1}).
extern "C" int recnow_cin_bwd(const float* const* weights_host, const float* dout, const void* saved, size_t saved_bytes, int64_t B,
                              int D, int F, const int* hidden_host, int L, int output_input, int sum_channel, float* demb,
                              float* const* dweights_host, void* ws, size_t ws_bytes, void* stream) {
    CinDims c;
    if (!cin_dims(B, D, F, hidden_host, L, output_input, &c)) return RECNOW_EINVAL;
    if (!dweights_host) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        for (int k = 0; k < L; ++k)
            RN_HIP(hipMemsetAsync(dweights_host[k], 0, (size_t)hidden_host[k] * F * hprev(c, k) * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!weights_host || !dout || !saved || !demb || !ws) return RECNOW_EINVAL;
    if (saved_bytes < recnow_cin_saved_bytes(B, D, F, hidden_host, L) || ws_bytes < recnow_cin_workspace_bytes(B, D, F, hidden_host, L))
        return RECNOW_EWORKSPACE;
    RnCarver sv(const_cast<void*>(saved), saved_bytes);
    const float* x0t = sv.take<float>((size_t)c.M * F);
    const float* X[64];
    if (L > 64) return RECNOW_EUNSUPPORTED;
    for (int k = 0; k < L; ++k) X[k] = sv.take<float>((size_t)c.M * c.H[k]);
    RnCarver wc(ws, ws_bytes);
    float* Wt = wc.take<float>((size_t)c.Hmax * F * c.Hmax);
    float* dXa = wc.take<float>((size_t)c.M * c.Hmax);
    float* dXb = wc.take<float>((size_t)c.M * c.Hmax);
    float* dx0t = wc.take<float>((size_t)c.M * F);
    void* gws = wc.base + wc.off;
    const size_t gws_bytes = ws_bytes - wc.off;

    // channel offset of every kept layer in concat mode
    int coffs[65];
    coffs[0] = output_input ? F : 0;
    for (int k = 0; k < L; ++k) coffs[k + 1] = coffs[k] + c.H[k];

    // gradient of `out` w.r.t. one stored layer: broadcast of dy (sum mode) or a gathered slice (concat mode)
    auto seed = [&](float* dst, int H, int coff) {
        if (sum_channel) hipLaunchKernelGGL(k_fill_rowbcast, ew_grid(c.M * H), 256, 0, st, dout, c.M, H, dst);
        else hipLaunchKernelGGL(k_cin_concat_bwd, ew_grid(c.M * H), 256, 0, st, dout, B, D, H, coff, c.ctot, dst);
    };
    if (output_input) seed(dx0t, F, 0);
    else RN_HIP(hipMemsetAsync(dx0t, 0, (size_t)c.M * F * sizeof(float), st));
    float* dXk = dXa;
    float* dXp = dXb;
    seed(dXk, c.H[L - 1], coffs[L - 1]);
    RN_LAUNCH_CHECK();
    int rc;
    for (int k = L - 1; k >= 0; --k) {
        const int Hk = c.H[k], Hp = hprev(c, k);
        const float* Xp = (k == 0) ? x0t : X[k - 1];
        {   // dW_k = dX_k^T Z,  Z[m][(f,h)] = x0t[m][f] * Xp[m][h]
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = dXk; d.lda = Hk; d.a_trans = 1;
            d.B = Xp; d.ldb = Hp; d.B2 = x0t; d.b_ld2 = F; d.b_hq = Hp; d.b_mode = RECNOW_OPMODE_OUTER; d.b_trans = 0;
            d.C = dweights_host[k]; d.ldc = (int64_t)F * Hp;
            d.M = Hk; d.N = F * Hp; d.K = (int)c.M;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        }
        // Data gradients.  Fused (csrc/cin_bwd.hip, round 4): T = dX_k W_k is formed ONCE on the matrix cores and both reductions -- over f with x0
        // for dX_{k-1}, over h with X_{k-1} for dx0 -- run on the accumulator tile: one forward-sized product instead of two.  Other shapes
        // (rows not a multiple of 128, H_{k-1} not 64 / 128, ...) keep the two products with generated outer-product operands.
        static const bool fused_on = []() { const char* e = getenv("RECNOW_CIN_FUSED"); return !e || e[0] != '0'; }();      // A/B switch
        if (fused_on && rn_cin_bwd_fused_supported(c.M, Hk, Hp, F)) {
            float* dst = (k == 0) ? dx0t : dXp;
            if (k > 0) seed(dst, Hp, coffs[k - 1]);          // X_{k-1}'s own share of the output gradient, then accumulate
            if ((rc = rn_cin_bwd_fused(dXk, weights_host[k], x0t, Xp, dst, dx0t, c.M, Hk, Hp, F, st))) return rc;
        } else {
        {   // gradient w.r.t. X_{k-1}:  (dX_k (x) x0t)[m][(c,f)] * W_k viewed [(c,f)][h]
            float* dst = (k == 0) ? dx0t : dXp;
            if (k > 0) seed(dst, Hp, coffs[k - 1]);          // X_{k-1}'s own share of the output gradient, then accumulate
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = x0t; d.lda = F; d.A2 = dXk; d.a_ld2 = Hk; d.a_hq = F; d.a_mode = RECNOW_OPMODE_OUTER; d.a_trans = 0;
            d.B = weights_host[k]; d.ldb = Hp; d.b_trans = 0;
            d.C = dst; d.ldc = Hp;
            d.M = (int)c.M; d.N = Hp; d.K = Hk * F;
            d.accumulate = 1;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        }
        {   // dx0t += (dX_k (x) Xp)[m][(c,h)] * W_k viewed [(c,h)][f]
            hipLaunchKernelGGL(k_cin_wt, ew_grid((int64_t)Hk * F * Hp), 256, 0, st, weights_host[k], Hk, F, Hp, Wt);
            RN_LAUNCH_CHECK();
            recnow_gemm_desc d = rn_gemm_desc_zero();
            d.A = Xp; d.lda = Hp; d.A2 = dXk; d.a_ld2 = Hk; d.a_hq = Hp; d.a_mode = RECNOW_OPMODE_OUTER; d.a_trans = 0;
            d.B = Wt; d.ldb = F; d.b_trans = 0;
            d.C = dx0t; d.ldc = F;
            d.M = (int)c.M; d.N = F; d.K = Hk * Hp;
            d.accumulate = 1;
            if ((rc = rn_gemm(&d, gws, gws_bytes, st))) return rc;
        }
        }
        float* t = dXk; dXk = dXp; dXp = t;
    }
    hipLaunchKernelGGL(k_cin_in_bwd, ew_grid(c.M * F), 256, 0, st, dx0t, B, D, F, demb);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
