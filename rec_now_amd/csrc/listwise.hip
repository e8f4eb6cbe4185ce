// In-batch listwise loss on sorted segments.  Replaces the four SparseTensor -> dense (G,B) materialisations, the row
// reductions and the (G_valid,B) softmax of /root/reference/rec_now/rec_block/listwise_loss_from_batch.py:89-173 by
// per-segment reductions (one wave per group) -- 16*B bytes of traffic instead of O(G*B).
//
//   valid group  : has a label > th AND a (label - th) < 0                                  (:135-137)
//   row g of the reference's dense matrices = members' logits, padded with `pad_logit`
//                  (= value_of_masked_logit when do_mask_logits else 0; B - n_g padded entries)   (:139-140)
//   p_i = y_i / sum_{k in g} y_k                                                            (:144)
//   l_g = softmax_cross_entropy(p, row) = lse(row) * sum_i p_i - sum_i p_i s_i              (:167)
//   loss = mean over valid groups (first-occurrence order) of w_g l_g, NaN -> 0             (:168-172)
#include "common.hpp"
#include "scan.hpp"

// one wave per segment.  Outputs indexed by segment g (sorted numbering):
//   seg_valid, seg_lse (log-sum-exp of the FULL padded row), seg_ysum; first_row[g] = smallest member row.
//   valid_at_row[first_row] = seg_valid  (valid_at_row must be zero-filled)
__global__ void __launch_bounds__(256)
k_lw_stats(const float* __restrict__ labels, const float* __restrict__ logits, const int32_t* __restrict__ order,
           const int32_t* __restrict__ seg_first, const int32_t* __restrict__ n_seg, int64_t B, float th, float pad_logit,
           int32_t* __restrict__ seg_valid, float* __restrict__ seg_lse, float* __restrict__ seg_ysum,
           float* __restrict__ seg_psum, float* __restrict__ seg_pdot, int32_t* __restrict__ first_row,
           int32_t* __restrict__ valid_at_row) {
    const int lane = threadIdx.x & 63;
    const int G = n_seg[0] < 0 ? (int)B : n_seg[0];      // -1: the grouping timed out and left the identity grouping (B one-row lists, none valid)
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g < G; g += gridDim.x * 4) {
        const int s = seg_first[g], e = seg_first[g + 1];
        float mx = -INFINITY, ysum = 0.f;
        int pos = 0, neg = 0;
        for (int k = s + lane; k < e; k += 64) {
            const int r = order[k];
            const float y = labels[r], v = logits[r];
            mx = fmaxf(mx, v);
            ysum += y;
            pos |= (y > th);
            neg |= ((y - th) < 0.f);
        }
        mx = wave_max(mx);
        ysum = wave_sum(ysum);
        pos = __any(pos);
        neg = __any(neg);
        const int n_pad = (int)(B - (e - s));
        if (n_pad > 0) mx = fmaxf(mx, pad_logit);
        float z = 0.f, psum = 0.f, pdot = 0.f;
        for (int k = s + lane; k < e; k += 64) {
            const int r = order[k];
            const float v = logits[r], p = labels[r] / ysum;
            z += expf(v - mx);
            psum += p;
            pdot += p * v;
        }
        z = wave_sum(z);
        psum = wave_sum(psum);
        pdot = wave_sum(pdot);
        if (n_pad > 0) z += (float)n_pad * expf(pad_logit - mx);
        if (lane == 0) {
            const int valid = (pos && neg) ? 1 : 0;
            const int fr = order[s];
            seg_valid[g] = valid;
            seg_lse[g] = mx + logf(z);
            seg_ysum[g] = ysum;
            seg_psum[g] = psum;
            seg_pdot[g] = pdot;
            first_row[g] = fr;
            valid_at_row[fr] = valid;
        }
    }
}

// valid_rank[g] = index of segment g among the VALID groups in first-occurrence order, or -1
__global__ void k_lw_rank(const int32_t* __restrict__ seg_valid, const int32_t* __restrict__ first_row,
                          const int32_t* __restrict__ vscan_excl, const int32_t* __restrict__ n_seg, int64_t B,
                          int32_t* __restrict__ valid_rank, int32_t* __restrict__ n_valid) {
    const int G = n_seg[0] < 0 ? (int)B : n_seg[0];      // -1: the grouping timed out and left the identity grouping (B one-row lists, none valid)
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x)
        valid_rank[g] = seg_valid[g] ? vscan_excl[first_row[g]] : -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) n_valid[0] = vscan_excl[B];
}

// per-row gradient base and per-valid-group weighted loss (by valid rank)
//   dbase[i]  = w_g * (softmax_i * psum_g - p_i)   for rows of valid groups, else 0
//   row_rank[i] = valid rank of i's group or -1;  group_loss[rank] = w_g * (lse_g * psum_g - pdot_g)
__global__ void __launch_bounds__(256)
k_lw_grad(const float* __restrict__ labels, const float* __restrict__ logits, const int32_t* __restrict__ order,
          const int32_t* __restrict__ seg_id, const int32_t* __restrict__ seg_first, const int32_t* __restrict__ seg_valid,
          const float* __restrict__ seg_lse, const float* __restrict__ seg_ysum, const float* __restrict__ seg_psum,
          const float* __restrict__ seg_pdot, const int32_t* __restrict__ valid_rank, const float* __restrict__ weights,
          int64_t B, float* __restrict__ dbase, int32_t* __restrict__ row_rank, float* __restrict__ group_loss) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int r = order[k], g = seg_id[k];
    const int vr = valid_rank[g];
    float d = 0.f;
    if (vr >= 0) {
        const float w = weights ? weights[vr] : 1.f;
        const float p = labels[r] / seg_ysum[g];
        d = w * (expf(logits[r] - seg_lse[g]) * seg_psum[g] - p);
        if (k == seg_first[g]) group_loss[vr] = w * (seg_lse[g] * seg_psum[g] - seg_pdot[g]);
    }
    dbase[r] = d;
    row_rank[r] = vr;
}

// loss = mean(group_loss[0..Gv)) with NaN -> 0 (nan_to_zero, :74-86); single block, fixed order
__global__ void __launch_bounds__(1024)
k_lw_mean(const float* __restrict__ group_loss, const int32_t* __restrict__ n_valid, float* __restrict__ loss) {
    __shared__ double red[16];
    const int n = n_valid[0];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)group_loss[i];
    s = block_sum<double>(s, red);
    if (threadIdx.x == 0) {
        float v = n > 0 ? (float)(s / (double)n) : NAN;
        if (isnan(v)) v = 0.f;
        *loss = v;
    }
}

// dense materialisation for API parity with to_listwise_sample (:131-148): buffers (Gv,B) are pre-filled by the caller
// (mask 0, labels 0, logits pad_logit); members of valid groups are scattered in.
__global__ void __launch_bounds__(256)
k_lw_dense(const float* __restrict__ labels, const float* __restrict__ logits, const int32_t* __restrict__ order,
           const int32_t* __restrict__ seg_id, const float* __restrict__ seg_ysum, const int32_t* __restrict__ valid_rank,
           int64_t B, uint8_t* __restrict__ mask_out, float* __restrict__ labels_out, float* __restrict__ logits_out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int r = order[k], g = seg_id[k];
    const int vr = valid_rank[g];
    if (vr < 0) return;
    const int64_t o = (int64_t)vr * B + r;
    mask_out[o] = 1;
    labels_out[o] = labels[r] / seg_ysum[g];
    logits_out[o] = logits[r];
}
__global__ void __launch_bounds__(256)
k_lw_dense_bwd(const float* __restrict__ ddense, const int32_t* __restrict__ row_rank, int64_t B, float* __restrict__ dlogits) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B) return;
    const int vr = row_rank[r];
    dlogits[r] = vr >= 0 ? ddense[(int64_t)vr * B + r] : 0.f;
}

// ---- softmax cross-entropy over the rows of dense (G,N) matrices (listwise_loss_via_..., :166-167) -----------------
// one 256-thread workgroup per row; float4 streaming when N % 4 == 0
__global__ void __launch_bounds__(256)
k_softmax_ce_rows_fwd(const float* __restrict__ labels, const float* __restrict__ logits, int64_t N, float* __restrict__ row_loss,
                      float* __restrict__ row_lse, float* __restrict__ row_psum) {
    __shared__ float red[16];
    const int64_t g = blockIdx.x;
    const float* lg = logits + g * N;
    const float* lb = labels + g * N;
    float mx = -INFINITY, psum = 0.f, pdot = 0.f;
    for (int64_t j = threadIdx.x; j < N; j += 256) {
        const float v = lg[j], p = lb[j];
        mx = fmaxf(mx, v);
        psum += p;
        pdot += p * v;
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float z = 0.f;
    for (int64_t j = threadIdx.x; j < N; j += 256) z += expf(lg[j] - mx);
    z = block_sum<float>(z, red);
    psum = block_sum<float>(psum, red);
    pdot = block_sum<float>(pdot, red);
    if (threadIdx.x == 0) {
        const float lse = mx + logf(z);
        row_lse[g] = lse;
        row_psum[g] = psum;
        row_loss[g] = lse * psum - pdot;
    }
}
// dlogits[g][j] = grow[g] * (exp(s - lse_g) * psum_g - p)
__global__ void __launch_bounds__(256)
k_softmax_ce_rows_bwd(const float* __restrict__ labels, const float* __restrict__ logits, const float* __restrict__ row_lse,
                      const float* __restrict__ row_psum, const float* __restrict__ grow, int64_t N, float* __restrict__ dlogits) {
    const int64_t g = blockIdx.y;
    const float lse = row_lse[g], ps = row_psum[g], gr = grow[g];
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < N; j += (int64_t)gridDim.x * 256)
        dlogits[g * N + j] = gr * (expf(logits[g * N + j] - lse) * ps - labels[g * N + j]);
}

// ---- host side -------------------------------------------------------------------------------------------------
extern "C" size_t recnow_listwise_workspace_bytes(int64_t B) {
    if (B < 0) return 0;
    const size_t n = (size_t)(B > 0 ? B : 1) + 1;
    return 3 * rn_align(n * sizeof(int32_t)) + rn_scan_ws_bytes(B) + 256;     // first_row, valid_at_row, vscan
}

// Per-segment statistics.  Arrays seg_* and valid_rank are sized B (indexed by segment), n_valid: [1].
extern "C" int recnow_listwise_segments(const float* labels, const float* logits, const int32_t* order, const int32_t* seg_first,
                                        const int32_t* n_seg, int64_t B, float pos_neg_th, float pad_logit, int32_t* seg_valid,
                                        float* seg_lse, float* seg_ysum, float* seg_psum, float* seg_pdot, int32_t* valid_rank,
                                        int32_t* n_valid, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || !n_valid) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(n_valid, 0, sizeof(int32_t), st));
        return RECNOW_OK;
    }
    if (!labels || !logits || !order || !seg_first || !n_seg || !seg_valid || !seg_lse || !seg_ysum || !seg_psum || !seg_pdot ||
        !valid_rank || !ws)
        return RECNOW_EINVAL;
    if (ws_bytes < recnow_listwise_workspace_bytes(B)) return RECNOW_EWORKSPACE;
    RnCarver c(ws, ws_bytes);
    int32_t* first_row = c.take<int32_t>(B + 1);
    int32_t* valid_at_row = c.take<int32_t>(B + 1);
    int32_t* vscan = c.take<int32_t>(B + 1);
    void* sws = c.base + c.off;
    const size_t sws_bytes = ws_bytes - c.off;
    RN_HIP(hipMemsetAsync(valid_at_row, 0, (size_t)(B + 1) * sizeof(int32_t), st));
    int g = rn_cdiv(B, 4);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_lw_stats, g, 256, 0, st, labels, logits, order, seg_first, n_seg, B, pos_neg_th, pad_logit, seg_valid, seg_lse,
                       seg_ysum, seg_psum, seg_pdot, first_row, valid_at_row);
    RN_LAUNCH_CHECK();
    int rc = rn_scan<int32_t, int32_t, 0>(valid_at_row, vscan, B, 1, sws, sws_bytes, st);
    if (rc) return rc;
    int g2 = rn_cdiv(B, 256);
    if (g2 > 1024) g2 = 1024;
    hipLaunchKernelGGL(k_lw_rank, g2, 256, 0, st, seg_valid, first_row, vscan, n_seg, B, valid_rank, n_valid);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// Fused loss + gradient base.  weights: [n_valid] by valid rank or NULL.  loss: [1] (mean with NaN -> 0).
// dbase [B], row_rank [B], group_loss [B] (first n_valid entries = per-list losses in first-occurrence order).
extern "C" int recnow_listwise_loss_fwdbwd(const float* labels, const float* logits, const int32_t* order, const int32_t* seg_id,
                                           const int32_t* seg_first, const int32_t* seg_valid, const float* seg_lse,
                                           const float* seg_ysum, const float* seg_psum, const float* seg_pdot,
                                           const int32_t* valid_rank, const int32_t* n_valid, const float* weights, int64_t B,
                                           float* loss, float* dbase, int32_t* row_rank, float* group_loss, void* stream) {
    if (B < 0 || !loss) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(loss, 0, sizeof(float), st));
        return RECNOW_OK;
    }
    if (!labels || !logits || !order || !seg_id || !seg_first || !seg_valid || !seg_lse || !seg_ysum || !seg_psum || !seg_pdot ||
        !valid_rank || !n_valid || !dbase || !row_rank || !group_loss)
        return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_lw_grad, rn_cdiv(B, 256), 256, 0, st, labels, logits, order, seg_id, seg_first, seg_valid, seg_lse, seg_ysum,
                       seg_psum, seg_pdot, valid_rank, weights, B, dbase, row_rank, group_loss);
    hipLaunchKernelGGL(k_lw_mean, 1, 1024, 0, st, group_loss, n_valid, loss);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// Dense (Gv,B) outputs of to_listwise_sample.  Caller pre-fills mask_out = 0, labels_out = 0, logits_out = pad_logit.
extern "C" int recnow_listwise_dense(const float* labels, const float* logits, const int32_t* order, const int32_t* seg_id,
                                     const float* seg_ysum, const int32_t* valid_rank, int64_t B, uint8_t* mask_out,
                                     float* labels_out, float* logits_out, void* stream) {
    if (B < 0) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!labels || !logits || !order || !seg_id || !seg_ysum || !valid_rank || !mask_out || !labels_out || !logits_out) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_lw_dense, rn_cdiv(B, 256), 256, 0, (hipStream_t)stream, labels, logits, order, seg_id, seg_ysum, valid_rank, B,
                       mask_out, labels_out, logits_out);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_listwise_dense_bwd(const float* ddense, const int32_t* row_rank, int64_t B, float* dlogits, void* stream) {
    if (B < 0) return RECNOW_EINVAL;
    if (B == 0) return RECNOW_OK;
    if (!ddense || !row_rank || !dlogits) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_lw_dense_bwd, rn_cdiv(B, 256), 256, 0, (hipStream_t)stream, ddense, row_rank, B, dlogits);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// Row-wise softmax cross-entropy of dense (G,N) labels/logits.  row_loss, row_lse, row_psum: [G].
extern "C" int recnow_softmax_ce_rows_fwd(const float* labels, const float* logits, int64_t G, int64_t N, float* row_loss,
                                          float* row_lse, float* row_psum, void* stream) {
    if (G < 0 || N < 1) return RECNOW_EINVAL;
    if (G == 0) return RECNOW_OK;
    if (G > 0x7fffffffll) return RECNOW_EUNSUPPORTED;
    if (!labels || !logits || !row_loss || !row_lse || !row_psum) return RECNOW_EINVAL;
    hipLaunchKernelGGL(k_softmax_ce_rows_fwd, (unsigned)G, 256, 0, (hipStream_t)stream, labels, logits, N, row_loss, row_lse, row_psum);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
extern "C" int recnow_softmax_ce_rows_bwd(const float* labels, const float* logits, const float* row_lse, const float* row_psum,
                                          const float* grow, int64_t G, int64_t N, float* dlogits, void* stream) {
    if (G < 0 || N < 1) return RECNOW_EINVAL;
    if (G == 0) return RECNOW_OK;
    if (G > 65535) return RECNOW_EUNSUPPORTED;
    if (!labels || !logits || !row_lse || !row_psum || !grow || !dlogits) return RECNOW_EINVAL;
    int gx = rn_cdiv(N, 256 * 4);
    if (gx > 256) gx = 256;
    dim3 grid(gx, (unsigned)G);
    hipLaunchKernelGGL(k_softmax_ce_rows_bwd, grid, 256, 0, (hipStream_t)stream, labels, logits, row_lse, row_psum, grow, N, dlogits);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}


// ---- the fused listwise loss in one call (include/recnow.h: recnow_listwise_loss) ----------------------------------------------
struct LwLossWs {
    int32_t *order, *seg_id, *seg_first, *super_id, *n_seg, *seg_valid, *valid_rank, *n_valid, *row_rank;
    float *seg_lse, *seg_ysum, *seg_psum, *seg_pdot, *dbase, *group_loss, *loss;
    uint32_t* words;
    uint8_t* solo;
    void* grp; size_t grp_bytes;
    void* lw; size_t lw_bytes;
    int n_words;
    size_t total;
};
static LwLossWs lw_loss_carve(void* ws, int64_t B, int key_dtype) {
    LwLossWs w;
    RnCarver c(ws, 0);
    const int64_t n = B > 0 ? B : 1;
    w.n_words = recnow_key_words(key_dtype);
    if (w.n_words < 1) w.n_words = 1;
    w.order = c.take<int32_t>(n); w.seg_id = c.take<int32_t>(n); w.seg_first = c.take<int32_t>(n + 1); w.super_id = c.take<int32_t>(n);
    w.n_seg = c.take<int32_t>(2); w.seg_valid = c.take<int32_t>(n); w.valid_rank = c.take<int32_t>(n); w.n_valid = c.take<int32_t>(1);
    w.row_rank = c.take<int32_t>(n);
    w.seg_lse = c.take<float>(n); w.seg_ysum = c.take<float>(n); w.seg_psum = c.take<float>(n); w.seg_pdot = c.take<float>(n);
    w.dbase = c.take<float>(n); w.group_loss = c.take<float>(n); w.loss = c.take<float>(1);
    w.words = c.take<uint32_t>((size_t)w.n_words * n);
    w.solo = c.take<uint8_t>(n);
    w.grp_bytes = recnow_group_segments_workspace_bytes(B, w.n_words);
    w.grp = c.take<char>(w.grp_bytes);
    w.lw_bytes = recnow_listwise_workspace_bytes(B);
    w.lw = c.take<char>(w.lw_bytes);
    w.total = c.off;
    return w;
}
extern "C" size_t recnow_listwise_loss_workspace_bytes(int64_t B, int key_dtype) {
    if (B < 0) return 0;
    return lw_loss_carve(nullptr, B, key_dtype).total + 256;
}
// dlogits = dbase / (number of valid lists) (0 when there is none); {loss, (float) number of valid lists}
__global__ void __launch_bounds__(256)
k_lw_norm(const float* __restrict__ dbase, const int32_t* __restrict__ n_valid, int64_t B, float* __restrict__ out, const float* __restrict__ loss,
          float* __restrict__ out2, const int32_t* __restrict__ n_seg) {
    const int nv = n_valid[0];
    // n_seg[0] < 0: the cooperative grouping launch timed out (scan_sort.hip): NaN loss and gradient, never a silent zero
    const bool bad = n_seg[0] < 0;
    const float sc = bad ? __int_as_float(0x7fc00000) : (nv > 0 ? 1.f / (float)nv : 0.f);
    if (out2 && blockIdx.x == 0 && threadIdx.x == 0) { out2[0] = bad ? sc : loss[0]; out2[1] = (float)nv; }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) out[i] = dbase[i] * sc;
}
struct RnTileFwd;
int rn_group_mid_raw(const void* group, int dtype, int64_t B, uint8_t* solo, int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id,
                     int32_t* n_seg, void* ws, size_t ws_bytes, hipStream_t st, const RnTileFwd* pack, unsigned long long* zero1, int* zeroed, int* packed);       // scan_sort.hip
extern "C" int recnow_listwise_loss(const void* groups, int key_dtype, const float* labels, const float* logits, const float* weights, int64_t B,
                                    float pos_neg_th, float pad_logit, float* out2, float* dlogits, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || !out2) return RECNOW_EINVAL;
    if (recnow_key_words(key_dtype) < 1) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) {
        RN_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(float), st));
        return RECNOW_OK;
    }
    if (!groups || !labels || !logits || !dlogits || !ws) return RECNOW_EINVAL;
    if (ws_bytes < recnow_listwise_loss_workspace_bytes(B, key_dtype)) return RECNOW_EWORKSPACE;
    const LwLossWs w = lw_loss_carve(ws, B, key_dtype);
    int rc;
    // one float32 / int32 id tensor above the one-workgroup size: keys and solo flags are formed inside the cooperative grouping launch (scan_sort.hip)
    rc = rn_group_mid_raw(groups, key_dtype, B, w.solo, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, w.grp, w.grp_bytes, st, nullptr, nullptr, nullptr, nullptr);
    if (rc != RECNOW_EUNSUPPORTED) {
        if (rc) return rc;
    } else {
    RN_HIP(hipMemsetAsync(w.solo, 0, (size_t)B, st));
    if ((rc = recnow_group_keys(groups, key_dtype, B, w.words, w.solo, stream))) return rc;
    if ((rc = recnow_group_segments(w.words, w.solo, B, w.n_words, w.n_words, w.order, w.seg_id, w.seg_first, w.super_id, w.n_seg, w.grp, w.grp_bytes,
                                    stream)))
        return rc;
    }
    if ((rc = recnow_listwise_segments(labels, logits, w.order, w.seg_first, w.n_seg, B, pos_neg_th, pad_logit, w.seg_valid, w.seg_lse, w.seg_ysum,
                                       w.seg_psum, w.seg_pdot, w.valid_rank, w.n_valid, w.lw, w.lw_bytes, stream)))
        return rc;
    if ((rc = recnow_listwise_loss_fwdbwd(labels, logits, w.order, w.seg_id, w.seg_first, w.seg_valid, w.seg_lse, w.seg_ysum, w.seg_psum, w.seg_pdot,
                                          w.valid_rank, w.n_valid, weights, B, w.loss, w.dbase, w.row_rank, w.group_loss, stream)))
        return rc;
    int G = rn_cdiv(B, 256);
    if (G > 2048) G = 2048;
    hipLaunchKernelGGL(k_lw_norm, G, 256, 0, st, w.dbase, w.n_valid, B, dlogits, w.loss, out2, w.n_seg);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
