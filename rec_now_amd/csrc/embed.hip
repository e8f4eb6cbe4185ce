// Pooled embedding lookup by slot:
//   /root/reference/rec_now/rec_block/embedding_util.py:239-324  embedding_using_sparse_batch_segment_ids
//   /root/reference/rec_now/rec_block/embedding_util.py:138-195  sparse_batch_segment_ids_of_targets (slot -> target index)
// The reference masks the (B,C) id matrix down to the entries whose slot is a target slot, runs tf.unique over them, looks
// the unique ids up, gathers back and pools with unsorted_segment_sum/mean into (B,T,D).  Here:
//   k_slot_targets   : slot -> target index (-1 = not pooled) and the sort key (id, or all-ones for unpooled entries)
//   k_embed_pool_fwd : one wave per batch row walks its C entries; 64/D lane groups take entries round-robin and
//                      accumulate into private (T,D) LDS tiles that are summed in a fixed order at the end - no atomics
//   k_embed_unique   : after the radix sort of the keys (scan_sort.hip): unique ids in sorted order + inverse index
//   k_embed_rows_*   : per-id sums of w * dout[b][t][:] over that id's entries, split by entry chunks (hot ids), fixed order
//   k_embed_scatter  : unique gradient rows -> dense (V,D) table gradient (each row written by exactly one wave)
// All of it is HBM/latency-bound integer + gather work; nothing here wants the MFMA pipe.
#include "common.hpp"

#define EMB_SENTINEL (-1ll)        // key of entries that are not pooled: all-ones sorts last in the unsigned radix order

template <typename ST>
__global__ void __launch_bounds__(256)
k_slot_targets(const ST* __restrict__ slots, const ST* __restrict__ targets, int T, const int64_t* __restrict__ ids, int64_t N,
               int32_t* __restrict__ seg, int64_t* __restrict__ key) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const ST s = slots[i];
        int t = -1;
        for (int j = 0; j < T; ++j)
            if (targets[j] == s) { t = j; break; }
        seg[i] = t;
        if (key) key[i] = t >= 0 ? ids[i] : EMB_SENTINEL;
    }
}

extern "C" int recnow_slot_targets(const void* slots, int slot_dtype, const void* targets, int T, const int64_t* ids, int64_t N,
                                   int32_t* seg, int64_t* key, void* stream) {
    if (N < 0 || T < 0 || (slot_dtype != RECNOW_KEY_I32 && slot_dtype != RECNOW_KEY_I64)) return RECNOW_EINVAL;
    if (N == 0) return RECNOW_OK;
    if (!slots || (T > 0 && !targets) || !seg || (key && !ids)) return RECNOW_EINVAL;
    int64_t g = (N + 255) / 256;
    if (g > 4096) g = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (slot_dtype == RECNOW_KEY_I32)
        hipLaunchKernelGGL(k_slot_targets<int32_t>, (int)g, 256, 0, st, (const int32_t*)slots, (const int32_t*)targets, T, ids, N, seg, key);
    else
        hipLaunchKernelGGL(k_slot_targets<int64_t>, (int)g, 256, 0, st, (const int64_t*)slots, (const int64_t*)targets, T, ids, N, seg, key);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// out[b][t][:] = sum_{c: seg[b][c] == t} w[b][c] * table[rows[b][c]][:]   (/ cnt[b][t] for 'mean'; empty segments stay 0)
__global__ void __launch_bounds__(256)
k_embed_pool_fwd(const float* __restrict__ table, int D, const int64_t* __restrict__ rows, const int32_t* __restrict__ seg,
                 const float* __restrict__ weights, int64_t B, int C, int T, int mean, float* __restrict__ out,
                 float* __restrict__ cnt_out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int GS = 1;                                    // lanes per entry: D rounded up to a power of two, at most 64
    while (GS < D && GS < 64) GS <<= 1;
    const int G = 64 / GS, grp = lane / GS, gl = lane % GS;
    float* acc = lds + (size_t)w * (G * T * D + T);          // [G][T][D] then cnt[T]
    float* cnt = acc + G * T * D;
    for (int64_t b = (int64_t)blockIdx.x * nw + w; b < B; b += (int64_t)gridDim.x * nw) {
        for (int i = lane; i < G * T * D + T; i += 64) acc[i] = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int c = grp; c < C; c += G) {
            const int t = seg[b * C + c];
            if (t < 0) continue;
            const float wt = weights ? weights[b * C + c] : 1.f;
            const float* row = table + rows[b * C + c] * (int64_t)D;
            float* a = acc + (grp * T + t) * D;
            for (int d = gl; d < D; d += GS) a[d] += wt * row[d];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mean || cnt_out) {                              // entry counts per target: lane t walks the row (C is small)
            for (int t = lane; t < T; t += 64) {
                int n = 0;
                for (int c = 0; c < C; ++c) n += seg[b * C + c] == t;
                cnt[t] = (float)n;
                if (cnt_out) cnt_out[b * T + t] = (float)n;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int i = lane; i < T * D; i += 64) {
            float s = 0.f;
            for (int g = 0; g < G; ++g) s += acc[g * T * D + i];
            if (mean) {
                const float n = cnt[i / D];
                s = n > 0.f ? s / n : 0.f;
            }
            out[b * T * D + i] = s;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

static int pool_cfg(int T, int D, int* waves, size_t* lds) {
    int GS = 1;
    while (GS < D && GS < 64) GS <<= 1;
    const size_t per_wave = ((size_t)(64 / GS) * T * D + T) * sizeof(float);
    int w = 4;
    while (w > 1 && per_wave * w > 64 * 1024) w >>= 1;
    if (per_wave * w > 64 * 1024) return RECNOW_EUNSUPPORTED;
    *waves = w;
    *lds = per_wave * w;
    return RECNOW_OK;
}

extern "C" int recnow_embed_pool_fwd(const float* table, int D, const int64_t* rows, const int32_t* seg, const float* weights,
                                     int64_t B, int C, int T, int mean, float* out, float* cnt, void* stream) {
    if (B < 0 || C < 0 || T < 0 || D < 1) return RECNOW_EINVAL;
    if (B == 0 || T == 0) return RECNOW_OK;
    if (!out || (C > 0 && (!table || !rows || !seg))) return RECNOW_EINVAL;
    int waves;
    size_t lds;
    int rc = pool_cfg(T, D, &waves, &lds);
    if (rc) return rc;
    int64_t g = (B + waves - 1) / waves;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_embed_pool_fwd, (int)g, waves * 64, lds, (hipStream_t)stream, table, D, rows, seg, weights, B, C, T, mean, out, cnt);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// After the stable radix sort of the keys: unique[s] = key of segment s, inverse[entry] = its segment, n_unique = number of
// segments that are not the sentinel segment (which, if present, is the last one).
__global__ void __launch_bounds__(256)
k_embed_unique(const int64_t* __restrict__ key, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_id,
               const int32_t* __restrict__ seg_first, const int32_t* __restrict__ n_seg, int64_t N, int64_t* __restrict__ unique,
               int64_t* __restrict__ inverse, int32_t* __restrict__ n_unique) {
    const int ns = n_seg[0];
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < N; k += (int64_t)gridDim.x * 256) {
        const int s = seg_id[k];
        inverse[order[k]] = s;
        if (k == seg_first[s]) unique[s] = key[order[k]];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int nu = ns;
        if (ns > 0 && key[order[seg_first[ns - 1]]] == EMB_SENTINEL) nu = ns - 1;
        *n_unique = nu;
    }
}
extern "C" int recnow_embed_unique(const int64_t* key, const int32_t* order, const int32_t* seg_id, const int32_t* seg_first,
                                   const int32_t* n_seg, int64_t N, int64_t* unique, int64_t* inverse, int32_t* n_unique,
                                   void* stream) {
    if (N < 0) return RECNOW_EINVAL;
    if (!n_unique) return RECNOW_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        RN_HIP(hipMemsetAsync(n_unique, 0, sizeof(int32_t), st));
        return RECNOW_OK;
    }
    if (!key || !order || !seg_id || !seg_first || !n_seg || !unique || !inverse) return RECNOW_EINVAL;
    int64_t g = (N + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_embed_unique, (int)g, 256, 0, st, key, order, seg_id, seg_first, n_seg, N, unique, inverse, n_unique);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// drows[s][:] = sum over the entries e of sorted segment s of  w_e * dout[b_e][t_e][:] (/ cnt[b_e][t_e] for 'mean').
// Ids are heavy-tailed (one hot id can own a third of the batch), so the work is split by ENTRIES, not by ids:
//   pass 1: one wave per chunk of EMB_CH consecutive sorted entries walks them in order and closes a run whenever the id
//           changes.  A run that is a whole segment goes straight to drows[s]; the (at most two) runs cut by the chunk
//           boundary go to part[chunk][0] (run touching the chunk start) / part[chunk][1] (run touching the chunk end).
//   pass 2: one workgroup per segment that crosses a chunk boundary adds its pieces in ascending chunk order: four waves
//           take the four quarters of the chunk range, the quarters are combined in wave order.  Fixed order => deterministic.
#define EMB_CH 128
__global__ void __launch_bounds__(256)
k_embed_rows_chunks(const int64_t* __restrict__ key, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_id,
                    const int32_t* __restrict__ seg_first, const int32_t* __restrict__ seg, const float* __restrict__ weights,
                    const float* __restrict__ cnt, const float* __restrict__ dout, int64_t N, int C, int T, int D, int mean,
                    float* __restrict__ drows, float* __restrict__ part, int64_t* __restrict__ row_ids) {
    const int lane = threadIdx.x & 63;
    const int64_t nchunk = (N + EMB_CH - 1) / EMB_CH;
    for (int64_t ch = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); ch < nchunk; ch += (int64_t)gridDim.x * 4) {
        const int64_t k0 = ch * EMB_CH, k1 = min(N, k0 + EMB_CH);
        for (int d0 = 0; d0 < D; d0 += 64) {
            const int d = d0 + lane;
            float a = 0.f;
            int64_t run0 = k0;
            int s = seg_id[k0];
            for (int64_t k = k0; k < k1; ++k) {
                const int e = order[k];
                const int t = seg[e];
                if (t >= 0) {                              // entries of the unpooled (-1) segment carry no gradient
                    const int64_t b = e / C;
                    float wt = weights ? weights[e] : 1.f;
                    if (mean) wt /= cnt[b * T + t];
                    if (d < D) a += wt * dout[(b * T + t) * (int64_t)D + d];
                }
                const bool last = k + 1 == k1;
                const int sn = last ? -1 : seg_id[k + 1];
                if (last || sn != s) {                     // close the run [run0, k]
                    const bool whole = run0 == seg_first[s] && k + 1 == seg_first[s + 1];
                    if (whole) {
                        if (d < D) drows[s * (int64_t)D + d] = a;
                        if (d0 == 0 && lane == 0 && row_ids) row_ids[s] = key[order[run0]];
                    } else {
                        // a cut run touches the chunk start, the chunk end, or both (then it is the whole chunk: slot 0)
                        const int slot = run0 == k0 ? 0 : 1;
                        if (d < D) part[(ch * 2 + slot) * (int64_t)D + d] = a;
                    }
                    a = 0.f;
                    run0 = k + 1;
                    s = sn;
                }
            }
        }
    }
}
__global__ void __launch_bounds__(256)
k_embed_rows_join(const int64_t* __restrict__ key, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_id,
                  const int32_t* __restrict__ seg_first, int64_t N, int D, const float* __restrict__ part, float* __restrict__ drows,
                  int64_t* __restrict__ row_ids) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t nchunk = (N + EMB_CH - 1) / EMB_CH;
    for (int64_t c = 1 + blockIdx.x; c < nchunk; c += gridDim.x) {         // boundary between chunks c-1 and c (block-uniform)
        const int s = seg_id[c * EMB_CH];
        if (seg_id[c * EMB_CH - 1] != s) continue;                          // no segment crosses this boundary
        const int64_t f = seg_first[s], l = seg_first[s + 1] - 1;           // first / last sorted position of the segment
        if (f / EMB_CH != c - 1) continue;                                  // another boundary block owns this segment
        const int64_t c0 = c - 1, c1 = l / EMB_CH;                          // chunks c0 .. c1 hold pieces of s
        // piece of chunk x: slot 1 in c0 unless the segment starts exactly at the chunk start, slot 0 in every later chunk
        const int64_t npiece = c1 - c0 + 1, per = (npiece + 3) / 4;
        for (int d0 = 0; d0 < D; d0 += 64) {
            const int d = d0 + lane;
            float a = 0.f;
            const int64_t x0 = c0 + w * per, x1 = min(c1 + 1, x0 + per);
            for (int64_t x = x0; x < x1; ++x) {
                const int slot = (x == c0 && f != c0 * EMB_CH) ? 1 : 0;
                if (d < D) a += part[(x * 2 + slot) * (int64_t)D + d];
            }
            __syncthreads();
            red[w][lane] = a;
            __syncthreads();
            if (w == 0 && d < D) drows[s * (int64_t)D + d] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
        }
        if (threadIdx.x == 0 && row_ids) row_ids[s] = key[order[f]];
    }
}
__global__ void k_embed_rows_tail(const int32_t* __restrict__ n_seg, int64_t N, int64_t* __restrict__ row_ids) {
    const int ns = n_seg[0];
    for (int64_t s = ns + (int64_t)blockIdx.x * 256 + threadIdx.x; s < N; s += (int64_t)gridDim.x * 256) row_ids[s] = EMB_SENTINEL;
}
extern "C" size_t recnow_embed_rows_bwd_workspace_bytes(int64_t N, int D) {
    const int64_t nchunk = (N + EMB_CH - 1) / EMB_CH;
    return rn_align((size_t)(nchunk > 0 ? nchunk : 1) * 2 * (size_t)D * sizeof(float));
}
extern "C" int recnow_embed_rows_bwd(const int64_t* key, const int32_t* order, const int32_t* seg_id, const int32_t* seg_first,
                                     const int32_t* n_seg, const int32_t* seg, const float* weights, const float* cnt, const float* dout,
                                     int64_t N, int C, int T, int D, int mean, float* drows, int64_t* row_ids, void* ws, size_t ws_bytes,
                                     void* stream) {
    if (N < 0 || C < 1 || T < 1 || D < 1) return RECNOW_EINVAL;
    if (N == 0) return RECNOW_OK;
    if (!key || !order || !seg_id || !seg_first || !n_seg || !seg || !dout || !drows || (mean && !cnt)) return RECNOW_EINVAL;
    if (!ws || ws_bytes < recnow_embed_rows_bwd_workspace_bytes(N, D)) return RECNOW_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nchunk = (N + EMB_CH - 1) / EMB_CH;
    int64_t g = (nchunk + 3) / 4;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_embed_rows_chunks, (int)g, 256, 0, st, key, order, seg_id, seg_first, seg, weights, cnt, dout, N, C, T, D, mean,
                       drows, (float*)ws, row_ids);
    if (nchunk > 1) {
        int64_t gj = nchunk - 1;
        if (gj > 16384) gj = 16384;
        hipLaunchKernelGGL(k_embed_rows_join, (int)gj, 256, 0, st, key, order, seg_id, seg_first, N, D, (const float*)ws, drows, row_ids);
    }
    if (row_ids) hipLaunchKernelGGL(k_embed_rows_tail, 64, 256, 0, st, n_seg, N, row_ids);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// dtable[row_ids[s]][:] = drows[s][:] for every used slot s (row ids are unique: one writer per table row)
__global__ void __launch_bounds__(256)
k_embed_scatter(const float* __restrict__ drows, const int64_t* __restrict__ row_ids, int64_t n_slots, int D, int64_t V,
                float* __restrict__ dtable) {
    const int lane = threadIdx.x & 63;
    for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < n_slots; s += (int64_t)gridDim.x * 4) {
        const int64_t id = row_ids[s];
        if (id < 0 || id >= V) continue;
        for (int d = lane; d < D; d += 64) dtable[id * D + d] = drows[s * (int64_t)D + d];
    }
}
extern "C" int recnow_embed_scatter_rows(const float* drows, const int64_t* row_ids, int64_t n_slots, int D, int64_t V, float* dtable,
                                         void* stream) {
    if (n_slots < 0 || D < 1 || V < 0) return RECNOW_EINVAL;
    if (n_slots == 0 || V == 0) return RECNOW_OK;
    if (!drows || !row_ids || !dtable) return RECNOW_EINVAL;
    int64_t g = (n_slots + 3) / 4;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_embed_scatter, (int)g, 256, 0, (hipStream_t)stream, drows, row_ids, n_slots, D, V, dtable);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
