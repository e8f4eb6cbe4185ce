// Pooled embedding lookup by slot:
//   /root/reference/rec_now/rec_block/embedding_util.py:239-324  embedding_using_sparse_batch_segment_ids
//   /root/reference/rec_now/rec_block/embedding_util.py:138-195  sparse_batch_segment_ids_of_targets (slot -> target index)
// The reference masks the (B,C) id matrix down to the entries whose slot is a target slot, runs tf.unique over them, looks
// the unique ids up, gathers back and pools with unsorted_segment_sum/mean into (B,T,D).  Here:
//   k_slot_targets   : slot -> target index (-1 = not pooled) and the sort key (id, or INT64_MIN for unpooled entries)
//   k_embed_pool_fwd : one wave per batch row; its C entries are described by all lanes at once, then each of the 64/D lane
//                      groups takes the entries of ITS targets (t % groups) in ascending order and adds their table rows
//                      into one (T,D) LDS tile - no float atomics on data (entry counts are whole numbers)
//   k_embed_unique   : after the radix sort of the keys (scan_sort.hip): unique ids in sorted order + inverse index
//   k_embed_rows_*   : per-id sums of w * dout[b][t][:] over that id's entries, split by entry chunks (hot ids), fixed order
//   k_embed_scatter  : unique gradient rows -> dense (V,D) table gradient (each row written by exactly one wave)
// All of it is HBM/latency-bound integer + gather work; nothing here wants the MFMA pipe.
#include "common.hpp"

// key of entries that are not pooled: INT64_MIN.  As an unsigned radix key it sorts after every id >= 0, and only its top
// byte differs from the zero high bytes of real ids, so it costs ONE extra 8-bit pass (all-ones made every pass non-trivial:
// 8 passes instead of 3 for 20-bit ids).
#define EMB_SENTINEL ((int64_t)0x8000000000000000ull)

// The target list lives in LDS (padded to a multiple of 4 with copies of its last entry) and is scanned from the back
// without an early exit, so the smallest matching index wins as in a forward search; reading targets[j] from memory inside
// a search loop with a break was a chain of up to T scalar-load latencies per entry (0.26 ms for 6.5 M entries, T = 64).
#define SLOT_PIECE 4096
template <typename ST>
__global__ void __launch_bounds__(256)
k_slot_targets(const ST* __restrict__ slots, const ST* __restrict__ targets, int T, const int64_t* __restrict__ ids, int64_t N,
               int32_t* __restrict__ seg, int64_t* __restrict__ key, int64_t not_pooled_key, int64_t id_limit, int32_t* __restrict__ key32) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tg_raw[];
    ST* tg = reinterpret_cast<ST*>(tg_raw);
    // target lists longer than SLOT_PIECE go through LDS piece by piece (front to back; an earlier piece's match stands)
    for (int p0 = 0; p0 < max(T, 1); p0 += SLOT_PIECE) {
        const int tp = min(SLOT_PIECE, T - p0), T4 = (tp + 3) & ~3;
        __syncthreads();
        for (int j = threadIdx.x; j < T4; j += 256) tg[j] = targets[p0 + min(j, tp - 1)];
        __syncthreads();
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
            const ST s = slots[i];
            int t = -1;
            for (int j = T4 - 4; j >= 0; j -= 4) {
                t = tg[j + 3] == s ? min(j + 3, tp - 1) : t;          // a padded copy reports the entry it copies
                t = tg[j + 2] == s ? min(j + 2, tp - 1) : t;
                t = tg[j + 1] == s ? min(j + 1, tp - 1) : t;
                t = tg[j] == s ? j : t;
            }
            t = t >= 0 ? p0 + t : -1;
            if (p0 > 0) {
                const int before = seg[i];
                t = before >= 0 ? before : t;
            }
            seg[i] = t;
            if (key && p0 + SLOT_PIECE >= T) {
                // id_limit > 0 (round 5, a table of V = id_limit rows): entries that are not pooled AND ids outside the table share the key
                // not_pooled_key = V -- one past the last row, so it sorts last and is dropped by the scatter like every id outside the table --
                // which keeps every key below 2^31: the sort runs on the 32-bit copy (three 8-bit digit passes for V = 2^20, where INT64_MIN in
                // a 64-bit key cost a fourth pass on a second key word, and the second word four empty pass iterations)
                const int64_t id = ids[i];
                const bool in = t >= 0 && (id_limit <= 0 || (id >= 0 && id < id_limit));
                const int64_t k = in ? id : not_pooled_key;
                key[i] = k;
                if (key32) key32[i] = (int32_t)k;
            }
        }
    }
}

extern "C" int recnow_slot_targets(const void* slots, int slot_dtype, const void* targets, int T, const int64_t* ids, int64_t N,
                                   int32_t* seg, int64_t* key, int64_t id_limit, int32_t* key32, void* stream) {
    if (N < 0 || T < 0 || (slot_dtype != RECNOW_KEY_I32 && slot_dtype != RECNOW_KEY_I64)) return RECNOW_EINVAL;
    if (N == 0) return RECNOW_OK;
    if (!slots || (T > 0 && !targets) || !seg || (key && !ids)) return RECNOW_EINVAL;
    if (id_limit < 0 || (key32 && (!key || id_limit <= 0 || id_limit >= 0x7fffffffll))) return RECNOW_EINVAL;
    const int64_t not_pooled_key = id_limit > 0 ? id_limit : EMB_SENTINEL;
    int64_t g = (N + 255) / 256;
    if (g > 4096) g = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (slot_dtype == RECNOW_KEY_I32)
        hipLaunchKernelGGL(k_slot_targets<int32_t>, (int)g, 256, (size_t)((min(T, SLOT_PIECE) + 3) & ~3) * sizeof(int32_t) + 16, st, (const int32_t*)slots, (const int32_t*)targets, T, ids, N, seg, key, not_pooled_key, id_limit, key32);
    else
        hipLaunchKernelGGL(k_slot_targets<int64_t>, (int)g, 256, (size_t)((min(T, SLOT_PIECE) + 3) & ~3) * sizeof(int64_t) + 16, st, (const int64_t*)slots, (const int64_t*)targets, T, ids, N, seg, key, not_pooled_key, id_limit, key32);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// out[b][t][:] = sum_{c: seg[b][c] == t} w[b][c] * table[rows[b][c]][:]   (/ cnt[b][t] for 'mean'; empty segments stay 0)
__global__ void __launch_bounds__(256)
k_embed_pool_fwd(const float* __restrict__ table, int D, int64_t V, const int64_t* __restrict__ rows, const int32_t* __restrict__ seg,
                 const float* __restrict__ weights, int64_t B, int C, int T, int mean, float* __restrict__ out,
                 float* __restrict__ cnt_out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int GS = 1;                                    // lanes per entry: D rounded up to a power of two, at most 64
    while (GS < D && GS < 64) GS <<= 1;
    const int G = 64 / GS, grp = lane / GS, gl = lane % GS;
    float* acc = lds + (size_t)w * (T * D + T);              // [T][D] then cnt[T]
    float* cnt = acc + T * D;
    for (int64_t b = (int64_t)blockIdx.x * nw + w; b < B; b += (int64_t)gridDim.x * nw) {
        for (int i = lane; i < T * D + T; i += 64) acc[i] = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // The row's entries, 64 at a time: lane l fetches (target, table row, weight) of entry c0 + l -- coalesced, all in
        // flight at once.  Lane group g then owns the targets t with t % G == g: it takes its entries in ascending order
        // from a ballot mask, four table rows in flight, and adds them into the ONE [T][D] tile -- groups never meet on a
        // target, so there are no per-group copies to clear and add up (those 16 KB per wave held the kernel to two waves
        // per SIMD) and no atomics.  (Before that: seg -> rows -> table row fetched per entry, three dependent latencies.)
        for (int c0 = 0; c0 < C; c0 += 64) {
            const int cl = c0 + lane;
            int m_t = -1;
            int64_t m_row = 0;
            float m_w = 0.f;
            if (cl < C) {
                m_t = seg[b * C + cl];
                if (m_t >= 0) {
                    m_row = rows[b * C + cl];
                    m_w = weights ? weights[b * C + cl] : 1.f;
                    atomicAdd(&cnt[m_t], 1.f);                               // whole numbers: exact in any order
                    // a row index outside the table contributes a zero row (what tf.nn.embedding_lookup returns on a GPU, and
                    // what the backward's scatter does with such an id); it still counts as a pooled entry for 'mean'
                    if (m_row < 0 || m_row >= V) m_row = -1;
                }
            }
            unsigned long long mine = 0ull;                                  // entries of this round that belong to this lane's group
            for (int g = 0; g < G; ++g) {
                const unsigned long long bal = __ballot(m_t >= 0 && (m_t % G) == g);
                mine = g == grp ? bal : mine;
            }
            while (__ballot(mine != 0ull)) {                                 // the whole wave stays in: the shuffles below read lanes of other groups
                float v[4], wv[4];
                int tv[4];
                int64_t rv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool have = mine != 0ull;
                    const int j = have ? __ffsll((long long)mine) - 1 : 0;
                    const int tj = __shfl(m_t, j, 64);                       // shuffles by every lane, selects afterwards
                    rv[u] = __shfl(m_row, j, 64);
                    wv[u] = __shfl(m_w, j, 64);
                    tv[u] = have ? tj : -1;
                    mine &= mine - 1ull;                                     // 0 stays 0
                }
                for (int d = gl; d < D; d += GS) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = (tv[u] >= 0 && rv[u] >= 0) ? table[rv[u] * (int64_t)D + d] : 0.f;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (tv[u] >= 0) acc[tv[u] * D + d] += wv[u] * v[u];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (cnt_out)
            for (int t = lane; t < T; t += 64) cnt_out[b * T + t] = cnt[t];
        for (int i = lane; i < T * D; i += 64) {
            float s = acc[i];
            if (mean) {
                const float n = cnt[i / D];
                s = n > 0.f ? s / n : 0.f;
            }
            out[b * T * D + i] = s;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// The same with 16-byte gathers (round 4): D / 4 lanes per entry, each lane one float4 of the table row, so a wave has 64 / (D / 4) lane groups -- 16 at
// D = 16 -- with EMB_V4_U table rows in flight each: 64 rows of 64 B in flight per wave instead of 16.  The gather of a pooled id is a 64-byte request
// from a table far larger than the L2; what bounds the kernel is how many such requests are outstanding, not the bytes.  Same arithmetic in the same order
// per output element (a target's entries are still taken in ascending order by ONE lane group).  D % 4 == 0, D / 4 a power of two <= 16, table 16-byte aligned.
typedef float emb_f4 __attribute__((ext_vector_type(4)));
#ifndef EMB_V4_U
#define EMB_V4_U 4          // table rows in flight per lane group (8 measured 0.213 against 0.201 ms at B 65 536, C 100)
#endif
__global__ void __launch_bounds__(256)
k_embed_pool_fwd_v4(const float* __restrict__ table, int D, int64_t V, const int64_t* __restrict__ rows, const int32_t* __restrict__ seg,
                    const float* __restrict__ weights, int64_t B, int C, int T, int mean, float* __restrict__ out, float* __restrict__ cnt_out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int GS = D / 4;                          // lanes per entry (power of two)
    const int G = 64 / GS, grp = lane / GS, gl = lane % GS;
    float* acc = lds + (size_t)w * (T * D + T);              // [T][D] then cnt[T]
    float* cnt = acc + T * D;
    for (int64_t b = (int64_t)blockIdx.x * nw + w; b < B; b += (int64_t)gridDim.x * nw) {
        for (int i = lane; i < T * D + T; i += 64) acc[i] = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int c0 = 0; c0 < C; c0 += 64) {
            const int cl = c0 + lane;
            int m_t = -1;
            int64_t m_row = 0;
            float m_w = 0.f;
            if (cl < C) {
                m_t = seg[b * C + cl];
                if (m_t >= 0) {
                    m_row = rows[b * C + cl];
                    m_w = weights ? weights[b * C + cl] : 1.f;
                    atomicAdd(&cnt[m_t], 1.f);                               // whole numbers: exact in any order
                    if (m_row < 0 || m_row >= V) m_row = -1;                 // outside the table: a zero row (as k_embed_pool_fwd)
                }
            }
            unsigned long long mine = 0ull;                                  // entries of this round that belong to this lane's group
            for (int g = 0; g < G; ++g) {
                const unsigned long long bal = __ballot(m_t >= 0 && (m_t % G) == g);
                mine = g == grp ? bal : mine;
            }
            while (__ballot(mine != 0ull)) {                                 // the whole wave stays in: the shuffles below read lanes of other groups
                emb_f4 v[EMB_V4_U];
                float wv[EMB_V4_U];
                int tv[EMB_V4_U];
                int64_t rv[EMB_V4_U];
#pragma unroll
                for (int u = 0; u < EMB_V4_U; ++u) {
                    const bool have = mine != 0ull;
                    const int j = have ? __ffsll((long long)mine) - 1 : 0;
                    const int tj = __shfl(m_t, j, 64);
                    rv[u] = __shfl(m_row, j, 64);
                    wv[u] = __shfl(m_w, j, 64);
                    tv[u] = have ? tj : -1;
                    mine &= mine - 1ull;
                }
#pragma unroll
                for (int u = 0; u < EMB_V4_U; ++u)
                    v[u] = (tv[u] >= 0 && rv[u] >= 0) ? *reinterpret_cast<const emb_f4*>(table + rv[u] * (int64_t)D + 4 * gl) : emb_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < EMB_V4_U; ++u)
                    if (tv[u] >= 0) {
                        emb_f4* a = reinterpret_cast<emb_f4*>(acc + tv[u] * D + 4 * gl);
                        *a = *a + v[u] * wv[u];
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (cnt_out)
            for (int t = lane; t < T; t += 64) cnt_out[b * T + t] = cnt[t];
        for (int i = lane; i < T * GS; i += 64) {                            // float4 per lane: whole 1 KiB pieces of the output row
            emb_f4 sv = *reinterpret_cast<const emb_f4*>(acc + 4 * i);
            if (mean) {
                const float n = cnt[(4 * i) / D];
                sv = n > 0.f ? sv / n : emb_f4{0.f, 0.f, 0.f, 0.f};
            }
            *reinterpret_cast<emb_f4*>(out + b * T * D + 4 * i) = sv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

static int pool_cfg(int T, int D, int* waves, size_t* lds) {
    int GS = 1;
    while (GS < D && GS < 64) GS <<= 1;
    const size_t per_wave = ((size_t)T * D + T) * sizeof(float);
    int w = 4;
    while (w > 1 && per_wave * w > 64 * 1024) w >>= 1;
    if (per_wave * w > 64 * 1024) return RECNOW_EUNSUPPORTED;
    *waves = w;
    *lds = per_wave * w;
    return RECNOW_OK;
}

extern "C" int recnow_embed_pool_fwd(const float* table, int D, int64_t V, const int64_t* rows, const int32_t* seg, const float* weights,
                                     int64_t B, int C, int T, int mean, float* out, float* cnt, void* stream) {
    if (B < 0 || C < 0 || T < 0 || D < 1 || V < 0) return RECNOW_EINVAL;
    if (B == 0 || T == 0) return RECNOW_OK;
    if (!out || (C > 0 && (!table || !rows || !seg))) return RECNOW_EINVAL;
    int waves;
    size_t lds;
    int rc = pool_cfg(T, D, &waves, &lds);
    if (rc) return rc;
    int64_t g = (B + waves - 1) / waves;
    if (g > 4096) g = 4096;
    static const bool v4_on = []() { const char* e = getenv("RECNOW_EMBED_V4"); return !e || e[0] != '0'; }();      // A/B switch
    const int gs4 = D / 4;
    // the accumulator tile of a wave starts at w * (T*D + T) floats: 16-byte aligned for every wave iff T is a multiple of 4
    const bool v4 = v4_on && D % 4 == 0 && gs4 >= 1 && gs4 <= 16 && (gs4 & (gs4 - 1)) == 0 && T % 4 == 0 && (((uintptr_t)table | (uintptr_t)out) & 15) == 0;
    if (v4) hipLaunchKernelGGL(k_embed_pool_fwd_v4, (int)g, waves * 64, lds, (hipStream_t)stream, table, D, V, rows, seg, weights, B, C, T, mean, out, cnt);
    else hipLaunchKernelGGL(k_embed_pool_fwd, (int)g, waves * 64, lds, (hipStream_t)stream, table, D, V, rows, seg, weights, B, C, T, mean, out, cnt);
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
//   pass 1: a group of LPE lanes (LPE = pow2 >= D, 16..64; 64/LPE groups per wave) walks one chunk of EMB_CH consecutive
//           sorted entries in order and closes a run whenever the id changes.  A run that is a whole segment goes
//           straight to drows[s]; the (at most two) runs cut by the chunk boundary go to part[chunk][0] (run touching the
//           chunk start) / part[chunk][1] (run touching the chunk end).  Everything about an entry except its gradient
//           row is known before the walk -- a run starts at max(chunk start, segment start), so "whole segment" and the
//           slot follow from seg_first alone -- and is fetched by the group's lanes in parallel, one or two entries per
//           lane, then handed around with shuffles; the walk itself only gathers dout rows, four in flight.  (Walking
//           with the whole wave and fetching order -> seg -> cnt -> dout per entry was a chain of four dependent
//           latencies per entry with a quarter of the lanes active at D = 16: 1.2 ms for 6.5 M entries.)
//   pass 2: one workgroup per segment that crosses a chunk boundary adds its pieces: 256/DL piece lanes x DL dims, four
//           loads in flight each, combined through LDS in lane order.  Fixed order => deterministic.
#define EMB_CH 32
#define EMB_JU 16
#define EMB_JSHORT 16      // segments spread over fewer chunks than this are joined by one lane group
template <int LPE>
__global__ void __launch_bounds__(256)
k_embed_rows_chunks(const int64_t* __restrict__ key, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_id,
                    const int32_t* __restrict__ seg_first, const int32_t* __restrict__ seg, const float* __restrict__ weights,
                    const float* __restrict__ cnt, const float* __restrict__ dout, int64_t N, int C, int T, int D, int mean,
                    float* __restrict__ drows, float* __restrict__ part, int64_t* __restrict__ row_ids) {
    constexpr int G = 64 / LPE;                                   // chunks walked side by side in one wave
    constexpr int NQ = (EMB_CH + LPE - 1) / LPE;                  // entries described per lane
    const int lane = threadIdx.x & 63, g = lane / LPE, dl = lane % LPE;
    const int64_t nchunk = (N + EMB_CH - 1) / EMB_CH;
    for (int64_t cb = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G; cb < nchunk; cb += (int64_t)gridDim.x * 4 * G) {
        const int64_t ch = cb + g;
        const int64_t k0 = ch * EMB_CH, k1 = ch < nchunk ? min(N, k0 + EMB_CH) : k0;
        const int n = (int)(k1 - k0);                             // group-uniform
        // ---- what this lane knows about entries q*LPE + dl of the chunk
        int64_t m_off[NQ], m_key[NQ];                             // first float of the gradient row (-1: none), id
        float m_wt[NQ];
        int m_sid[NQ], m_flag[NQ];                                // segment; bit0 closes a run, bit1 whole segment, bit2 slot
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = q * LPE + dl;
            m_off[q] = -1; m_key[q] = EMB_SENTINEL; m_wt[q] = 0.f; m_sid[q] = 0; m_flag[q] = 0;
            if (i < n) {
                const int64_t k = k0 + i;
                const int e = order[k];
                const int t = seg[e];
                const int sid = seg_id[k];
                if (t >= 0) {                                     // entries of the unpooled (-1) segment carry no gradient
                    const int64_t b = e / C;
                    float wt = weights ? weights[e] : 1.f;
                    if (mean) wt /= cnt[b * T + t];
                    m_wt[q] = wt;
                    m_off[q] = (b * T + t) * (int64_t)D;
                }
                const bool closes = k + 1 == k1 || seg_id[k + 1] != sid;
                const int64_t first = seg_first[sid], end = seg_first[sid + 1];
                const bool whole = first >= k0 && end <= k1;      // the run starts at max(k0, first)
                m_sid[q] = sid;
                m_flag[q] = (closes ? 1 : 0) | (whole ? 2 : 0) | (first > k0 ? 4 : 0);
                if (row_ids && closes && whole) m_key[q] = key[e];          // all entries of a segment carry its id
            }
        }
        for (int d0 = 0; d0 < D; d0 += LPE) {                     // more than one slice only for D > 64
            const int d = d0 + dl;
            float a = 0.f;
            for (int i0 = 0; i0 < n; i0 += 4) {
                float v[4], wt[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u, q = NQ > 1 && i >= LPE ? 1 : 0, srcl = i % LPE;
                    const int64_t off = __shfl(q ? m_off[NQ - 1] : m_off[0], srcl, LPE);
                    wt[u] = __shfl(q ? m_wt[NQ - 1] : m_wt[0], srcl, LPE);
                    v[u] = (i < n && off >= 0 && d < D) ? dout[off + d] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u;
                    if (i >= n) break;
                    const int q = NQ > 1 && i >= LPE ? 1 : 0, srcl = i % LPE;
                    a += wt[u] * v[u];
                    const int flag = __shfl(q ? m_flag[NQ - 1] : m_flag[0], srcl, LPE);
                    if (flag & 1) {                               // close the run ending at entry i
                        if (flag & 2) {
                            const int sid = __shfl(q ? m_sid[NQ - 1] : m_sid[0], srcl, LPE);
                            const int64_t id = __shfl(q ? m_key[NQ - 1] : m_key[0], srcl, LPE);
                            if (d < D) drows[sid * (int64_t)D + d] = a;
                            if (d == 0 && row_ids) row_ids[sid] = id;
                        } else if (d < D) {
                            // a cut run touches the chunk start, the chunk end, or both (then it is the whole chunk: slot 0)
                            part[(((flag & 4) ? nchunk : 0) + ch) * (int64_t)D + d] = a;     // slot-major: [2][nchunk][D]
                        }
                        a = 0.f;
                    }
                }
            }
        }
    }
}
__global__ void __launch_bounds__(256)
k_embed_rows_join(const int64_t* __restrict__ key, const int32_t* __restrict__ order, const int32_t* __restrict__ seg_id,
                  const int32_t* __restrict__ seg_first, int64_t N, int D, int DL, const float* __restrict__ part,
                  float* __restrict__ drows, int64_t* __restrict__ row_ids) {
    __shared__ float red[256];
    __shared__ int64_t own[256];
    __shared__ int nown;
    const int pl = threadIdx.x / DL, dl = threadIdx.x % DL, NPL = 256 / DL;
    const int64_t nchunk = (N + EMB_CH - 1) / EMB_CH;
    for (int64_t base = 1 + (int64_t)blockIdx.x * NPL; base < nchunk; base += (int64_t)gridDim.x * NPL) {
        if (threadIdx.x == 0) nown = 0;
        __syncthreads();
        {   // boundary between chunks c-1 and c, one per group of DL lanes: does a segment cross it, and does it start in
            // chunk c-1 (then this boundary owns it)?  Few pieces (the usual case: ids with a handful of entries) are added
            // right here in chunk order; long segments are left to the whole workgroup below.
            const int64_t c = base + pl;
            if (c < nchunk) {
                const int s = seg_id[c * EMB_CH];
                const int64_t f = seg_first[s];
                if (seg_id[c * EMB_CH - 1] == s && f / EMB_CH == c - 1) {
                    const int64_t c0 = c - 1, c1 = (seg_first[s + 1] - 1) / EMB_CH;
                    if (c1 - c0 < EMB_JSHORT) {
                        const int64_t first_slot = f != c0 * EMB_CH ? 1 : 0;
                        for (int d = dl; d < D; d += DL) {
                            float t = part[(first_slot * nchunk + c0) * (int64_t)D + d];
                            for (int64_t x = c0 + 1; x <= c1; ++x) t += part[x * (int64_t)D + d];
                            drows[s * (int64_t)D + d] = t;
                        }
                        if (dl == 0 && row_ids) row_ids[s] = key[order[f]];
                    } else if (dl == 0) {
                        own[atomicAdd(&nown, 1)] = c;
                    }
                }
            }
        }
        __syncthreads();
        const int no = nown;
        for (int o = 0; o < no; ++o) {                            // list order varies from run to run, the sums do not
            const int64_t c = own[o];
            const int s = seg_id[c * EMB_CH];
            const int64_t f = seg_first[s], l = seg_first[s + 1] - 1;       // first / last sorted position of the segment
            const int64_t c0 = c - 1, c1 = l / EMB_CH;                      // chunks c0 .. c1 hold pieces of s
            // piece of chunk x: slot 1 in c0 unless the segment starts exactly at the chunk start, slot 0 in every later chunk
            const int64_t first_slot = f != c0 * EMB_CH ? 1 : 0;
            for (int d0 = 0; d0 < D; d0 += DL) {
                const int d = d0 + dl;
                float acc[EMB_JU];                                          // EMB_JU loads in flight per thread (one hot id: 65 K pieces)
#pragma unroll
                for (int u = 0; u < EMB_JU; ++u) acc[u] = 0.f;
                if (d < D) {
                    int64_t x = c0 + pl;
                    if (x == c0 && x <= c1) { acc[0] = part[(first_slot * nchunk + x) * (int64_t)D + d]; x += NPL; }
                    for (; x + (EMB_JU - 1) * NPL <= c1; x += EMB_JU * NPL) {
#pragma unroll
                        for (int u = 0; u < EMB_JU; ++u) acc[u] += part[(x + u * NPL) * (int64_t)D + d];
                    }
                    for (; x <= c1; x += NPL) acc[0] += part[x * (int64_t)D + d];
                }
                float a0 = 0.f;
#pragma unroll
                for (int u = 0; u < EMB_JU; ++u) a0 += acc[u];
                __syncthreads();
                red[threadIdx.x] = a0;
                __syncthreads();
                if (pl == 0 && d < D) {
                    float t = red[dl];
                    for (int u = 1; u < NPL; ++u) t += red[u * DL + dl];
                    drows[s * (int64_t)D + d] = t;
                }
            }
            if (threadIdx.x == 0 && row_ids) row_ids[s] = key[order[f]];
        }
        __syncthreads();
    }
}
__global__ void k_embed_rows_tail(const int32_t* __restrict__ n_seg, int64_t N, int64_t* __restrict__ row_ids) {
    const int64_t ns = n_seg[0] < 0 ? N : n_seg[0];      // -1: timed-out grouping = the identity grouping (N segments); the host raises on it
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
    const int LPE = D <= 16 ? 16 : D <= 32 ? 32 : 64;             // lanes per entry; also the dims lanes of the join
    int64_t g = (nchunk + 4 * (64 / LPE) - 1) / (4 * (64 / LPE));
    if (g > 16384) g = 16384;
    if (LPE == 16) hipLaunchKernelGGL(k_embed_rows_chunks<16>, (int)g, 256, 0, st, key, order, seg_id, seg_first, seg, weights, cnt, dout, N, C, T, D, mean, drows, (float*)ws, row_ids);
    else if (LPE == 32) hipLaunchKernelGGL(k_embed_rows_chunks<32>, (int)g, 256, 0, st, key, order, seg_id, seg_first, seg, weights, cnt, dout, N, C, T, D, mean, drows, (float*)ws, row_ids);
    else hipLaunchKernelGGL(k_embed_rows_chunks<64>, (int)g, 256, 0, st, key, order, seg_id, seg_first, seg, weights, cnt, dout, N, C, T, D, mean, drows, (float*)ws, row_ids);
    if (nchunk > 1) {
        const int64_t per = 256 / LPE;                            // boundaries per workgroup and step
        int64_t gj = (nchunk - 1 + per - 1) / per;
        if (gj > 16384) gj = 16384;
        hipLaunchKernelGGL(k_embed_rows_join, (int)gj, 256, 0, st, key, order, seg_id, seg_first, N, D, LPE, (const float*)ws, drows, row_ids);
    }
    if (row_ids) hipLaunchKernelGGL(k_embed_rows_tail, 64, 256, 0, st, n_seg, N, row_ids);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}

// dtable[row_ids[s]][:] = drows[s][:] for every used slot s (row ids are unique: one writer per table row)
__global__ void __launch_bounds__(256)
k_embed_scatter(const float* __restrict__ drows, const int64_t* __restrict__ row_ids, int64_t n_slots, int D, int64_t V, int LPE,
                float* __restrict__ dtable, const int32_t* __restrict__ n_seg) {
    // a group of LPE lanes (pow2 >= D, 16..64) per slot: at D = 16 a wave moves four rows per step
    const int gl = threadIdx.x % LPE;
    const int64_t per = 256 / LPE;
    // only the first n_seg slots hold a segment (the rest carry the sentinel): with the count on the device the sweep stops there instead
    // of reading all N row ids (6.5 M for 0.3 M distinct ids: 79 -> ~10 us)
    if (n_seg && n_seg[0] >= 0 && n_seg[0] < n_slots) n_slots = n_seg[0];
    for (int64_t s = (int64_t)blockIdx.x * per + threadIdx.x / LPE; s < n_slots; s += (int64_t)gridDim.x * per) {
        const int64_t id = row_ids[s];
        if (id < 0 || id >= V) continue;
        for (int d = gl; d < D; d += LPE) dtable[id * D + d] = drows[s * (int64_t)D + d];
    }
}
extern "C" int recnow_embed_scatter_rows(const float* drows, const int64_t* row_ids, int64_t n_slots, int D, int64_t V, float* dtable,
                                         const int32_t* n_seg, void* stream) {
    if (n_slots < 0 || D < 1 || V < 0) return RECNOW_EINVAL;
    if (n_slots == 0 || V == 0) return RECNOW_OK;
    if (!drows || !row_ids || !dtable) return RECNOW_EINVAL;
    const int LPE = D <= 16 ? 16 : D <= 32 ? 32 : 64;
    int64_t g = (n_slots + 256 / LPE - 1) / (256 / LPE);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_embed_scatter, (int)g, 256, 0, (hipStream_t)stream, drows, row_ids, n_slots, D, V, LPE, dtable, n_seg);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}


// Gradient w.r.t. the per-id weights (TF autodiff of rec_now/rec_block/embedding_util.py:315-317, `embeddings * expand_dims(sp_weights)`):
//   dweights[b][c] = <dout[b][seg[b][c]][:], table[rows[b][c]][:]>  (/ cnt[b][t] for 'mean'),  0 for entries that are not pooled.
// LPE lanes per entry (pow2 >= min(D, 64)); both rows are contiguous: coalesced per lane group.
__global__ void __launch_bounds__(256)
k_embed_pool_bwd_weights(const float* __restrict__ table, int D, int64_t V, const int64_t* __restrict__ rows, const int32_t* __restrict__ seg,
                         const float* __restrict__ cnt, const float* __restrict__ dout, int64_t N, int C, int T, int mean, int LPE,
                         float* __restrict__ dweights) {
    const int gl = threadIdx.x % LPE;
    const int64_t per = 256 / LPE;
    // every lane of a group walks the same entry, so the loop trip count is group-uniform; shuffles stay inside the group
    for (int64_t e0 = (int64_t)blockIdx.x * per; e0 < N; e0 += (int64_t)gridDim.x * per) {
        const int64_t e = e0 + threadIdx.x / LPE;
        float p = 0.f;
        int t = -1;
        int64_t b = 0;
        if (e < N) {
            t = seg[e];
            b = e / C;
            if (t >= 0) {
                const int64_t r = rows[e];
                if (r >= 0 && r < V) {
                    const float* g = dout + (b * T + t) * (int64_t)D;
                    const float* tr = table + r * (int64_t)D;
                    for (int d = gl; d < D; d += LPE) p += g[d] * tr[d];
                }
            }
        }
        for (int o = LPE >> 1; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
        if (e < N && gl == 0) {
            if (t >= 0 && mean) {
                const float n = cnt[b * T + t];
                p = n > 0.f ? p / n : 0.f;
            }
            dweights[e] = t >= 0 ? p : 0.f;
        }
    }
}
extern "C" int recnow_embed_pool_bwd_weights(const float* table, int D, int64_t V, const int64_t* rows, const int32_t* seg, const float* cnt,
                                             const float* dout, int64_t B, int C, int T, int mean, float* dweights, void* stream) {
    if (B < 0 || C < 0 || T < 0 || D < 1 || V < 0) return RECNOW_EINVAL;
    const int64_t N = B * C;
    if (N == 0) return RECNOW_OK;
    if (!rows || !seg || !dweights || (T > 0 && (!dout || (V > 0 && !table))) || (mean && !cnt)) return RECNOW_EINVAL;
    const int LPE = D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64;
    int64_t g = (N + 256 / LPE - 1) / (256 / LPE);
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_embed_pool_bwd_weights, (int)g, 256, 0, (hipStream_t)stream, table, D, V, rows, seg, cnt, dout, N, C, T, mean, LPE, dweights);
    RN_LAUNCH_CHECK();
    return RECNOW_OK;
}
