/*
 * librecnow_hip.so -- C ABI of the MI355X (gfx950) implementation of rec_now's in-batch ranking-loss and
 * feature-interaction hot path.
 *
 * The reference (ChaoLiangTHU/rec_now) is pure Python on TensorFlow: it has NO FFI/plugin layer, so there is no
 * reference C interface to copy.  Each entry point below therefore cites the reference PYTHON function/lines whose
 * arithmetic it replaces; the Python mirror of the reference's own call signatures lives in rec_now_amd/ and reaches
 * these symbols through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is DEVICE memory (hipMalloc'ed / torch CUDA tensor storage) unless
 *     the parameter name ends in _host.  The caller owns every buffer; the library never allocates or frees.
 *   - every entry point is asynchronous on `stream` (a hipStream_t passed as void*), stateless and re-entrant.
 *   - return value: 0 = ok, negative = RECNOW_E*, positive = a forwarded hipError_t.  No C++ exception crosses.
 *   - scratch memory: `*_workspace_bytes(...)` is queried first, the caller passes `ws`/`ws_bytes`.
 *   - all matrices are row-major, fp32 unless stated; index outputs are int32, counts are int64.
 *   - reductions are deterministic (no floating-point atomics anywhere): same inputs -> same bits.
 */
#ifndef RECNOW_H_
#define RECNOW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RECNOW_OK 0
#define RECNOW_EINVAL (-1)       /* bad argument (null pointer, negative size, unsupported combination) */
#define RECNOW_EWORKSPACE (-2)   /* ws_bytes smaller than *_workspace_bytes() */
#define RECNOW_EUNSUPPORTED (-3) /* shape outside what the kernels implement (documented per function) */

/* activations fused into kernels (keras names: None/'linear', 'relu', 'tanh', 'sigmoid') */
#define RECNOW_ACT_LINEAR 0
#define RECNOW_ACT_RELU 1
#define RECNOW_ACT_TANH 2
#define RECNOW_ACT_SIGMOID 3

/* group-id dtypes accepted by recnow_group_keys */
#define RECNOW_KEY_F32 0
#define RECNOW_KEY_F64 1
#define RECNOW_KEY_I32 2
#define RECNOW_KEY_I64 3

/* pair predicate flags */
#define RECNOW_PAIR_LABEL_GT 1    /* keep (i,j) only if label_i > label_j   (pairwise_loss_from_batch.py:189)     */
#define RECNOW_PAIR_WRONG_ORDER 2 /* keep (i,j) only if score_i < score_j   (pairwise_loss_from_batch.py:197-203) */

int recnow_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Grouping: replaces the dense (B,B) same-group mask of rec_now/rec_block/pairwise_loss_from_batch.py:16-40,43-74
 * and tf.unique_with_counts of listwise_loss_from_batch.py:109 by canonical keys + stable radix sort + segments.
 * ---------------------------------------------------------------------------------------------------------- */

/* Number of 32-bit key words one group tensor of `dtype` contributes (1 or 2); <0 on bad dtype. */
int recnow_key_words(int dtype);

/* Canonicalise one group-id tensor into key words.  Float semantics follow `g_i - g_j == 0.0`
 * (pairwise_loss_from_batch.py:33-35): -0.0 == +0.0; NaN and +-inf rows pair with nobody -> solo[i] |= 1.
 * words: [recnow_key_words(dtype)][B] (word-major, most significant word first).  solo: [B], OR-accumulated. */
int recnow_group_keys(const void* group, int dtype, int64_t B, uint32_t* words, uint8_t* solo, void* stream);

size_t recnow_group_segments_workspace_bytes(int64_t B, int n_words);

/* Stable LSD radix sort of rows by the composite key (all words, lexicographic) and segment detection.
 *   words        [n_words][B]  composite key, words of groups[0] first (n_words_first of them)
 *   solo         [B]           rows that pair with nobody (each becomes a segment of its own)
 * outputs
 *   order        [B]    original row index at sorted position k (ascending row index inside a segment)
 *   seg_id       [B]    segment (= group) index of sorted position k, segments numbered in sorted order
 *   seg_first    [B+1]  first sorted position of segment g; seg_first[n_seg] = B
 *   super_id     [B]    index of the groups[0]-only segment containing sorted position k
 *   n_seg        [2]    {number of segments, number of groups[0]-only segments}
 */
int recnow_group_segments(const uint32_t* words, const uint8_t* solo, int64_t B, int n_words, int n_words_first,
                          int32_t* order, int32_t* seg_id, int32_t* seg_first, int32_t* super_id, int32_t* n_seg,
                          void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * In-batch pairwise loss: rec_now/rec_block/pairwise_loss_from_batch.py
 * ---------------------------------------------------------------------------------------------------------- */
size_t recnow_pairwise_workspace_bytes(int64_t B);

/* Per-row count of valid partners (replaces the mask algebra of pairwise_loss_from_batch.py:254-264).
 *   valid(i,j) = same segment, i != j, mask_i && mask_j, [label_i > label_j], [score_i < score_j]
 *   cnt_row   [B]  c_i by ORIGINAL row index
 *   cnt_super [B]  number of surviving pairs whose positive row lies in groups[0]-segment s  (:282-291)
 *   n_pair    [1]
 * mask may be NULL (all rows valid). */
int recnow_pair_count(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                      const int32_t* seg_id, const int32_t* seg_first, const int32_t* super_id, int64_t B, int flags,
                      int32_t* cnt_row, int64_t* cnt_super, int64_t* n_pair, void* ws, size_t ws_bytes, void* stream);

/* Exclusive scan of cnt_row over ORIGINAL row order -> offsets[B+1] (offsets[B] = n_pair). */
int recnow_pair_offsets(const int32_t* cnt_row, int64_t B, int64_t* offsets, void* ws, size_t ws_bytes, void* stream);

/* Materialise the pair list in the reference's order: row-major over the dense mask = ascending i, then ascending j
 * (tf.boolean_mask of the flattened (B,B) mask, pairwise_loss_from_batch.py:217,272-273).  Bit-exact integer path.
 * pos_idx/neg_idx: [capacity]; pairs beyond capacity are not written. */
int recnow_pair_emit(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                     const int32_t* seg_id, const int32_t* seg_first, int64_t B, int flags, const int64_t* offsets,
                     int32_t* pos_idx, int32_t* neg_idx, int64_t capacity, void* ws, size_t ws_bytes, void* stream);

/* Fused BPR/logistic pairwise loss + gradient without materialising pairs (pairwise_loss_from_batch.py:96-127 applied
 * to the pair set of :254-274; occurrence weights of :130-151,282-291).
 *   loss   = sum_p w_p * softplus(-factor*(s_i - s_j)) / (float(P) + 1e-10)      (or the raw sum if !reduce_mean)
 *   w_p    = cnt_super[super(i_p)] ** power   (power == 0 -> 1)
 *   dscores[k] = d loss / d scores[k]   (pair set and weights are constants, :264,:270)
 * loss: [1] fp32, dscores: [B]. */
int recnow_pair_bpr_fwdbwd(const float* scores, const float* labels, const uint8_t* mask, const int32_t* order,
                           const int32_t* seg_id, const int32_t* seg_first, const int32_t* super_id,
                           const int64_t* cnt_super, const int64_t* n_pair, int64_t B, int flags, float factor,
                           float power, int reduce_mean, float* loss, float* dscores, void* ws, size_t ws_bytes,
                           void* stream);

/* bpr_loss_func on explicit (P,) vectors (pairwise_loss_from_batch.py:96-127).  weights may be NULL.
 * dpos = d loss/d outputs_pos, dneg = -dpos.  P may be 0 (loss = 0). */
int recnow_bpr_loss_fwdbwd(const float* pos, const float* neg, const float* weights, int64_t P, float factor,
                           int reduce_mean, float* loss, float* dpos, void* ws, size_t ws_bytes, void* stream);

/* Dense (B,B) bool mask, for API parity with generate_pair_mask (pairwise_loss_from_batch.py:43-74).
 * only_upper_band keeps only j == i+1 (band_part(m,0,1) minus the diagonal, :38-39).  mask_out must be zero-filled. */
int recnow_pair_mask_dense(const int32_t* order, const int32_t* seg_id, const int32_t* seg_first, int64_t B,
                           int only_upper_band, uint8_t* mask_out, void* stream);

/* occurance_power_weight (pairwise_loss_from_batch.py:130-151): w[i] = (size of i's segment) ** power. */
int recnow_occurance_power_weight(const int32_t* order, const int32_t* seg_id, const int32_t* seg_first, int64_t B,
                                  float power, float* w_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * FMLayer: rec_now/layers/fm_layer.py:24-42.   HBM-bound (12*B*F*D bytes fwd+bwd).
 *   y[b] = 0.5 * sum_d [ (sum_f x_f[b][d])^2 - sum_f x_f[b][d]^2 ]
 * fields: DEVICE array of F device pointers, each a contiguous (B,D) fp32 tensor (the reference's list input);
 * y: [B]; S: [B*D] field sums saved for backward (NULL = forward only).
 * ---------------------------------------------------------------------------------------------------------- */
int recnow_fm_fwd(const float* const* fields, int F, int64_t B, int D, float* y, float* S, void* stream);
/* dfields[f][b][d] = gy[b] * (S[b][d] - x_f[b][d]);  dfields: DEVICE array of F device pointers to (B,D) buffers. */
int recnow_fm_bwd(const float* const* fields, float* const* dfields, int F, int64_t B, int D, const float* S,
                  const float* gy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RECNOW_H_ */
