"""TEST INFRASTRUCTURE ONLY (oracle/): never imported, called or linked by the product package.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, as the checker.

CPU restatement of the reference's hot path IN THE REFERENCE'S OWN DENSE FORMULATION, op by op,
on torch CPU tensors (fp32 by default; pass fp64 tensors for gradient references).  torch autograd
of these functions is the backward oracle (the reference leaves backward to TF autodiff).

Every function cites the reference file:line it follows (paths relative to /root/reference/).
Parity status: PINNED -- tests/test_oracle_golden.py checks these functions against every golden
of the reference's nine hot-path test files, with seeded inputs regenerated
by oracle/tf_seeded_rng.py.  TensorFlow itself is not installable here.
"""
import torch

SMALL_POSIVITE_FLOAT = 1.0e-10   # rec_now/rec_block/pairwise_loss_from_batch.py:13


# ----------------------------------------------------------------------------------------------
# rec_now/rec_block/pairwise_loss_from_batch.py
# ----------------------------------------------------------------------------------------------
def _generate_pair_mask(group, only_upper_band=False):
    """pairwise_loss_from_batch.py:16-40 -- float compare g_i - g_j == 0.0, minus eye, cast bool."""
    n = group.numel()
    g = group.reshape(-1, 1)
    diff = g - g.t()                                            # :33
    same = (diff == 0.0).to(torch.float32)                      # :35
    pair_mask = (same - torch.eye(n, dtype=torch.float32)).to(torch.bool)   # :36-37
    if only_upper_band:
        # tf.linalg.band_part(m, 0, 1): keep main diagonal and FIRST super-diagonal only  :38-39
        i = torch.arange(n).reshape(-1, 1)
        j = torch.arange(n).reshape(1, -1)
        pair_mask = pair_mask & ((j - i) >= 0) & ((j - i) <= 1)
    return pair_mask


def generate_pair_mask(group_tensor_or_list, only_upper_band=False):
    """pairwise_loss_from_batch.py:43-74 -- AND over the list of group tensors."""
    if not isinstance(group_tensor_or_list, list):
        group_tensor_or_list = [group_tensor_or_list]
    pair_mask = None
    for group in group_tensor_or_list:
        one = _generate_pair_mask(group, only_upper_band)
        pair_mask = one if pair_mask is None else (pair_mask & one)
    return pair_mask


def vec_to_matrix_pair(vec):
    """pairwise_loss_from_batch.py:77-93 -- M[i,j]=v_i and its transpose."""
    vec = vec.reshape(-1, 1)
    mat = vec.repeat(1, vec.numel())
    return mat, mat.t()


def bpr_loss_func(outputs_pos, outputs_neg, weights=None, factor=1.0, reduce_mean=True):
    """pairwise_loss_from_batch.py:96-127.
    sigmoid_cross_entropy_with_logits(labels=1, x) = max(x,0) - x + log1p(exp(-|x|))."""
    logits = outputs_pos - outputs_neg
    if factor != 1.0:
        logits = logits * factor
    losses = torch.clamp(logits, min=0) - logits + torch.log1p(torch.exp(-torch.abs(logits)))
    if weights is not None:
        losses = losses * weights
    loss = losses.sum()
    if reduce_mean:
        loss = loss / (float(losses.numel()) + SMALL_POSIVITE_FLOAT)
    return loss


def occurance_power_weight(group_id, power=0.0):
    """pairwise_loss_from_batch.py:130-151 -- count**power gathered back per element."""
    group_id = torch.as_tensor(group_id)
    _, idx, count = torch.unique(group_id, return_inverse=True, return_counts=True)
    weights = count.to(torch.float32)
    if power != 1.0:
        weights = torch.pow(weights, power)
    return weights[idx]


def _apply_pair_mask(mat, flat_mask):
    """pairwise_loss_from_batch.py:206-217 -- boolean_mask over the row-major flattening."""
    if mat is None:
        return None
    return mat.reshape(-1)[flat_mask]


def pairwise_loss(outputs, labels, groups, pairloss_func=bpr_loss_func, only_use_wrong_order_pair=False,
                  return_num_pair=False, click_occurance_power=0.0, mask=None,
                  label_pair_to_weight_func=None, **kwargs):
    """pairwise_loss_from_batch.py:228-279 (+ helpers :154-203, :282-291)."""
    pair_mask = generate_pair_mask(groups)                                   # :254
    if mask is not None:                                                     # :255 -> :154-172
        mm, mmt = vec_to_matrix_pair(mask.to(torch.bool))
        pair_mask = pair_mask & (mm & mmt)
    om, omt = vec_to_matrix_pair(outputs)                                    # :256
    lm, lmt = vec_to_matrix_pair(labels)                                     # :257 -> :175-194
    if label_pair_to_weight_func is None:
        label_cond = lm > lmt
        weights_mat = None
    else:
        weights_mat = label_pair_to_weight_func(lm, lmt, **kwargs)
        label_cond = weights_mat > 0
    pair_mask = pair_mask & label_cond                                       # :259
    if only_use_wrong_order_pair:                                            # :260 -> :197-203
        pair_mask = pair_mask & (om.detach() < omt.detach())
    flat = pair_mask.reshape(-1)                                             # :263-264 (stop_gradient)
    weights = _apply_pair_mask(weights_mat, flat)                            # :266
    if click_occurance_power != 0.0:                                         # :267 -> :282-291
        group = groups[0] if isinstance(groups, list) else groups
        gm, _ = vec_to_matrix_pair(group)
        groups_pos = _apply_pair_mask(gm, flat)
        occ = occurance_power_weight(groups_pos, power=click_occurance_power)
        weights = occ if weights is None else weights * occ                  # :220-225
    if weights is not None:
        weights = weights.detach()                                           # :269-270
    outputs_pos = _apply_pair_mask(om, flat)                                 # :272
    outputs_neg = _apply_pair_mask(omt, flat)                                # :273
    loss = pairloss_func(outputs_pos, outputs_neg, weights)                  # :274
    if return_num_pair:
        return loss, float(outputs_pos.numel())                              # :275-277
    return loss


def pair_indices(labels, groups, only_use_wrong_order_pair=False, outputs=None, mask=None):
    """(pos_idx, neg_idx) of the surviving pairs in the reference's order = row-major nonzero of the
    dense mask (pairwise_loss_from_batch.py:217,272-273)."""
    pair_mask = generate_pair_mask(groups)
    if mask is not None:
        mm, mmt = vec_to_matrix_pair(mask.to(torch.bool))
        pair_mask = pair_mask & (mm & mmt)
    lm, lmt = vec_to_matrix_pair(labels)
    pair_mask = pair_mask & (lm > lmt)
    if only_use_wrong_order_pair:
        om, omt = vec_to_matrix_pair(outputs)
        pair_mask = pair_mask & (om < omt)
    nz = torch.nonzero(pair_mask)
    return nz[:, 0].contiguous(), nz[:, 1].contiguous()


# ----------------------------------------------------------------------------------------------
# rec_now/rec_block/listwise_loss_from_batch.py
# ----------------------------------------------------------------------------------------------
def nan_to_zero(val):
    """listwise_loss_from_batch.py:74-86."""
    if val.dim() != 0:
        raise ValueError('input muust be a scalar tf.Tensor')
    return torch.zeros((), dtype=val.dtype) if torch.isnan(val) else val


def to_listwise_sample(group_ids, labels, logits, do_mask_logits=True, value_of_masked_logit=-1e9,
                       pos_neg_th=0.5):
    """listwise_loss_from_batch.py:89-148. unique_with_counts -> first-occurrence group order."""
    group_ids = group_ids.reshape(-1)
    labels = labels.reshape(-1)
    logits = logits.reshape(-1)
    n = group_ids.numel()
    # tf.unique: y in first-occurrence order; idx = rank of each row's group           :109
    seen = {}
    idx = []
    for v in group_ids.tolist():
        if v not in seen:
            seen[v] = len(seen)
        idx.append(seen[v])
    idx = torch.tensor(idx, dtype=torch.long)
    g = len(seen)
    cols = torch.arange(n)

    def gen_dense(values):                                                             # :123-129
        d = torch.zeros((g, n), dtype=values.dtype)
        d[idx, cols] = values.reshape(-1)
        return d

    dense_mask = gen_dense(torch.ones(n, dtype=torch.bool))                           # :131
    dense_labels = gen_dense(labels)                                                  # :132
    dense_logits = gen_dense(logits)                                                  # :133
    has_pos = ((dense_labels.to(torch.float32) > pos_neg_th).to(torch.int32).sum(-1) > 0)      # :135
    has_neg = ((gen_dense(labels - pos_neg_th).to(torch.float32) < 0.0).to(torch.int32).sum(-1) > 0)  # :136
    row_mask = has_pos & has_neg                                                      # :137
    if do_mask_logits:                                                                # :139-140
        dense_logits = dense_logits + (1.0 - dense_mask.to(dense_logits.dtype)) * value_of_masked_logit
    dense_mask = dense_mask[row_mask]                                                 # :142
    dense_labels = dense_labels[row_mask]                                             # :143
    dense_labels = dense_labels / dense_labels.sum(-1, keepdim=True)                  # :144
    dense_logits = dense_logits[row_mask]                                             # :145
    return dense_mask, dense_labels.detach(), dense_logits                            # :147-148


def listwise_loss_via_softmax_cross_entropy_with_logits(labels_for_softmax, logits_for_softmax,
                                                        weights=None, do_reduce=True):
    """listwise_loss_from_batch.py:151-173. softmax_cross_entropy_with_logits = -sum p*log_softmax."""
    labels_for_softmax = labels_for_softmax.detach()
    loss = -(labels_for_softmax * torch.log_softmax(logits_for_softmax, dim=-1)).sum(-1)
    if weights is not None:
        loss = loss * weights
    if do_reduce:
        loss = loss.mean() if loss.numel() > 0 else torch.full((), float('nan'), dtype=logits_for_softmax.dtype)
        loss = nan_to_zero(loss)
    return loss


# ----------------------------------------------------------------------------------------------
# rec_now/layers/*  (functional: weights are explicit arguments, named as the reference names them)
# ----------------------------------------------------------------------------------------------
def _act(name):
    if name is None or name == 'linear':
        return lambda v: v
    if callable(name):                      # keras.activations.get passes callables through
        return name
    return {'relu': torch.relu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid}[name]


def fm_layer(inputs):
    """layers/fm_layer.py:24-42. inputs: list of F (B,D) tensors (or one tensor, wrapped :33-34)."""
    if not isinstance(inputs, list):
        inputs = [inputs]
    x = torch.stack(inputs, 0)
    s = x.sum(0)                                                # :36
    second = s * s - (x * x).sum(0)                             # :37-40
    return 0.5 * second.sum(1, keepdim=True)                    # :41


def cin_layer(inputs, weights, num_field, embedding_dim, output_input=True, sum_channel=True):
    """layers/cin_layer.py:72-122. weights[k-1] has shape (1,1,H_k,H_{k-1}*F) ("weight_of_layer{k}")."""
    emb = torch.cat(inputs, dim=1) if isinstance(inputs, list) else inputs        # :88-91
    layer0 = emb.reshape(-1, num_field, embedding_dim).permute(0, 2, 1)           # :96-97 (B,D,F)
    layers = [layer0]
    for w in weights:                                                             # :101-110
        prev = layers[-1]
        hidden = torch.einsum('bdf,bdh->bdfh', layer0, prev)                      # :103
        hidden = hidden.reshape(hidden.shape[0], embedding_dim, -1, 1)            # :105-106
        hidden = torch.matmul(w, hidden).squeeze(-1)                              # :108-109 (B,D,H_k)
        layers.append(hidden)
    if not output_input:
        layers = layers[1:]                                                       # :112-113
    out = torch.cat(layers, dim=-1)                                               # :115
    if sum_channel:
        return out.sum(-1)                                                        # :116-117
    out = out.permute(0, 2, 1)                                                    # :119
    return out.reshape(out.shape[0], -1)                                          # :120-121


def cin_layer_gemm_form(inputs, weights, num_field, embedding_dim, output_input=True, sum_channel=True):
    """layers/cin_layer.py:72-122 once more, for FULL-SIZE use by the chunked oracle (tests/_chunked_oracle.py): the per-(b, d)
    matrix-vector product of :108-109, `matmul(w (1,1,H_k,H_{k-1}*F), hidden (B,D,H_{k-1}*F,1))`, written as ONE matrix product
    `hidden.reshape(B*D, H_{k-1}*F) @ w^T` -- the same contraction over the column index f*H_{k-1}+h, term by term, but a GEMM
    instead of B*D mat-vecs that each re-read the weight (at B = 16 384, F = 64, H = 128 the line-by-line form moves terabytes).
    tests/test_oracle_golden.py holds it to `cin_layer` (and through it to the reference's golden)."""
    emb = torch.cat(inputs, dim=1) if isinstance(inputs, list) else inputs        # :88-91
    layer0 = emb.reshape(-1, num_field, embedding_dim).permute(0, 2, 1)           # :96-97 (B,D,F)
    B = layer0.shape[0]
    layers = [layer0]
    for w in weights:                                                             # :101-110
        prev = layers[-1]
        hidden = torch.einsum('bdf,bdh->bdfh', layer0, prev)                      # :103
        hidden = hidden.reshape(B * embedding_dim, -1)                            # :105-106 (B*D, F*H_{k-1})
        hk = w.shape[2]
        layers.append((hidden @ w.reshape(hk, -1).t()).reshape(B, embedding_dim, hk))      # :108-109
    if not output_input:
        layers = layers[1:]                                                       # :112-113
    out = torch.cat(layers, dim=-1)                                               # :115
    if sum_channel:
        return out.sum(-1)                                                        # :116-117
    out = out.permute(0, 2, 1)                                                    # :119
    return out.reshape(out.shape[0], -1)                                          # :120-121


def dcn_layer(inputs, kernels, biases=None, activation=None):
    """layers/dcn_layer.py:79-103. kernels[l] (D,1), biases[l] (1,D). No residual term."""
    act = _act(activation)
    layer_input = inputs
    for l, kernel in enumerate(kernels):
        cross = layer_input @ kernel                        # :94
        out = inputs * cross                                # :95
        if biases is not None:
            out = out + biases[l]                           # :96-98
        out = act(out)                                      # :99
        layer_input = out
    return out


def dcn_mix_layer(inputs, origin_to_sub, sub_to_sub, sub_to_origin, biases, gate_kernels,
                  activation_inner='tanh', activation_outer='tanh'):
    """layers/dcn_mix_layer.py:114-151. Per layer l: origin_to_sub[l] (N,D,S), sub_to_sub[l] (N,S,S),
    sub_to_origin[l] (N,S,D), biases[l] (1,N,D), gate_kernels[l] (D,N) (Dense, use_bias=False :101)."""
    ai, ao = _act(activation_inner), _act(activation_outer)
    ext = inputs.unsqueeze(1)                                                       # :123
    layer_input = inputs
    for U, V, W, b, K in zip(origin_to_sub, sub_to_sub, sub_to_origin, biases, gate_kernels):
        sub = torch.einsum('bd,nds->bns', layer_input, U)                           # :135 tensordot
        sub = ai(sub)                                                               # :136
        sub = torch.einsum('bns,nst->bnt', sub, V)                                  # :137
        sub = ao(sub)                                                               # :138
        org = torch.einsum('bns,nsd->bnd', sub, W)                                  # :141
        org = org + b                                                               # :142
        org = ext * org                                                             # :143
        gates = torch.softmax(layer_input @ K, dim=-1)                              # :146-147
        layer_input = torch.einsum('bnd,bn->bd', org, gates)                        # :149-150
    return layer_input


def multi_dense_layer(inputs, kernel, bias=None, activation=None):
    """layers/multi_dense_layer.py:80-94. kernel (N,D,U), bias (N,1,U)."""
    if inputs.dim() == 2:
        inputs = inputs.unsqueeze(0)                       # :88-89
    if inputs.shape[0] not in (1, kernel.shape[0]):
        raise ValueError('batch dims %s vs. %s' % (list(inputs.shape), list(kernel.shape)))
    out = torch.matmul(inputs, kernel)                     # :90
    if bias is not None:
        out = out + bias                                   # :91-92
    return _act(activation)(out)                           # :93


def mmoe_layer(inputs, expert_kernels, expert_biases, gate_kernel, gate_bias, activation=None,
               merge_output=True):
    """layers/mmoe_layer.py:96-126. experts: stack of MultiDense (:64-86; last layer no activation),
    gates: MultiDense(num_experts, num_task) with bias + Softmax (:46-62)."""
    h = inputs
    nl = len(expert_kernels)
    for i, (k, b) in enumerate(zip(expert_kernels, expert_biases)):
        h = multi_dense_layer(h, k, b, None if i == nl - 1 else activation)         # :109
    experts = h.unsqueeze(0)                                                        # :110 (1,N,B,U)
    gates = torch.softmax(multi_dense_layer(inputs, gate_kernel, gate_bias), -1)    # :112 (T,B,N)
    gates = gates.permute(0, 2, 1).unsqueeze(-1)                                    # :113-114
    out = (experts * gates).sum(1)                                                  # :116-117
    if merge_output:
        return out
    return [out[t] for t in range(out.shape[0])]                                    # :122-126


def ple_layer(inputs, layers, is_shared_tasks, activation=None):
    """layers/ple_layer.py:295-321.  `layers` = list over PLE layers of dicts:
      'dnn':  list over task groups (shared first) of list of (kernel (N,Din,U), bias (N,1,U)|None)
      'gate': list over task groups of (kernel (Din,units), bias (units,)) or None
    """
    outputs = []
    num_layer = len(layers)
    for li, layer in enumerate(layers):
        is_first, is_last = li == 0, li == num_layer - 1
        dnn_outputs, task_inputs = [], []
        for ti, (shared, dnn) in enumerate(zip(is_shared_tasks, layer['dnn'])):     # :259-272
            if is_first:
                x = inputs
            else:
                last = outputs[-1]
                if shared:
                    x = torch.cat(last, dim=-1)                                     # :228-229
                else:
                    parts = [last[ti]] + [o for o, s in zip(last, is_shared_tasks) if s]   # :231-236
                    x = torch.cat(parts, dim=-1)
            task_inputs.append(x)
            h = x
            for i, (k, b) in enumerate(dnn):
                h = multi_dense_layer(h, k, b, None if i == len(dnn) - 1 else activation)
            dnn_outputs.append(h)
        gated = []
        for ti, (shared, gate) in enumerate(zip(is_shared_tasks, layer['gate'])):   # :274-293
            if shared and is_last:
                gated.append(None)
                continue
            gk, gb = gate
            g = torch.softmax(task_inputs[ti] @ gk + gb, dim=-1)                    # :283-284 (B,N)
            g = g.t().unsqueeze(2)                                                  # :285-286 (N,B,1)
            if shared:
                e = torch.cat(dnn_outputs, dim=0)                                   # :249-250
            else:
                parts = [dnn_outputs[ti]] + [o for o, s in zip(dnn_outputs, is_shared_tasks) if s]
                e = torch.cat(parts, dim=0)                                         # :251-257
            gated.append((e * g).sum(0))                                            # :289-290
        outputs.append(gated)
    return [o for o in outputs[-1] if o is not None]                                # :318-321


# ----------------------------------------------------------------------------------------------
# SURVEY.md section 8f rows (the callers / neighbours of the hot path), same conventions as above
# ----------------------------------------------------------------------------------------------
def inner_pnn_layer(inputs):
    """layers/inner_pnn_layer.py:25-53. inputs: list of F (B,D) tensors -> (B, F*(F-1)/2); pair order r < c, r-major (:41-45)."""
    x = torch.stack(inputs, 0)                                  # (F,B,D), :38-39
    F = x.shape[0]
    row = [r for r in range(F - 1) for _ in range(r + 1, F)]
    col = [c for r in range(F - 1) for c in range(r + 1, F)]
    prod = x[row] * x[col]                                      # :47-49
    return prod.sum(-1).transpose(0, 1)                         # :51-52


def senet_layer(inputs, kernels, biases, activation_inner='tanh', activation_outer='tanh'):
    """layers/senet_layer.py:93-119. inputs: list of F (B,D_f); kernels = [(F,mid), (mid,F)], biases or None.
    mid = max(round(F * reduction_ratio), 1) (:68-74) is implied by the kernel shapes."""
    if not isinstance(inputs, list):
        inputs = [inputs]
    sq = torch.cat([x.mean(-1, keepdim=True) for x in inputs], -1)      # :104-110
    h = sq @ kernels[0]
    if biases is not None and biases[0] is not None:
        h = h + biases[0]
    h = _act(activation_inner)(h)
    w = h @ kernels[1]
    if biases is not None and biases[1] is not None:
        w = w + biases[1]
    w = _act(activation_outer)(w)                                        # :111
    pos_idx = [f for f, x in enumerate(inputs) for _ in range(x.shape[-1])]   # :86-88
    ew = w[:, pos_idx]                                                   # embedding_wise_weight.py:27-36
    return torch.cat(inputs, -1) * ew                                    # :114-117


def focal_crossentropy_loss(labels, logits, alpha=0.25, gamma=2.0, stop_weight_gradient=False, return_mean=True):
    """rec_block/focal_loss.py:12-66."""
    if alpha and (alpha <= 0.0 or alpha >= 1.0):
        raise ValueError('Value of alpha should be greater than zero and less than one.')      # :43-44
    if gamma and gamma < 0:
        raise ValueError('Value of gamma should be greater than or equal to zero.')            # :45-46
    fl = torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))   # :48 (TF's stable form)
    if alpha:
        fl = (labels * alpha + (1 - labels) * (1 - alpha)) * fl                                 # :50-53
    if gamma:
        p = torch.sigmoid(logits)
        sim = labels * p + (1 - labels) * (1 - p)                                               # :56-57
        mod = torch.pow(1.0 - sim, gamma)                                                       # :59
        if stop_weight_gradient:
            mod = mod.detach()                                                                  # :60-61
        fl = mod * fl
    return fl.mean() if return_mean else fl                                                     # :64-66


def attention_by_dot_product(user_emb, doc_emb, filter_neg=False):
    """rec_block/attention.py:12-38. user_emb (B,L,D), doc_emb (B,D) -> (B,D), (B,1)."""
    score = (user_emb * doc_emb[:, None, :]).sum(2, keepdim=True)       # :28-30
    if filter_neg:
        score = torch.clamp(score, min=0.0)                              # :31-32
    mat = (user_emb * score).sum(1)                                      # :33-34
    return mat, score.squeeze(2).sum(1, keepdim=True)                    # :36-38


def embedding_using_sparse_batch_segment_ids(params, slots, target_slots, ids, weights=None, method='sum'):
    """rec_block/embedding_util.py:239-324 with embedding_func = row lookup in `params` (V,D) (the docstring's own example,
    :254-256).  slots, ids: (B,C) integer; target_slots: list of T ints; weights (B,C) or None -> (B,T,D).
    Column c of row b is pooled into target t iff slots[b][c] == target_slots[t]; 'mean' divides by the number of pooled
    entries (unsorted_segment_mean: empty segments give 0)."""
    B, C = ids.shape
    T = len(target_slots)
    emb = params[ids.reshape(-1).long()].reshape(B, C, -1)
    if weights is not None:
        emb = emb * weights[:, :, None]
    ts = torch.as_tensor(list(target_slots), dtype=slots.dtype)
    m = (slots[:, :, None] == ts[None, None, :]).to(emb.dtype)           # (B,C,T)
    out = torch.einsum('bct,bcd->btd', m, emb)
    if method == 'mean':
        cnt = m.sum(1)                                                   # (B,T)
        out = torch.where(cnt[:, :, None] > 0, out / cnt.clamp(min=1)[:, :, None], torch.zeros_like(out))
    return out


def calc_sum_of_abs_diff(arr1, arr2):
    """util/numpy_tools.py:12-27 -- the parity metric of every reference test."""
    import numpy as np
    a = np.array(arr1, dtype=np.float64)
    b = np.array(arr2, dtype=np.float64)
    return float(np.sum(np.abs(a - b)))
