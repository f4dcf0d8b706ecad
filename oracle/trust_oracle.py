"""TEST INFRASTRUCTURE ONLY — CPU restatement of the trust head's path attention layer in torch (autograd gives the
gradients the HIP kernels are compared with).  Reference: LightGCN_SPEX/code/utility2/layers.py:15-71.  Pinned twice:
tests/test_host_logic.py holds this closed form against the reference's own loop structure, and the G11 golden
(tests/golden/trust_tiny.npz, minted from the reference) pins the whole dual-task model that uses it.  Nothing under
spex_amd/ imports this file.
"""
import torch


def path_attention(src, seq, seq_l, a, concat):
    """One GraphAttentionLayer.  concat=True: src = embedding table [rows, H], seq = [B, L] indices (layers.py:17-40);
    concat=False: src is ignored, seq = dense [B, L, H] (layers.py:41-70).  a: [2H] or [2H, 1]."""
    a = a.reshape(-1)
    H = a.numel() // 2
    seq_l = seq_l.long()
    if concat:
        raw = src[seq.long()]                                                     # [B, L, H]
        L = seq.shape[1]
        pos = (seq_l[:, None] - torch.arange(L)[None, :]).to(raw.dtype)           # l - i        (:22, :27)
        x = raw + pos[..., None]
        y = torch.cat([raw[:, 1:] + (pos[:, :-1] - 1)[..., None], torch.zeros_like(raw[:, :1])], dim=1)   # (:23, :28)
    else:
        raw = seq
        L = raw.shape[1]
        x = raw
        y = torch.cat([raw[:, 1:], torch.zeros_like(raw[:, :1])], dim=1)          # (:58-59)
    a1, a2 = a[:H], a[H:]
    att = torch.softmax(torch.stack([x @ a1 + x @ a2, x @ a1 + y @ a2], dim=-1), dim=-1)      # [h|h].a, [h|h'].a  (:29-31)
    mixed = att[..., :1] * x + att[..., 1:] * y                                   # (:32-33)
    valid = torch.arange(L)[None, :] < (seq_l[:, None] - 1)
    return torch.where(valid[..., None], mixed, raw)                              # (:35-39): tail positions keep the input


def dual_task_losses(rowptr, col, val, P, users, items, labels, inputs, mask, targets, n_layers=3, nonhybrid=False, dtype=torch.float64):
    """The whole dual-task forward, model_expert_s.LightGCN.forward(flag=0) — LightGCN_SPEX/code/utility1/model_expert_s.py:
    computer() (:95-126), the two-expert gate (:154-161), dot + BCE (:163-168), the trust head (:170-192) with compute_scores
    (:128-148) — as torch ops on the CPU in `dtype` (fp64 by default), autograd included.
    P: {state_dict name: tensor}; tensors that require grad receive the gradients of whatever the caller backpropagates.
    inputs / mask: [T, L] padded paths and their 0/1 mask (utility2/utils.py:36-51), targets: [T].  Returns (loss1, loss2).
    Pinned by tests/test_oracle_golden.py against the reference's own losses and gradients (G11, trust_tiny.npz)."""
    import numpy as np
    F = torch.nn.functional
    cast = lambda x: x.to(dtype)
    uw, iw = cast(P["embedding_user.weight"]), cast(P["embedding_item.weight"])
    n = uw.shape[0] + iw.shape[0]
    rows = np.repeat(np.arange(n), np.diff(np.asarray(rowptr)))
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, np.asarray(col, np.int64)])), cast(torch.from_numpy(np.asarray(val))), (n, n))
    all_emb = torch.cat([uw, iw])
    embs = [all_emb]
    for _ in range(n_layers):                                                     # :112-121
        all_emb = torch.sparse.mm(A, all_emb)
        embs.append(all_emb)
    light = torch.stack(embs, 1).mean(1)
    ex_u, ex_i = light[: uw.shape[0]], light[uw.shape[0]:]
    att1 = torch.softmax(torch.cat([uw, ex_u], 1) @ cast(P["att_exp1"]), 1)       # :158-161
    att2 = torch.softmax(torch.cat([iw, ex_i], 1) @ cast(P["att_exp2"]), 1)
    all_u = uw * att1[:, :1] + ex_u * att1[:, 1:2]
    all_i = iw * att2[:, :1] + ex_i * att2[:, 1:2]
    users, items = torch.as_tensor(users).long(), torch.as_tensor(items).long()
    gamma = (all_u[users] * all_i[items]).sum(1)                                  # :163-166
    loss1 = F.binary_cross_entropy_with_logits(gamma, cast(torch.as_tensor(labels)))
    inputs, mask, targets = torch.as_tensor(inputs).long(), torch.as_tensor(mask).long(), torch.as_tensor(targets).long()
    seq_l = mask.sum(1)
    n_heads = sum(1 for k in P if k.startswith("attention_") and k.endswith(".a"))
    mul_seq = torch.cat([path_attention(uw, inputs, seq_l, cast(P[f"attention_{h}.a"]), True) for h in range(n_heads)], dim=2)   # :179
    H = uw.shape[1]
    mul_one = F.elu(mul_seq.reshape(-1, mul_seq.shape[2]) @ cast(P["w"]))         # :180-182
    hidden = path_attention(None, mul_one.view(mul_seq.shape[0], mul_seq.shape[1], H), seq_l, cast(P["out_att.a"]), False)       # :183-185
    B = mask.shape[0]
    ht = hidden[torch.arange(B), seq_l - 1]                                       # :129
    q1 = (ht @ cast(P["linear_one.weight"]).t() + cast(P["linear_one.bias"])).view(B, 1, -1)
    q2 = hidden @ cast(P["linear_two.weight"]).t() + cast(P["linear_two.bias"])
    alpha = torch.sigmoid(q1 + q2) @ cast(P["linear_three.weight"]).t()          # :132
    a = (alpha * hidden * mask.view(B, -1, 1).to(dtype)).sum(1)
    p_a = a if nonhybrid else torch.cat([a, ht], 1) @ cast(P["linear_transform.weight"]).t() + cast(P["linear_transform.bias"])
    p_i = uw[inputs] * mask.unsqueeze(2).to(dtype)                                # :140-142
    p_max = p_i.max(dim=1)[0]
    att = torch.softmax(torch.cat([p_a, p_max], 1) @ cast(P["att_t"]), 1)         # :144-145
    a2 = p_a * att[:, :1] + p_max * att[:, 1:2]
    scores = a2 @ uw[:-1].t()                                                     # :137, :147
    loss2 = F.cross_entropy(scores, targets)                                      # :192
    return loss1, loss2
