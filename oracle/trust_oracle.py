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
