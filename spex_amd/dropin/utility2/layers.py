"""Path attention layer behind the reference's `utility2.layers.GraphAttentionLayer` (same parameter `a` [2H, 1],
same outputs), without the reference's Python loops.

Reference: LightGCN_SPEX/code/utility2/layers.py:15-71 walks batch x path positions in Python, building 2x2H matrices
and calling mm/softmax per position (with .item() syncs and host->device copies, :20-23).  In closed form, for a path
x_0 .. x_{l-1} (position i < l-1):

    concat=True :  a = emb[x_i] + (l - i),  b = emb[x_{i+1}] + (l - i - 1)        (scalar position offsets, :22-31)
    concat=False:  a = seq_i,               b = seq_{i+1}                         (:58-63)
    att = softmax([ [a|a] . A, [a|b] . A ]);   out_i = att_0 a + att_1 b;          out_i = raw input for i >= l-1

which is a handful of batched tensor ops over [B, Lmax, H].  SURVEY.md 8f "next" #1: expressed with torch ops on the
device for now (differentiable for free); the per-row work is tiny next to the graph propagation.
"""
import torch
import torch.nn as nn


class GraphAttentionLayer(nn.Module):
    def __init__(self, hidden_size, concat=True):
        super().__init__()
        self.concat = concat
        self.hidden_size = hidden_size
        self.a = nn.Parameter(torch.zeros(size=(2 * hidden_size, 1)))
        nn.init.xavier_uniform_(self.a.data, gain=1.414)
        self.leakyrelu = nn.LeakyReLU(inplace=True)   # defined, never applied, in the reference too (:13)

    def forward(self, emb, seq, seq_l):
        H = self.hidden_size
        seq_l = seq_l.to(self.a.device)
        if self.concat:
            seq = seq.to(self.a.device).long()
            raw = emb.index_select(0, seq.reshape(-1)).view(*seq.shape, H)    # [B, L, H]; index_select: atomic backward
            L = seq.shape[1]
            pos = (seq_l[:, None] - torch.arange(L, device=raw.device)[None, :]).to(raw.dtype)   # l - i
            a = raw + pos[..., None]
            b = torch.cat([raw[:, 1:] + (pos[:, :-1] - 1)[..., None], torch.zeros_like(raw[:, :1])], dim=1)
        else:
            raw = seq.to(self.a.device)
            L = raw.shape[1]
            a = raw
            b = torch.cat([raw[:, 1:], torch.zeros_like(raw[:, :1])], dim=1)
        a1, a2 = self.a[:H, 0], self.a[H:, 0]
        att0 = a @ a1 + a @ a2                                               # [a|a] . A
        att1 = a @ a1 + b @ a2                                               # [a|b] . A
        w = torch.softmax(torch.stack([att0, att1], dim=-1), dim=-1)
        mixed = w[..., :1] * a + w[..., 1:] * b
        valid = torch.arange(L, device=raw.device)[None, :] < (seq_l[:, None] - 1)
        return torch.where(valid[..., None], mixed, raw)
