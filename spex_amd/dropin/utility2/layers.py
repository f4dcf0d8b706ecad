"""Path attention layer behind the reference's `utility2.layers.GraphAttentionLayer` (same parameter `a` [2H, 1],
same outputs), as one HIP kernel launch instead of the reference's Python loops.

Reference: LightGCN_SPEX/code/utility2/layers.py:15-71 walks batch x path positions in Python, building 2x2H matrices
and calling mm/softmax per position (with .item() syncs and host->device copies, :20-23).  In closed form, for a path
x_0 .. x_{l-1} (position i < l-1):

    concat=True :  a = emb[x_i] + (l - i),  b = emb[x_{i+1}] + (l - i - 1)        (scalar position offsets, :22-31)
    concat=False:  a = seq_i,               b = seq_{i+1}                         (:58-63)
    att = softmax([ [a|a] . A, [a|b] . A ]);   out_i = att_0 a + att_1 b;          out_i = raw input for i >= l-1

`spex_path_attention_f32` / `_bwd_f32` (spex_amd/csrc/path.hip; SURVEY.md 8f "next" #1) evaluate that with one wave per
position; the dual-task model calls the kernel once for its three heads (model_expert_s.py).
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
        from spex_amd import ops
        if not self.a.is_cuda:
            raise RuntimeError("spex_amd GraphAttentionLayer runs on the GPU only (no CPU fallback): call .to('cuda')")
        dev = self.a.device
        a = self.a.view(1, -1)
        if self.concat:
            return ops.path_attention(emb, seq.to(dev), seq_l.to(dev), a, True)
        return ops.path_attention(seq.to(dev), None, seq_l.to(dev), a, False)
