"""Rec branch of the dual-task model behind the reference's `utility1.model_expert_s` surface.

Reference: LightGCN_SPEX/code/utility1/model_expert_s.py — LightGCN propagation (`computer`, :95-126, identical to
model.py), then a two-expert gate between raw and propagated tables (:154-161), then dot + BCE (:163-168).  The
trust-path head of the same class (:170-192, utility2/layers.py) is SURVEY.md 8f "next" #1 and is not built yet:
`flag=2`, and `flag=0` with trust data, raise NotImplementedError.  Without the head's parameters the torch RNG stream
at construction differs from the reference's, so tests inject weights instead of comparing by seed.

    att = softmax([E0 | out] @ att_exp, dim=1);  mixed = E0 * att[:, 0] + out * att[:, 1]     (users and items apart)

Inference runs the fused gate kernel (`spex_expert_gate_f32`); under autograd the gate (a [rows,128]x[128,2] product,
negligible next to the propagation) is expressed with torch ops so it is differentiated for free.
"""
import torch
from torch import nn

from spex_amd import ops
from utility1.model import LightGCN as _LightGCN


class LightGCN(_LightGCN):
    def __init__(self, args_r, dataset):
        super().__init__(args_r, dataset)
        self.hidden_size = args_r.hiddenSize
        self.task_weights = nn.Parameter(torch.zeros(2))
        self.rec_loss = nn.BCEWithLogitsLoss()
        self.att_exp1 = nn.Parameter(torch.zeros(2 * self.hidden_size, 2))
        self.att_exp2 = nn.Parameter(torch.zeros(2 * self.hidden_size, 2))
        nn.init.xavier_uniform_(self.att_exp1.data, gain=1)
        nn.init.xavier_uniform_(self.att_exp2.data, gain=1)

    def _gated_tables(self):
        light_out = self._light_out()
        n_u = self.num_users + 1
        raw_u, raw_i = self.embedding_user.weight, self.embedding_item.weight
        out_u, out_i = light_out[:n_u], light_out[n_u:]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            a1 = torch.softmax(torch.cat([raw_u, out_u], 1) @ self.att_exp1, 1)
            a2 = torch.softmax(torch.cat([raw_i, out_i], 1) @ self.att_exp2, 1)
            return raw_u * a1[:, :1] + out_u * a1[:, 1:], raw_i * a2[:, :1] + out_i * a2[:, 1:]
        return (ops.expert_gate(raw_u.detach().contiguous(), out_u.contiguous(), self.att_exp1.detach()),
                ops.expert_gate(raw_i.detach().contiguous(), out_i.contiguous(), self.att_exp2.detach()))

    def forward(self, users, items, labels, slice_indices=None, trust_data=None, flag=0):
        if flag == 2 or (flag == 0 and trust_data is not None):
            raise NotImplementedError("the trust-path head (model_expert_s.py:170-192) is not part of this build yet")
        all_users, all_items = self._gated_tables()
        dev = all_users.device
        if flag == 1:
            gamma, _ = ops.score_bce(all_users.detach().contiguous(), all_items.detach().contiguous(), users, items)
            return gamma
        users_emb, items_emb = all_users[ops._idx(users, dev)], all_items[ops._idx(items, dev)]
        gamma = torch.sum(users_emb * items_emb, dim=1)
        return self.rec_loss(gamma, labels.to(dev).float())
