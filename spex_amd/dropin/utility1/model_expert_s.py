"""Dual-task model (recommendation + trust-path prediction, shared user table) behind the reference's
`utility1.model_expert_s` surface, so LightGCN_SPEX/code/main_auto_expert_s.py runs on it unchanged.

Reference: LightGCN_SPEX/code/utility1/model_expert_s.py.  Same constructor, parameter names (state_dicts are
interchangeable), creation order (same values for the same torch seed) and `forward(users, items, labels,
slice_indices, trust_data, flag)` contract:

  flag 0  -> (rec BCE loss, trust cross-entropy loss)          flag 1 -> rec scores          flag 2 -> (trust scores, negs)

What runs where
  * rec branch: HIP propagation (`computer`, :95-126 == model.py:66-97), two-expert gate between raw and propagated
    tables (:154-161, `spex_expert_gate_f32` and its backward kernel), dot + BCE (:163-168);
  * trust branch (SURVEY.md 8f "next" #1, :170-192 + compute_scores :128-148): three path-attention heads, [B*L,192] x
    [192,64] + ELU, an output attention layer, soft-attention readout, max-pool gate, logits against the whole user
    table and cross-entropy.  The reference evaluates the attention layers with Python loops over batch x path
    position; here each attention layer is one HIP kernel launch over all heads (spex_path_attention_f32, with its
    backward kernel); the small dense layers around it ([B, <=6, 64]) are torch ops on the device.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from spex_amd import ops
from utility1.model import LightGCN as _RecLightGCN
from utility2.layers import GraphAttentionLayer


def _rows(table, idx):
    """table[idx] for an index tensor of any shape, through index_select: its backward is an atomic index_add_, where
    advanced indexing's backward (index_put_ with accumulate) sorts the indices — 120 us per call here, a quarter of the
    dual-task step's GPU time."""
    return table.index_select(0, idx.reshape(-1)).view(*idx.shape, table.shape[1])


class BasicModel(nn.Module):
    def getUsersRating(self, users):
        raise NotImplementedError


class LightGCN(_RecLightGCN):
    def __init__(self, args_r, dataset):
        # creation order follows model_expert_s.py:19-70 line by line: the torch RNG is consumed identically
        nn.Module.__init__(self)
        self.args_r, self.dataset = args_r, dataset
        self.hidden_size = args_r.hiddenSize
        self.batch_size = args_r.batchSize
        self.nonhybrid = args_r.nonhybrid
        self.linear_one = nn.Linear(self.hidden_size, self.hidden_size, bias=True)
        self.linear_two = nn.Linear(self.hidden_size, self.hidden_size, bias=True)
        self.linear_three = nn.Linear(self.hidden_size, 1, bias=False)
        self.linear_transform = nn.Linear(self.hidden_size * 2, self.hidden_size, bias=True)
        self.reset_parameters()
        self.in_att = [GraphAttentionLayer(self.hidden_size, concat=True) for _ in range(args_r.nb_heads)]
        for i, attention in enumerate(self.in_att):
            self.add_module("attention_{}".format(i), attention)
        self.out_att = GraphAttentionLayer(self.hidden_size, concat=False)
        self.w = nn.Parameter(torch.zeros(size=(args_r.nb_heads * self.hidden_size, self.hidden_size)))
        nn.init.xavier_uniform_(self.w.data, gain=1.414)

        self.bcel = nn.BCEWithLogitsLoss()
        self.num_users, self.num_items = dataset.n_users, dataset.m_items
        self.latent_dim, self.n_layers = args_r.recdim, args_r.layer
        self.keep_prob, self.A_split = args_r.keepprob, args_r.A_split
        self.embedding_user = nn.Embedding(self.num_users + 1, self.latent_dim)
        self.embedding_item = nn.Embedding(self.num_items, self.latent_dim)
        nn.init.xavier_uniform_(self.embedding_user.weight, gain=1)
        nn.init.xavier_uniform_(self.embedding_item.weight, gain=1)
        self._fuse_tables()
        self.f = nn.Sigmoid()
        self.Graph = dataset.getSparseGraph()
        self._graph_t, self._dropout_calls, self._injected_mask, self._cache = None, 0, None, None

        self.task_weights = nn.Parameter(torch.FloatTensor([0.0, 0.0]))
        self.rec_loss = nn.BCEWithLogitsLoss()
        self.loss_function = nn.CrossEntropyLoss()
        self.att_exp1 = nn.Parameter(torch.zeros(size=(2 * self.hidden_size, 2)))
        self.att_exp2 = nn.Parameter(torch.zeros(size=(2 * self.hidden_size, 2)))
        nn.init.xavier_uniform_(self.att_exp1.data, gain=1)
        nn.init.xavier_uniform_(self.att_exp2.data, gain=1)
        self.att_t = nn.Parameter(torch.zeros(size=(2 * self.hidden_size, 2)))
        nn.init.xavier_normal_(self.att_t.data, gain=1)

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hidden_size)
        for weight in self.parameters():
            weight.data.uniform_(-stdv, stdv)

    # ------------------------------------------------------------------ rec branch
    def _gated_tables(self):
        light_out = self._light_out()
        n_u = self.num_users + 1
        raw_u, raw_i = self.embedding_user.weight, self.embedding_item.weight
        out_u, out_i = light_out[:n_u], light_out[n_u:]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return (ops.expert_gate_autograd(raw_u, out_u, self.att_exp1), ops.expert_gate_autograd(raw_i, out_i, self.att_exp2))
        return (ops.expert_gate(raw_u.detach().contiguous(), out_u.contiguous(), self.att_exp1.detach()),
                ops.expert_gate(raw_i.detach().contiguous(), out_i.contiguous(), self.att_exp2.detach()))

    # ------------------------------------------------------------------ trust branch
    def compute_scores(self, hidden, inputs, mask, user_table=None):
        """user_table: the [n_users + 1, H] user embedding table the logits are taken against (default: this module's own
        parameter; the row-partitioned model passes the all-gathered block)."""
        table = self.embedding_user.weight if user_table is None else user_table
        B = mask.shape[0]
        last = torch.sum(mask, 1) - 1
        ht = _rows(hidden.reshape(-1, hidden.shape[2]), torch.arange(B, device=hidden.device) * hidden.shape[1] + last)
        q1 = self.linear_one(ht).view(B, 1, -1)
        q2 = self.linear_two(hidden)
        alpha = self.linear_three(torch.sigmoid(q1 + q2))
        a = torch.sum(alpha * hidden * mask.view(B, -1, 1).float(), 1)
        # the reference leaves p_a undefined under --nonhybrid (NameError at :141); use the pooled vector there
        p_a = a if self.nonhybrid else self.linear_transform(torch.cat([a, ht], 1))
        b = table[:-1]
        p_i = _rows(table, inputs) * mask.unsqueeze(2)
        p_maxpool = torch.max(p_i, dim=1)[0]
        att = torch.softmax(torch.cat([p_a, p_maxpool], 1) @ self.att_t, 1)
        a = p_a * att[:, 0].unsqueeze(1) + p_maxpool * att[:, 1].unsqueeze(1)
        return a @ b.t()

    def _trust_param_tensors(self):
        """The trust head's parameters in the order of the fused kernels' flat block (include/spex_hip.h, trust head)."""
        return ([att.a for att in self.in_att]
                + [self.out_att.a, self.w, self.linear_one.weight, self.linear_one.bias, self.linear_two.weight,
                   self.linear_two.bias, self.linear_three.weight, self.linear_transform.weight, self.linear_transform.bias,
                   self.att_t])

    def _trust_fused_ok(self, width):
        return ops.trust_head_supported(self.hidden_size, width, len(self.in_att)) and self.latent_dim == self.hidden_size

    def trust_loss(self, inputs, mask, targets, user_table=None):
        """flag 0's second output (:176-192): mean cross-entropy of the trust logits.  With hidden size 64 the whole branch,
        forward and backward, is four launches (ops.TrustHeadLoss); other shapes take the layer-by-layer path."""
        table = self.embedding_user.weight if user_table is None else user_table
        inputs, mask = np.asarray(inputs), np.asarray(mask)
        if not self._trust_fused_ok(inputs.shape[1]):
            scores = self._trust_scores(inputs, mask, user_table)
            return self.loss_function(scores, torch.as_tensor(np.asarray(targets), device=scores.device).long())
        params = torch.cat([t.reshape(-1) for t in self._trust_param_tensors()])
        return ops.TrustHeadLoss.apply(table, params, torch.from_numpy(inputs.astype(np.int64)),
                                       torch.from_numpy(mask.sum(1).astype(np.int64)),
                                       torch.from_numpy(np.asarray(targets).astype(np.int64)), len(self.in_att),
                                       not self.nonhybrid)

    def _trust_scores(self, inputs, mask, user_table=None):
        emb = self.embedding_user.weight if user_table is None else user_table
        dev = emb.device
        inputs = torch.as_tensor(np.asarray(inputs), device=dev).long()
        mask = torch.as_tensor(np.asarray(mask), device=dev).long()
        seq_l = torch.sum(mask, 1)
        heads = torch.stack([att.a.view(-1) for att in self.in_att])                           # [heads, 2H]
        mul_seq = ops.path_attention(emb, inputs, seq_l, heads, True)                          # [B, L, heads*H], one launch
        mul_one = F.elu(mul_seq.reshape(-1, mul_seq.shape[2]) @ self.w)
        hidden = self.out_att(emb, mul_one.view(mul_seq.shape[0], mul_seq.shape[1], self.hidden_size), seq_l)
        return self.compute_scores(hidden, inputs, mask, user_table)

    # ------------------------------------------------------------------ forward (:150-193)
    def forward(self, users, items, labels, slice_indices=None, trust_data=None, flag=0):
        loss1 = None
        if flag in (0, 1):
            all_users, all_items = self._gated_tables()
            dev = all_users.device
            if flag == 1:
                gamma, _ = ops.score_bce(all_users.detach().contiguous(), all_items.detach().contiguous(), users, items)
                return gamma
            users_emb, items_emb = _rows(all_users, ops._idx(users, dev)), _rows(all_items, ops._idx(items, dev))
            loss1 = self.rec_loss(torch.sum(users_emb * items_emb, dim=1), labels.to(dev).float())
        if flag in (0, 2):
            if flag == 0:
                inputs, mask, targets = trust_data.get_slice(slice_indices)
            else:
                inputs, mask, targets, negs = trust_data.get_slice(slice_indices)
            if flag == 2:
                scores = self._trust_scores(inputs, mask)
                return scores, torch.as_tensor(np.asarray(negs), device=scores.device).long()
            loss2 = self.trust_loss(inputs, mask, targets)
        return loss1, loss2
