"""Diffnet++ influence / interest diffusion on the HIP kernels.

Reference: Diffnet++_SPEX/code/utility/Model.py (`diffnetplus`, a TensorFlow keras Model) with its three sparsity
patterns built by utility/DataModule.py:242-399 — social neighbours [U,U], consumed items [U,I], item customers [I,U],
each in row-major sorted order.  TensorFlow is not part of this image, so this is a torch module with the same
attribute and method names, the same parameter set and the same arithmetic; it cannot be a drop-in for a keras Model
and its parity is unpinned (tests/test_gpu_diffnet.py holds it against an fp64 CPU restatement written from the
source text).

Per training step the reference evaluates (Model.py:194-290, 293-397)
  * six per-edge value vectors  exp(sigmoid(w p_e + b))  and their row softmax (`tf.sparse.softmax`)
        -> elementwise torch ops + `spex_edge_softmax_f32`
  * eight sparse x dense products with those learned values (`tf.sparse.sparse_dense_matmul`; Model.py:341-343 and
    381-384 evaluate the item-customer product twice, once is enough)
        -> `spex_graph_set_values` + `spex_spmm_f32`; backward: the transposed handle (same values through the edge
           ids) and `spex_sddmm_f32` for the gradient w.r.t. the values
  * the node-level attention MLPs ([rows, 2H] x [2H, 1] etc.) and the fusion arithmetic
        -> `spex_attn_fuse_f32` (+ backward): one launch per side and layer instead of ~40 small tensor ops.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .graph import SpexGraph, csr_transpose


def pairs_to_csr(rows, cols, n_rows, n_cols):
    """Sorted, de-duplicated row-major pattern with the reference's `avg` values 1/len(row)
    (DataModule.py:279-281, 345-347, 388-390).  Returns (rowptr, col, val)."""
    key = np.unique(np.asarray(rows, np.int64) * np.int64(n_cols) + np.asarray(cols, np.int64))
    r, c = key // n_cols, key % n_cols
    deg = np.bincount(r, minlength=n_rows)
    rowptr = np.zeros(n_rows + 1, np.int64)
    np.cumsum(deg, out=rowptr[1:])
    val = (1.0 / deg[r]).astype(np.float32)
    return rowptr.astype(np.int32), c.astype(np.int32), val


class LearnedGraph:
    """A sparsity pattern on the device: the handle, its transposed copy sharing one per-edge numbering, nnz."""

    def __init__(self, rowptr, col, val, n_cols, device="cuda"):
        self.host = (rowptr, col, val)
        self.n_rows, self.n_cols, self.nnz = len(rowptr) - 1, int(n_cols), len(col)
        self.g = SpexGraph(rowptr, col, val, n_cols=n_cols, device=device)
        t_rowptr, t_col, t_val, perm = csr_transpose(rowptr, col, val, n_cols)
        self.gt = SpexGraph(t_rowptr, t_col, t_val, n_cols=self.n_rows, edge_id=perm, device=device)

    def softmax(self, v):
        return ops.edge_softmax(v, self.g)

    def matmul(self, val, X):
        return ops.spmm_learned(val, X, self.g, self.gt)


class _Dense(nn.Module):
    """tf.keras.layers.Dense(units=1, activation=...): glorot-uniform kernel, zero bias."""

    def __init__(self, in_features, activation):
        super().__init__()
        self.kernel = nn.Parameter(torch.empty(in_features, 1))
        self.bias = nn.Parameter(torch.zeros(1))
        nn.init.xavier_uniform_(self.kernel)
        self.activation = activation

    def forward(self, x):
        # units == 1: a scale (in_features == 1, applied to a per-edge vector) or a row-wise dot product — not a GEMM
        # (hipBLASLt's N = 1, K = 1 GEMM takes ~1 ms on a 209 k-entry vector; the broadcast multiply takes ~5 us)
        if self.kernel.shape[0] == 1:
            return self.activation(x * self.kernel.view(1, 1) + self.bias)
        # (rocBLAS gemv on a [12 k, 128] operand: ~50 us; multiply + row-sum: ~10 us)
        return self.activation((x * self.kernel.view(1, -1)).sum(dim=1, keepdim=True) + self.bias)


def _leaky(x):
    return F.leaky_relu(x, 0.2)          # tf.nn.leaky_relu's default alpha


class DiffnetPlusPlus(nn.Module):
    """`diffnetplus` (Model.py:6).  social / consumed / customer: LearnedGraph of the three patterns."""

    def __init__(self, num_users, num_items, hidden_size, social, consumed, customer):
        super().__init__()
        H = hidden_size
        self.num_users, self.num_items, self.hidden_size = num_users, num_items, H
        self.social, self.consumed, self.customer = social, consumed, customer
        # initializeNodes, Model.py:85-192 (the *_layer2 low-level Denses and reduce_dimension_layer are created by
        # the reference but never used in call(); they are left out)
        self.user_embedding = nn.Parameter(torch.randn(num_users, H) * 0.01)
        self.item_embedding = nn.Parameter(torch.randn(num_items, H) * 0.01)
        mk = lambda n, act: _Dense(n, act)
        for lvl in ("first", "second"):
            setattr(self, f"{lvl}_user_part_social_graph_att_layer1", mk(2 * H, torch.tanh))
            setattr(self, f"{lvl}_user_part_social_graph_att_layer2", mk(1, _leaky))
            setattr(self, f"{lvl}_user_part_interest_graph_att_layer1", mk(2 * H, torch.tanh))
            setattr(self, f"{lvl}_user_part_interest_graph_att_layer2", mk(1, _leaky))
            setattr(self, f"{lvl}_item_part_itself_graph_att_layer1", mk(H, torch.tanh))
            setattr(self, f"{lvl}_item_part_itself_graph_att_layer2", mk(1, _leaky))
            setattr(self, f"{lvl}_item_part_user_graph_att_layer1", mk(H, torch.tanh))
            setattr(self, f"{lvl}_item_part_user_graph_att_layer2", mk(1, _leaky))
            setattr(self, f"{lvl}_low_att_layer_for_social_neighbors_layer1", mk(1, torch.sigmoid))
            setattr(self, f"{lvl}_low_att_layer_for_user_item_layer1", mk(1, torch.sigmoid))
            setattr(self, f"{lvl}_low_att_layer_for_item_user_layer1", mk(1, torch.sigmoid))
        # per-edge parameters, Model.py:181-186
        self.snii1 = nn.Parameter(torch.randn(social.nnz))
        self.snii2 = nn.Parameter(torch.randn(social.nnz))
        self.ciii1 = nn.Parameter(torch.randn(consumed.nnz))
        self.ciii2 = nn.Parameter(torch.randn(consumed.nnz))
        self.icii1 = nn.Parameter(torch.randn(customer.nnz))
        self.icii2 = nn.Parameter(torch.randn(customer.nnz) * 0.01)

    # Model.py:195-286: per-edge values and their row softmax
    def computer_somenode(self):
        def att(graph, dense, p):
            return graph.softmax(torch.exp(dense(p.view(-1, 1))).view(-1))    # (reduce_sum over a size-1 axis, Model.py:197-199)
        self.first_social_neighbors_low_level_att_matrix = att(
            self.social, self.first_low_att_layer_for_social_neighbors_layer1, self.snii1)
        self.second_social_neighbors_low_level_att_matrix = att(
            self.social, self.second_low_att_layer_for_social_neighbors_layer1, self.snii2)
        self.first_consumed_items_low_level_att_matrix = att(
            self.consumed, self.first_low_att_layer_for_user_item_layer1, self.ciii1)
        self.second_consumed_items_low_level_att_matrix = att(
            self.consumed, self.second_low_att_layer_for_user_item_layer1, self.ciii2)
        self.first_items_users_neighborslow_level_att_matrix = att(
            self.customer, self.first_low_att_layer_for_item_user_layer1, self.icii1)
        self.second_items_users_neighborslow_level_att_matrix = att(
            self.customer, self.second_low_att_layer_for_item_user_layer1, self.icii2)

    def _branch(self, lvl, name):
        """Parameter block [w1 | b1 | w2 | b2] of one attention branch (its two Dense(1) layers, Model.py:100-140)."""
        l1, l2 = getattr(self, f"{lvl}_{name}_layer1"), getattr(self, f"{lvl}_{name}_layer2")
        return torch.cat([l1.kernel.view(-1), l1.bias, l2.kernel.view(-1), l2.bias])

    def _layer(self, lvl, user_emb, item_emb, v_social, v_consumed, v_customer):
        """One influence + interest diffusion layer (Model.py:303-345 and 349-385): three SpMMs with learned values, then
        the node-level attention fusion of users and of items — one fused kernel each (spex_attn_fuse_f32)."""
        from_items = self.consumed.matmul(v_consumed, item_emb)
        from_social = self.social.matmul(v_social, user_emb)
        new_user = ops.attn_fuse(user_emb, from_items, from_social, self._branch(lvl, "user_part_interest_graph_att"),
                                 self._branch(lvl, "user_part_social_graph_att"), 0.7, 0.3, 0.5, 0.5)
        from_customers = self.customer.matmul(v_customer, user_emb)
        new_item = ops.attn_fuse(None, item_emb, from_customers, self._branch(lvl, "item_part_itself_graph_att"),
                                 self._branch(lvl, "item_part_user_graph_att"), 1.0, 1.0, 0.0, 1.0)
        return new_user, new_item

    def final_embeddings(self):
        self.computer_somenode()
        u0, i0 = self.user_embedding, self.item_embedding
        u1, i1 = self._layer("first", u0, i0, self.first_social_neighbors_low_level_att_matrix,
                             self.first_consumed_items_low_level_att_matrix,
                             self.first_items_users_neighborslow_level_att_matrix)
        u2, i2 = self._layer("second", u1, i1, self.second_social_neighbors_low_level_att_matrix,
                             self.second_consumed_items_low_level_att_matrix,
                             self.second_items_users_neighborslow_level_att_matrix)
        self.final_user_embedding = torch.cat([u1, u2, u0], 1)          # Model.py:388-391
        self.final_item_embedding = torch.cat([i1, i2, i0], 1)
        return self.final_user_embedding, self.final_item_embedding

    def forward(self, user_input, item_input, labels_input=None, flag=0):
        """call(), Model.py:293-397: flag 1 -> scores; flag 0 -> (scores, float labels)."""
        fu, fi = self.final_embeddings()
        dev = fu.device
        u = torch.as_tensor(np.asarray(user_input) if not torch.is_tensor(user_input) else user_input, device=dev).long().view(-1)
        i = torch.as_tensor(np.asarray(item_input) if not torch.is_tensor(item_input) else item_input, device=dev).long().view(-1)
        score = (fu[u] * fi[i]).sum(dim=1)
        if flag == 1:
            return score
        labels = torch.as_tensor(np.asarray(labels_input) if not torch.is_tensor(labels_input) else labels_input,
                                 device=dev).float().view(-1)
        return score, labels


def loss_fn(score, labels):
    """main_rec.py:34: mean sigmoid cross-entropy with logits."""
    return F.binary_cross_entropy_with_logits(score, labels)
