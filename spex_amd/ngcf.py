"""NGCF on libspexhip — the model the reference defines inside its driver (`Model_Wrapper`,
NGCF_SPEX/code/main_rec.py:36-113), with the same constructor arguments, parameter names and `forward` contract, so
the driver's training / evaluation loops (main_rec.py:116-148, utility/batch_test.py:27-32) work on it as they are:

    model = NGCF(data_config={'n_users':…, 'n_items':…, 'norm_adj': scipy_csr}, device, args)   # args: ngcf_parser
    loss = model(user, item, labels, flag=0);  ua, ia = model(None, None, None, flag=1)
    Model_Wrapper(data_config, device)            # the reference's two-argument form (flags from ngcf_parser / sys.argv)

Per layer the reference runs   side = A·ego (sparse.mm, and a host→device copy of A on EVERY call, :76);
sum = LeakyReLU(W_gc side + b); bi = LeakyReLU(W_bi (ego ⊙ side) + b); ego' = dropout(sum + bi);
all ‖= normalize(ego').   Here A (= D⁻¹(A+I), not symmetric) and Aᵀ live in HBM once and a layer is two launches in
each direction: the SpMM and one fused layer kernel (forward: both 64×64 products on the matrix cores, bias,
LeakyReLU, message dropout, L2 normalisation, concat write; backward: the layer recomputed, both input-gradient
products, both weight-gradient products — `spex_ngcf_layer_fwd_f32` / `_bwd_f32`), scoring + BCE + its gradient rows
one launch.  Message dropout is counter-based (seed, step, layer): `message_dropout_seed` / `dropout_step` name the
stream, nothing is stored between forward and backward.

Storage: both embedding tables sit back to back in one [n_users + 1 + n_items, d] buffer (the reference's unused pad
user row, :67, included), and the graph handle has that many nodes with the pad row isolated — the kernels read the
parameters in place, where the reference concatenates `user_w[:-1]` and `item_w` every step (:72).
Layer widths other than 64 take the same HIP SpMM with the dense layer expressed in torch ops on the device.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import SpexGraph, csr_transpose


class SpMM(torch.autograd.Function):
    """y = A x with A, Aᵀ resident as SpexGraph handles; backward is Aᵀ g (what torch.sparse.mm's autograd does)."""

    @staticmethod
    def forward(ctx, x, graph, graph_t):
        ctx.graph_t = graph_t
        return graph.spmm(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ctx.graph_t.spmm(g.contiguous()), None, None


def insert_isolated_node(rowptr, col, val, at):
    """CSR of the matrix with one empty row and column inserted at index `at` (the pad user row between the user block
    and the item block)."""
    rowptr, col = np.asarray(rowptr, np.int64), np.asarray(col, np.int64)
    new_rowptr = np.concatenate([rowptr[:at + 1], rowptr[at:]])
    return new_rowptr.astype(np.int32), (col + (col >= at)).astype(np.int32), np.asarray(val, np.float32)


class NGCF(nn.Module):
    def __init__(self, data_config, device, args):
        super().__init__()
        self.device = torch.device(device)
        self.n_users, self.n_items = data_config["n_users"], data_config["n_items"]
        self.embedding_dim = args.embed_size
        self.weight_size = [self.embedding_dim] + list(eval(args.layer_size))
        self.n_layers = len(self.weight_size) - 1
        self.mess_dropout = list(eval(args.mess_dropout))
        self.decay = eval(args.regs)[0]
        # same sub-module names / creation order as main_rec.py:54-66 => same parameters for the same torch seed
        self.dropout_list, self.GC_Linear_list, self.Bi_Linear_list = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(self.n_layers):
            self.GC_Linear_list.append(nn.Linear(self.weight_size[i], self.weight_size[i + 1]))
            self.Bi_Linear_list.append(nn.Linear(self.weight_size[i], self.weight_size[i + 1]))
            self.dropout_list.append(nn.Dropout(self.mess_dropout[i]))
        self.user_embedding = nn.Embedding(self.n_users + 1, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        self.item_embedding = nn.Embedding(self.n_items, self.embedding_dim)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self._fuse_tables()
        self.rec_loss_function = nn.BCEWithLogitsLoss()
        self.message_dropout_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.dropout_step = 0          # counts training forwards: names the dropout mask of the step
        # "reference": replay nn.Dropout's own noise draw instead (see _reference_noise); SPEX_DROPOUT_STREAM sets the default, as
        # for the LightGCN drop-in's edge dropout
        self.dropout_stream = "reference" if os.environ.get("SPEX_DROPOUT_STREAM") == "reference" else "counter"

        adj = data_config["norm_adj"].tocsr().astype(np.float32)
        adj.sort_indices()
        n = self.n_users + self.n_items
        if adj.shape != (n, n):
            raise ValueError(f"norm_adj is {adj.shape}, expected {(n, n)}")
        rowptr, col, val = insert_isolated_node(adj.indptr, adj.indices, adj.data, self.n_users)
        self.graph = SpexGraph(rowptr, col, val, n_cols=n + 1, device=self.device)
        t_rowptr, t_col, t_val, _ = csr_transpose(rowptr, col, val, n + 1)
        self.graph_t = SpexGraph(t_rowptr, t_col, t_val, n_cols=n + 1, device=self.device)

    # ------------------------------------------------------------------ parameter storage
    def _fuse_tables(self):
        """Both tables in one contiguous [n_users + 1 + n_items, d] buffer (users first): the kernels' ego table."""
        u, i = self.user_embedding.weight, self.item_embedding.weight
        flat = torch.cat([u.data, i.data])
        u.data = flat[: u.shape[0]]
        i.data = flat[u.shape[0]:]

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._fuse_tables()
        return out

    def flat_table(self):
        return ops._flat_tables(self.user_embedding.weight, self.item_embedding.weight, strict=True)

    def _fused_ok(self):
        return all(w == 64 for w in self.weight_size)

    def _layer_weights(self):
        ws = []
        for gc, bi in zip(self.GC_Linear_list, self.Bi_Linear_list):
            ws += [gc.weight, gc.bias, bi.weight, bi.bias]
        return ws

    # ------------------------------------------------------------------ propagation (main_rec.py:71-86)
    def _all_embeddings(self):
        """The concatenated table over n_users + 1 + n_items rows (the pad row's values are never read)."""
        uw, iw = self.user_embedding.weight, self.item_embedding.weight
        if not uw.is_cuda:
            raise RuntimeError("spex_amd NGCF runs on the GPU only: call .to('cuda') first (no CPU fallback)")
        drop = None
        if self.training and any(p > 0 for p in self.mess_dropout):
            drop = (tuple(self.mess_dropout), self.message_dropout_seed, self.dropout_step, self._reference_noise())
            self.dropout_step += 1
        if self._fused_ok():
            return ops.NGCFPropagate.apply(uw, iw, self.graph, self.graph_t, drop, self.n_users, *self._layer_weights())
        # other widths: HIP SpMM in both directions, the dense layer through torch ops on the device
        ego = ops._flat_tables(uw, iw) if not (uw.requires_grad or iw.requires_grad) else torch.cat((uw, iw), dim=0)
        parts = [ego]
        for i in range(self.n_layers):
            side = SpMM.apply(ego, self.graph, self.graph_t)
            ego = F.leaky_relu(self.GC_Linear_list[i](side)) + F.leaky_relu(self.Bi_Linear_list[i](ego * side))
            ego = self.dropout_list[i](ego)
            parts.append(F.normalize(ego, p=2, dim=1))
        return torch.cat(parts, dim=1)

    def _reference_noise(self):
        """dropout_stream == "reference" (validation mode): the step's message-dropout noise is the reference's own — nn.Dropout on
        each layer's [N, 64] output (main_rec.py:81) is at::dropout's empty_like(x).bernoulli_(1 - p) from torch's global CPU
        generator, one draw per layer in layer order.  The same calls on reused pinned buffers consume the generator identically;
        the keep bytes go up and the layer kernels (forward and the recomputing backward) read them through
        spex_ngcf_message_mask.  Returns one uint8 [N, 64] tensor per layer (None where p == 0), or None for the counter stream."""
        if getattr(self, "dropout_stream", "counter") != "reference":
            return None
        n_ref, dev = self.n_users + self.n_items, self.user_embedding.weight.device
        bufs = self.__dict__.setdefault("_noise_host", {})
        masks = []
        for l, p in enumerate(self.mess_dropout[: self.n_layers]):
            if p <= 0:
                masks.append(None)
                continue
            if l not in bufs:
                bufs[l] = torch.empty((n_ref, self.weight_size[l + 1]), dtype=torch.float32).pin_memory()
            bufs[l].bernoulli_(1.0 - float(p))
            masks.append((bufs[l].to(dev) != 0).to(torch.uint8))
        return masks

    def forward(self, user, item, labels_list, flag):
        all_emb = self._all_embeddings()
        n_u = self.n_users + 1
        if flag == 1:
            return all_emb[: self.n_users], all_emb[n_u:]
        dev = all_emb.device
        return ops.ScoreBCELoss.apply(all_emb, n_u, ops._idx(user, dev).reshape(-1), ops._idx(item, dev).reshape(-1),
                                      labels_list.to(device=dev, dtype=torch.float32).reshape(-1))

    def compute_rec_loss(self, u_g_embeddings, i_g_embeddings, labels_list):
        predict = torch.sum(torch.mul(u_g_embeddings, i_g_embeddings), dim=1)
        return self.rec_loss_function(predict, labels_list.float())


def Model_Wrapper(data_config, device):
    """The reference's constructor form (main_rec.py:169): flags come from ngcf_parser, as its module-global `args` do."""
    from .dropin.ngcf.ngcf_parser import parse_known
    return NGCF(data_config, device, parse_known())
