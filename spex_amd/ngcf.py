"""NGCF on libspexhip — the model the reference defines inside its driver (`Model_Wrapper`,
NGCF_SPEX/code/main_rec.py:36-113), with the same constructor arguments, parameter names and `forward` contract, so
the driver's training / evaluation loops (main_rec.py:116-148, utility/batch_test.py:27-32) work on it as they are:

    model = NGCF(data_config={'n_users':…, 'n_items':…, 'norm_adj': scipy_csr}, device, args)   # args: ngcf_parser
    loss = model(user, item, labels, flag=0);  ua, ia = model(None, None, None, flag=1)

Per layer the reference runs   side = A·ego (sparse.mm, and a host→device copy of A on EVERY call, :76);
sum = LeakyReLU(W_gc side + b); bi = LeakyReLU(W_bi (ego ⊙ side) + b); ego' = dropout(sum + bi);
all ‖= normalize(ego').   Here A (= D⁻¹(A+I), not symmetric) and Aᵀ live in HBM once; the SpMM is the HIP kernel in
both directions (autograd Function below); in inference the whole dense epilogue is the fused `spex_ngcf_layer_f32`
kernel; in training the epilogue is expressed with torch ops so autograd differentiates it (two 64×64 GEMMs per
layer — 128 MFLOP on Epinion2 — are not the bottleneck; the SpMM is).
"""
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
        self.rec_loss_function = nn.BCEWithLogitsLoss()

        adj = data_config["norm_adj"].tocsr().astype(np.float32)
        adj.sort_indices()
        rowptr, col, val = adj.indptr.astype(np.int32), adj.indices.astype(np.int32), adj.data.astype(np.float32)
        self.graph = SpexGraph(rowptr, col, val, n_cols=adj.shape[1], device=self.device)
        t_rowptr, t_col, t_val, _ = csr_transpose(rowptr, col, val, adj.shape[1])
        self.graph_t = SpexGraph(t_rowptr, t_col, t_val, n_cols=adj.shape[0], device=self.device)

    def _propagate(self):
        ego = torch.cat((self.user_embedding.weight[:-1], self.item_embedding.weight), dim=0)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        parts = [ego]
        for i in range(self.n_layers):
            gc, bi = self.GC_Linear_list[i], self.Bi_Linear_list[i]
            fused_ok = (not need_grad and not (self.training and self.mess_dropout[i] > 0)
                        and ego.shape[1] == 64 and gc.out_features == 64)
            if fused_ok:  # inference: SpMM + one fused epilogue kernel
                side = self.graph.spmm(ego.contiguous())
                out, e1 = ops.ngcf_layer(ego.contiguous(), side, gc.weight, gc.bias, bi.weight, bi.bias, want_e1=True)
                parts.append(out[:, 64:])
                ego = e1
            else:         # training: HIP SpMM in both directions, dense epilogue through autograd
                side = SpMM.apply(ego, self.graph, self.graph_t)
                ego = F.leaky_relu(gc(side)) + F.leaky_relu(bi(ego * side))
                ego = self.dropout_list[i](ego)
                parts.append(F.normalize(ego, p=2, dim=1))
        all_emb = torch.cat(parts, dim=1)
        return torch.split(all_emb, [self.n_users, self.n_items], dim=0)

    def forward(self, user, item, labels_list, flag):
        ua, ia = self._propagate()
        if flag == 1:
            return ua, ia
        dev = ua.device
        # index_select: its backward is an atomic index_add_ (advanced indexing's sorts the indices: ~120 us per call)
        u_g, i_g = ua.index_select(0, user.to(dev).reshape(-1)), ia.index_select(0, item.to(dev).reshape(-1))
        return self.compute_rec_loss(u_g, i_g, labels_list.to(dev))

    def compute_rec_loss(self, u_g_embeddings, i_g_embeddings, labels_list):
        predict = torch.sum(torch.mul(u_g_embeddings, i_g_embeddings), dim=1)
        return self.rec_loss_function(predict, labels_list.float())
