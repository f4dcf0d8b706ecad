"""LightGCN behind the reference's `utility1.model` surface, running on libspexhip (MI355X).

Public surface = LightGCN_SPEX/code/utility1/model.py:18-121: `LightGCN(args_r, dataset)` is an nn.Module with
`embedding_user` / `embedding_item` (nn.Embedding, xavier-uniform), `Graph`, `computer()` and
`forward(users, items, labels, flag)`; `loss.backward()` and any torch optimiser work on it unchanged
(main_rec.py:30-37).  `bpr_loss()` is the north-star extension (upstream LightGCN-PyTorch semantics; the reference
has no BPR).

How it runs instead of `torch.sparse.mm` in a Python loop:
  * both tables live back-to-back in one [N, d] HBM buffer, so the propagation reads them without the per-step
    `torch.cat` (model.py:72);
  * `computer()` is one autograd Function: L SpMM launches with the layer mean fused into their epilogue (no
    stack/mean pass, model.py:94-95); its backward is L SpMM launches with the g/(L+1) term fused;
  * in eval mode the propagated tables are cached until a parameter changes, so `test()`'s one-forward-per-user
    loop (batch_test.py:28-33) costs one propagation per epoch instead of 3 185;
  * scoring + BCE + its gradient rows are one kernel.
"""
import os

import torch
from torch import nn

from spex_amd import ops
from spex_amd.graph import SpexGraph, csr_transpose


class BasicModel(nn.Module):
    def getUsersRating(self, users):
        raise NotImplementedError


class LightGCN(BasicModel):
    def __init__(self, args_r, dataset):
        super().__init__()
        self.args_r = args_r
        self.dataset = dataset
        self.bcel = nn.BCEWithLogitsLoss()  # kept for API parity; the fused kernel computes the same loss
        self.num_users = dataset.n_users
        self.num_items = dataset.m_items
        self.latent_dim = args_r.recdim
        self.n_layers = args_r.layer
        self.keep_prob = args_r.keepprob
        self.A_split = args_r.A_split

        # Same construction order and initialiser as model.py:32-35 => same values for the same torch seed.
        self.embedding_user = nn.Embedding(self.num_users + 1, self.latent_dim)
        self.embedding_item = nn.Embedding(self.num_items, self.latent_dim)
        nn.init.xavier_uniform_(self.embedding_user.weight, gain=1)
        nn.init.xavier_uniform_(self.embedding_item.weight, gain=1)
        self._fuse_tables()

        self.f = nn.Sigmoid()
        self.Graph = dataset.getSparseGraph()
        self._graph_t = None          # A^T with the edge-id permutation, built on first use under dropout
        self._dropout_calls = 0
        self._injected_mask = None    # test hook: (uint8 device tensor) replaces the sampled mask
        # Which random stream picks the dropped edges (only read when --dropout 1):
        #   "philox"    (default, the fast path) a counter-based mask regenerated inside the SpMM kernel: same law as the
        #               reference's, a different stream;
        #   "reference" the reference's OWN stream: `torch.rand(nnz)` drawn on the CPU from the global generator once per
        #               training `computer()` call, exactly where model.py:50 draws it, uploaded as a keep mask (418 KB per step
        #               on Epinion2) — the validation mode: with the same seed a run drops the same edges as main_rec.py
        #               --dropout 1 and reproduces its losses step by step (tests: G12-dropout goldens).
        self.dropout_stream = os.environ.get("SPEX_DROPOUT_STREAM", "philox")
        self._cache = None            # (version_u, version_i, light_out) for eval mode

    # ------------------------------------------------------------------ parameter storage
    def _fuse_tables(self):
        """Place both tables in one contiguous [N, d] buffer (users first): the kernels' E0."""
        u, i = self.embedding_user.weight, self.embedding_item.weight
        flat = torch.cat([u.data, i.data])
        u.data = flat[: u.shape[0]]
        i.data = flat[u.shape[0]:]

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)   # .to(device) moves each table separately ...
        self._fuse_tables()                 # ... so re-join them
        self._cache = None
        return out

    def flat_table(self):
        """The two embedding tables as ONE [N, d] tensor sharing the parameters' storage (users first): what
        spex_amd.trainer.LightGCNStepper trains in place, so the module's own weights are the trained ones."""
        return ops._flat_tables(self.embedding_user.weight, self.embedding_item.weight, strict=True)

    # ------------------------------------------------------------------ dropout (model.py:46-64)
    def set_edge_mask(self, keep):
        """Test hook: inject a keep mask (bool/uint8 per stored entry, reference entry order) instead of sampling."""
        self._injected_mask = None if keep is None else keep.to(device=self.embedding_user.weight.device,
                                                                dtype=torch.uint8).contiguous()

    def _transposed(self):
        if self._graph_t is None:
            rowptr, col, val = self.Graph.host
            t_rowptr, t_col, t_val, eid = csr_transpose(rowptr, col, val, self.Graph.n_cols)
            self._graph_t = SpexGraph(t_rowptr, t_col, t_val, n_cols=self.Graph.n_rows, edge_id=eid,
                                      device=self.embedding_user.weight.device)
        return self._graph_t

    def _mask_for_step(self):
        if not (self.args_r.dropout and self.training):
            return None
        if self._injected_mask is not None:
            return (1, self._injected_mask, float(self.keep_prob), 0)
        if self.dropout_stream == "reference":
            return (1, self._reference_stream_mask(), float(self.keep_prob), 0)
        if self.dropout_stream != "philox":
            raise ValueError(f"dropout_stream must be 'philox' or 'reference' (got {self.dropout_stream!r})")
        self._dropout_calls += 1
        seed = (int(getattr(self.args_r, "seed", 0)) << 32) | (self._dropout_calls & 0xFFFFFFFF)
        return (2, None, float(self.keep_prob), seed)

    def _reference_stream_mask(self):
        """The keep mask model.py:46-55 would draw for this step: `torch.rand(len(values)) + keep_prob`, `.int().bool()`, from
        the global CPU generator — one draw per fold under --A_split (model.py:57-64 calls __dropout_x fold by fold), in the
        stored-entry order of the coalesced adjacency (= the handles' edge ids)."""
        from spex_amd.trainer import reference_keep_mask
        graphs = self.Graph if isinstance(self.Graph, (list, tuple)) else [self.Graph]
        return reference_keep_mask([g.nnz for g in graphs], self.keep_prob, self.embedding_user.weight.device)

    # ------------------------------------------------------------------ propagation (model.py:66-97)
    def _light_out(self):
        uw, iw = self.embedding_user.weight, self.embedding_item.weight
        if not uw.is_cuda:
            raise RuntimeError("spex_amd LightGCN runs on the GPU only: call .to('cuda') first (no CPU fallback)")
        use_cache = not self.training and not torch.is_grad_enabled()
        if use_cache and self._cache is not None and self._cache[0] == (uw._version, iw._version, uw.data_ptr()):
            return self._cache[1]
        if isinstance(self.Graph, (list, tuple)):
            out = self._light_out_folds(uw, iw)
        else:
            mask = self._mask_for_step()
            graph_t = self._transposed() if mask is not None else self.Graph  # A_hat is symmetric without dropout
            out = ops.PropagateMean.apply(uw, iw, self.Graph, graph_t, self.n_layers, mask)
        if use_cache:
            self._cache = ((uw._version, iw._version, uw.data_ptr()), out)
        return out

    def _transposed_folds(self):
        """Row blocks of A^T (same block boundaries as the forward folds), each entry carrying its edge id in A."""
        if getattr(self, "_folds_t", None) is None:
            rowptr, col, val = self.dataset.build_adjacency()
            n = len(rowptr) - 1
            t_rowptr, t_col, t_val, eid = csr_transpose(rowptr, col, val, n)
            bounds = [0]
            for g in self.Graph:
                bounds.append(bounds[-1] + g.n_rows)
            from spex_amd.graph import row_block
            dev = self.embedding_user.weight.device
            self._folds_t = []
            for k in range(len(self.Graph)):
                r, c, v, e = row_block(t_rowptr, t_col, t_val, bounds[k], bounds[k + 1], eid)
                self._folds_t.append(SpexGraph(r, c, v, n_cols=n, edge_id=e, device=dev))
        return self._folds_t

    def _light_out_folds(self, uw, iw):
        """--A_split: the adjacency as row blocks, one SpMM per block per layer (model.py:84-89), training included: the
        backward pass runs on the row blocks of A^T."""
        mask = self._mask_for_step()
        need_grad = torch.is_grad_enabled() and (uw.requires_grad or iw.requires_grad)
        folds_t = self._transposed_folds() if need_grad else None
        return ops.PropagateMeanFolds.apply(uw, iw, self.Graph, folds_t, self.n_layers, mask)

    def computer(self):
        light_out = self._light_out()
        return torch.split(light_out, [self.num_users + 1, self.num_items])

    # ------------------------------------------------------------------ scoring (model.py:111-121)
    def forward(self, users, items, labels, flag=0):
        if flag not in (0, 1):
            raise ValueError("flag must be 0 (loss) or 1 (scores)")
        uw, iw = self.embedding_user.weight, self.embedding_item.weight
        if (flag == 0 and uw.is_cuda and torch.is_grad_enabled() and (uw.requires_grad or iw.requires_grad)
                and not isinstance(self.Graph, (list, tuple))):
            # training step: propagation + scoring as one autograd node
            dev = uw.device
            mask = self._mask_for_step()
            graph_t = self._transposed() if mask is not None else self.Graph
            return ops.LightGCNBCELoss.apply(uw, iw, self.Graph, graph_t, self.n_layers, mask, ops._idx(users, dev),
                                             ops._idx(items, dev), labels.to(device=dev, dtype=torch.float32))
        light_out = self._light_out()
        n_u = self.num_users + 1
        if flag == 1:
            gamma, _ = ops.score_bce(light_out[:n_u].detach(), light_out[n_u:].detach(), users, items)
            return gamma
        dev = light_out.device
        return ops.ScoreBCELoss.apply(light_out, n_u, ops._idx(users, dev), ops._idx(items, dev),
                                      labels.to(device=dev, dtype=torch.float32))

    def getUsersRating(self, users):
        all_users, all_items = self.computer()
        return self.f(all_users[users.long()] @ all_items.t())

    # ------------------------------------------------------------------ north-star extension
    def bpr_loss(self, users, pos, neg):
        """(mean softplus(<u,neg> - <u,pos>), 0.5 * (|u0|^2 + |p0|^2 + |n0|^2) / B) on propagated / raw tables."""
        light_out = self._light_out()
        dev = light_out.device
        users, pos, neg = ops._idx(users, dev), ops._idx(pos, dev), ops._idx(neg, dev)
        loss = ops.BPRLoss.apply(light_out, self.num_users + 1, users, pos, neg)
        u0, p0, n0 = self.embedding_user(users), self.embedding_item(pos), self.embedding_item(neg)
        reg = 0.5 * (u0.pow(2).sum() + p0.pow(2).sum() + n0.pow(2).sum()) / float(users.numel())
        return loss, reg
