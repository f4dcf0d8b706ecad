"""The dual-task model (recommendation + trust-path prediction, LightGCN_SPEX/code/main_auto_expert_s.py with
utility1/model_expert_s.py) on a 1-D row-partitioned graph — BASELINE config 5's multi-GPU form (last row).

What is partitioned and what is replicated
  * the embedding table E0 (users incl. the pad row, then items) and the adjacency: row blocks, one per rank
    (spex_amd.dist.PartitionedLightGCN: all-gather of a layer's rows -> local SpMM);
  * the two-expert gate (model_expert_s.py:154-161) runs on the rank's LOCAL rows of (E0, propagated table): it is a
    per-row operation; the rank's rows below the user/item boundary use att_exp1, the others att_exp2;
  * the rec batch is replicated: the batch's rows of the gated table are exchanged owner-computes (one launch + one
    small all-reduce), every rank evaluates the same BCE loss and keeps the gradient rows it owns;
  * the trust head needs user rows by path index and the whole user table for its logits (`a . user_w^T`,
    model_expert_s.py:147): it reuses the all-gathered E0 of the FIRST propagation layer (no extra collective), is
    evaluated redundantly on every rank (a step's <= 15 paths x 6 nodes), and each rank keeps the rows it owns of the
    user-table gradient — identical on all ranks by construction, so nothing is reduced;
  * the ~40 small dense parameters are replicated; their gradients are identical on every rank except those of the two
    gate matrices, whose local-row contributions are summed with one 1 KB all-reduce.
Collectives per step: L all-gathers forward, L backward, one all-reduce of the batch's rows, one of the gate gradients.

Every rank must draw the same batch and the same paths (same seeds), as with any replicated-batch scheme.
"""
import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import ops as _default_ops
from .dist import PartitionedLightGCN


class _PropagateAndUserBlock(torch.autograd.Function):
    """(E0_local) -> (propagated local rows, all-gathered user block).  Backward: the partitioned propagation backward
    for the first output; of the second output's gradient — the same on every rank — the rank keeps its own rows."""

    @staticmethod
    def forward(ctx, E0_local, model):
        P = model.P
        light = P.propagate(E0_local, keep_first=True).clone()
        users = P.gathered0.index_select(0, model.user_pos)
        ctx.model = model
        return light, users

    @staticmethod
    def backward(ctx, g_light, g_users):
        m = ctx.model
        P = m.P
        g = P.propagate_bwd(g_light.contiguous(), grad_out=torch.empty_like(g_light))
        if g_users is not None and m.n_local_users > 0:
            g[: m.n_local_users] += g_users[P.r0: P.r0 + m.n_local_users]
        return g, None


class _FetchRows(torch.autograd.Function):
    """rows of a row-partitioned table at replicated positions (owner-computes gather + all-reduce); backward scatters the
    (replicated) gradient rows to their owners."""

    @staticmethod
    def forward(ctx, table_local, pos, P):
        ctx.P, ctx.n_local = P, table_local.shape[0]
        ctx.save_for_backward(pos)
        out = torch.empty((pos.numel(), table_local.shape[1]), dtype=table_local.dtype, device=table_local.device)
        return P.fetch_rows_at(pos, out, table=table_local.contiguous())

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        grad = torch.zeros((ctx.n_local, g.shape[1]), dtype=g.dtype, device=g.device)
        ctx.P.add_owned_rows(g.contiguous().clone(), pos, grad, clear=False)
        return grad, None, None


class PartitionedDualTask(nn.Module):
    """One rank of the row-partitioned dual-task model.

    core: a `utility1.model_expert_s.LightGCN` built from the same seed on every rank (its dense parameters are used as
    they are — same names, same initial values; its two embedding tables only supply this rank's rows and are released).
    csr: host CSR of the normalised adjacency (Loader.build_adjacency()).
    graph_factory / kernels: injection points of the CPU schedule tests (defaults: SpexGraph, spex_amd.ops).
    """

    def __init__(self, core, csr, rank, world, device, group=None, graph_factory=None, kernels=None):
        super().__init__()
        self.core = core
        self.k = kernels if kernels is not None else _default_ops
        self.rank, self.world, self.group = rank, world, group
        dev = torch.device(device)
        n_u = core.num_users + 1
        if graph_factory is None:
            from .graph import SpexGraph
            graph_factory = lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev)
        self.P = PartitionedLightGCN(*csr, n_u, core.n_layers, core.latent_dim, rank, world, graph_factory, dev, group=group)
        P = self.P
        full = torch.cat([core.embedding_user.weight.detach(), core.embedding_item.weight.detach()])
        self.E0_local = nn.Parameter(full[P.r0:P.r1].clone().to(dev))
        core.embedding_user.weight.requires_grad_(False)
        core.embedding_item.weight.requires_grad_(False)
        self.n_user_rows = n_u
        self.n_local_users = int(min(max(n_u - P.r0, 0), P.n_local))        # this rank's rows that are user rows
        self.user_pos = torch.from_numpy(P.part.to_padded(np.arange(n_u))).to(dev)
        self.task_weights = core.task_weights

    def trained_parameters(self):
        """What the optimiser steps: the rank's table rows + the replicated dense parameters."""
        skip = {id(self.core.embedding_user.weight), id(self.core.embedding_item.weight)}
        return [self.E0_local] + [p for p in self.core.parameters() if id(p) not in skip]

    def _gated_local(self, light):
        c, ku = self.core, self.n_local_users
        raw = self.E0_local
        parts = []
        if ku > 0:
            parts.append(self.k.expert_gate_autograd(raw[:ku], light[:ku], c.att_exp1))
        if ku < raw.shape[0]:
            parts.append(self.k.expert_gate_autograd(raw[ku:], light[ku:], c.att_exp2))
        return parts[0] if len(parts) == 1 else torch.cat(parts)

    def forward(self, users, items, labels, slice_indices, trust_data):
        """flag=0 of model_expert_s.LightGCN.forward: (rec BCE loss, trust cross-entropy loss), the same values on every
        rank."""
        c, P = self.core, self.P
        dev = self.E0_local.device
        light, user_table = _PropagateAndUserBlock.apply(self.E0_local, self)
        mixed = self._gated_local(light)
        users, items = users.to(dev).long(), items.to(dev).long()
        pu, pi = P.padded_index(users, items)
        rows = _FetchRows.apply(mixed, torch.cat([pu, pi]), P)
        B = users.numel()
        ar = torch.arange(B, device=dev)
        loss1 = self.k.ScoreBCELoss.apply(rows, B, ar, ar, labels.to(device=dev, dtype=torch.float32))
        inputs, mask, targets = trust_data.get_slice(slice_indices)
        loss2 = c.trust_loss(inputs, mask, targets, user_table=user_table)
        return loss1, loss2

    def reduce_gate_gradients(self):
        """Call between backward() and optimizer.step(): the gate matrices saw only this rank's rows."""
        if self.world > 1:
            for p in (self.core.att_exp1, self.core.att_exp2):
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                dist.all_reduce(p.grad, group=self.group)
