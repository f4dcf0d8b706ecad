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

    def __init__(self, core, csr, rank, world, device, group=None, graph_factory=None, kernels=None, edge_ids=False):
        super().__init__()
        self.core = core
        self.k = kernels if kernels is not None else _default_ops
        self.rank, self.world, self.group = rank, world, group
        dev = torch.device(device)
        n_u = core.num_users + 1
        if graph_factory is None:
            from .graph import SpexGraph
            graph_factory = lambda r, c, v, n_cols, edge_id=None: SpexGraph(r, c, v, n_cols=n_cols, edge_id=edge_id, device=dev)
        # edge_ids: blocks with global edge ids, for edge dropout on the rec branch (model_expert_s.py:104-109; P.set_edge_mask per step)
        self.P = PartitionedLightGCN(*csr[:3], n_u, core.n_layers, core.latent_dim, rank, world, graph_factory, dev, group=group,
                                     edge_ids=edge_ids)
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


class PartitionedDualTaskStepper:
    """The dual-task training step on the row partition as ONE native call (spex_partitioned_dual_task_step_f32) — what
    DualTaskStepper is to the single-GPU model: no autograd, no allocation, no host work between the launches; the exchanges go
    through the model's NativeComm (spex_comm_*, RCCL bound inside the library).  main_auto_expert_s.py:53-91.

    model: a PartitionedDualTask (every rank built from the same seed).  Its parameters are re-homed into the rank's flat arena
    [n_local table rows | trust block | att_exp1 | att_exp2 | task_weights] and trained in place: `model.E0_local` and the core's dense
    parameters stay valid views, so evaluation / checkpointing read the trained values.  Every rank must pass the same batch and
    the same paths.  The gate runs on the batch's 2B rows (replicated), so no gate-gradient collective is needed: per step the
    wire carries 2 L exchanges and one all-reduce of 4B rows.
    exchange: "native" (equal padded shards, ncclAllGather) or "native-p2p" (real rows, grouped send / recv)."""

    def __init__(self, model, path_capacity, path_len, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, n_rec=5, batch_capacity=256,
                 exchange="native-p2p", two_streams=True, deterministic=False, fixed_task_weights=False, comm=None, fast=True):
        from . import _lib, ops
        core, P = model.core, model.P
        dev = model.E0_local.device
        assert dev.type == "cuda", "PartitionedDualTaskStepper: GPU only (no CPU fallback)"
        d = P.d
        n_heads = len(core.in_att)
        if d != 64 or core.hidden_size != d or not ops.trust_head_supported(d, path_len, n_heads):
            raise ValueError(f"PartitionedDualTaskStepper needs hidden size 64, <= 16 path positions, <= 4 heads (got {d}, {path_len}, {n_heads})")
        self.model, self.P, self.dev, self.d, self.L = model, P, dev, d, P.L
        self.n_u = model.n_user_rows
        self.path_capacity, self.path_len, self.n_heads = int(path_capacity), int(path_len), n_heads
        self.lr, self.betas, self.eps, self.n_rec = lr, betas, eps, n_rec
        self.deterministic, self.fixed_task_weights = bool(deterministic), bool(fixed_task_weights)
        self.fast = bool(fast)          # False: the launch-by-launch schedule (2 L exchanges, nine batch-sized launches) — tests
        if comm is not None:
            P.native = comm
        P.set_allgather(exchange)
        self._side = torch.cuda.Stream(device=dev) if two_streams else None
        n_loc = P.n_local
        Pn = ops.trust_param_count(n_heads, d)
        self.n_trust = Pn
        total = n_loc * d + Pn + 512 + 4
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.arena, self.m, self.v = z(total), z(total), z(total)
        off = 0

        def place(param, n):
            nonlocal off
            view = self.arena[off: off + n].view(param.shape)
            view.copy_(param.data)
            param.data = view
            off += n

        place(model.E0_local, n_loc * d)
        for t in core._trust_param_tensors():
            place(t, t.numel())
        assert off == n_loc * d + Pn
        place(core.att_exp1, 256); place(core.att_exp2, 256); place(core.task_weights, 2)
        self.light, self.g_prop, self.g_raw, self.gs, self.g_E0 = (z(n_loc, d) for _ in range(5))
        self.gathered0 = z(P.part.n_padded, d)
        self.user_table = z(self.n_u, d)
        self.g_user, self.g_small = z(self.n_u, d), z(Pn + 512)
        T = self.path_capacity
        self.a2 = z(T, d)
        self.trust_ws = z(max(1, int(_lib.load().spex_trust_workspace_floats(T, self.path_len, d, n_heads, self.n_u))))
        self.dscore, self.loss_b = z(T, self.n_u - 1), z(T)
        self.loss, self.loss_acc, self.precision = z(2), z(2), z(2, 2)
        self.t = 0
        self.slot_capacity = 0
        self._desc = None
        self._slots(2 * int(batch_capacity))
        self.refresh_precision()

    def _slots(self, n):
        if self.slot_capacity < n:
            from . import _lib
            z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
            self.rows = z(2 * n, self.d)
            self.mixed_slots, self.grad_slots, self.g_prop_slots, self.g_raw_slots = (z(n, self.d) for _ in range(4))
            self.loss_rows = z(n)
            self.att_parts = z(int(_lib.load().spex_expert_gate_rows_bwd_parts(n)) * 512)
            self.arange = torch.arange(n, dtype=torch.int64, device=self.dev)
            self.slot_capacity = n
            self._release()

    def _release(self):
        d = self._desc
        self._desc = None
        if d is not None and (d.ev_fork or d.ev_join):
            try:
                torch.cuda.synchronize()
                from . import _lib
                _lib.release_step_events(d)
            except Exception:
                pass

    def __del__(self):
        self._release()

    def refresh_precision(self):
        self.precision[(self.t + 1) & 1] = torch.exp(-2.0 * self.model.core.task_weights.detach())

    def positions(self, users, items):
        pu, pi = self.P.padded_index(users.to(self.dev).long(), items.to(self.dev).long())
        return torch.cat([pu, pi]).contiguous()

    def step(self, users, items, labels, seq, seq_l, targets, pos=None):
        """users / items: int64 [B]; labels: fp32 [B]; seq: int64 [T, path_len] padded with the pad row's index; seq_l, targets:
        int64 [T] — the same on every rank.  Both losses are added to `loss_acc`."""
        import ctypes
        from . import _lib
        from .graph import _bump, _launch
        P, model = self.P, self.model
        B, T = users.numel(), (0 if seq is None else seq.shape[0])
        self._slots(2 * B)
        if pos is None:
            pos = self.positions(users, items)
        labels = labels.to(device=self.dev, dtype=torch.float32).contiguous()
        if T:
            seq, seq_l, targets = (t.to(self.dev).long().contiguous() for t in (seq, seq_l, targets))
            if seq.shape[1] != self.path_len or T > self.path_capacity:
                raise ValueError(f"PartitionedDualTaskStepper.step: {T} paths of width {seq.shape[1]} (capacity {self.path_capacity} x {self.path_len})")
        if self._desc is None:
            p = lambda t: t.data_ptr()
            self._desc = _lib.PartitionedDualStepDesc(
                graph=P.graph._h.value, graph_t=P.graph_t._h.value, comm=P.native._h.value, rows_per_rank=None,
                params=p(self.arena), m=p(self.m), v=p(self.v), light=p(self.light), g_prop=p(self.g_prop), g_raw=p(self.g_raw), gs=p(self.gs),
                g_E0=p(self.g_E0), gathered1=p(P.table(1)), gathered=p(P.gathered), gathered0=p(self.gathered0), user_pos=p(model.user_pos),
                user_table=p(self.user_table), rows=p(self.rows), mixed_slots=p(self.mixed_slots), grad_slots=p(self.grad_slots),
                g_prop_slots=p(self.g_prop_slots), g_raw_slots=p(self.g_raw_slots), loss_rows=p(self.loss_rows), att_parts=p(self.att_parts),
                arange=p(self.arange), g_user=p(self.g_user), g_small=p(self.g_small), a2=p(self.a2), trust_ws=p(self.trust_ws),
                dscore=p(self.dscore), loss_b=p(self.loss_b), loss=p(self.loss), loss_acc=p(self.loss_acc), precision=p(self.precision),
                n_local=P.n_local, max_rows=P.part.max_rows, n_local_users=model.n_local_users,
                user_lo=P.r0 if model.n_local_users > 0 else 0, slot_capacity=self.slot_capacity, path_capacity=self.path_capacity,
                path_len=self.path_len, n_user_rows=self.n_u, L=self.L, d=self.d, n_heads=self.n_heads,
                hybrid=0 if model.core.nonhybrid else 1, n_rec=self.n_rec, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                t=self.t, side_stream=None if self._side is None else self._side.cuda_stream, ev_fork=None, ev_join=None, flags=0,
                graph_push=P.push_graph()._h.value if self.fast and P.n_local > 0 and self.L >= 2 else None,
                gathered2=p(P.table(2)) if self.fast else None)
        dsc = self._desc
        dsc.t, dsc.lr = self.t, self.lr
        dsc.flags = (_lib.STEP_DETERMINISTIC if self.deterministic else 0) | (_lib.STEP_FIXED_TASK_WEIGHTS if self.fixed_task_weights else 0)
        self._rpr_keep = P._rows_per_rank
        dsc.rows_per_rank = None if P._rows_per_rank is None else ctypes.cast(P._rows_per_rank, ctypes.c_void_p).value
        dsc.comm = P.native._h.value
        if T and self._side is not None:
            for t in (seq, seq_l, targets):
                t.record_stream(self._side)
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
        _launch(self.dev, "spex_partitioned_dual_task_step_f32", ctypes.byref(dsc), vp(pos), vp(labels), B,
                vp(seq) if T else None, vp(seq_l) if T else None, vp(targets) if T else None, T)
        self.t = dsc.t
        _bump(self.arena, self.m, self.v, self.loss_acc)
