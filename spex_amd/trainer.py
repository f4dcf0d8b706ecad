"""Autograd-free training steps on pre-allocated HBM buffers — what a production loop (and bench.py) drives.

The drop-in model keeps torch autograd + torch.optim so the reference drivers run unchanged; this class is the same
mathematics as main_rec.py:30-37 issued as a fixed sequence of libspexhip launches with no allocation, no host
synchronisation and no Python between kernels beyond the ctypes calls (capturable in a HIP graph):

    exact step  (reference semantics):  L x SpMM (mean fused)  ->  score + BCE + grad rows  ->  L x SpMM^T (g/(L+1)
                                        fused)  ->  fused Adam over the whole [N, d] table
    bpr step    (north-star extension): L x SpMM (mean fused)  ->  fused gather + dot + sigmoid + SGD over triples,
                                        scores read from the propagated table, updates applied to E0
"""
import gc
import os

import numpy as np
import torch

from . import ops


class LightGCNStepper:
    def __init__(self, graph, E0, n_user_rows, n_layers=3, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graph_t=None, deterministic=None):
        """deterministic (default: SPEX_DETERMINISTIC=1 in the environment): the step takes every sum in a fixed order — no float
        atomics (spex_lightgcn_step_t.flags & SPEX_STEP_DETERMINISTIC): per-sample gradient rows added per table row in ascending
        slot order, the whole backward in pull form.  Two runs then end in bit-identical tables; ~10 us more per step."""
        assert E0.is_cuda and E0.dtype == torch.float32 and E0.is_contiguous()
        self.deterministic = (os.environ.get("SPEX_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        self.graph, self.graph_t = graph, (graph_t if graph_t is not None else graph)
        self.E0, self.n_u, self.L = E0, int(n_user_rows), int(n_layers)
        self.lr, self.betas, self.eps = lr, betas, eps
        n, d = E0.shape
        dev = E0.device
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.light_out = z(n, d)
        self.ws_fwd = z(2, n, d)
        self.g_out = z(n, d)
        self.ws_bwd = z(3, n, d)
        self.grad_E0 = z(n, d)
        self.m, self.v = z(n, d), z(n, d)
        self.loss_acc = z(1)          # running loss sum of the fused BPR steps (read it when you need it)
        self.lo_batch = None          # propagated rows of the current batch (propagate_for_batch), allocated on first use
        self.t = 0

    # -- pieces
    def propagate(self):
        return self.graph.propagate(self.E0, self.L, mean_out=self.light_out, ws=self.ws_fwd)

    def propagate_for_batch(self, users, items):
        """The propagation as the training step needs it: layers 1 .. L-1 over the whole graph, the LAST layer only at the
        batch's rows (the loss reads light_out nowhere else, model.py:115-116) — spex_spmm_rowlist_f32 instead of a
        launch over the whole matrix.  Returns a table that is valid at those rows only.  Falls back to the full propagation
        when the row-list kernel does not apply (d != 64, L == 0)."""
        L, g, lo = self.L, self.graph, self.light_out
        if L == 0 or self.E0.shape[1] != 64:
            return self.propagate()
        cur = self.E0
        for l in range(L - 1):
            nxt = self.ws_fwd[l & 1]
            g.spmm(cur, Y=nxt, acc_in=self.E0 if l == 0 else lo, acc_out=lo)
            cur = nxt
        # NOT in place: a batch lists the same user several times, and a second workgroup must not read a row the
        # first one has already finished — the running sum stays in `lo`, the finished rows go to `lo_batch`
        if self.lo_batch is None:
            self.lo_batch = torch.zeros_like(lo)
        g.spmm_rows(cur, users, items, 0, self.n_u, acc_in=self.E0 if L == 1 else lo, acc_out=self.lo_batch,
                    acc_div=float(L + 1))
        return self.lo_batch

    def step_bce(self, users, items, labels, loss_acc=None, batch_rows_only=False):
        """One exact reference training step (main_rec.py:32-37).  Returns the mean BCE loss (device tensor) — or, with
        `loss_acc` (a 1-element device buffer), accumulates the batch's loss SUM into it and returns None: then the
        step is exactly eight launches (3 SpMM, scoring, 3 SpMM, Adam) with nothing between them.  The gradient table
        is cleared by the Adam pass of the previous step.  batch_rows_only: compute the last forward layer at the
        batch's rows only (same arithmetic for those rows; `self.light_out` is then not the whole table)."""
        if batch_rows_only and loss_acc is not None and self._one_call_ok(users, items, labels):
            return self._step_bce_one_call(users, items, labels, loss_acc)
        masked = getattr(self.graph, "mask_mode", 0) != 0        # (the row-list kernel of the launch-by-launch form takes no mask)
        lo = self.propagate_for_batch(users, items) if batch_rows_only and not masked else self.propagate()
        B = users.numel()
        self._slots(B)
        if self.deterministic and self.E0.shape[1] == 64:
            # launch-by-launch form of the deterministic step: per-sample rows only (no dense atomics), added per table row in
            # slot order, then the all-pull backward
            _, loss_sum = ops.score_bce(lo[:self.n_u], lo[self.n_u:], users, items, labels, None, None, 1.0 / B, loss_sum=loss_acc,
                                        want_gamma=False, grad_slots=self.grad_slots)
            ops.reduce_slots(users, items, self.n_u, self.E0.shape[0], self.grad_slots, self.g_out)
            self._ws0_clean = False
            self.graph_t.propagate_bwd(self.g_out, self.L, grad_E0=self.grad_E0, ws=self.ws_bwd)
        else:
            _, loss_sum = ops.score_bce(lo[:self.n_u], lo[self.n_u:], users, items, labels, self.g_out[:self.n_u],
                                        self.g_out[self.n_u:], 1.0 / B, loss_sum=loss_acc, want_gamma=False, grad_slots=self.grad_slots)
            self.backward_from_batch_rows(users, items)
        self.t += 1
        ops.adam_step(self.E0, self.grad_E0, self.m, self.v, self.t, self.lr, self.betas[0], self.betas[1], self.eps,
                      zero=self.g_out)
        return None if loss_acc is not None else loss_sum / B

    def _slots(self, B):
        """The batch's per-sample gradient rows (operands of the push-form first backward product)."""
        if getattr(self, "grad_slots", None) is None or self.grad_slots.shape[0] < 2 * B:
            self.grad_slots = torch.zeros((2 * B, self.E0.shape[1]), dtype=torch.float32, device=self.E0.device)
            self._desc = None
        return self.grad_slots

    # -- the whole step as one library call (spex_lightgcn_step_bce_f32): same launches, issued from native code
    def _one_call_ok(self, users, items, labels):
        masked = getattr(self.graph, "mask_mode", 0) != 0 or getattr(self.graph_t, "mask_mode", 0) != 0
        return (self.L >= 1 and self.E0.shape[1] == 64 and (not masked or (self.L >= 2 and self.graph_t is not self.graph))
                and users.is_cuda and items.is_cuda and labels.is_cuda
                and users.dtype == torch.int64 and items.dtype == torch.int64 and labels.dtype == torch.float32
                and users.is_contiguous() and items.is_contiguous() and labels.is_contiguous()
                and users.numel() == items.numel() == labels.numel() and users.numel() >= 1)

    def _prepare_desc(self, B):
        """The one-call step's descriptor for batches of up to B samples (built once, refreshed per call)."""
        from . import _lib
        if self.lo_batch is None:
            self.lo_batch = torch.zeros_like(self.light_out)
        self._slots(B)
        if not getattr(self, "_ws0_clean", False):       # another path used the workspace: restore the all-zero push target
            self.ws_bwd[0].zero_()
            self._ws0_clean = True
        if getattr(self, "_desc", None) is None:
            p = lambda t: t.data_ptr()
            self._desc = _lib.LightGCNStepDesc(
                graph=self.graph._h.value, graph_t=self.graph_t._h.value, E0=p(self.E0), m=p(self.m), v=p(self.v),
                light_out=p(self.light_out), ws_fwd=p(self.ws_fwd), lo_batch=p(self.lo_batch), g_out=p(self.g_out),
                ws_bwd=p(self.ws_bwd), grad_E0=p(self.grad_E0), grad_slots=p(self.grad_slots),
                slot_capacity=self.grad_slots.shape[0], n_user_rows=self.n_u, L=self.L, d=self.E0.shape[1],
                lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, t=self.t, flags=0)
        d = self._desc
        d.t, d.lr = self.t, self.lr
        d.flags = _lib.STEP_DETERMINISTIC if self.deterministic else 0
        return d

    def _step_bce_one_call(self, users, items, labels, loss_acc):
        import ctypes
        from .graph import _bump, _launch
        B = users.numel()
        d = self._prepare_desc(B)
        _launch(self.E0.device, "spex_lightgcn_step_bce_f32", ctypes.byref(d), ctypes.c_void_p(users.data_ptr()),
                ctypes.c_void_p(items.data_ptr()), ctypes.c_void_p(labels.data_ptr()), B, ctypes.c_void_p(loss_acc.data_ptr()))
        self.t = d.t
        _bump(self.E0, self.m, self.v, loss_acc)
        return None

    def epoch_bce(self, users, items, labels, batch_size, loss_full, loss_ragged, max_steps=None, keep_prob=1.0, drop_seed=0):
        """A whole pre-shuffled, device-resident epoch as ONE native call (spex_lightgcn_epoch_bce_f32; main_rec.py:30-37): batch k =
        samples [k B, (k+1) B) through the one-call step, nothing but its launches issued by the host.  loss_full / loss_ragged:
        1-element device accumulators (loss sums of the full batches / of a shorter last one).  keep_prob < 1: the in-kernel sampled
        edge mask, a fresh one per step (seed (drop_seed << 32) | step — trainer.edge_dropout_mask's "philox" stream)."""
        import ctypes
        from .graph import _bump, _launch
        if not self._one_call_ok(users[:1], items[:1], labels[:1]):
            raise ValueError("LightGCNStepper.epoch_bce: needs d == 64 and contiguous int64 / fp32 device tensors")
        if keep_prob < 1.0 and (self.L < 2 or self.graph_t is self.graph):
            raise ValueError("LightGCNStepper.epoch_bce: edge dropout needs L >= 2 and graph_t = the transposed handle with the edge-id permutation")
        d = self._prepare_desc(min(int(batch_size), users.numel()))
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        _launch(self.E0.device, "spex_lightgcn_epoch_bce_f32", ctypes.byref(d), vp(users), vp(items), vp(labels), users.numel(), int(batch_size),
                -1 if max_steps is None else int(max_steps), float(keep_prob), int(drop_seed) & 0xFFFFFFFF, vp(loss_full), vp(loss_ragged))
        self.t = d.t
        _bump(self.E0, self.m, self.v, loss_full, loss_ragged)

    def backward_from_batch_rows(self, users, items):
        """grad_E0 from g_out (non-zero on the batch's rows only).  The first product of the backward pass, A^T g, touches
        only the stored entries of those <= 2B rows OF A: it is taken in push form (~14 k entries on
        Epinion2, spex_spmm_push_batch_f32 over the forward handle) instead of a pull-form SpMM over all 418 k; the remaining L - 1
        products are dense, over A^T.  Falls back to the all-pull form where the push form does not apply (edge dropout on the
        handle, L < 2)."""
        L, gt = self.L, self.graph_t
        self._ws0_clean = False
        if L < 2 or self.E0.shape[1] != 64 or getattr(gt, "mask_mode", 0) != 0 or getattr(self.graph, "mask_mode", 0) != 0:
            gt.propagate_bwd(self.g_out, L, grad_E0=self.grad_E0, ws=self.ws_bwd)
            return
        inv = 1.0 / float(L + 1)
        G = self.ws_bwd[0]
        G.zero_()
        ops.spmm_push_batch(self.graph, users, items, self.n_u, self.grad_slots, G, add=self.grad_slots, scale=inv)   # G_{L-1}
        cur = G
        for l in range(L - 2, -1, -1):
            nxt = self.grad_E0 if l == 0 else self.ws_bwd[1 + ((L - 2 - l) & 1)]      # ws_bwd[1], [2], [1] ...: never the source
            gt.spmm(cur, Y=nxt, add_in=self.g_out, add_div=float(L + 1))                                      # g/(L+1) + A^T G
            cur = nxt

    def step_bpr_sgd(self, users, pos, neg, lr=None, reg=0.0):
        """Propagation + fused BPR-SGD kernel (scores from the propagated table, update on E0).  Returns the running
        loss-sum buffer `loss_acc` (sum over every triple since it was last zeroed; no per-step allocation or sync)."""
        T = users.numel()
        if (1 <= self.L <= 3 and self.E0.shape[1] == 64 and getattr(self.graph, "mask_mode", 0) == 0 and T < ops.GROUPED_BPR_MIN_TRIPLES
                and all(t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() for t in (users, pos, neg))):
            # one native call, L + 1 launches: the layer mean is formed by the BPR kernel at its triples' rows only, so the
            # whole-graph launches behind layer 1 run in the plain form (spex_lightgcn_step_bpr_f32).  self.light_out then holds
            # E^0 + E^1, not the propagated table.
            import ctypes
            from .graph import _bump, _launch
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            _launch(self.E0.device, "spex_lightgcn_step_bpr_f32", self.graph._h, p(self.E0), p(self.light_out), p(self.ws_fwd), self.n_u,
                    self.L, 64, p(users), p(pos), p(neg), T, float(self.lr if lr is None else lr), float(reg), p(self.loss_acc))
            _bump(self.E0, self.loss_acc, self.light_out)
            return self.loss_acc
        self.propagate()
        lo = self.light_out
        ops.bpr_sgd_step(lo[:self.n_u], lo[self.n_u:], self.E0[:self.n_u], self.E0[self.n_u:], users, pos, neg,
                         self.lr if lr is None else lr, reg, loss_sum=self.loss_acc)
        return self.loss_acc

    def step_bpr_exact(self, users, pos, neg):
        """BPR loss differentiated through the propagation, Adam update (upstream LightGCN training semantics)."""
        self.propagate()
        T = users.numel()
        lo = self.light_out
        loss_sum = ops.bpr_loss_grad(lo[:self.n_u], lo[self.n_u:], users, pos, neg, self.g_out[:self.n_u],
                                     self.g_out[self.n_u:], 1.0 / T)
        self._ws0_clean = False
        self.graph_t.propagate_bwd(self.g_out, self.L, grad_E0=self.grad_E0, ws=self.ws_bwd)
        self.t += 1
        ops.adam_step(self.E0, self.grad_E0, self.m, self.v, self.t, self.lr, self.betas[0], self.betas[1], self.eps,
                      zero=self.g_out)          # invariant of both exact steps: g_out is all-zero between steps
        return loss_sum / T


def dataloader_epoch_order(n):
    """The index order `DataLoader(dataset, shuffle=True)` walks in one epoch, drawn from the GLOBAL torch RNG exactly as
    torch's own iterator draws it (a base seed at iterator creation, then RandomSampler's seed, then a randperm from a
    private generator) — so a loop built on it sees the batches main_rec.py:20,30 would see for the same torch.manual_seed.
    tests/test_host_logic.py checks it against the installed torch's DataLoader."""
    torch.empty((), dtype=torch.int64).random_()                      # _BaseDataLoaderIter: _base_seed
    seed = int(torch.empty((), dtype=torch.int64).random_().item())    # RandomSampler.__iter__
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g)


_rand_staging = {}


def reference_keep_mask(nnz_list, keep_prob, device):
    """The keep mask `(torch.rand(nnz) + keep_prob).int().bool()` of model.py:46-55 for one or several handles (one draw per fold
    under --A_split), as a device uint8 tensor.  Only the DRAW stays on the CPU — it has to: it is the reference's global
    generator stream — into a reused pinned buffer; `+ keep_prob`, the truncation and the byte conversion run on the device on the
    uploaded floats (the same IEEE operations, bit for bit).  Doing those three elementwise passes with CPU tensor ops cost 18 ms per
    step on the GPU box (torch's intra-op thread pool, sized for the whole machine, on a 16-core share); the draw itself is < 1 ms."""
    total = int(sum(int(n) for n in nnz_list))
    if torch.device(device).type != "cuda":
        keep = torch.cat([(torch.rand(int(n)) + keep_prob).int().bool() for n in nnz_list])
        return keep.to(torch.uint8).to(device).contiguous()
    buf = _rand_staging.get(total)
    if buf is None:
        buf = _rand_staging[total] = torch.empty(total, dtype=torch.float32).pin_memory()
    off = 0
    for n in nnz_list:                                   # torch.rand(n, out=...) draws exactly what torch.rand(n) draws
        torch.rand(int(n), out=buf[off: off + int(n)])
        off += int(n)
    r = buf.to(device)                                   # (synchronous: the pinned buffer is free again when this returns)
    return (r + keep_prob).int().bool().to(torch.uint8)


def edge_dropout_mask(graph, keep_prob, stream, seed=0, step=0):
    """The edge-dropout mask tuple of one training step (SpexGraph.set_edge_mask arguments) — model.py:46-55.
    stream "reference": the reference's own draw, `torch.rand(nnz) + keep_prob` on the CPU from the global generator (one call
    per step, exactly where model.py:50 makes it), uploaded as a keep mask; "philox": the in-kernel counter-based mask."""
    if stream == "reference":
        return (1, reference_keep_mask([graph.nnz], keep_prob, graph.device), float(keep_prob), 0)
    if stream != "philox":
        raise ValueError(f"edge dropout stream must be 'reference' or 'philox' (got {stream!r})")
    return (2, None, float(keep_prob), (int(seed) << 32) | (int(step) & 0xFFFFFFFF))


def epoch_arrays(train_data, resample=True):
    """One epoch's samples as the reference's loop would meet them (main_rec.py:26-31): negatives drawn by `train_data.ng_sample()`
    (NumPy's global stream), then the DataLoader's shuffle (torch's global stream) applied — three host arrays (users, items, labels)."""
    if resample:
        train_data.ng_sample()
    order = dataloader_epoch_order(len(train_data)).numpy()
    return train_data.users_fill[order], train_data.items_fill[order], train_data.labels_fill_np[order]


def train_epochs(stepper, train_data, n_epochs, batch_size=256, edge_dropout=None, after_epoch=None):
    """n_epochs x train_epoch with the NEXT epoch's negatives and shuffle prepared on a second host thread while the GPU trains the
    current one (an Epinion2 epoch: 0.14 s of sampling beside 0.38 s of steps — the native epoch call releases the GIL, and the
    sampler's NumPy / torch generators are touched by that thread alone meanwhile, in the order a sequential loop would touch them:
    same negatives, same shuffles, same run).  after_epoch(epoch, loss_sum_tensor) runs between epochs (evaluation: the reference's
    Test() draws no random numbers — a callback that does would see them drawn AFTER the next epoch's).  Returns the per-epoch loss
    sums (main_rec.py:36)."""
    import threading
    nxt = epoch_arrays(train_data)
    totals = []
    for ep in range(n_epochs):
        arrays, box, th = nxt, {}, None
        if ep + 1 < n_epochs:
            def work():
                try:
                    box["v"] = epoch_arrays(train_data)
                except BaseException as e:       # noqa: BLE001 — re-raised on the caller's thread
                    box["e"] = e
            th = threading.Thread(target=work)
            th.start()
        try:
            total = train_epoch(stepper, train_data, batch_size=batch_size, edge_dropout=edge_dropout, arrays=arrays)
        finally:
            if th is not None:
                th.join()
        if "e" in box:
            raise box["e"]
        nxt = box.get("v")
        totals.append(total)
        if after_epoch is not None:
            after_epoch(ep, total)
    return [float(t) for t in totals]


def train_epoch(stepper, train_data, batch_size=256, resample=True, pause_gc=True, edge_dropout=None, max_steps=None,
                step_losses=None, arrays=None):
    """Train() of main_rec.py:25-38 without the per-step host work of its DataLoader loop: negatives are drawn like the
    reference's (`train_data.ng_sample()`, NumPy global RNG), the epoch's sample order is the DataLoader's own
    (dataloader_epoch_order), the whole shuffled epoch is moved to the device once, and every batch is one
    LightGCNStepper.step_bce (no allocation, no synchronisation).  Same batches, same arithmetic — ≈105 us per step instead
    of ≈550 us.  Returns the epoch's summed loss as a device tensor (main_rec.py:36 accumulates the same sum).
    edge_dropout: None, or (keep_prob, stream[, seed]) for `--dropout 1 --keepprob p` (README.md:119-123): a fresh mask per
    step (edge_dropout_mask) on the stepper's graph and graph_t — which must then be the transposed handle carrying the edge-id
    permutation, since the masked operator is not symmetric.  max_steps: stop after that many batches; step_losses: a list that
    receives every step's mean loss (synchronises per step: a validation aid).  arrays: the epoch's (users, items, labels) host arrays
    already sampled and shuffled (epoch_arrays) — train_epochs prepares the next epoch's while this one runs."""
    users_h, items_h, labels_h = arrays if arrays is not None else epoch_arrays(train_data, resample)
    n = len(users_h)
    if edge_dropout is not None and stepper.graph_t is stepper.graph:
        raise ValueError("train_epoch(edge_dropout=...): the stepper needs graph_t = the transposed handle with the edge-id "
                         "permutation (LightGCN._transposed()): a masked adjacency is not symmetric")
    dev = stepper.E0.device
    users = torch.from_numpy(users_h).to(dev)
    items = torch.from_numpy(items_h).to(dev)
    labels = torch.from_numpy(labels_h).to(device=dev, dtype=torch.float32)
    # A full (generation-2) collection of Python's cyclic GC walks every object torch / scipy / pandas created at import:
    # ~40 ms, i.e. ~400 steps' worth of launches, whenever it triggers inside the loop (tools/stall_probe.py).  The loop
    # creates no reference cycles: pause the collector for its duration.
    gc_was_on = pause_gc and gc.isenabled()
    if gc_was_on:
        gc.disable()
    n_full = n // batch_size * batch_size
    acc = torch.zeros(2, 1, dtype=torch.float32, device=dev)     # loss sums of the full batches / of the ragged last one
    starts = list(range(0, n, batch_size))
    if max_steps is not None:
        starts = starts[:max_steps]
    # the whole epoch as ONE native call where nothing has to happen on the host between two steps: no per-step loss read-back, no
    # host-drawn mask (the reference-stream dropout mask is drawn on the CPU per step; the sampled one is keyed in the kernel)
    native_ok = (step_losses is None and (edge_dropout is None or edge_dropout[1] == "philox") and hasattr(stepper, "epoch_bce")
                 and n > 0 and stepper._one_call_ok(users[:1], items[:1], labels[:1])
                 and (edge_dropout is None or (stepper.L >= 2 and stepper.graph_t is not stepper.graph)))
    if native_ok:
        try:
            kp = 1.0 if edge_dropout is None else float(edge_dropout[0])
            seed = 0 if edge_dropout is None or len(edge_dropout) < 3 else int(edge_dropout[2])
            stepper.epoch_bce(users, items, labels, batch_size, acc[0], acc[1], max_steps=max_steps, keep_prob=kp, drop_seed=seed)
        finally:
            if gc_was_on:
                gc.enable()
        n_done = min(n, len(starts) * batch_size)
        ragged = n_done - n_done // batch_size * batch_size
        return acc[0, 0] / batch_size + (acc[1, 0] / ragged if ragged else 0.0)
    try:
        for k, s in enumerate(starts):
            e = min(s + batch_size, n)
            slot = acc[0] if e - s == batch_size else acc[1]
            tmp = torch.zeros(1, dtype=torch.float32, device=dev) if step_losses is not None else None
            if edge_dropout is not None:
                mask = edge_dropout_mask(stepper.graph, edge_dropout[0], edge_dropout[1], edge_dropout[2] if len(edge_dropout) > 2 else 0, k + 1)
                stepper.graph.set_edge_mask(*mask)
                stepper.graph_t.set_edge_mask(*mask)
            stepper.step_bce(users[s:e], items[s:e], labels[s:e], loss_acc=slot if tmp is None else tmp, batch_rows_only=True)
            if tmp is not None:
                step_losses.append(tmp.item() / (e - s))
                slot += tmp
    finally:
        if edge_dropout is not None:
            stepper.graph.set_edge_mask(0)
            stepper.graph_t.set_edge_mask(0)
        if gc_was_on:
            gc.enable()
    total = acc[0, 0] / batch_size + (acc[1, 0] / (n - n_full) if n_full < n else 0.0)   # sum of per-batch mean losses
    return total


class NGCFStepper:
    """The NGCF training step (NGCF_SPEX/code/main_rec.py:122-128: forward, BCE, backward, Adam) as a fixed sequence of
    libspexhip launches on pre-allocated buffers — no autograd, no allocation, no host synchronisation:

        per layer:  SpMM (side = A ego)  ->  fused layer kernel (both 64x64 products, LeakyReLU, message dropout, L2 norm,
                    concat slice)
        scoring:    gather . dot . BCE + gradient rows into the [N, 64 (L + 1)] gradient table
        per layer, last to first:  fused layer backward (recompute, input gradients, weight gradients)  ->  SpMM on A^T
                    with the direct part added in its epilogue
        Adam:       one pass over the embedding table, one over the flat block of layer weights (each also clears its
                    gradient buffer for the next step)

    model: a spex_amd.ngcf.NGCF with 64-wide layers; its parameters are trained IN PLACE (the table is the module's own
    flat buffer; the layer weights are re-homed into one flat block whose views replace the nn.Linear parameters' data),
    so `model` can be evaluated / saved as usual at any point.  Message dropout uses the model's counter-based stream
    (message_dropout_seed, dropout_step); `stepper.dropout_stream = "reference"` (single-layer one-call step): the noise is the
    reference's own nn.Dropout draw instead — empty(N, 64).bernoulli_(1 - p) from the global CPU generator, once per step where
    main_rec.py:81 draws it, uploaded and read by the kernels through spex_ngcf_message_mask (validation mode: ~1.5 ms per step).
    """

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, deterministic=None):
        """deterministic (default: SPEX_DETERMINISTIC=1 in the environment): no float atomics in the step — the slots' compact
        gradient rows are added per table row in ascending slot order and A^T g_side is a pull-form product over A^T
        (spex_ngcf_step_t.flags & SPEX_STEP_DETERMINISTIC; single-layer models, the one-call step)."""
        if not model._fused_ok():
            raise ValueError("NGCFStepper needs 64-wide layers (the fused layer kernels)")
        self.model, self.lr, self.betas, self.eps, self.t = model, lr, betas, eps, 0
        self.deterministic = (os.environ.get("SPEX_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        # spex_ngcf_step_t.side_stream (the layer weights' Adam pass on a second stream beside the push-form product and the table's
        # pass) stays unused: measured on the MI355X it LOSES 10 us per step (69.4 vs 59.3 us) — the 5 us pass it hides costs a fork and
        # a join, and a cross-stream event wait takes ~5 us to propagate between the two hardware queues.
        self._side = None
        self.E0 = model.flat_table()
        n, d = self.E0.shape
        dev = self.E0.device
        L = self.L = model.n_layers
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        # layer weights in one flat block: [W_gc | b_gc | W_bi | b_bi] per layer
        per = 2 * (d * d + d)
        self.W = z(L * per)
        self.views = []
        for l, (gc, bi) in enumerate(zip(model.GC_Linear_list, model.Bi_Linear_list)):
            o = l * per
            vs = (self.W[o:o + d * d].view(d, d), self.W[o + d * d:o + d * d + d],
                  self.W[o + d * d + d:o + 2 * d * d + d].view(d, d), self.W[o + 2 * d * d + d:o + per])
            for v, p in zip(vs, (gc.weight, gc.bias, bi.weight, bi.bias)):
                v.copy_(p.data)
                p.data = v                                    # the module's parameters now live in the flat block
            self.views.append(vs)
        self.gW = z(L * per)
        self.g_views = [tuple(self.gW[l * per + a:l * per + b].view(*shape) for a, b, shape in
                              ((0, d * d, (d, d)), (d * d, d * d + d, (d,)), (d * d + d, 2 * d * d + d, (d, d)),
                               (2 * d * d + d, per, (d,)))) for l in range(L)]
        self.mW, self.vW = z(L * per), z(L * per)
        self.mE, self.vE = z(n, d), z(n, d)
        self.all_emb = z(n, d * (L + 1))
        self.g_all = z(n, d * (L + 1))                          # invariant: all-zero between steps
        self._sides_all, self._egos_all, self._g_next_all = z(L, n, d), z(max(L - 1, 1), n, d), z(2, n, d)   # (contiguous: the deep step's descriptor)
        self.sides = [self._sides_all[l] for l in range(L)]
        self.egos = [self.E0] + [self._egos_all[l] for l in range(L - 1)]
        self.g_side, self.g_ego = z(n, d), z(n, d)
        self.g_next = [self._g_next_all[0], self._g_next_all[1]]   # g_next[(L-1) & 1] is all-zero between steps (push target)
        self.g_slots, self.g_side_c, self.g_ego_c, self.gW_parts = None, None, None, None
        self.loss_acc = z(1)
        self.n_u = model.n_users + 1

    def step(self, users, items, labels, loss_acc=None):
        """One training step on a batch (device tensors).  Accumulates the batch's loss SUM into `loss_acc` (default: the
        stepper's own running buffer) and returns it."""
        m, L = self.model, self.L
        acc = self.loss_acc if loss_acc is None else loss_acc
        if L == 1 and self._one_call_ok(users, items, labels):
            return self._step_one_call(users, items, labels, acc)
        if (L >= 2 and not self.deterministic and self._one_call_ok(users, items, labels)
                and getattr(self, "dropout_stream", getattr(m, "dropout_stream", "counter")) != "reference"):
            return self._step_deep_one_call(users, items, labels, acc)
        if self.deterministic:
            raise ValueError("NGCFStepper(deterministic=True): single-layer models with contiguous int64 / fp32 device batches only")
        drop = None
        if any(p > 0 for p in m.mess_dropout):
            drop = (m.mess_dropout, m.message_dropout_seed, m.dropout_step)
            m.dropout_step += 1
        pad = m.n_users
        for l in range(L):
            m.graph.spmm(self.egos[l], Y=self.sides[l])
            dl = None if drop is None or drop[0][l] <= 0 else (drop[0][l], drop[1], drop[2])
            ops.ngcf_layer_fwd(self.egos[l], self.sides[l], *self.views[l], self.all_emb, l, l == 0,
                               self.egos[l + 1] if l < L - 1 else None, drop=dl, pad_row=pad)
        B = users.numel()
        if self.g_side_c is None or self.g_side_c.shape[0] < 2 * B:
            dev, d = self.E0.device, self.E0.shape[1]
            self.g_slots = torch.zeros((2 * B, d * (L + 1)), dtype=torch.float32, device=dev)     # per-sample gradient rows
            self.g_side_c = torch.zeros((2 * B, d), dtype=torch.float32, device=dev)
            self.g_ego_c = torch.zeros_like(self.g_side_c)
            self.gW_parts = torch.zeros((ops.ngcf_bwd_rows_parts(2 * B), 2 * (d * d + d)), dtype=torch.float32, device=dev)
        # scoring: per-sample gradient rows; the table form too only if earlier layers (dense backward) need their slices
        tab = (self.g_all[:self.n_u], self.g_all[self.n_u:]) if L > 1 else (None, None)
        ops.score_bce(self.all_emb[:self.n_u], self.all_emb[self.n_u:], users, items, labels, tab[0], tab[1], 1.0 / B, loss_sum=acc,
                      want_gamma=False, grad_slots=self.g_slots)
        # backward.  Only the batch's <= 2B rows carry a gradient behind the LAST layer: its backward runs on the batch's
        # slots (compact tiles, each slot with its own gradient row), and A^T g_side is a push over their stored entries
        # instead of a pull-form SpMM.
        l = L - 1
        dl = None if drop is None or drop[0][l] <= 0 else (drop[0][l], drop[1], drop[2])
        parts = self.gW_parts[: ops.ngcf_bwd_rows_parts(2 * B)]      # (a ragged last batch writes fewer blocks)
        ops.ngcf_layer_bwd_rows(self.egos[l], self.sides[l], *self.views[l], self.g_slots, l, None, users, items, self.n_u,
                                self.g_side_c, self.g_ego_c, parts, drop=dl, pad_row=pad)
        g_next = self.g_next[l & 1]
        if L > 1:
            g_next.zero_()
        ops.spmm_push_batch(m.graph, users, items, self.n_u, self.g_side_c, g_next, add=self.g_ego_c)
        for l in range(L - 2, -1, -1):
            dl = None if drop is None or drop[0][l] <= 0 else (drop[0][l], drop[1], drop[2])
            ops.ngcf_layer_bwd(self.egos[l], self.sides[l], *self.views[l], self.g_all, l, g_next, self.g_side, self.g_ego,
                               *self.g_views[l], drop=dl, pad_row=pad)
            out = self.g_next[l & 1]
            m.graph_t.spmm(self.g_side, Y=out, add_in=self.g_ego, add_div=1.0)
            g_next = out
        self.t += 1
        b1, b2 = self.betas
        # (L == 1: the table gradient is the push target, cleared again by the Adam pass that consumes it)
        ops.adam_step(self.E0, g_next, self.mE, self.vE, self.t, self.lr, b1, b2, self.eps, zero=g_next if L == 1 else None)
        # layer weights: the last layer's gradient arrives as partial blocks (summed inside its Adam pass); the other layers'
        # (dense backward, atomics into gW) go through the plain pass
        per = self.gW_parts.shape[1]
        lo = (L - 1) * per
        ops.adam_step_sum(self.W[lo:lo + per], parts, self.mW[lo:lo + per], self.vW[lo:lo + per], self.t, self.lr, b1, b2, self.eps)
        if L > 1:
            ops.adam_step(self.W[:lo], self.gW[:lo], self.mW[:lo], self.vW[:lo], self.t, self.lr, b1, b2, self.eps, zero=self.gW[:lo])
            self.g_all.zero_()
        return acc


def _drop_desc(stepper):
    """Forget a two-stream stepper's native step descriptor (rebuilt on the next step), releasing the fork / join events the
    library keeps in it.  The steps already queued hold no reference to the events once recorded / waited on."""
    d = getattr(stepper, "_desc", None)
    stepper._desc = None
    if d is not None and (getattr(d, "ev_fork", None) or getattr(d, "ev_join", None)):
        try:
            torch.cuda.synchronize()
            from . import _lib
            _lib.release_step_events(d)
        except Exception:
            pass


def _ngcf_one_call_ok(users, items, labels):
    return (users.is_cuda and items.is_cuda and labels.is_cuda and users.dtype == torch.int64 and items.dtype == torch.int64
            and labels.dtype == torch.float32 and users.is_contiguous() and items.is_contiguous() and labels.is_contiguous()
            and users.numel() == items.numel() == labels.numel() and users.numel() >= 1)


def _ngcf_prepare_desc(self, B):
    """The single-layer one-call step's descriptor for batches of up to B samples (built once, refreshed per call)."""
    from . import _lib
    m = self.model
    if self.g_side_c is None or self.g_side_c.shape[0] < 2 * B:
        dev, d = self.E0.device, self.E0.shape[1]
        self.g_slots = torch.zeros((2 * B, 2 * d), dtype=torch.float32, device=dev)
        self.g_side_c = torch.zeros((2 * B, d), dtype=torch.float32, device=dev)
        self.g_ego_c = torch.zeros_like(self.g_side_c)
        self.gW_parts = torch.zeros((ops.ngcf_bwd_rows_parts(2 * B), 2 * (d * d + d)), dtype=torch.float32, device=dev)
        _drop_desc(self)
    if getattr(self, "_desc", None) is None:
        p = lambda t: t.data_ptr()
        self._desc = _lib.NGCFStepDesc(
            graph=m.graph._h.value, E0=p(self.E0), mE=p(self.mE), vE=p(self.vE), W=p(self.W), mW=p(self.mW), vW=p(self.vW),
            all_emb=p(self.all_emb), side=p(self.sides[0]), g_slots=p(self.g_slots), g_side_c=p(self.g_side_c),
            g_ego_c=p(self.g_ego_c), gW_parts=p(self.gW_parts), grad=p(self.g_next[0]), slot_capacity=self.g_slots.shape[0],
            n_user_rows=self.n_u, pad_row=m.n_users, slope=0.01, p_drop=float(m.mess_dropout[0]), seed=int(m.message_dropout_seed),
            dropout_step=m.dropout_step, t=self.t, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
            side_stream=None if self._side is None else self._side.cuda_stream, ev_fork=None, ev_join=None, graph_t=None,
            g_side_dense=None, g_ego_dense=None, flags=0)
        if self.deterministic:
            self._desc.graph_t = m.graph_t._h.value
            self._desc.g_side_dense, self._desc.g_ego_dense = p(self.g_side), p(self.g_ego)     # all-zero between steps
            self._desc.flags = _lib.STEP_DETERMINISTIC
    d = self._desc
    d.t, d.lr, d.dropout_step, d.seed, d.p_drop = self.t, self.lr, m.dropout_step, int(m.message_dropout_seed), float(m.mess_dropout[0])
    return d


def _ngcf_epoch(self, users, items, labels, batch_size, loss_full, loss_ragged, max_steps=None):
    """A whole pre-shuffled, device-resident epoch of the single-layer model as ONE native call (spex_ngcf_epoch_bce_f32; train() of
    NGCF_SPEX/code/main_rec.py:116-131): nothing but the steps' launches is issued by the host.  Counter-based message dropout only
    (the reference-stream validation mode draws its noise on the host per step)."""
    import ctypes
    from .graph import _bump, _launch
    m = self.model
    if self.L != 1 or not self._one_call_ok(users[:1], items[:1], labels[:1]) \
            or (getattr(self, "dropout_stream", getattr(m, "dropout_stream", "counter")) == "reference" and m.mess_dropout[0] > 0):
        raise ValueError("NGCFStepper.epoch: the single-layer model with the counter-based dropout stream and contiguous int64 / fp32 device tensors")
    d = _ngcf_prepare_desc(self, min(int(batch_size), users.numel()))
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    _launch(self.E0.device, "spex_ngcf_epoch_bce_f32", ctypes.byref(d), vp(users), vp(items), vp(labels), users.numel(), int(batch_size),
            -1 if max_steps is None else int(max_steps), vp(loss_full), vp(loss_ragged))
    self.t, m.dropout_step = d.t, d.dropout_step
    _bump(self.E0, self.W, loss_full, loss_ragged)


def _ngcf_step_one_call(self, users, items, labels, acc):
    """The single-layer step as one library call (spex_ngcf_step_bce_f32): the same launches, issued from native code."""
    import ctypes
    from . import _lib
    from .graph import _bump, _launch
    m = self.model
    B = users.numel()
    d = _ngcf_prepare_desc(self, B)
    ref_stream = getattr(self, "dropout_stream", getattr(m, "dropout_stream", "counter")) == "reference" and m.mess_dropout[0] > 0
    if ref_stream:
        # validation mode: the step's message-dropout noise is the REFERENCE's — nn.Dropout on the [N, 64] layer output
        # (main_rec.py:81) is at::dropout: empty_like(x).bernoulli_(1 - p) from the global CPU generator.  The same call on a
        # reused pinned buffer consumes the generator identically; the bytes go up and the kernels read them through
        # spex_ngcf_message_mask instead of drawing their counter-based mask.
        n_ref = m.n_users + m.n_items
        buf = getattr(self, "_noise_host", None)
        if buf is None or buf.shape[0] != n_ref:
            buf = self._noise_host = torch.empty((n_ref, 64), dtype=torch.float32).pin_memory()
        buf.bernoulli_(1.0 - float(m.mess_dropout[0]))
        self._msg_mask = (buf.to(self.E0.device) != 0).to(torch.uint8)
        _lib.call("spex_ngcf_message_mask", ctypes.c_void_p(self._msg_mask.data_ptr()))
    try:
        _launch(self.E0.device, "spex_ngcf_step_bce_f32", ctypes.byref(d), ctypes.c_void_p(users.data_ptr()),
                ctypes.c_void_p(items.data_ptr()), ctypes.c_void_p(labels.data_ptr()), B, ctypes.c_void_p(acc.data_ptr()))
    finally:
        if ref_stream:
            _lib.call("spex_ngcf_message_mask", None)
    self.t, m.dropout_step = d.t, d.dropout_step
    _bump(self.E0, self.W, acc)
    return acc


def _ngcf_step_deep_one_call(self, users, items, labels, acc):
    """L >= 2 layers as one library call (spex_ngcf_deep_step_bce_f32): whole-table forward of layers 0 .. L-2, the last layer at the
    batch's rows, scoring, rows backward + push, the dense layer backwards with their A^T products, the Adam passes — main_rec.py:71-93,
    116-128 for `--layer_size [64,64,..]`."""
    import ctypes
    from . import _lib
    from .graph import _bump, _launch
    m, L = self.model, self.L
    B = users.numel()
    d = self.E0.shape[1]
    if self.g_side_c is None or self.g_side_c.shape[0] < 2 * B or self.g_slots.shape[1] != d * (L + 1):
        dev = self.E0.device
        self.g_slots = torch.zeros((2 * B, d * (L + 1)), dtype=torch.float32, device=dev)
        self.g_side_c = torch.zeros((2 * B, d), dtype=torch.float32, device=dev)
        self.g_ego_c = torch.zeros_like(self.g_side_c)
        self.gW_parts = torch.zeros((ops.ngcf_bwd_rows_parts(2 * B), 2 * (d * d + d)), dtype=torch.float32, device=dev)
        self._deep_desc = None
    if getattr(self, "_deep_desc", None) is None:
        p = lambda t: t.data_ptr()
        self._p_drop = (ctypes.c_float * L)(*[float(x) for x in m.mess_dropout[:L]])
        self._deep_desc = _lib.NGCFDeepStepDesc(
            graph=m.graph._h.value, graph_t=m.graph_t._h.value, E0=p(self.E0), mE=p(self.mE), vE=p(self.vE), W=p(self.W), mW=p(self.mW),
            vW=p(self.vW), gW=p(self.gW), all_emb=p(self.all_emb), g_all=p(self.g_all), sides=p(self._sides_all), egos=p(self._egos_all),
            g_slots=p(self.g_slots), g_side_c=p(self.g_side_c), g_ego_c=p(self.g_ego_c), gW_parts=p(self.gW_parts), g_side=p(self.g_side),
            g_ego=p(self.g_ego), g_next=p(self._g_next_all), p_drop=ctypes.cast(self._p_drop, ctypes.c_void_p).value, L=L,
            slot_capacity=self.g_slots.shape[0], n_user_rows=self.n_u, pad_row=m.n_users, slope=0.01, seed=int(m.message_dropout_seed),
            dropout_step=m.dropout_step, t=self.t, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps)
    dsc = self._deep_desc
    for l in range(L):
        self._p_drop[l] = float(m.mess_dropout[l])
    dsc.t, dsc.lr, dsc.dropout_step, dsc.seed = self.t, self.lr, m.dropout_step, int(m.message_dropout_seed)
    _launch(self.E0.device, "spex_ngcf_deep_step_bce_f32", ctypes.byref(dsc), ctypes.c_void_p(users.data_ptr()),
            ctypes.c_void_p(items.data_ptr()), ctypes.c_void_p(labels.data_ptr()), B, ctypes.c_void_p(acc.data_ptr()))
    self.t, m.dropout_step = dsc.t, dsc.dropout_step
    _bump(self.E0, self.W, acc)
    return acc


NGCFStepper._one_call_ok = staticmethod(_ngcf_one_call_ok)
NGCFStepper._step_deep_one_call = _ngcf_step_deep_one_call
NGCFStepper._step_one_call = _ngcf_step_one_call
NGCFStepper.epoch = _ngcf_epoch
NGCFStepper.__del__ = _drop_desc


def epoch_arrays_ngcf(data):
    """One NGCF epoch's samples as the reference's loop meets them (main_rec.py:118-121): Data.sample_epoch (the `random` stream), then
    the DataLoader's shuffle (torch's global stream) — three host arrays."""
    us, vs, rs = data.sample_epoch()
    order = dataloader_epoch_order(len(us)).numpy()
    return us[order], vs[order], rs[order]


def train_epochs_ngcf(stepper, data, n_epochs, batch_size=None, after_epoch=None):
    """n_epochs x train_epoch_ngcf with the NEXT epoch's samples and shuffle prepared on a second host thread while the current epoch
    runs as one native call (see train_epochs: same generators, same order of draws, same run).  Returns the per-epoch loss sums."""
    import threading
    nxt = epoch_arrays_ngcf(data)
    totals = []
    for ep in range(n_epochs):
        arrays, box, th = nxt, {}, None
        if ep + 1 < n_epochs:
            def work():
                try:
                    box["v"] = epoch_arrays_ngcf(data)
                except BaseException as e:       # noqa: BLE001 — re-raised on the caller's thread
                    box["e"] = e
            th = threading.Thread(target=work)
            th.start()
        try:
            total = train_epoch_ngcf(stepper, data, batch_size=batch_size, arrays=arrays)
        finally:
            if th is not None:
                th.join()
        if "e" in box:
            raise box["e"]
        nxt = box.get("v")
        totals.append(total)
        if after_epoch is not None:
            after_epoch(ep, total)
    return [float(t) for t in totals]


def train_epoch_ngcf(stepper, data, batch_size=None, pause_gc=True, callbacks=None, step_losses=None, max_steps=None, arrays=None):
    """train() of NGCF_SPEX/code/main_rec.py:116-131 without the per-step host work of its DataLoader loop: the epoch's
    samples are drawn like the reference's (Data.sample_epoch: `random` stream), the sample order is the DataLoader's own
    (dataloader_epoch_order: global torch RNG), the shuffled epoch is moved to the device once, and every batch is one
    NGCFStepper.step.  Returns the epoch's summed per-batch mean loss (main_rec.py:129 accumulates the same sum).
    callbacks: {k: fn} — fn() is called in front of the epoch's k-th batch (e.g. a mid-epoch evaluation; it must leave the
    model in training mode); step_losses: a list that receives every step's mean loss (synchronises per step); max_steps: stop
    after that many batches.  (stepper.dropout_stream = "reference": the message-dropout noise is the reference's own per-step
    draw from the global generator — a validation mode, see NGCFStepper.)"""
    us, vs, rs = arrays if arrays is not None else epoch_arrays_ngcf(data)
    n = len(us)
    bs = batch_size or data.batch_size
    dev = stepper.E0.device
    users, items = torch.from_numpy(np.ascontiguousarray(us)).to(dev), torch.from_numpy(np.ascontiguousarray(vs)).to(dev)
    labels = torch.from_numpy(np.ascontiguousarray(rs)).to(dev)
    gc_was_on = pause_gc and gc.isenabled()
    if gc_was_on:
        gc.disable()
    n_full = n // bs * bs
    acc = torch.zeros(2, 1, dtype=torch.float32, device=dev)
    m_ = stepper.model
    native_ok = (not callbacks and step_losses is None and n > 0 and getattr(stepper, "L", 0) == 1 and hasattr(stepper, "epoch")
                 and stepper._one_call_ok(users[:1], items[:1], labels[:1])
                 and not (getattr(stepper, "dropout_stream", getattr(m_, "dropout_stream", "counter")) == "reference" and m_.mess_dropout[0] > 0))
    if native_ok:
        try:
            stepper.epoch(users, items, labels, bs, acc[0], acc[1], max_steps=max_steps)
        finally:
            if gc_was_on:
                gc.enable()
        n_done = n if max_steps is None else min(n, int(max_steps) * bs)
        ragged = n_done - n_done // bs * bs
        return acc[0, 0] / bs + (acc[1, 0] / ragged if ragged else 0.0)
    try:
        for k, s in enumerate(range(0, n, bs)):
            if max_steps is not None and k >= max_steps:
                break
            if callbacks and k in callbacks:
                callbacks[k]()
            e = min(s + bs, n)
            slot = acc[0] if e - s == bs else acc[1]
            if step_losses is None:
                stepper.step(users[s:e], items[s:e], labels[s:e], loss_acc=slot)
            else:
                tmp = torch.zeros(1, dtype=torch.float32, device=dev)
                stepper.step(users[s:e], items[s:e], labels[s:e], loss_acc=tmp)
                step_losses.append(tmp.item() / (e - s))
                slot += tmp
    finally:
        if gc_was_on:
            gc.enable()
    return acc[0, 0] / bs + (acc[1, 0] / (n - n_full) if n_full < n else 0.0)


class DualTaskStepper:
    """The dual-task training step (LightGCN_SPEX/code/main_auto_expert_s.py:63-89: rec branch + trust branch of
    utility1/model_expert_s.py, uncertainty-weighted loss, backward, torch Adam over every parameter) as ONE library call
    of 2 L + 7 launches on pre-allocated buffers (spex_dual_task_step_f32) — no autograd, no allocation, no host
    synchronisation.  ≈2.7 ms per step through the reference-shaped autograd path in round 1, ≈1.0 ms with the fused
    trust head under autograd, and the GPU time of the launches here.

    model: a `utility1.model_expert_s.LightGCN` on the GPU (hidden size 64).  Its parameters are re-homed into one flat arena
    [table | trust block | att_exp1 | att_exp2 | task_weights] and trained IN PLACE, so `model` can be evaluated
    (rec_test, trust_test5) or saved at any point.  Edge dropout: a model built with --dropout 1, set_edge_dropout per step.
    path_capacity: the largest number of paths a step may carry (3 x trust_batch_size in the reference driver, :70-71)."""

    def __init__(self, model, path_capacity, path_len, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, n_rec=5, batch_capacity=256,
                 two_streams=None, deterministic=None, fixed_task_weights=False, pipelined=False):
        """two_streams (default on): the trust branch's two launches — a latency chain on
        <= path_capacity workgroups — run on a second HIP stream beside the rec branch and join it in front of the Adam
        pass (spex_dual_task_step_t.side_stream).  Same results as the one-stream order.
        pipelined (needs two streams; flags & SPEX_STEP_PIPELINED): the Adam pass split by owner over the two streams, so that no
        fork / join sits on the trust branch's cycle.  The side stream then runs AHEAD of the current stream between steps: the
        path inputs of a step must be complete before the previous `join()`, and `join()` must come before anything but the
        next step touches the model, the moments or `loss_acc` (train_epoch_dual stages an epoch's inputs up front and, with
        SPEX_DUAL_PIPELINED=1, turns the mode on for its loop and joins at the end).  Worth it when the trust branch is the
        longer of the two; on Epinion2 with 15 paths the fork / join form is faster (93 vs 105 us per step).
        deterministic (default: SPEX_DETERMINISTIC=1 in the environment): no float atomics in the step (flags &
        SPEX_STEP_DETERMINISTIC).  fixed_task_weights: loss = loss1 + loss2 as in main_11.py:69 (flags &
        SPEX_STEP_FIXED_TASK_WEIGHTS; the task weights stay where they are)."""
        self.deterministic = (os.environ.get("SPEX_DETERMINISTIC", "0") == "1") if deterministic is None else bool(deterministic)
        self.fixed_task_weights = bool(fixed_task_weights)
        from . import _lib
        table = model.flat_table()
        assert table.is_cuda, "DualTaskStepper: the model must be on the GPU (no CPU fallback)"
        N, d = table.shape
        n_heads = len(model.in_att)
        if not ops.trust_head_supported(d, path_len, n_heads) or model.hidden_size != d:
            raise ValueError(f"DualTaskStepper needs hidden size 64, <= 16 path positions, <= 4 heads (got {d}, {path_len}, {n_heads})")
        dev = table.device
        self.model, self.dev = model, dev
        self.N, self.d, self.n_u, self.L = N, d, model.num_users + 1, model.n_layers
        self.path_capacity, self.path_len, self.n_heads = int(path_capacity), int(path_len), n_heads
        self.lr, self.betas, self.eps, self.n_rec = lr, betas, eps, n_rec
        if two_streams is None:
            two_streams = True
        self._side = torch.cuda.Stream(device=dev) if two_streams else None
        self.pipelined = bool(pipelined) and self._side is not None
        P = ops.trust_param_count(n_heads, d)
        self.n_trust = P
        total = N * d + P + 512 + 4
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.arena, self.m, self.v = z(total), z(total), z(total)
        # ---- re-home every parameter into the arena (same values, same Parameter objects)
        off = 0

        def place(param, n):
            nonlocal off
            view = self.arena[off: off + n].view(param.shape)
            view.copy_(param.data)
            param.data = view
            off += n

        u, i = model.embedding_user.weight, model.embedding_item.weight
        place(u, u.numel()); place(i, i.numel())
        assert off == N * d
        for t in model._trust_param_tensors():
            place(t, t.numel())
        assert off == N * d + P
        place(model.att_exp1, 256); place(model.att_exp2, 256); place(model.task_weights, 2)
        model._cache = None
        # ---- work buffers
        self.light, self.lo_batch = z(N, d), z(N, d)
        self.g_raw, self.g_prop, self.g_E0 = z(N, d), z(N, d), z(N, d)
        self.ws_fwd, self.ws_bwd = z(2, N, d), z(3, N, d)
        self.g_user, self.g_small = z(self.n_u, d), z(P + 512)
        self.slot_capacity = 0
        self._slots(2 * int(batch_capacity))
        T = self.path_capacity
        self.a2 = z(T, d)
        self.trust_ws = z(max(1, int(_lib.load().spex_trust_workspace_floats(T, self.path_len, d, n_heads, self.n_u))))
        self.dscore, self.loss_b = z(T, self.n_u - 1), z(T)
        self.loss, self.loss_acc, self.precision = z(2), z(2), z(2, 2)
        self.t = 0
        self._desc = None
        # the LightGCN adjacency is symmetric — but a MASKED one is not: a model built with --dropout 1 gets the transposed handle
        # (with its edge-id permutation) for the backward products, so that edge dropout can be set per step (set_edge_dropout)
        self._graph_t = model._transposed() if getattr(model.args_r, "dropout", 0) else model.Graph
        self.refresh_precision()

    def _slots(self, n):
        """Per-slot buffers of the rec branch (the batch's 2B gated rows, their gradients, the slot index list)."""
        if self.slot_capacity < n:
            z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.dev)
            self.mixed_slots, self.grad_slots, self.g_prop_slots = z(n, self.d), z(n, self.d), z(n, self.d)
            self.g_raw_slots, self.loss_rows = z(n, self.d), z(n)
            self.att_parts = z(int(_lib_mod().load().spex_expert_gate_rows_bwd_parts(n)) * 512)
            self.arange = torch.arange(n, dtype=torch.int64, device=self.dev)
            self.slot_capacity = n
            _drop_desc(self)

    def __del__(self):
        _drop_desc(self)

    def refresh_precision(self):
        """Call after changing task_weights from outside the stepper (the step keeps exp(-2 s) snapshots on the device)."""
        self.precision[(self.t + 1) & 1] = torch.exp(-2.0 * self.model.task_weights.detach())

    def step(self, users, items, labels, seq, seq_l, targets):
        """One training step.  users / items: device int64 [B]; labels: device fp32 [B]; seq: device int64 [T, path_len]
        padded with the pad row index; seq_l, targets: device int64 [T].  Both losses are added to `loss_acc`."""
        import ctypes
        from . import _lib
        from .graph import _bump, _launch
        B, T = users.numel(), (0 if seq is None else seq.shape[0])
        for t, dt in ((users, torch.int64), (items, torch.int64), (labels, torch.float32)):
            if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() == B):
                raise ValueError("DualTaskStepper.step: users / items (int64) and labels (fp32) must be contiguous device tensors of one length")
        if T:
            for t in (seq, seq_l, targets):
                if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
                    raise ValueError("DualTaskStepper.step: seq / seq_l / targets must be contiguous device int64 tensors")
            if seq.shape[1] != self.path_len or seq_l.numel() != T or targets.numel() != T or T > self.path_capacity:
                raise ValueError(f"DualTaskStepper.step: {T} paths of width {seq.shape[1]} (capacity {self.path_capacity} x {self.path_len})")
        if getattr(self.model.Graph, "mask_mode", 0) != 0 and (self._graph_t is self.model.Graph or self.L < 2):
            raise ValueError("DualTaskStepper.step: edge dropout needs the transposed handle (a model built with --dropout 1) and L >= 2")
        dsc = self._prepare_desc(B)
        if T and self._side is not None:                     # the side stream reads them: keep the allocator from recycling
            for t in (seq, seq_l, targets):                  # their memory under a step still in flight
                t.record_stream(self._side)
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
        _launch(self.dev, "spex_dual_task_step_f32", ctypes.byref(dsc), vp(users), vp(items), vp(labels), B,
                vp(seq) if T else None, vp(seq_l) if T else None, vp(targets) if T else None, T)
        self.t = dsc.t
        _bump(self.arena, self.m, self.v, self.loss_acc)
        self.model._cache = None

    def _prepare_desc(self, B):
        """The one-call step's descriptor for batches of up to B samples (built once, refreshed per call)."""
        from . import _lib
        self._slots(2 * B)
        if self._desc is None:
            p = lambda t: t.data_ptr()
            self._desc = _lib.DualTaskStepDesc(
                graph=self.model.Graph._h.value, graph_t=self._graph_t._h.value, params=p(self.arena), m=p(self.m), v=p(self.v),
                light=p(self.light), ws_fwd=p(self.ws_fwd), lo_batch=p(self.lo_batch), g_prop=p(self.g_prop), g_raw=p(self.g_raw),
                g_E0=p(self.g_E0), ws_bwd=p(self.ws_bwd), mixed_slots=p(self.mixed_slots), grad_slots=p(self.grad_slots),
                g_prop_slots=p(self.g_prop_slots), arange=p(self.arange), g_user=p(self.g_user), g_small=p(self.g_small),
                a2=p(self.a2), trust_ws=p(self.trust_ws), dscore=p(self.dscore), loss_b=p(self.loss_b),
                loss=p(self.loss), loss_acc=p(self.loss_acc), precision=p(self.precision), slot_capacity=self.slot_capacity,
                path_capacity=self.path_capacity, path_len=self.path_len, n_user_rows=self.n_u, L=self.L, d=self.d,
                n_heads=self.n_heads, hybrid=0 if self.model.nonhybrid else 1, n_rec=self.n_rec, lr=self.lr, beta1=self.betas[0],
                beta2=self.betas[1], eps=self.eps, t=self.t,
                side_stream=None if self._side is None else self._side.cuda_stream, ev_fork=None, ev_join=None,
                g_raw_slots=p(self.g_raw_slots), att_parts=p(self.att_parts), loss_rows=p(self.loss_rows), flags=0, side_pending=0)
        dsc = self._desc
        dsc.t, dsc.lr = self.t, self.lr
        dsc.flags = ((_lib.STEP_DETERMINISTIC if self.deterministic else 0) | (_lib.STEP_FIXED_TASK_WEIGHTS if self.fixed_task_weights else 0)
                     | (_lib.STEP_PIPELINED if self.pipelined and self._side is not None else 0))
        return dsc

    def epoch(self, users, items, labels, batch_size, seq, seq_l, targets, path_off, max_steps=None, keep_prob=1.0, drop_seed=0):
        """A whole pre-shuffled, device-resident epoch as ONE native call (spex_dual_task_epoch_f32; Train() of
        main_auto_expert_s.py:60-91): batch k = samples [k B, (k+1) B) with the staged paths [path_off[k], path_off[k+1]) (path_off: a
        host int64 array of n_batches + 1 offsets).  Both losses are added to `loss_acc`.  keep_prob < 1: the in-kernel sampled edge
        mask on the rec branch, a fresh one per step (seed (drop_seed << 32) | step)."""
        import ctypes
        import numpy as np
        from .graph import _bump, _launch
        for t, dt in ((users, torch.int64), (items, torch.int64), (labels, torch.float32)):
            if not (t.is_cuda and t.dtype == dt and t.is_contiguous() and t.numel() == users.numel()):
                raise ValueError("DualTaskStepper.epoch: users / items (int64) and labels (fp32) must be contiguous device tensors of one length")
        path_off = np.ascontiguousarray(path_off, dtype=np.int64)
        n_paths = int(path_off[-1])
        if n_paths:
            for t in (seq, seq_l, targets):
                if not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
                    raise ValueError("DualTaskStepper.epoch: seq / seq_l / targets must be contiguous device int64 tensors")
            if seq.shape[1] != self.path_len or seq.shape[0] < n_paths or seq_l.numel() < n_paths or targets.numel() < n_paths \
                    or int(np.diff(path_off).max()) > self.path_capacity:
                raise ValueError("DualTaskStepper.epoch: the staged paths do not match path_off / the stepper's capacity")
        n_batches = (users.numel() + int(batch_size) - 1) // int(batch_size)
        if len(path_off) < (n_batches if max_steps is None else min(n_batches, int(max_steps))) + 1:
            raise ValueError("DualTaskStepper.epoch: path_off needs one offset per batch + 1")
        if keep_prob < 1.0 and (self._graph_t is self.model.Graph or self.L < 2):
            raise ValueError("DualTaskStepper.epoch: edge dropout needs the transposed handle (a model built with --dropout 1) and L >= 2")
        dsc = self._prepare_desc(min(int(batch_size), users.numel()))
        if n_paths and self._side is not None:
            for t in (seq, seq_l, targets):
                t.record_stream(self._side)
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
        _launch(self.dev, "spex_dual_task_epoch_f32", ctypes.byref(dsc), vp(users), vp(items), vp(labels), users.numel(), int(batch_size),
                -1 if max_steps is None else int(max_steps), vp(seq) if n_paths else None, vp(seq_l) if n_paths else None,
                vp(targets) if n_paths else None, path_off.ctypes.data_as(ctypes.c_void_p), float(keep_prob), int(drop_seed) & 0xFFFFFFFF)
        self.t = dsc.t
        _bump(self.arena, self.m, self.v, self.loss_acc)
        self.model._cache = None

    def set_edge_dropout(self, mask=None):
        """The next steps' edge-dropout mask on both handles (an edge_dropout_mask(...) tuple; None: off) — model_expert_s.py:104-109."""
        g, gt = self.model.Graph, self._graph_t
        if mask is None:
            g.set_edge_mask(0)
            if gt is not g:
                gt.set_edge_mask(0)
            return
        if gt is g:
            raise ValueError("DualTaskStepper.set_edge_dropout: build the model with --dropout 1 (the stepper then keeps the transposed handle)")
        g.set_edge_mask(*mask)
        gt.set_edge_mask(*mask)

    def join(self):
        """Order everything a pipelined step left on the side stream in front of the current stream (spex_dual_task_step_join);
        a no-op for the other forms.  Call before reading the model, the moments or `loss_acc` and before changing them."""
        if self._desc is not None and self._desc.side_pending:
            import ctypes
            from .graph import _launch
            _launch(self.dev, "spex_dual_task_step_join", ctypes.byref(self._desc))


def _lib_mod():
    from . import _lib
    return _lib


def dual_task_epoch_paths(batch_users, by_user, cap):
    """The per-batch path selection of main_auto_expert_s.py:64-71, batch by batch in order: every path starting at one
    of the batch's users, cut to `cap` paths with random.sample (Python's global `random`, like the reference).  Returns the
    chosen path indices of every batch."""
    import random
    chosen_all = []
    for users in batch_users:
        chosen = []
        for u in set(users.tolist()):
            chosen.extend(by_user[u])
        if len(chosen) > cap:
            chosen = random.sample(chosen, cap)
        chosen_all.append(chosen)
    return chosen_all


def epoch_arrays_dual(train_data, trust_data, by_user, cap, batch_size=256, resample=True, max_steps=None):
    """One dual-task epoch's inputs as the reference's loop meets them (main_auto_expert_s.py:56-71): negatives (`ng_sample`: NumPy's
    global stream), the DataLoader's shuffle (torch's), every batch's paths cut to `cap` by random.sample (Python's `random`) — host
    arrays (users, items, labels, path inputs, path lengths, path targets) and the per-batch path lists."""
    import numpy as np
    users_h, items_h, labels_h = epoch_arrays(train_data, resample)
    starts = list(range(0, len(users_h), batch_size))
    if max_steps is not None:
        starts = starts[:max_steps]
    chosen = dual_task_epoch_paths([users_h[s:s + batch_size] for s in starts], by_user, cap)
    flat = np.fromiter((k for c in chosen for k in c), dtype=np.int64, count=sum(len(c) for c in chosen))
    inputs, mask, targets = trust_data.get_slice(flat)
    return (users_h, items_h, labels_h, np.ascontiguousarray(inputs, dtype=np.int64), np.asarray(mask).sum(1).astype(np.int64),
            np.asarray(targets).astype(np.int64), chosen)


def train_epochs_dual(stepper, train_data, trust_data, by_user, cap, n_epochs, batch_size=256, edge_dropout=None, after_epoch=None):
    """n_epochs x train_epoch_dual with the NEXT epoch's negatives, shuffle and path selection prepared on a second host thread while
    the current epoch runs as one native call (see train_epochs: the three generators are drawn by that thread alone meanwhile, in the
    order a sequential loop draws them — the same run).  Returns the per-epoch (loss1, loss2) sums."""
    import threading
    nxt = epoch_arrays_dual(train_data, trust_data, by_user, cap, batch_size)
    totals = []
    for ep in range(n_epochs):
        arrays, box, th = nxt, {}, None
        if ep + 1 < n_epochs:
            def work():
                try:
                    box["v"] = epoch_arrays_dual(train_data, trust_data, by_user, cap, batch_size)
                except BaseException as e:       # noqa: BLE001 — re-raised on the caller's thread
                    box["e"] = e
            th = threading.Thread(target=work)
            th.start()
        try:
            total = train_epoch_dual(stepper, train_data, trust_data, by_user, cap, batch_size=batch_size, edge_dropout=edge_dropout, arrays=arrays)
        finally:
            if th is not None:
                th.join()
        if "e" in box:
            raise box["e"]
        nxt = box.get("v")
        totals.append(total)
        if after_epoch is not None:
            after_epoch(ep, total)
    return [t.cpu().numpy() for t in totals]


def train_epoch_dual(stepper, train_data, trust_data, by_user, cap, batch_size=256, resample=True, pause_gc=True, max_steps=None,
                     cum_every=None, cum_out=None, n_paths_out=None, edge_dropout=None, arrays=None):
    """Train() of main_auto_expert_s.py:53-91 on the device: negatives drawn like the reference's (`ng_sample`), the
    epoch's sample order is the shuffled DataLoader's own, the per-batch paths are chosen by the reference's rule
    (dual_task_epoch_paths), everything is moved to the device once and every batch is one DualTaskStepper.step.
    Returns (sum of loss1, sum of loss2) over the epoch's steps as a device tensor.  cum_every / cum_out: every cum_every steps a
    copy of the running (loss1, loss2) sums is appended to the list cum_out (device tensors: no synchronisation); n_paths_out: a
    list that receives every step's path count.  edge_dropout: None, or (keep_prob, stream[, seed]) for `--dropout 1 --keepprob p`
    (README.md:119-123; stream "reference" replays the reference's per-step `torch.rand(nnz)`, "philox" draws in-kernel) — a fresh
    mask on the rec branch's handles per step; the model must have been built with --dropout 1."""
    import numpy as np
    if arrays is None:
        arrays = epoch_arrays_dual(train_data, trust_data, by_user, cap, batch_size, resample, max_steps)
    users_h, items_h, labels_h, seq_h, seq_l_h, tgt_h, chosen = arrays
    n = len(users_h)
    dev = stepper.dev
    starts = list(range(0, n, batch_size))
    if max_steps is not None:
        starts = starts[:max_steps]
        chosen = chosen[:max_steps]
    seq, seq_l, tgt = torch.from_numpy(seq_h).to(dev), torch.from_numpy(seq_l_h).to(dev), torch.from_numpy(tgt_h).to(dev)
    users = torch.from_numpy(users_h).to(dev)
    items = torch.from_numpy(items_h).to(dev)
    labels = torch.from_numpy(labels_h).to(device=dev, dtype=torch.float32)
    stepper.join()
    stepper.loss_acc.zero_()
    # the whole epoch as ONE native call where nothing has to happen on the host between two steps (no running sums asked for, no
    # host-drawn mask, not the pipelined form — whose side stream the loop below manages step by step)
    if (not cum_every and (edge_dropout is None or edge_dropout[1] == "philox") and hasattr(stepper, "epoch") and n > 0
            and not (stepper._side is not None and os.environ.get("SPEX_DUAL_PIPELINED", "0") == "1") and not stepper.pipelined):
        gc_was_on = pause_gc and gc.isenabled()
        if gc_was_on:
            gc.disable()
        try:
            path_off = np.zeros(len(chosen) + 1, dtype=np.int64)
            np.cumsum([len(c) for c in chosen], out=path_off[1:])
            kp = 1.0 if edge_dropout is None else float(edge_dropout[0])
            seed = 0 if edge_dropout is None or len(edge_dropout) < 3 else int(edge_dropout[2])
            stepper.epoch(users, items, labels, batch_size, seq, seq_l, tgt, path_off, max_steps=max_steps, keep_prob=kp, drop_seed=seed)
            if n_paths_out is not None:
                n_paths_out.extend(len(c) for c in chosen)
        finally:
            stepper.join()
            if gc_was_on:
                gc.enable()
        return stepper.loss_acc.clone()
    # the epoch's inputs are complete on the device before its first step and nothing but the steps touches the model inside the
    # loop: the pipelined form's contract (DualTaskStepper.__init__), so the loop MAY run in it — SPEX_DUAL_PIPELINED=1.  It pays
    # when the trust branch is the longer one (round 3's first measurement: 119 -> 102 us per step on Epinion2); with this round's
    # trust kernel (46 us against the rec branch's ~70) the fork / join form is the faster one again (93 vs 105 us), so that is the
    # default.  (A pipelined loop's first step forks the side stream from the current one, behind the uploads above.)
    was_pipelined = stepper.pipelined
    if stepper._side is not None and os.environ.get("SPEX_DUAL_PIPELINED", "0") == "1":
        stepper.pipelined = True
    gc_was_on = pause_gc and gc.isenabled()
    if gc_was_on:
        gc.disable()
    try:
        p0 = 0
        for k, (s, c) in enumerate(zip(starts, chosen)):
            e, p1 = min(s + batch_size, n), p0 + len(c)
            if edge_dropout is not None:
                stepper.set_edge_dropout(edge_dropout_mask(stepper.model.Graph, edge_dropout[0], edge_dropout[1],
                                                           edge_dropout[2] if len(edge_dropout) > 2 else 0, k + 1))
            stepper.step(users[s:e], items[s:e], labels[s:e], seq[p0:p1] if c else None, seq_l[p0:p1], tgt[p0:p1])
            p0 = p1
            if cum_every and cum_out is not None and (k + 1) % cum_every == 0:
                stepper.join()
                cum_out.append(stepper.loss_acc.clone())
        if n_paths_out is not None:
            n_paths_out.extend(len(c) for c in chosen)
    finally:
        stepper.join()
        if edge_dropout is not None:
            stepper.set_edge_dropout(None)
        stepper.pipelined = was_pipelined
        if gc_was_on:
            gc.enable()
    return stepper.loss_acc.clone()
