"""1-D row partition of the propagation across the GPUs of one node, one process per GPU.

Rank p owns a contiguous block of rows of A_hat (CSR slice, balanced by stored entries) and the same rows of every
layer's embedding matrix.  A layer is:  all-gather of the current layer's rows over xGMI (RCCL, through
torch.distributed)  ->  local SpMM on the rank's row block.  This is the schedule the reference's own `A_split` path
runs serially on one device (LightGCN_SPEX/code/utility1/dataloader.py:167-177, model.py:84-89).

Layout trick: shards are padded to the largest shard so the collective is a plain equal-size all-gather, and the
rank's column indices are rewritten once, at partition time, into that padded layout — the gathered buffer is used
by the SpMM as-is, no unpack pass.

Backward (G_l = g/(L+1) + A^T G_{l+1}) uses the same schedule on the row blocks of A^T.  Scoring is owner-computes:
every rank scores the (small, replicated) batch against the gathered output table and keeps the gradient rows it
owns, so the only exchange step on the data path is the per-layer all-gather.
"""
import numpy as np
import os

import torch
import torch.distributed as dist

from .graph import row_block


def balanced_row_bounds(rowptr, world, row_cost=None):
    """Row boundaries r_0=0 < ... < r_P=N giving each rank about the same `entries + row_cost * rows`.

    What a rank pays per layer: its SpMM (~264 B of gather traffic per stored entry at ~7 TB/s, plus ~512 B per output row) and
    the all-gather, whose equal-size form moves world * max_rows rows of 256 B at the collective's rate (xGMI: ~0.15-0.3 TB/s
    effective).  In entry units a row therefore costs ~2 for the SpMM alone — and ~45 * world once the exchange is counted
    (256 B * world / 0.15 TB/s against 264 B / 7 TB/s): with more than one rank the exchange dominates and the bounds should
    equalise ROWS (no padding in the gathered table: a [users; items] table balanced by entries alone has user shards a quarter
    the length of item shards, i.e. 1.6 x the bytes on the wire), accepting SpMM shards that differ ~2.4 x in entries.
    row_cost=None picks 4 for one rank and 45 * world otherwise; results do not depend on the bounds (tests pass explicit ones)."""
    if row_cost is None:
        row_cost = 4 if world == 1 else 45 * world
    n = len(rowptr) - 1
    cost = np.asarray(rowptr, np.int64) + row_cost * np.arange(n + 1, dtype=np.int64)
    targets = cost[-1] * np.arange(1, world, dtype=np.float64) / world
    inner = np.searchsorted(cost, targets, side="left")
    bounds = np.concatenate([[0], inner, [n]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


class RowPartition:
    """Static description of the partition (identical on every rank)."""

    def __init__(self, rowptr, world, bounds=None):
        self.n = len(rowptr) - 1
        self.world = world
        self.bounds = np.asarray(bounds if bounds is not None else balanced_row_bounds(rowptr, world), np.int64)
        assert len(self.bounds) == world + 1 and self.bounds[0] == 0 and self.bounds[-1] == self.n
        self.rows = np.diff(self.bounds)
        self.max_rows = int(max(1, self.rows.max()))
        self.n_padded = self.max_rows * world

    def owner(self, g):
        return np.searchsorted(self.bounds, g, side="right") - 1

    def to_padded(self, g):
        """Global row id -> position in the gathered, padded [world * max_rows, d] buffer."""
        g = np.asarray(g, np.int64)
        o = self.owner(g)
        return o * self.max_rows + (g - self.bounds[o])

    def to_padded_torch(self, g):
        b = torch.as_tensor(self.bounds[1:-1], device=g.device)
        o = torch.bucketize(g, b, right=True)
        return o * self.max_rows + (g - torch.as_tensor(self.bounds, device=g.device)[o])

    def local_block(self, rowptr, col, val, rank, edge_id=None):
        r0, r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        lrowptr, lcol, lval, leid = row_block(rowptr, col, val, r0, r1, edge_id)
        return lrowptr, self.to_padded(lcol).astype(np.int32), lval, leid


class PeerAllGather:
    """All-gather of equal row shards by direct peer writes: every rank copies its shard straight into its slot of every
    peer's gather buffer — world - 1 concurrent device-to-device copies on separate streams, one per xGMI link of the
    full mesh (7 links x ~50 GB/s usable on an 8-GPU node) — where a ring moves the same bytes through one link at a time
    (world - 1 sequential steps).  SURVEY.md 5 ("Distributed communication backend") asks for this schedule.

    The peers' buffers are reached through CUDA/HIP IPC handles exchanged once at construction (torch's own tensor
    sharing: hipIpcGetMemHandle / hipIpcOpenMemHandle underneath; HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool).
    Synchronisation needs no custom device code: a rank's copies are ordered before a one-element all-reduce on its
    compute stream, and that collective completes on any rank only after every rank has entered it — i.e. after every
    rank's copies have completed — so the SpMM queued behind it reads a complete table.  Two buffers alternate from call
    to call: a fast rank's copies for call c + 1 land in the other buffer while a slow rank is still reading call c's
    (it cannot get two calls ahead: it would have to pass the barrier of call c + 1, which the slow rank enters only
    after its SpMM of call c).
    """

    def __init__(self, n_padded, max_rows, d, rank, world, device, group=None):
        import torch.multiprocessing.reductions as reductions
        self.rank, self.world, self.group, self.max_rows = rank, world, group, max_rows
        # Every rank walks the same sequence of collectives whatever fails locally (exporting or opening an IPC handle can fail
        # on one rank only): a failure is agreed on with a MIN all-reduce and raised on ALL ranks, so the caller's fallback to
        # the collective path is taken everywhere — a rank that left early would leave the others waiting in a collective.
        err = None
        self.bufs, mine = None, None
        try:
            self.bufs = [torch.zeros((n_padded, d), dtype=torch.float32, device=device) for _ in range(2)]
            mine = [reductions.reduce_tensor(b) for b in self.bufs]
        except Exception as e:                               # noqa: BLE001
            err = e
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        self.peer = []                                   # peer[q][k]: rank q's buffer k as a tensor usable from this process
        if err is None and all(e is not None for e in everyone):
            try:
                for q in range(world):
                    self.peer.append(self.bufs if q == rank else [fn(*args) for fn, args in everyone[q]])
                self.streams = [torch.cuda.Stream(device=device) for _ in range(world)]
                self.token = torch.zeros(2, dtype=torch.float32, device=device)
                self.token_step = torch.tensor([1.0, -1.0], dtype=torch.float32, device=device)
            except Exception as e:                           # noqa: BLE001
                err = e
        elif err is None:
            err = RuntimeError("a peer could not export its buffers")
        ok = torch.tensor([0.0 if err is not None else 1.0], dtype=torch.float32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)   # (also the barrier: every handle is opened before anyone writes)
        if ok.item() != 1.0:
            self.peer = []
            raise RuntimeError("PeerAllGather: IPC set-up failed on at least one rank" + (f" (here: {err!r})" if err is not None else ""))
        self.calls = 0
        # SPEX_PEER_CHECK=1 (tests): the barrier's token carries every rank's call counter as (count, -count) under a MAX
        # all-reduce, and the host checks after each call that the largest and the smallest counter at the barrier are this
        # rank's own — i.e. that no rank was a call ahead of another when it wrote (the two-buffer argument above).  It costs a
        # host synchronisation per call, so it is off otherwise (the token is still reduced: that IS the barrier).
        self.check = os.environ.get("SPEX_PEER_CHECK", "0") == "1"
        self.checked = 0

    def all_gather(self, send, n_rows=None):
        """send: this rank's padded shard [max_rows, d]; n_rows: its real rows (default: all).  Only the real rows cross
        the links — the slots keep the padded stride, their tails stay zero as allocated and are never read — so the
        exchange moves N rows in total where the equal-size collective moves world * max_rows (1.6 x as many on a
        [users; items] table balanced by stored entries).  Returns the gathered [world * max_rows, d] table (one of the
        two alternating buffers), valid for kernels queued on the current stream after this call."""
        n_rows = self.max_rows if n_rows is None else int(n_rows)
        k = self.calls & 1
        self.calls += 1
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        lo = self.rank * self.max_rows
        for q in range(self.world):
            st = self.streams[q]
            st.wait_event(ready)
            with torch.cuda.stream(st):
                self.peer[q][k][lo: lo + n_rows].copy_(send[:n_rows], non_blocking=True)
            cur.wait_stream(st)
        self.token.add_(self.token_step)                 # (count, -count): equal on every rank after the previous barrier's MAX
        dist.all_reduce(self.token, op=dist.ReduceOp.MAX, group=self.group)    # the barrier described above (stream-ordered after the copies)
        if self.check:
            hi, neg_lo = self.token.tolist()
            if hi != float(self.calls) or -neg_lo != float(self.calls):
                raise RuntimeError(f"PeerAllGather: ranks met at the barrier of call {self.calls} with counters {int(-neg_lo)}..{int(hi)}")
            self.checked += 1
        return self.bufs[k]


class NativeComm:
    """libspexhip's own communicator (spex_comm_*: RCCL bound inside the library, include/spex_hip.h) — the collectives of the
    partitioned path issued from native code instead of through torch.distributed.  The 128-byte id is created on rank 0 and
    handed to the other ranks by whatever the host has: here a torch.distributed object broadcast over `group` (gloo or nccl),
    which is only the bootstrap — no tensor of the data path goes through it afterwards.  Creation is collective.
    One process per GPU (RCCL refuses two ranks on one device); world size 1 works anywhere."""

    def __init__(self, rank, world, device, group=None, unique_id=None):
        import ctypes
        from . import _lib
        lib = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        if unique_id is None:
            buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
            if rank == 0:
                _lib.check(lib.spex_comm_unique_id(buf), "spex_comm_unique_id")
            box = [bytes(buf.raw)]
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            unique_id = box[0]
        assert len(unique_id) == _lib.COMM_ID_BYTES
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.spex_comm_create(self.rank, self.world, unique_id, ctypes.byref(self._h)), "spex_comm_create")
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.spex_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def allgather_rows(self, send, recv, max_rows, rows_per_rank=None):
        """recv [world * max_rows, d] <- every rank's send [<= max_rows, d]; rows_per_rank (a ctypes int32 array or None)."""
        from .graph import _bump, _launch, _ptr
        _launch(self.device, "spex_comm_allgather_rows_f32", self._h, _ptr(send), _ptr(recv), int(max_rows), send.shape[1], rows_per_rank)
        _bump(recv)
        return recv

    def allreduce_sum(self, buf):
        from .graph import _bump, _launch, _ptr
        _launch(self.device, "spex_comm_allreduce_sum_f32", self._h, _ptr(buf), buf.numel())
        _bump(buf)
        return buf


class PartitionedLightGCN:
    """LightGCN propagation + scoring step on one rank of a row-partitioned graph.

    graph_factory(rowptr, col, val, n_cols=...) builds the local device graph (SpexGraph on a GPU; the CPU tests of
    the schedule inject a stand-in with the same .spmm signature).  `group` is a torch.distributed process group
    (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests).
    """

    def __init__(self, rowptr, col, val, n_user_rows, n_layers, d, rank, world, graph_factory, device, group=None,
                 t_csr=None, bounds=None, always_collective=False, allgather="collective", edge_ids=False):
        self.part = RowPartition(rowptr, world, bounds)
        self.rank, self.world, self.L, self.d, self.group = rank, world, n_layers, d, group
        self.always_collective = always_collective      # issue the collectives even at world size 1 (backend smoke tests)
        self.n_user_rows = n_user_rows
        self.device = torch.device(device)
        p = self.part
        # edge_ids (edge dropout, utility1/model.py:46-64): the blocks carry their entries' GLOBAL edge ids — the entry's index in A;
        # for the block of A^T the transposition's permutation — so that a mask (injected, indexed by edge id; or sampled, keyed by
        # it) drops the same edges of A and of A^T on every rank.  A masked operator is not symmetric: graph_t is then a handle of
        # its own even for a symmetric A.  graph_factory must accept edge_id=.
        self.edge_ids = bool(edge_ids)
        if self.edge_ids:
            from .graph import csr_transpose
            lr, lc, lv, le = p.local_block(rowptr, col, val, rank, np.arange(len(col), dtype=np.int32))
            self.graph = graph_factory(lr, lc, lv, n_cols=p.n_padded, edge_id=le)
            full_t = csr_transpose(rowptr, col, val, len(rowptr) - 1) if t_csr is None else t_csr
            if len(full_t) < 4 or full_t[3] is None:
                raise ValueError("PartitionedLightGCN(edge_ids=True): t_csr must carry the edge-id permutation (graph.csr_transpose)")
            tr, tc, tv, te = p.local_block(*full_t[:3], rank, full_t[3])
            self.graph_t = graph_factory(tr, tc, tv, n_cols=p.n_padded, edge_id=te)
            self._t_block = (tr, tc, tv)
            self._t_block_eid = te
            self.nnz_global = int(len(col))
        else:
            lr, lc, lv, _ = p.local_block(rowptr, col, val, rank)
            self.graph = graph_factory(lr, lc, lv, n_cols=p.n_padded)
            if t_csr is None:       # symmetric adjacency: A^T == A
                self.graph_t = self.graph
                self._t_block = (lr, lc, lv)
            else:
                tr, tc, tv, _ = p.local_block(*t_csr[:3], rank)
                self.graph_t = graph_factory(tr, tc, tv, n_cols=p.n_padded)
                self._t_block = (tr, tc, tv)
        self._graph_factory, self._graph_push, self._tables = graph_factory, None, {}
        self._edge_mask = (0, None, 1.0, 0)            # the mask last set (re-applied to the push structure when it is built)
        self.r0, self.r1 = int(p.bounds[rank]), int(p.bounds[rank + 1])
        self.n_local = self.r1 - self.r0
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.device)
        self.gathered = z(p.n_padded, d)                 # all-gather target == SpMM gather source
        self.send = z(p.max_rows, d)                     # padded local shard
        self.light_out = z(self.n_local, d)
        self.out_gathered = z(p.n_padded, d)
        # allgather: "collective" = torch.distributed's all_gather_into_tensor (RCCL on the GPU); "peer" = direct peer
        # writes through IPC-mapped buffers (PeerAllGather) for the per-layer exchange
        # "native" / "native-p2p" = libspexhip's own RCCL communicator (NativeComm): the plain equal-size collective, or the
        # grouped point-to-point form that moves the real rows only
        self.peer, self.use_peer = None, False
        self.native, self.use_native, self._rows_per_rank = None, False, None
        if allgather not in ("collective", "peer", "native", "native-p2p"):
            raise ValueError("allgather must be 'collective', 'peer', 'native' or 'native-p2p'")
        if allgather != "collective":
            self.set_allgather(allgather)

    def set_edge_mask(self, mode=0, keep=None, keep_prob=1.0, seed=0):
        """The step's edge-dropout mask on both blocks (SpexGraph.set_edge_mask's arguments; trainer.edge_dropout_mask(...) builds
        them — for an injected mask `keep` is indexed by GLOBAL edge id, length = the whole matrix's stored entries).  Every rank
        must set the same mask.  mode 0 clears it."""
        if not self.edge_ids and mode != 0:
            raise ValueError("PartitionedLightGCN.set_edge_mask: build the model with edge_ids=True")
        self._edge_mask = (mode, keep, keep_prob, seed)
        self.graph.set_edge_mask(mode, keep, keep_prob, seed)
        if self.graph_t is not self.graph:
            self.graph_t.set_edge_mask(mode, keep, keep_prob, seed)
        if self._graph_push is not None and self.edge_ids:   # (the fast path's push reads it entry by entry with the same keep rule)
            self._graph_push.set_edge_mask(mode, keep, keep_prob, seed)

    # -- what the one-call native steps need beside the Python-issued schedule's buffers
    def table(self, k):
        """The k-th further [n_padded, d] exchange table (all-zero when first asked for): the native steps exchange IN PLACE — a layer's
        SpMM writes into the rank's own slot of the table the next exchange completes — and alternate between `gathered` and these."""
        if k not in self._tables:
            self._tables[k] = torch.zeros_like(self.gathered)
        return self._tables[k]

    def push_graph(self):
        """The (n_padded x n_local) transpose of the rank's block of A^T: row p holds A[p, c] for the columns c this rank owns — the
        structure the backward's first product walks in push form (every rank holds the batch's gradient rows, so that product needs
        no exchange: include/spex_hip.h, spex_partitioned_dual_step_t).  Built on first use."""
        if self._graph_push is None:
            import numpy as np
            import scipy.sparse as sp
            tr, tc, tv = self._t_block
            # (transposed through the entries' INDICES, so that values and — under edge dropout — global edge ids follow their entry)
            idx = sp.csr_matrix((np.arange(1, len(tc) + 1, dtype=np.int64), np.asarray(tc), np.asarray(tr)),
                                shape=(self.n_local, self.part.n_padded)).T.tocsr()          # (stored entries kept one by one: nothing is summed)
            idx.sort_indices()
            perm = idx.data - 1
            kw = {"edge_id": np.ascontiguousarray(np.asarray(self._t_block_eid)[perm]).astype(np.int32)} if self.edge_ids else {}
            self._graph_push = self._graph_factory(idx.indptr.astype(np.int64), idx.indices.astype(np.int32),
                                                   np.ascontiguousarray(np.asarray(tv, np.float32)[perm]), n_cols=self.n_local, **kw)
            if self.edge_ids:
                self._graph_push.set_edge_mask(*self._edge_mask)
        return self._graph_push

    # -- the one exchange step of the data path
    def all_gather_rows(self, local, out=None):
        if local.data_ptr() != self.send.data_ptr():      # a layer's SpMM writes straight into the send buffer
            self.send[: self.n_local].copy_(local)
        if self.use_peer and out is None:
            return self.peer.all_gather(self.send, self.n_local)
        out = self.gathered if out is None else out
        if self.use_native:
            return self.native.allgather_rows(self.send, out, self.part.max_rows, self._rows_per_rank)
        if self.world == 1 and not self.always_collective:
            out.copy_(self.send)
        else:
            dist.all_gather_into_tensor(out, self.send, group=self.group)
        return out

    def set_allgather(self, mode):
        """Switch the per-layer exchange between "collective" and "peer" (every rank must make the same call: building the
        peer path exchanges IPC handles).  At world size 1 both are a local copy."""
        if mode in ("native", "native-p2p"):
            if self.native is None:
                self.native = NativeComm(self.rank, self.world, self.device, group=self.group)
            import ctypes
            self._rows_per_rank = (ctypes.c_int32 * self.world)(*[int(r) for r in self.part.rows]) if mode == "native-p2p" else None
            self.use_native, self.use_peer = True, False
            return
        self.use_native = False
        if mode == "peer" and self.world > 1:
            if self.peer is None:
                self.peer = PeerAllGather(self.part.n_padded, self.part.max_rows, self.d, self.rank, self.world, self.device,
                                          group=self.group)
            self.use_peer = True
        elif mode in ("collective", "peer"):
            self.use_peer = False
        else:
            raise ValueError("allgather must be 'collective', 'peer', 'native' or 'native-p2p'")

    def propagate(self, E0_local, keep_first=False, first_gathered=None):
        """mean_l(A^l E0) for this rank's rows (LightGCN.computer(), model.py:66-97).  keep_first: the first layer's
        all-gathered table (= E0 of every rank, padded layout) is gathered into `self.gathered0` and survives the call
        (the dual-task model reads the whole user block from it).  first_gathered: that table, kept up to date by the caller
        (gather_first once, refresh_first_rows after every sparse update) — layer 0's exchange is then skipped."""
        cur = E0_local
        if keep_first and getattr(self, "gathered0", None) is None:
            self.gathered0 = torch.zeros_like(self.gathered)
        if self.use_native and self.L >= 1 and E0_local.is_cuda:
            return self._propagate_in_place(E0_local, keep_first, first_gathered)
        for l in range(self.L):
            if l == 0 and first_gathered is not None:
                X = first_gathered
            else:
                X = self.all_gather_rows(cur, out=self.gathered0 if (keep_first and l == 0) else None)
            last = l == self.L - 1
            nxt = None if last else self.send[: self.n_local]     # next layer's all-gather source, no copy
            self.graph.spmm(X, Y=nxt, acc_in=E0_local if l == 0 else self.light_out, acc_out=self.light_out,
                            acc_div=float(self.L + 1) if last else 1.0)
            cur = nxt
        if self.L == 0:
            self.light_out.copy_(E0_local)
            if keep_first:
                self.all_gather_rows(E0_local, out=self.gathered0)
        return self.light_out

    # -- the native exchange forms IN PLACE (what the one-call native steps do, comm.hip: exchange_in_place): a layer's SpMM writes its
    #    output into the rank's own slot of the table the NEXT exchange completes — `gathered` and table(1) alternate — and the
    #    collective sends from the slot it also receives around (ncclAllGather's in-place form; the point-to-point form sends the real
    #    rows from where they lie): no send buffer, no copy of a layer's rows.
    def _own_slot(self, table):
        lo = self.rank * self.part.max_rows
        return table[lo: lo + self.part.max_rows]

    def _exchange_in_place(self, table, local=None):
        if local is not None:
            table[self.rank * self.part.max_rows: self.rank * self.part.max_rows + self.n_local].copy_(local)
        return self.native.allgather_rows(self._own_slot(table), table, self.part.max_rows, self._rows_per_rank)

    def _propagate_in_place(self, E0_local, keep_first, first_gathered):
        T = [self.gathered, self.table(1)]
        lo = self.rank * self.part.max_rows
        for l in range(self.L):
            if l == 0:
                X = first_gathered if first_gathered is not None else \
                    self._exchange_in_place(self.gathered0 if keep_first else T[1], E0_local)
            else:
                X = self._exchange_in_place(T[(l - 1) & 1])
            last = l == self.L - 1
            nxt = None if last else T[l & 1][lo: lo + self.n_local]       # layer l's output: own slot of the next exchange's table
            self.graph.spmm(X, Y=nxt, acc_in=E0_local if l == 0 else self.light_out, acc_out=self.light_out,
                            acc_div=float(self.L + 1) if last else 1.0)
        return self.light_out

    def _propagate_bwd_in_place(self, gs, grad_out):
        T = [self.gathered, self.table(1)]
        lo = self.rank * self.part.max_rows
        for k, l in enumerate(range(self.L - 1, -1, -1)):
            X = self._exchange_in_place(T[k & 1], gs if k == 0 else None)
            nxt = grad_out if l == 0 else T[(k + 1) & 1][lo: lo + self.n_local]
            self.graph_t.spmm(X, Y=nxt, add_in=gs, add_div=1.0)
        return grad_out

    def propagate_bwd(self, g_local, grad_out=None):
        """d loss / d E0 (local rows) from d loss / d light_out (local rows).  No allocation after the first call: the
        scaled gradient g/(L+1) lives in a per-model buffer, every intermediate layer is written straight into the
        all-gather's send buffer."""
        if getattr(self, "_gs", None) is None or self._gs.shape != g_local.shape or self._gs.device != g_local.device:
            self._gs = torch.empty_like(g_local)
        gs = torch.div(g_local, float(self.L + 1), out=self._gs)
        if grad_out is None:
            grad_out = torch.empty_like(gs)
        if self.use_native and self.L >= 1 and gs.is_cuda:
            return self._propagate_bwd_in_place(gs, grad_out)
        cur = gs
        for l in range(self.L - 1, -1, -1):
            X = self.all_gather_rows(cur)
            nxt = grad_out if l == 0 else self.send[: self.n_local]
            self.graph_t.spmm(X, Y=nxt, add_in=gs, add_div=1.0)
            cur = nxt
        if self.L == 0:
            grad_out.copy_(gs)
        return grad_out

    # -- E0 replicated in its gathered form and kept current by DELTAS: a step that changes few rows of E0 (the BPR-SGD update
    #    touches <= 3 T rows) need not all-gather the whole table for the next step's first layer — the owners publish the
    #    post-update values of the touched rows (the same small owner-computes exchange that fetches a batch's rows) and
    #    every rank patches its copy.  One all-gather of the table per step less.
    def gather_first(self, E0_local):
        if getattr(self, "gathered0", None) is None:
            self.gathered0 = torch.zeros_like(self.gathered)
        return self.all_gather_rows(E0_local, out=self.gathered0)

    def refresh_first_rows(self, padded_pos, E0_local, scratch):
        """gathered0[padded_pos] = the owners' current rows of E0 (scratch: [len(padded_pos), d]; repeated positions receive the
        same value twice)."""
        rows = self.fetch_rows_at(padded_pos, scratch, table=E0_local)
        self.gathered0.index_copy_(0, padded_pos, rows)
        return self.gathered0

    def gather_output(self):
        """Full (padded) propagated table on every rank, for scoring."""
        return self.all_gather_rows(self.light_out, out=self.out_gathered)

    def plan_rows(self, padded_pos):
        """Which of the given padded positions this rank owns: (indices into padded_pos, local row indices).  Compute
        once per batch (it sizes tensors from data, i.e. synchronises)."""
        owner = torch.div(padded_pos, self.part.max_rows, rounding_mode="floor")
        own_idx = torch.nonzero(owner == self.rank).reshape(-1)
        return own_idx, padded_pos[own_idx] - self.rank * self.part.max_rows

    def fetch_rows(self, plan, out, table=None):
        """Rows of the propagated table at a batch's positions, on every rank: each rank writes the rows it owns into
        the zeroed buffer and an all-reduce adds the contributions up — a few MB per batch instead of an all-gather of
        the whole table."""
        own_idx, local = plan
        out.zero_()
        out.index_copy_(0, own_idx, (self.light_out if table is None else table).index_select(0, local))
        if self.world > 1 or self.always_collective:
            dist.all_reduce(out, group=self.group)
        return out

    def fetch_rows_at(self, padded_pos, out, table=None):
        """fetch_rows() without a plan: on the GPU one launch writes the owned rows and zero-fills the others
        (spex_gather_owned_rows_f32), then the same all-reduce; on CPU tensors (the gloo tests of the schedule) the
        plan-based tensor ops.  table: the rank's rows of the table to read (default: the propagated table)."""
        table = self.light_out if table is None else table
        if out.is_cuda:
            from . import ops
            ops.gather_owned_rows(table, padded_pos, self.rank * self.part.max_rows, out)
            if self.use_native:
                self.native.allreduce_sum(out)
            elif self.world > 1 or self.always_collective:
                dist.all_reduce(out, group=self.group)
            return out
        return self.fetch_rows(self.plan_rows(padded_pos), out, table=table)

    def add_owned_rows(self, upd, padded_pos, table_local, clear=True):
        """table_local[owned rows of padded_pos] += upd rows (and clear upd): the owner-computes update."""
        if upd.is_cuda:
            from . import ops
            return ops.scatter_add_owned_rows(upd, padded_pos, self.rank * self.part.max_rows, table_local, clear)
        own_idx, local = self.plan_rows(padded_pos)
        table_local.index_add_(0, local, upd.index_select(0, own_idx))
        if clear:
            upd.zero_()
        return table_local

    def padded_index(self, users, items):
        """Batch indices (user ids, item ids) -> rows of the padded gathered table."""
        return self.part.to_padded_torch(users), self.part.to_padded_torch(items + self.n_user_rows)

    def own_slice(self, padded):
        s = self.rank * self.part.max_rows
        return padded[s: s + self.n_local]


class PartitionedStepper:
    """The exact reference training step (BCE, backward through the propagation, Adam — main_rec.py:32-37) on a
    row-partitioned model.  Every rank holds its rows of E0 and of the Adam moments; the batch is replicated; per step
    the exchanges are the per-layer all-gathers (forward and backward) plus one all-reduce of the batch's 2B
    propagated rows.  Gradient rows are computed everywhere and each rank keeps the ones it owns (owner-computes).

    A step is a fixed sequence of launches and collectives on pre-allocated buffers — no host synchronisation, no
    allocation: L x (all-gather, SpMM), gather-owned-rows, all-reduce, scoring, scatter-add-owned-rows, one scaling
    pass, L x (all-gather, SpMM), Adam (which also clears the gradient table for the next step).
    `ops`: the kernel namespace (spex_amd.ops on the GPU; the CPU tests of the schedule inject a stand-in)."""

    def __init__(self, part_model, E0_local, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, ops=None, fast=True):
        self.fast = bool(fast)          # the one-call native step's fast path (False: its launch-by-launch schedule) — every rank alike
        if ops is None:
            from . import ops
        self.ops, self.P, self.E0 = ops, part_model, E0_local
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.m, self.v = torch.zeros_like(E0_local), torch.zeros_like(E0_local)
        self.g_local = torch.zeros_like(E0_local)          # invariant: all-zero between steps (cleared by the Adam pass)
        self.grad_E0 = torch.zeros_like(E0_local)
        self._B = -1

    def _buffers(self, B, dev):
        if B != self._B:
            d = self.P.d
            self.rows = torch.zeros((2 * B, d), dtype=torch.float32, device=dev)
            self.grad_rows = torch.zeros((2 * B, d), dtype=torch.float32, device=dev)   # all-zero between steps too
            ar = torch.arange(B, device=dev)
            self.ar_u, self.ar_i = ar, ar + B
            self._B = B

    def positions(self, users, items):
        """The batch's rows in the padded gathered layout ([users | items], int64 on the model's device).  A loop that
        knows its batches ahead (an epoch's shuffled samples) computes these once for the whole epoch."""
        dev = self.E0.device
        pu, pi = self.P.padded_index(users.to(dev), items.to(dev))
        return torch.cat([pu, pi])

    def _native_step(self, pos, labels, B, loss_acc, deterministic):
        """The whole step as ONE library call (spex_partitioned_step_bce_f32): the exchanges go through the model's NativeComm,
        nothing is issued from Python between the launches."""
        import ctypes
        from . import _lib
        from .graph import _bump, _launch
        P = self.P
        if getattr(self, "_desc", None) is None or self._desc_B != self._B:
            z = lambda *s: torch.zeros(s, dtype=torch.float32, device=self.E0.device)
            if getattr(self, "_gs", None) is None:
                self._gs = z(*self.E0.shape)
            self._arange = torch.arange(2 * self._B, dtype=torch.int64, device=self.E0.device)
            p = lambda t: t.data_ptr()
            self._desc = _lib.PartitionedStepDesc(
                graph=P.graph._h.value, graph_t=P.graph_t._h.value, comm=P.native._h.value,
                rows_per_rank=None if P._rows_per_rank is None else ctypes.cast(P._rows_per_rank, ctypes.c_void_p).value,
                E0=p(self.E0), m=p(self.m), v=p(self.v), light_out=p(P.light_out), g_local=p(self.g_local), gs=p(self._gs),
                grad_E0=p(self.grad_E0), gathered1=p(P.table(1)), gathered=p(P.gathered), rows=p(self.rows), grad_rows=p(self.grad_rows),
                arange=p(self._arange), n_local=P.n_local, max_rows=P.part.max_rows, slot_capacity=2 * self._B, L=P.L, d=P.d,
                lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, t=self.t, flags=0,
                graph_push=P.push_graph()._h.value if self.fast and P.n_local > 0 and P.L >= 2 else None,
                gathered2=p(P.table(2)) if self.fast else None)
            self._desc_B = self._B
        d = self._desc
        d.t, d.lr, d.flags = self.t, self.lr, (_lib.STEP_DETERMINISTIC if deterministic else 0)
        # the exchange form may have been switched since the descriptor was built (set_allgather): refreshed like t / lr / flags; the
        # ctypes array is kept alive by the stepper for as long as the descriptor points at it
        self._rpr_keep = P._rows_per_rank
        d.rows_per_rank = None if P._rows_per_rank is None else ctypes.cast(P._rows_per_rank, ctypes.c_void_p).value
        d.comm = P.native._h.value
        _launch(self.E0.device, "spex_partitioned_step_bce_f32", ctypes.byref(d), ctypes.c_void_p(pos.data_ptr()),
                ctypes.c_void_p(labels.data_ptr()), B, ctypes.c_void_p(loss_acc.data_ptr()))
        self.t = d.t
        _bump(self.E0, self.m, self.v, loss_acc, P.light_out)

    def step_bce(self, users, items, labels, pos=None, loss_acc=None, deterministic=False):
        """One training step.  Returns the batch's mean BCE loss (device tensor) — or, with `loss_acc` (a 1-element device
        buffer), accumulates the loss SUM into it and returns None.  With the model's exchange set to "native" / "native-p2p"
        (PartitionedLightGCN.set_allgather) the whole step is one native call; deterministic selects its atomic-free mode."""
        P, ops = self.P, self.ops
        dev = self.E0.device
        B = users.numel()
        self._buffers(B, dev)
        if pos is None:
            pos = self.positions(users, items)
        if getattr(P, "use_native", False) and P.d == 64 and P.L >= 1 and self.E0.is_cuda:
            acc = loss_acc if loss_acc is not None else torch.zeros(1, dtype=torch.float32, device=dev)
            self._native_step(pos.contiguous(), labels.to(device=dev, dtype=torch.float32).contiguous(), B, acc, deterministic)
            return None if loss_acc is not None else acc / B
        P.propagate(self.E0)
        rows = P.fetch_rows_at(pos, self.rows)
        _, loss_sum = ops.score_bce(rows, rows, self.ar_u, self.ar_i, labels.to(dev), self.grad_rows, self.grad_rows, 1.0 / B,
                                    loss_sum=loss_acc, want_gamma=False)
        P.add_owned_rows(self.grad_rows, pos, self.g_local, clear=True)
        P.propagate_bwd(self.g_local, grad_out=self.grad_E0)
        self.t += 1
        ops.adam_step(self.E0, self.grad_E0, self.m, self.v, self.t, self.lr, self.betas[0], self.betas[1], self.eps,
                      zero=self.g_local)
        return None if loss_acc is not None else loss_sum / B
