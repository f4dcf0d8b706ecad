"""Run in a child process with SPEX_RCCL_LIB = the recording stub (tests/stubs/rccl_record_stub.c): drives libspexhip's native
exchange for world sizes a one-GPU box cannot host and asserts the exact sequence of RCCL calls it issues.  No byte crosses a
wire — this pins the peer / count / offset / stream / grouping arithmetic of spex_amd/csrc/comm.hip, not RCCL.

  python rccl_stub_driver.py cpu    no GPU needed: the rank's rows already sit in its own slot (send == recv + rank * slot), so
                                    the library issues no device copy and the pointers are never dereferenced
  python rccl_stub_driver.py gpu    world = 4, rank = 2 on cuda:0: both all-gather forms incl. the own-slot copy, the all-reduce,
                                    spex_partitioned_propagate_f32, spex_partitioned_step_bce_f32 and the dual-task step (in-place
                                    exchanges, 2L - 1 (fast path) or 2L of them + one all-reduce per step, the caller's stream)

The schedule's reference analogue: the serial fold loop of --A_split, LightGCN_SPEX/code/utility1/model.py:84-89."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

OP = dict(unique_id=1, init=2, destroy=3, allgather=4, allreduce=5, send=6, recv=7, group_start=8, group_end=9)
NCCL_FLOAT32, NCCL_SUM = 7, 0


class Rec(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int), ("peer", ctypes.c_int), ("dtype", ctypes.c_int), ("group_depth", ctypes.c_int),
                ("count", ctypes.c_longlong), ("send", ctypes.c_void_p), ("recv", ctypes.c_void_p), ("stream", ctypes.c_void_p)]


class Stub:
    def __init__(self):
        self.lib = ctypes.CDLL(os.environ["SPEX_RCCL_LIB"])       # the same instance comm.hip dlopen()s
        self.lib.stub_log_get.argtypes = [ctypes.c_int, ctypes.POINTER(Rec)]

    def take(self):
        out = []
        for i in range(self.lib.stub_log_count()):
            r = Rec()
            assert self.lib.stub_log_get(i, ctypes.byref(r)) == 0
            out.append(r)
        self.lib.stub_log_clear()
        return out


def expect_p2p(log, rank, world, rows, max_rows, d, send, recv, stream):
    """One grouped exchange: GroupStart, then per peer q != rank in ascending order Send(own rows -> q) and Recv(q's rows -> slot q),
    all inside the group, GroupEnd.  Ranks without rows are neither sent to ... nor received from."""
    slot_bytes = max_rows * d * 4
    want = [("group_start",)]
    for q in range(world):
        if q == rank:
            continue
        if rows[rank]:
            want.append(("send", q, rows[rank] * d, send))
        if rows[q]:
            want.append(("recv", q, rows[q] * d, recv + q * slot_bytes))
    want.append(("group_end",))
    assert len(log) == len(want), ([(r.op, r.peer, r.count) for r in log], want)
    for r, w in zip(log, want):
        assert r.op == OP[w[0]], (r.op, w)
        if w[0] == "group_start":
            assert r.group_depth == 0
        elif w[0] == "group_end":
            assert r.group_depth == 0                      # (logged after the depth went back down)
        else:
            assert r.group_depth == 1 and r.peer == w[1] and r.count == w[2] and r.dtype == NCCL_FLOAT32, (r.peer, r.count, w)
            assert (r.send if w[0] == "send" else r.recv) == w[3], (w, r.send, r.recv)
            assert (r.stream or 0) == stream
    n_send = sum(1 for r in log if r.op == OP["send"])
    n_recv = sum(1 for r in log if r.op == OP["recv"])
    return n_send, n_recv


def cpu():
    from spex_amd import _lib
    lib = _lib.load()
    stub = Stub()
    idb = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    _lib.check(lib.spex_comm_unique_id(idb))
    assert idb.raw == b"\x5a" * 128 and [r.op for r in stub.take()] == [OP["unique_id"]]
    rng = np.random.default_rng(5)
    d, STREAM = 64, 0x7000
    for world in (2, 4, 8):
        for rank in range(world):
            rows = [int(x) for x in rng.integers(1, 900, world)]
            if world >= 4:
                rows[(rank + 1) % world] = 0                          # a peer without rows
            max_rows = max(rows)
            h = ctypes.c_void_p()
            _lib.check(lib.spex_comm_create(rank, world, idb, ctypes.byref(h)))
            r = stub.take()
            assert len(r) == 1 and r[0].op == OP["init"] and r[0].peer == rank and r[0].count == world
            table = np.zeros((world * max_rows, d), np.float32)       # never dereferenced by the stub
            recv = table.ctypes.data
            send = recv + rank * max_rows * d * 4                     # in place: the rank's rows already sit in its slot
            rpr = (ctypes.c_int32 * world)(*rows)
            _lib.check(lib.spex_comm_allgather_rows_f32(h, send, recv, max_rows, d, rpr, STREAM))
            ns, nr = expect_p2p(stub.take(), rank, world, rows, max_rows, d, send, recv, STREAM)
            assert ns == world - 1 - 0 and nr == sum(1 for q in range(world) if q != rank and rows[q])
            # a rank WITHOUT rows still receives everybody else's
            rows0 = list(rows)
            rows0[rank] = 0
            _lib.check(lib.spex_comm_allgather_rows_f32(h, send, recv, max_rows, d, (ctypes.c_int32 * world)(*rows0), STREAM))
            ns, nr = expect_p2p(stub.take(), rank, world, rows0, max_rows, d, send, recv, STREAM)
            assert ns == 0
            # equal padded shards: ONE ncclAllGather of max_rows * d floats per rank, outside any group
            _lib.check(lib.spex_comm_allgather_rows_f32(h, send, recv, max_rows, d, None, STREAM))
            lg = stub.take()
            assert len(lg) == 1 and lg[0].op == OP["allgather"] and lg[0].count == max_rows * d and lg[0].send == send \
                and lg[0].recv == recv and lg[0].stream == STREAM and lg[0].dtype == NCCL_FLOAT32 and lg[0].group_depth == 0
            # all-reduce: in place, float32 sum
            _lib.check(lib.spex_comm_allreduce_sum_f32(h, recv, 12345, STREAM))
            lg = stub.take()
            assert len(lg) == 1 and lg[0].op == OP["allreduce"] and lg[0].peer == NCCL_SUM and lg[0].count == 12345 \
                and lg[0].send == recv and lg[0].recv == recv and lg[0].stream == STREAM
            # bad row counts are refused before anything is issued
            bad = (ctypes.c_int32 * world)(*[max_rows + 1] * world)
            assert lib.spex_comm_allgather_rows_f32(h, send, recv, max_rows, d, bad, STREAM) == -1 and stub.take() == []
            # a failing send inside the group: an error, and the group is CLOSED again (an open group would swallow every later call)
            stub.lib.stub_fail(OP["send"], 0)
            assert lib.spex_comm_allgather_rows_f32(h, send, recv, max_rows, d, rpr, STREAM) == -3
            assert b"ncclSend" in lib.spex_last_error() and stub.lib.stub_group_depth() == 0
            lg = stub.take()
            assert lg[0].op == OP["group_start"] and lg[-1].op == OP["group_end"] and sum(1 for x in lg if x.op == OP["send"]) == 1
            stub.lib.stub_fail(OP["recv"], 0)
            assert lib.spex_comm_allgather_rows_f32(h, send, recv, max_rows, d, rpr, STREAM) == -3 and stub.lib.stub_group_depth() == 0
            stub.take()
            _lib.check(lib.spex_comm_destroy(h))
            assert [x.op for x in stub.take()] == [OP["destroy"]]
    print("rccl stub (cpu): ok")


def gpu():
    import torch
    from spex_amd import _lib
    from spex_amd.dist import NativeComm, PartitionedLightGCN, PartitionedStepper
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    lib = _lib.load()
    stub = Stub()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(8)
    n_user, n_item, L, d, world, rank = 600, 1400, 3, 64, 4, 2
    uu, ii = rng.integers(0, n_user, 12000), rng.integers(0, n_item, 12000)
    pairs = np.unique(np.stack([uu, ii], 1), axis=0)
    csr = lightgcn_norm_adj(pairs[:, 0], pairs[:, 1], n_user, n_item)
    n = len(csr[0]) - 1
    bounds = np.array([0, 300, 601, 1500, n])                       # uneven: 300 / 301 / 899 / rest
    P = PartitionedLightGCN(*csr, n_user + 1, L, d, rank, world,
                            lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev, bounds=bounds)
    idb = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    _lib.check(lib.spex_comm_unique_id(idb))
    P.native = NativeComm(rank, world, dev, unique_id=bytes(idb.raw))
    stub.take()
    rows = [int(r) for r in P.part.rows]
    max_rows, n_loc = P.part.max_rows, P.n_local
    assert rows[rank] == n_loc == 899 and max_rows == 899
    side = torch.cuda.Stream(device=dev)
    E0_local = torch.randn(n_loc, d, device=dev) * 0.1
    recv, send = P.gathered.data_ptr(), P.send.data_ptr()
    own = slice(rank * max_rows, rank * max_rows + n_loc)
    with torch.cuda.stream(side):
        # ---- both all-gather forms from a buffer that is NOT the slot: the rank's rows are copied into its own slot, and every
        #      RCCL call carries the caller's stream
        P.set_allgather("native-p2p")
        P.gathered.zero_()
        P.all_gather_rows(E0_local)
        ns, nr = expect_p2p(stub.take(), rank, world, rows, max_rows, d, send, recv, side.cuda_stream)
        assert (ns, nr) == (3, 3)
        side.synchronize()
        assert torch.equal(P.gathered[own], E0_local) and float(P.gathered[: own.start].abs().sum()) == 0.0
        P.set_allgather("native")
        P.all_gather_rows(E0_local)
        lg = stub.take()
        assert len(lg) == 1 and lg[0].op == OP["allgather"] and lg[0].count == max_rows * d and lg[0].send == send and lg[0].recv == recv \
            and lg[0].stream == side.cuda_stream
        buf = torch.ones(777, device=dev)
        P.native.allreduce_sum(buf)
        lg = stub.take()
        assert len(lg) == 1 and lg[0].op == OP["allreduce"] and lg[0].count == 777 and lg[0].send == buf.data_ptr() == lg[0].recv \
            and lg[0].stream == side.cuda_stream
        # ---- the one-call partitioned propagation and training step
        for mode in ("native-p2p", "native"):
            P.set_allgather(mode)
            st = PartitionedStepper(P, E0_local.clone(), lr=1e-3)
            B = 64
            users = torch.from_numpy(rng.integers(0, n_user, B)).to(dev)
            items = torch.from_numpy(rng.integers(0, n_item, B)).to(dev)
            labels = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(dev)
            acc = torch.zeros(1, device=dev)
            for det in (False, True):
                st.step_bce(users, items, labels, loss_acc=acc, deterministic=det)
                lg = stub.take()
                assert all((r.stream or 0) == side.cuda_stream for r in lg if r.op not in (OP["group_start"], OP["group_end"]))
                reduces = [k for k, r in enumerate(lg) if r.op == OP["allreduce"]]
                assert len(reduces) == 1 and lg[reduces[0]].count == 2 * B * d and lg[reduces[0]].send == st.rows.data_ptr()
                before, after = lg[: reduces[0]], lg[reduces[0] + 1:]
                # every exchange is IN PLACE: the rows go out from the rank's own slot of the table they are received around (the
                # layer's SpMM wrote them there; E^0 and the scaled gradient are copied in), and the two tables alternate
                tables = [recv, P.table(1).data_ptr()]
                slot_of = lambda t: t + rank * max_rows * d * 4
                if det:     # launch by launch: L exchanges forward, L backward (2L + one all-reduce per step), the two tables alternating
                    fwd = bwd = [tables[k & 1] for k in range(L)]
                else:       # fast path: E^0 through table 1, then alternating; the backward's first product is the push — NO exchange —,
                            # its L - 1 pull products start from the push target's table (2L - 1 + one all-reduce per step)
                    fwd = [tables[(k + 1) & 1] for k in range(L)]
                    bwd = [P.table(2).data_ptr()] + [tables[k & 1] for k in range(L - 2)]
                if mode == "native-p2p":
                    per = 2 + 3 + 3                                      # GroupStart, 3 sends, 3 receives, GroupEnd
                    assert len(before) == len(fwd) * per and len(after) == len(bwd) * per, (det, len(before), len(after))
                    for k, tb in enumerate(fwd):
                        expect_p2p(before[k * per:(k + 1) * per], rank, world, rows, max_rows, d, slot_of(tb), tb, side.cuda_stream)
                    for k, tb in enumerate(bwd):
                        expect_p2p(after[k * per:(k + 1) * per], rank, world, rows, max_rows, d, slot_of(tb), tb, side.cuda_stream)
                else:
                    assert len(before) == len(fwd) and len(after) == len(bwd)
                    assert all(r.op == OP["allgather"] and r.count == max_rows * d for r in before + after)
                    assert [(r.send, r.recv) for r in before] == [(slot_of(tb), tb) for tb in fwd]
                    assert [(r.send, r.recv) for r in after] == [(slot_of(tb), tb) for tb in bwd]
            side.synchronize()
            assert st.t == 2 and bool(torch.isfinite(st.E0).all()) and bool(torch.isfinite(acc).all())
            # switching the exchange form takes effect on the NEXT step of the same stepper (the descriptor is refreshed)
            other = "native" if mode == "native-p2p" else "native-p2p"
            P.set_allgather(other)
            st.step_bce(users, items, labels, loss_acc=acc)
            lg = stub.take()
            n_ag = sum(1 for r in lg if r.op == OP["allgather"])
            n_sr = sum(1 for r in lg if r.op in (OP["send"], OP["recv"]))
            assert (n_ag, n_sr) == ((2 * L - 1, 0) if other == "native" else (0, (2 * L - 1) * 6)), (other, n_ag, n_sr)
        # spex_partitioned_propagate_f32 on its own: L exchanges, nothing else
        P.set_allgather("native-p2p")
        st = PartitionedStepper(P, E0_local.clone(), lr=1e-3)
        st._buffers(64, dev)
        st.step_bce(users, items, labels, loss_acc=acc)
        stub.take()
        from spex_amd.graph import _launch
        _launch(dev, "spex_partitioned_propagate_f32", ctypes.byref(st._desc))
        lg = stub.take()
        assert len(lg) == L * 8 and sum(1 for r in lg if r.op == OP["allreduce"]) == 0
        # ---- the one-call partitioned DUAL-TASK step (spex_partitioned_dual_task_step_f32): ONE all-reduce of the batch's rows of E^0
        #      and of the propagated table together (4B rows) — no gate-gradient collective; the first exchange's table is the kept
        #      one (gathered0), the trust branch issues no collective.  Fast path: L exchanges in front of the all-reduce, L - 1
        #      behind it (the backward's first product is the push: no exchange), the first of them on gathered2 (the push target's
        #      table); launch-by-launch schedule (fast=False): L + L
        import argparse
        sys.path.insert(0, os.path.join(REPO, "spex_amd", "dropin"))
        import utility1.model_expert_s as mex
        from spex_amd.dist_dual import PartitionedDualTask, PartitionedDualTaskStepper

        class _DS:
            n_users, m_items = n_user, n_item
            getSparseGraph = staticmethod(lambda: None)
        dargs = argparse.Namespace(hiddenSize=64, batchSize=100, nonhybrid=False, nb_heads=3, recdim=64, layer=L, keepprob=0.6, A_split=False,
                                   dropout=0)
        torch.manual_seed(1)
        core = mex.LightGCN(dargs, _DS).to(dev)
        dmodel = PartitionedDualTask(core, csr, rank, world, dev)        # (its own PartitionedLightGCN; default row bounds)
        dcomm = NativeComm(rank, world, dev, unique_id=bytes(idb.raw))
        seq = torch.tensor([[1, 2, n_user, n_user], [3, 4, 5, n_user], [7, n_user, n_user, n_user]], dtype=torch.int64, device=dev)
        seq_l = torch.tensor([2, 3, 1], dtype=torch.int64, device=dev)
        tgt = torch.tensor([9, 10, 11], dtype=torch.int64, device=dev)
        for fast in (True, False):
            dst = PartitionedDualTaskStepper(dmodel, path_capacity=5, path_len=4, exchange="native-p2p", two_streams=False, comm=dcomm, fast=fast)
            stub.take()
            DP = dmodel.P
            drows = [int(r) for r in DP.part.rows]
            dmax = DP.part.max_rows
            dst.step(users, items, labels, seq, seq_l, tgt)
            lg = stub.take()
            reduces = [k for k, r in enumerate(lg) if r.op == OP["allreduce"]]
            assert len(reduces) == 1 and lg[reduces[0]].count == 4 * B * d and lg[reduces[0]].send == dst.rows.data_ptr(), [(r.op, r.count) for r in lg]
            before, after = lg[: reduces[0]], lg[reduces[0] + 1:]
            per = 2 + 3 + 3
            tb = [DP.gathered.data_ptr(), DP.table(1).data_ptr()]
            fwd = [dst.gathered0.data_ptr()] + [tb[(l - 1) & 1] for l in range(1, L)]           # E^0 -> gathered0, then the layers' tables
            bwd = ([DP.table(2).data_ptr()] + [tb[k & 1] for k in range(L - 2)]) if fast else [tb[k & 1] for k in range(L)]
            assert len(before) == len(fwd) * per and len(after) == len(bwd) * per, (fast, len(before), len(after))
            for k, t in enumerate(fwd):
                expect_p2p(before[k * per:(k + 1) * per], rank, world, drows, dmax, d, t + rank * dmax * d * 4, t, side.cuda_stream)
            for k, t in enumerate(bwd):
                expect_p2p(after[k * per:(k + 1) * per], rank, world, drows, dmax, d, t + rank * dmax * d * 4, t, side.cuda_stream)
            side.synchronize()
            assert dst.t == 1 and bool(torch.isfinite(dst.arena).all()) and bool(torch.isfinite(dst.loss_acc).all())
        DP.native.close()
        stub.take()
    side.synchronize()
    P.native.close()
    assert [x.op for x in stub.take()] == [OP["destroy"]]
    print("rccl stub (gpu): ok")


if __name__ == "__main__":
    {"cpu": cpu, "gpu": gpu}[sys.argv[1]]()
