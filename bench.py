#!/usr/bin/env python3
"""Headline benchmark: LightGCN-SPEX graph convolution + BPR step on MI355X (BASELINE.json's metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch, inputs resident in HBM:
    3-layer propagation over the normalised user-item graph (3 SpMM launches, layer mean fused)
  + one fused BPR gather+dot+sigmoid+SGD kernel over T = 2048 triples scored on the propagated table.
N = 1: the Epinion2 graph (BASELINE configs[1]: N = 15 593 nodes, nnz = 418 608, d = 64).
N > 1: weak scaling — "Epinion2 x N": N replicas, the N copies of every interaction joined across replicas by a
random permutation (every node keeps its Epinion2 degree), 1-D row partition, RCCL all-gather of the layer's rows
before each SpMM.

value = graph-conv edges/s = L * nnz * K / t over the whole job; BPR triples/s of the same timed region and the
stand-alone kernel rates ride along in "extra".
"roofline" is the SpMM kernel timed with hipEvents INSIDE the timed region — on Epinion2, whose 4 MB table is cache-resident:
its `frac` (algorithmic bytes / time / 8 TB/s) is a cache-bandwidth figure, NOT an HBM utilisation; it is also reported
as `cache_algorithmic_frac`, next to the L2-miss traffic a rocprofv3 --pmc pass measured for the same launch (a stored
profile figure: profiles/hbm_traffic.json) and its ratio to the compulsory bytes.
"roofline_hbm" repeats the measurement on the SURVEY.md 8d graph (N = 2^24 nodes, 4.3 GB table >> 256 MB Infinity
Cache): that is where an HBM-roofline fraction means something.
"cpu_baseline": the C restatement of the reference's algorithm (OpenMP) on the host cores — all cores and one thread,
forward step and the exact training step — plus what the reference literally executes (stock torch.sparse.mm), each on
a bounded sample.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); ~6.3 TB/s achievable
L, D, T_TRIPLES = 3, 64, 2048
BPR_BYTES_PER_TRIPLE = 1548    # 12 B of indices + 3 rows read + 3 rows written


def algorithmic_bytes(nnz, n_rows, d=D):
    """SURVEY.md 8d gather model: per entry int32 col + fp32 val + one gathered fp32 row; per row int32 rowptr + one
    written fp32 row."""
    return nnz * (4 + 4 + 4 * d) + n_rows * (4 + 4 * d)


def compulsory_bytes(nnz, n_rows, d=D):
    """every matrix entry, every table row read once and written once."""
    return nnz * 8 + n_rows * (4 + 8 * d)


def time_events(fn, iters, finish=None):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    if finish is not None:      # e.g. a pipelined stepper's join: the side stream's last launches are inside the timed region
        finish()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-roofline", action="store_true")
    ap.add_argument("--no-standalone", action="store_true")
    ap.add_argument("--hbm-log2-nodes", type=int, default=24)     # the scaled roofline graph is N = 2^24
    ap.add_argument("--no-strong-scaling", action="store_true")   # N > 1: skip the strong-scaling leg on the 2^24-node graph
    a = ap.parse_args()
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner on stdout when its first communicator is created (the
    # N > 1 runs; the world-size-1 communicator of `extra.partitioned_one_call_steps_world1`): keep the real stdout for the line and
    # point file descriptor 1 at stderr for everything a library may print in between.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # Rehearsal hook for a one-GPU box: SPEX_BENCH_BACKEND=gloo SPEX_BENCH_SHARE_GPU=1 runs the N-rank code path with
    # every rank on cuda:0 (collectives through gloo).  The driver's real runs use one rank per GPU over RCCL.
    share = os.environ.get("SPEX_BENCH_SHARE_GPU") == "1"
    # SPEX_BENCH_FORCE_PARTITIONED=1 (tests): run the N > 1 code path — row partition, exchange selection, strong-scaling leg —
    # at world size 1 through the real `nccl` backend, the closest a one-GPU box gets to the driver's multi-GPU runs
    part = world > 1 or os.environ.get("SPEX_BENCH_FORCE_PARTITIONED") == "1"
    backend = os.environ.get("SPEX_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if part:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from spex_amd import ops
    from spex_amd.datasets import epinion2_replicated, xavier_uniform_np
    from spex_amd.graph import SpexGraph, lightgcn_norm_adj
    from spex_amd.trainer import LightGCNStepper

    uu, ii, n_user, m_item = epinion2_replicated(world, seed=2020)   # identical on every rank (CPU generator)
    rowptr, col, val = lightgcn_norm_adj(uu.numpy(), ii.numpy(), n_user, m_item)
    n_nodes, nnz = len(rowptr) - 1, len(col)
    n_u = n_user + 1
    rng = np.random.default_rng(2020)
    E0_host = np.concatenate([xavier_uniform_np(n_u, D, rng), xavier_uniform_np(m_item, D, rng)])
    trng = np.random.default_rng(2021)
    tu = torch.from_numpy(trng.integers(0, n_user, T_TRIPLES)).to(dev)
    tp = torch.from_numpy(trng.integers(0, m_item, T_TRIPLES)).to(dev)
    tn = torch.from_numpy(trng.integers(0, m_item, T_TRIPLES)).to(dev)
    lr = 1e-3

    if not part:
        graph = SpexGraph(rowptr, col, val, device=dev)
        stepper = LightGCNStepper(graph, torch.from_numpy(E0_host).to(dev), n_u, n_layers=L, lr=lr)
        local_nnz, local_rows = nnz, n_nodes

        def step():
            return stepper.step_bpr_sgd(tu, tp, tn)
    else:
        from spex_amd.dist import PartitionedLightGCN
        P = PartitionedLightGCN(rowptr, col, val, n_u, L, D, rank, world,
                                lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev)
        graph = P.graph
        E0_local = torch.from_numpy(E0_host[P.r0:P.r1].copy()).to(dev)
        # The per-layer exchange: RCCL's all-gather, or direct peer writes over the xGMI mesh (spex_amd.dist.PeerAllGather).
        # SPEX_ALLGATHER = collective | peer | auto (default): auto builds the peer path, checks that it gathers the same
        # table, times ten propagations each way and keeps the faster; any failure leaves the collective in place.
        ag_mode = os.environ.get("SPEX_ALLGATHER", "auto")
        ag_info = {"requested": ag_mode, "used": "collective"}
        if ag_mode in ("peer", "auto"):
            try:
                ref = P.propagate(E0_local).clone()
                P.set_allgather("peer")
                same = torch.equal(P.propagate(E0_local), ref)
                ok = torch.tensor([1.0 if same else 0.0], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if ok.item() != 1.0:
                    raise RuntimeError("peer all-gather produced a different table")
                times = {}
                for mode in ("collective", "peer"):
                    P.set_allgather(mode)
                    for _ in range(3):
                        P.propagate(E0_local)
                    torch.cuda.synchronize(); dist.barrier()
                    t_ = time.perf_counter()
                    for _ in range(10):
                        P.propagate(E0_local)
                    torch.cuda.synchronize()
                    tt_ = torch.tensor([time.perf_counter() - t_], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    times[mode] = tt_.item() / 10
                use = "peer" if (ag_mode == "peer" or times["peer"] < times["collective"]) else "collective"
                P.set_allgather(use)
                ag_info.update(used=use, propagate_ms_collective=times["collective"] * 1e3, propagate_ms_peer=times["peer"] * 1e3)
            except Exception as e:      # noqa: BLE001 — the collective path always works
                P.use_peer = False
                ag_info["peer_error"] = repr(e)[:300]
        # The same exchange behind the C ABI (spex_comm_*: RCCL bound inside libspexhip, include/spex_hip.h): the equal-size
        # collective ("native") and the grouped point-to-point form that moves the real rows only ("native-p2p").  Both are
        # checked against the table the torch.distributed path gathered and timed like the peer path; the fastest verified
        # schedule runs the timed region.  SPEX_BENCH_NATIVE=0 skips this; one GPU shared by several ranks (the rehearsal
        # mode) cannot host an RCCL communicator and skips it too.
        # (SPEX_RCCL_LIB set: the library binds a stand-in — tests/stubs/rccl_shm_stub.c moves the data between ranks that share a GPU —
        #  so the rehearsal can run this selection code with real data before the driver's multi-GPU run does)
        if os.environ.get("SPEX_BENCH_NATIVE", "1") != "0" and ((not share and backend == "nccl") or os.environ.get("SPEX_RCCL_LIB")):
            native_info = {}
            try:
                import ctypes
                from spex_amd import _lib as _slib
                probe = ctypes.create_string_buffer(_slib.COMM_ID_BYTES)
                ok = torch.tensor([1.0 if _slib.load().spex_comm_unique_id(probe) == 0 else 0.0], device=dev)   # librccl loads here?
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if ok.item() != 1.0:
                    raise RuntimeError("librccl could not be bound on every rank")
                keep_mode = "peer" if P.use_peer else "collective"
                best = ag_info.get("propagate_ms_" + keep_mode)
                P.set_allgather("collective")
                ref = P.propagate(E0_local).clone()
                if best is None:
                    for _ in range(3):
                        P.propagate(E0_local)
                    torch.cuda.synchronize(); dist.barrier()
                    t_ = time.perf_counter()
                    for _ in range(10):
                        P.propagate(E0_local)
                    torch.cuda.synchronize()
                    tt_ = torch.tensor([time.perf_counter() - t_], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    best = tt_.item() / 10 * 1e3
                    ag_info["propagate_ms_collective"] = best
                use = keep_mode
                for mode in ("native", "native-p2p"):
                    P.set_allgather(mode)
                    same = torch.equal(P.propagate(E0_local), ref)
                    ok = torch.tensor([1.0 if same else 0.0], device=dev)
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                    native_info[mode + "_equal"] = ok.item() == 1.0
                    if ok.item() != 1.0:
                        continue
                    for _ in range(3):
                        P.propagate(E0_local)
                    torch.cuda.synchronize(); dist.barrier()
                    t_ = time.perf_counter()
                    for _ in range(10):
                        P.propagate(E0_local)
                    torch.cuda.synchronize()
                    tt_ = torch.tensor([time.perf_counter() - t_], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                    ms_ = tt_.item() / 10 * 1e3
                    native_info["propagate_ms_" + mode] = ms_
                    if ms_ < best:
                        best, use = ms_, mode
                P.set_allgather(use)
                ag_info["used"] = use
            except Exception as e:      # noqa: BLE001 — torch.distributed's path stays in place
                native_info["error"] = repr(e)[:300]
                try:
                    P.set_allgather("peer" if ag_info.get("used") == "peer" else "collective")
                except Exception:       # noqa: BLE001
                    P.use_native = False
            ag_info["native"] = native_info
        pu, pp = P.padded_index(tu, tp)
        _, pn = P.padded_index(tu, tn)
        pos_all = torch.cat([pu, pp, pn])
        fetched = torch.zeros(3 * T_TRIPLES, D, device=dev)
        upd = torch.zeros(3 * T_TRIPLES, D, device=dev)
        ar = torch.arange(T_TRIPLES, device=dev)
        cu, cp, cn = ar, ar + T_TRIPLES, ar + 2 * T_TRIPLES        # compact ids into `fetched`
        loss_acc = torch.zeros(1, device=dev)
        local_nnz, local_rows = graph.nnz, graph.n_rows

        # The BPR-SGD update touches <= 3 T rows of E0, so the next step's FIRST layer does not need a fresh all-gather of the
        # whole table: every rank keeps the gathered E0 and the owners publish the touched rows' new values (the same small
        # exchange that fetches the batch's rows).  Two table all-gathers + two row exchanges per step instead of three + one.
        # SPEX_BENCH_DELTA=0 restores the plain schedule; the replica is checked against a real all-gather before the timed
        # region (below) and the plain schedule is used if it ever differs.
        delta = {"on": os.environ.get("SPEX_BENCH_DELTA", "1") != "0"}
        delta_info = {}
        refreshed = torch.zeros(3 * T_TRIPLES, D, device=dev)
        if delta["on"]:
            P.gather_first(E0_local)

        def step():
            P.propagate(E0_local, first_gathered=P.gathered0 if delta["on"] else None)
            # owner-computes: the (replicated) batch's rows are exchanged (one launch + one small all-reduce), every
            # rank scores the batch and applies the updates of the rows it owns (one launch) — no gradient exchange
            rows = P.fetch_rows_at(pos_all, fetched)
            ops.bpr_sgd_step(rows, rows, upd, upd, cu, cp, cn, lr, 0.0, loss_sum=loss_acc, grouped=False)
            P.add_owned_rows(upd, pos_all, E0_local, clear=True)      # leaves `upd` all-zero for the next step
            if delta["on"]:
                P.refresh_first_rows(pos_all, E0_local, refreshed)
            return loss_acc

    def barrier():
        if part:
            dist.barrier()
        torch.cuda.synchronize()

    # Python's cyclic GC: a full collection walks every object torch / scipy / numpy created at import — ~40 ms, which
    # lands once inside any loop longer than ~900 steps (tools/stall_probe.py: 100-step chunks take 2.9 ms of host time,
    # the chunk with the collection 40-44 ms).  The step loops create no cycles: move everything allocated so far out
    # of the collector's reach, as a long-running training loop would — before ANY measurement below.
    gc.collect()
    gc.freeze()

    # ---- stand-alone rates of the pieces (same process, same data).  They run BEFORE the timed region: every bench run
    # makes them anyway, and in front they also bring the GPU to its steady clocks before the K timed steps (the
    # driver's 20-step timed region is 1 ms long).
    aux = {}
    if not part and not a.no_standalone:
        try:
            for _ in range(300):                      # bring the GPU to its steady clocks first (the set-up above left it idle)
                stepper.propagate()
            aux["propagate_only_ms"] = time_events(stepper.propagate, 200)
            aux["propagate_only_edges_per_s"] = L * nnz / (aux["propagate_only_ms"] * 1e-3)
            lo = stepper.light_out
            Tb = 1 << 20

            def bpr_rate(u_, p_, n_, grouped):
                fn = lambda: ops.bpr_sgd_step(lo[:n_u], lo[n_u:], stepper.E0[:n_u], stepper.E0[n_u:], u_, p_, n_, 1e-6, 0.0,
                                              grouped=grouped)
                fn()
                ms = time_events(fn, 20)
                return {"triples_per_s": u_.numel() / (ms * 1e-3), "ms": ms, "T": u_.numel(),
                        "algorithmic_GBs": u_.numel() * BPR_BYTES_PER_TRIPLE / (ms * 1e-3) / 1e9,
                        "frac_of_hbm_peak": u_.numel() * BPR_BYTES_PER_TRIPLE / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            bu = torch.randint(0, n_user, (Tb,), device=dev)
            bp = torch.randint(0, m_item, (Tb,), device=dev)
            bn = torch.randint(0, m_item, (Tb,), device=dev)
            # triples in the sampler's order (5 negatives per training pair, pairs sorted by user — dataloader.py:250-265)
            order = np.lexsort((ii.numpy(), uu.numpy()))
            su = torch.from_numpy(np.repeat(uu.numpy()[order], 5)).to(dev)
            sp_ = torch.from_numpy(np.repeat(ii.numpy()[order], 5)).to(dev)
            sn_ = torch.randint(0, m_item, (su.numel(),), device=dev)
            aux["bpr_roofline"] = {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_triple": BPR_BYTES_PER_TRIPLE,
                "kernel": "grouped form (bpr_count / scan / bin_scatter / bucket_update<users> / <items>), T = 2^20 on "
                          "Epinion2's tables; the tables are cache-resident, so this is an algorithmic-bytes rate against the "
                          "HBM peak like the SpMM's Epinion2 figure",
                "random_order": bpr_rate(bu, bp, bn, True), "sampler_order": bpr_rate(su, sp_, sn_, True),
                "atomic_form_random_order": bpr_rate(bu, bp, bn, False),
                "atomic_form_sampler_order": bpr_rate(su, sp_, sn_, False)}
            aux["bpr_kernel_triples_per_s_T2e20"] = aux["bpr_roofline"]["random_order"]["triples_per_s"]
            aux["bpr_kernel_algorithmic_GBs"] = aux["bpr_roofline"]["random_order"]["algorithmic_GBs"]
            aux["bpr_kernel_triples_per_s_sampler_order"] = aux["bpr_roofline"]["sampler_order"]["triples_per_s"]
            del bu, bp, bn, su, sp_, sn_
            yb = (torch.rand(256, device=dev) < 1 / 6).float()
            ub, ib = tu[:256], tp[:256]
            acc = torch.zeros(1, device=dev)
            fn2 = lambda: stepper.step_bce(ub, ib, yb, loss_acc=acc, batch_rows_only=True)
            fn2()
            ms2 = time_events(fn2, 200)
            aux["exact_train_step_ms_B256"] = ms2
            # the same step on batches shaped like the reference's (dataloader.py:250-277: every observed pair once + 5 uniform
            # negatives each, shuffled): the positives' items arrive in proportion to their degree, so most batches carry a hub
            # row, and the batch kernel's push is bound by the per-CU atomic rate
            trng2 = np.random.default_rng(2022)
            pairs = trng2.integers(0, uu.numel(), (64, 256))                    # a sample = a random observed pair (u, i) ...
            tb_y = (trng2.random((64, 256)) < 1 / 6).astype(np.float32)         # ... itself (label 1) or one of its 5 negatives:
            tb_u = uu.numpy()[pairs]                                            # the SAME user with a uniform item (label 0)
            tb_i = np.where(tb_y > 0, ii.numpy()[pairs], trng2.integers(0, m_item, (64, 256)))
            tb = [(torch.from_numpy(tb_u[k].copy()).to(dev), torch.from_numpy(tb_i[k].copy()).to(dev), torch.from_numpy(tb_y[k].copy()).to(dev))
                  for k in range(64)]
            state = {"k": 0}

            def fn2b():
                u_, i_, y_ = tb[state["k"] & 63]
                state["k"] += 1
                return stepper.step_bce(u_, i_, y_, loss_acc=acc, batch_rows_only=True)
            fn2b()
            aux["exact_train_step_ms_B256_training_batches"] = time_events(fn2b, 256)
            # the same step in the deterministic accumulation mode (SPEX_STEP_DETERMINISTIC: no float atomics, all-pull backward)
            stepper.deterministic = True
            fn2b()
            aux["exact_train_step_ms_B256_training_batches_deterministic"] = time_events(fn2b, 256)
            stepper.deterministic = False
            fn2b()
            # the same step under the reference's recommended edge dropout (`--dropout 1 --keepprob 0.3`, README.md:119-123): a fresh
            # in-kernel (Philox) mask per step on the forward handle and on the transposed handle with its edge-id permutation
            try:
                from spex_amd.graph import csr_transpose
                from spex_amd.trainer import edge_dropout_mask
                t_rp, t_c, t_v, t_eid = csr_transpose(rowptr, col, val, n_nodes)
                graph_tr = SpexGraph(t_rp, t_c, t_v, n_cols=n_nodes, edge_id=t_eid, device=dev)
                dstp = LightGCNStepper(graph, stepper.E0.clone(), n_u, n_layers=L, lr=lr, graph_t=graph_tr)
                kd = {"k": 0}

                def fn2d():
                    kd["k"] += 1
                    mask = edge_dropout_mask(graph, 0.3, "philox", 7, kd["k"])
                    graph.set_edge_mask(*mask)
                    graph_tr.set_edge_mask(*mask)
                    return dstp.step_bce(ub, ib, yb, loss_acc=acc, batch_rows_only=True)
                for _ in range(10):
                    fn2d()
                aux["exact_train_step_ms_B256_edge_dropout_0.3"] = time_events(fn2d, 256)
            finally:
                graph.set_edge_mask(0)
            del dstp, graph_tr
            aux["exact_train_step_samples_per_s"] = 256 / (ms2 * 1e-3)
            aux["exact_train_step_edges_per_s"] = 2 * L * nnz / (ms2 * 1e-3)
            # NGCF (BASELINE configs[3] shape): one layer = SpMM on D^-1(A+I) + fused layer kernel; and the whole training
            # step of the model (forward, scoring, fused layer backward, SpMM^T, torch Adam)
            from spex_amd.graph import ngcf_norm_adj
            ncsr = ngcf_norm_adj(uu.numpy(), ii.numpy(), n_user, m_item)
            gn = SpexGraph(*ncsr, device=dev)
            ego = torch.rand(n_user + m_item, D, device=dev) - 0.5
            Wg, Wb = (torch.rand(D, D, device=dev) - 0.5 for _ in range(2))
            bg, bb = (torch.rand(D, device=dev) - 0.5 for _ in range(2))
            side = torch.empty_like(ego)

            def ngcf_fwd():
                gn.spmm(ego, Y=side)
                return ops.ngcf_layer(ego, side, Wg, bg, Wb, bb)
            ngcf_fwd()
            aux["ngcf_layer_forward_ms"] = time_events(ngcf_fwd, 200)
            aux["ngcf_layer_forward_edges_per_s"] = len(ncsr[1]) / (aux["ngcf_layer_forward_ms"] * 1e-3)
            del gn
            import scipy.sparse as sp
            from spex_amd.ngcf import NGCF
            nargs = argparse.Namespace(embed_size=64, layer_size="[64]", mess_dropout="[0.1]", regs="[1e-5]")
            net = NGCF({"n_users": n_user, "n_items": m_item,
                        "norm_adj": sp.csr_matrix((ncsr[2], ncsr[1], ncsr[0]), shape=(n_user + m_item,) * 2)}, dev, nargs).to(dev)
            opt = torch.optim.Adam(net.parameters(), lr=1e-3)
            net.train()

            def ngcf_train():
                opt.zero_grad()
                net(ub, ib, yb, flag=0).backward()
                opt.step()
            for _ in range(5):
                ngcf_train()
            aux["ngcf_train_step_ms_B256"] = time_events(ngcf_train, 100)
            from spex_amd.trainer import NGCFStepper
            nst = NGCFStepper(net, lr=1e-3)
            nacc = torch.zeros(1, device=dev)
            for _ in range(5):
                nst.step(ub, ib, yb, loss_acc=nacc)
            aux["ngcf_stepper_step_ms_B256"] = time_events(lambda: nst.step(ub, ib, yb, loss_acc=nacc), 200)
            del net, opt, nst
            # BASELINE configs[5]: the dual-task step (rec branch + expert gate + trust head over the shared user table +
            # uncertainty-weighted loss + Adam over every parameter) as one library call; 15 synthetic paths per step
            # (the reference driver's cap, 3 x trust_batch_size on Epinion2)
            import os as _os, sys as _sys
            _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "spex_amd", "dropin"))
            import utility1.model_expert_s as mex
            from spex_amd.trainer import DualTaskStepper

            class _DS:                                   # what model_expert_s.LightGCN reads from its dataset
                n_users, m_items = n_user, m_item
                getSparseGraph = staticmethod(lambda: graph)
            dargs = argparse.Namespace(hiddenSize=64, batchSize=100, nonhybrid=False, nb_heads=3, recdim=64, layer=L, keepprob=0.6,
                                       A_split=False, dropout=0)
            dnet = mex.LightGCN(dargs, _DS).to(dev)
            T_PATHS, P_LEN = 15, 6
            dst = DualTaskStepper(dnet, path_capacity=T_PATHS, path_len=P_LEN, lr=1e-3)
            prng = np.random.default_rng(11)
            plen = prng.integers(2, P_LEN + 1, T_PATHS)
            pseq = np.full((T_PATHS, P_LEN), n_user, dtype=np.int64)
            for r, l in enumerate(plen):
                pseq[r, :l] = prng.choice(n_user, size=l, replace=False)
            pseq_d, plen_d = torch.from_numpy(pseq).to(dev), torch.from_numpy(plen.astype(np.int64)).to(dev)
            ptgt = torch.from_numpy(prng.integers(0, n_user, T_PATHS)).to(dev)
            fn3 = lambda: dst.step(ub, ib, yb, pseq_d, plen_d, ptgt)
            for _ in range(20):
                fn3()
            aux["dual_task_step_ms_B256_T15"] = time_events(fn3, 500)      # two streams, fork / join (the default form)
            dst.pipelined = True                         # SPEX_STEP_PIPELINED: Adam split by owner over the two streams (pays when
            for _ in range(20):                          # the trust branch is the longer one; not on Epinion2 with 15 paths)
                fn3()
            dst.join()
            aux["dual_task_step_ms_B256_T15_pipelined"] = time_events(fn3, 500, finish=dst.join)
            dst.join()
            del dst
            # the ROW-PARTITIONED one-call steps (spex_partitioned_step_bce_f32 / spex_partitioned_dual_task_step_f32) at world
            # size 1, where every exchange is the library's local-copy shortcut (in place: nothing is copied but E^0): what the
            # partition's schedule itself costs beside the one-GPU steps above — 2L - 1 exchanges + one all-reduce per step on N GPUs
            try:
                from spex_amd.dist import PartitionedLightGCN, PartitionedStepper
                from spex_amd.dist_dual import PartitionedDualTask, PartitionedDualTaskStepper
                pP = PartitionedLightGCN(rowptr, col, val, n_u, L, D, 0, 1, lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev),
                                         dev, allgather="native-p2p")
                pst = PartitionedStepper(pP, torch.from_numpy(E0_host).to(dev), lr=lr)
                ppos = pst.positions(ub, ib)
                fnp = lambda: pst.step_bce(ub, ib, yb, pos=ppos, loss_acc=acc)
                for _ in range(10):
                    fnp()
                part1 = {"exact_train_step_ms_B256": time_events(fnp, 300)}
                pP.native.close()
                del pst, pP
                pmodel = PartitionedDualTask(dnet, (rowptr, col, val), 0, 1, dev)
                pdst = PartitionedDualTaskStepper(pmodel, path_capacity=T_PATHS, path_len=P_LEN, lr=1e-3)
                dpos = pdst.positions(ub, ib)
                fnpd = lambda: pdst.step(ub, ib, yb, pseq_d, plen_d, ptgt, pos=dpos)
                for _ in range(20):
                    fnpd()
                part1["dual_task_step_ms_B256_T15"] = time_events(fnpd, 300)
                pmodel.P.native.close()
                del pdst, pmodel
                aux["partitioned_one_call_steps_world1"] = part1
            except Exception as e:      # noqa: BLE001 — RCCL could not be bound, ...: the one-GPU figures stand
                aux["partitioned_one_call_steps_world1"] = {"error": repr(e)[:200]}
            del dnet
            # BASELINE configs[3] / [5] on their OWN shape (Trust_SPEX/code/main_trust.py:42: 6 812 Weibo users; synthetic interactions
            # with log-normal user activity and Zipf item popularity, hub rows beyond 1 024 entries): the exact LightGCN step and the
            # dual-task step (15 paths against the 6 812-user table)
            from spex_amd.datasets import synthetic_interactions
            wu_, wi_ = synthetic_interactions(6812, 20000, 400000, seed=7, sigma=1.4)
            wcsr = lightgcn_norm_adj(wu_.numpy(), wi_.numpy(), 6812, 20000)
            wgraph = SpexGraph(*wcsr, device=dev)
            wrng = np.random.default_rng(13)
            wE0 = torch.from_numpy(np.concatenate([xavier_uniform_np(6813, D, wrng), xavier_uniform_np(20000, D, wrng)])).to(dev)
            wst = LightGCNStepper(wgraph, wE0.clone(), 6813, n_layers=L, lr=lr)
            wub = torch.from_numpy(wrng.integers(0, 6812, 256)).to(dev)
            wib = torch.from_numpy(wrng.integers(0, 20000, 256)).to(dev)
            wacc = torch.zeros(1, device=dev)
            fnw = lambda: wst.step_bce(wub, wib, yb, loss_acc=wacc, batch_rows_only=True)
            for _ in range(10):
                fnw()
            aux["weibo_shape"] = {"graph": "6812 users x 20000 items, nnz %d (synthetic: log-normal activity, Zipf popularity), max row %d entries"
                                           % (len(wcsr[1]), int(np.diff(wcsr[0]).max())),
                                  "exact_train_step_ms_B256": time_events(fnw, 300)}

            class _WDS:
                n_users, m_items = 6812, 20000
                getSparseGraph = staticmethod(lambda: wgraph)
            wnet = mex.LightGCN(dargs, _WDS).to(dev)
            wdst = DualTaskStepper(wnet, path_capacity=T_PATHS, path_len=P_LEN, lr=1e-3)
            wseq = np.full((T_PATHS, P_LEN), 6812, dtype=np.int64)
            for r, l in enumerate(plen):
                wseq[r, :l] = wrng.choice(6812, size=l, replace=False)
            wseq_d, wtgt = torch.from_numpy(wseq).to(dev), torch.from_numpy(wrng.integers(0, 6812, T_PATHS)).to(dev)
            fnd = lambda: wdst.step(wub, wib, yb, wseq_d, plen_d, wtgt)
            for _ in range(20):
                fnd()
            aux["weibo_shape"]["dual_task_step_ms_B256_T15"] = time_events(fnd, 300)
            del wnet, wdst, wst
            # ... and both under the reference's recommended edge dropout (`--dropout 1 --keepprob 0.3`, README.md:119-123): the masked
            # launches fold their hub rows inside the launch (spmm_chunk_kernel<.., MASKED, .., FOLD>), no fix-up launch per product
            try:
                from spex_amd.graph import csr_transpose
                from spex_amd.trainer import edge_dropout_mask
                w_rp, w_c, w_v, w_eid = csr_transpose(*wcsr, len(wcsr[0]) - 1)
                wgraph_tr = SpexGraph(w_rp, w_c, w_v, n_cols=len(wcsr[0]) - 1, edge_id=w_eid, device=dev)
                wdstp = LightGCNStepper(wgraph, wE0.clone(), 6813, n_layers=L, lr=lr, graph_t=wgraph_tr)
                kw = {"k": 0}

                def fnwd():
                    kw["k"] += 1
                    mask = edge_dropout_mask(wgraph, 0.3, "philox", 7, kw["k"])
                    wgraph.set_edge_mask(*mask)
                    wgraph_tr.set_edge_mask(*mask)
                    return wdstp.step_bce(wub, wib, yb, loss_acc=wacc, batch_rows_only=True)
                for _ in range(10):
                    fnwd()
                aux["weibo_shape"]["exact_train_step_ms_B256_edge_dropout_0.3"] = time_events(fnwd, 256)
                wgraph.set_edge_mask(0)
                del wdstp, wgraph_tr
                dargs_d = argparse.Namespace(**dict(vars(dargs), dropout=1, keepprob=0.3))
                wnet_d = mex.LightGCN(dargs_d, _WDS).to(dev)
                wdst_d = DualTaskStepper(wnet_d, path_capacity=T_PATHS, path_len=P_LEN, lr=1e-3)

                def fndd():
                    kw["k"] += 1
                    wdst_d.set_edge_dropout(edge_dropout_mask(wgraph, 0.3, "philox", 7, kw["k"]))
                    return wdst_d.step(wub, wib, yb, wseq_d, plen_d, wtgt)
                for _ in range(20):
                    fndd()
                aux["weibo_shape"]["dual_task_step_ms_B256_T15_edge_dropout_0.3"] = time_events(fndd, 256)
                wdst_d.set_edge_dropout(None)
                del wnet_d, wdst_d
            finally:
                wgraph.set_edge_mask(0)
            del wgraph
            # BASELINE configs[3] on its own shape (Trust_SPEX/code/main_trust.py:44: 8 930 Twitter users): the NGCF training step
            tu_, ti_ = synthetic_interactions(8930, 20000, 400000, seed=9, sigma=1.4)
            tcsr = ngcf_norm_adj(tu_.numpy(), ti_.numpy(), 8930, 20000)
            tnet = NGCF({"n_users": 8930, "n_items": 20000,
                         "norm_adj": sp.csr_matrix((tcsr[2], tcsr[1], tcsr[0]), shape=(28930, 28930))}, dev, nargs).to(dev)
            tst = NGCFStepper(tnet, lr=1e-3)
            tub = torch.from_numpy(wrng.integers(0, 8930, 256)).to(dev)
            for _ in range(10):
                tst.step(tub, wib, yb, loss_acc=nacc)
            aux["twitter_shape"] = {"graph": "8930 users x 20000 items, NGCF adjacency D^-1(A+I), nnz %d (synthetic), max row %d entries"
                                             % (len(tcsr[1]), int(np.diff(tcsr[0]).max())),
                                    "ngcf_stepper_step_ms_B256": time_events(lambda: tst.step(tub, wib, yb, loss_acc=nacc), 300)}
            del tnet, tst
        except Exception as e:  # never lose the headline line to an auxiliary measurement
            aux["aux_error"] = repr(e)

    # one hipEvent pair around the SpMM launches of every n-th propagation of the timed region (three back-to-back
    # launches share the ~3 us an event pair costs on the stream; bracketing single launches charged it to each).
    # At most ~100 brackets per run: with 400 outstanding timing events the runtime's bookkeeping slowed every step
    # of a 2 000-step run by 14 us.  The timer is created before the warm-up (creating its events takes ~1 ms) and
    # reset right before the timed region, so that nothing but a synchronisation separates the warm-up from the K steps:
    # a gap of tens of milliseconds there (it used to hold a 40 ms garbage collection) lets the GPU clock down and made
    # a 20-step timed region read 8 us per step slower than a 200-step one.
    every = max(5, a.steps // 100) * (1 if not part else L)
    graph.attach_timer(128, every=every)
    for _ in range(a.warmup):
        step()
    if part and delta["on"]:
        # the replicated E0 must equal a real all-gather of the owners' rows, bit for bit, on every rank
        same = torch.equal(P.gathered0, P.all_gather_rows(E0_local, out=torch.empty_like(P.gathered0)))
        ok_ = torch.tensor([1.0 if same else 0.0], device=dev)
        dist.all_reduce(ok_, op=dist.ReduceOp.MIN)
        delta_info = {"checked_equal": ok_.item() == 1.0}
        if ok_.item() != 1.0:
            delta["on"] = False
        else:
            # keep whichever schedule is faster HERE (one 1.5 MB row exchange against one all-gather of the table: the
            # answer depends on world size and on the collective's latency) — ten steps each way, the slowest rank counts
            t_mode = {}
            for mode in (True, False):
                delta["on"] = mode
                if mode:
                    P.gather_first(E0_local)
                for _ in range(3):
                    step()
                barrier()
                t_ = time.perf_counter()
                for _ in range(10):
                    step()
                torch.cuda.synchronize()
                tt_ = torch.tensor([time.perf_counter() - t_], device=dev, dtype=torch.float64)
                dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
                t_mode[mode] = tt_.item() / 10
            delta["on"] = t_mode[True] <= t_mode[False]
            delta_info.update(step_ms_deltas=t_mode[True] * 1e3, step_ms_allgather=t_mode[False] * 1e3)
            if delta["on"]:
                P.gather_first(E0_local)
        for _ in range(3):
            step()
    graph.read_timer(reset=True)
    # a hipEvent pair on the steps' stream around the K steps, beside the wall clock: the wall-clock region also carries the first
    # launch's latency after the synchronisation and the closing synchronisation (~90 us in all: 4-5 us per step of a 20-step region,
    # nothing of a 2 000-step one) — `extra.timed_region` reports both so the two figures can be told apart
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev_a.record(); ev_b.record()          # (torch creates the hipEvent at an event's FIRST record: not inside the timed region)
    barrier()
    t0 = time.perf_counter()
    ev_a.record()
    for _ in range(a.steps):
        step()
    ev_b.record()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    region_event_ms = ev_a.elapsed_time(ev_b)
    if part:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    bracket_ms, bracket_launches = graph.read_timer(per_launch=False)
    graph.detach_timer()

    # ---- N > 1: STRONG scaling on the one size where a row partition can win: the 2^24-node graph (450 M stored
    #      entries, a 4.3 GB table) partitioned over the N ranks — 18.7 / N ms of SpMM per layer against the all-gather of the
    #      4.3 GB table.  Reported in `extra` (the headline `value` stays the weak-scaling figure the contract names).
    strong = None
    if part and not a.no_strong_scaling:
        try:
            from spex_amd.datasets import scaled_graph
            del P, E0_local
            torch.cuda.empty_cache()
            t_b = time.perf_counter()
            rp, cc, vv, n_u2 = scaled_graph(a.hbm_log2_nodes, device=dev)          # identical on every rank (seeded)
            n2, nnz2 = len(rp) - 1, len(cc)
            from spex_amd.dist import PartitionedLightGCN as _PL
            P2 = _PL(rp, cc, vv, n_u2, L, D, rank, world, lambda r, c, v, n_cols: SpexGraph(r, c, v, n_cols=n_cols, device=dev), dev)
            del rp, cc, vv
            build_s = time.perf_counter() - t_b
            if ag_info.get("used") in ("native", "native-p2p"):
                P2.set_allgather(ag_info["used"])
            X2 = (torch.rand(P2.n_local, D, device=dev) - 0.5) * 0.1
            for _ in range(2):
                P2.propagate(X2)
            barrier()
            t_ = time.perf_counter()
            n_it = 5
            for _ in range(n_it):
                P2.propagate(X2)
            torch.cuda.synchronize()
            tt_ = torch.tensor([time.perf_counter() - t_], device=dev, dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            ms_prop = tt_.item() / n_it * 1e3
            strong = {"workload": "Epinion2 x %d replicas, N=%d nodes ~2^%d, nnz=%d, d=64: ONE graph row-partitioned over %d ranks"
                                  % (round((1 << a.hbm_log2_nodes) / 15593), n2, a.hbm_log2_nodes, nnz2, world),
                      "propagation_ms": ms_prop, "edges_per_s": L * nnz2 / (ms_prop * 1e-3), "layers": L,
                      "exchange": ag_info.get("used", "collective"), "local_rows": P2.n_local, "local_nnz": P2.graph.nnz,
                      "table_gb": n2 * D * 4 / 1e9, "build_s": build_s,
                      "single_gpu_reference": "roofline_hbm of the N = 1 run of the same bench (one launch of the whole graph = one layer)"}
            del P2, X2
        except Exception as e:      # noqa: BLE001
            strong = {"error": repr(e)[:300]}

    if rank != 0:
        if part:
            dist.destroy_process_group()
        return

    edges_per_s = L * nnz * a.steps / dt
    n_timed = int(bracket_launches.sum())
    spmm_ms = float(bracket_ms.sum() / n_timed) if n_timed else float("nan")
    bytes_launch = algorithmic_bytes(local_nnz, local_rows)
    achieved = bytes_launch / (spmm_ms * 1e-3) / 1e9
    out = {
        "metric": "graph-conv edges/sec + BPR-triples/sec, LightGCN-SPEX Epinion2 d=64",
        "value": edges_per_s, "unit": "edges/s", "bpr_triples_per_s": T_TRIPLES * a.steps / dt, "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("epinion2 (reference preprocessing of the shipped Epinions .mat; N=%d nnz=%d d=%d L=%d)"
                                % (n_nodes, nnz, D, L)) if not part else
                               ("epinion2 x%d replicas cross-linked by per-interaction permutations; N=%d nnz=%d d=%d L=%d; "
                                "1-D row partition + RCCL all-gather per layer" % (world, n_nodes, nnz, D, L)),
                   "bpr_triples_per_step": T_TRIPLES, "embeddings": "xavier-uniform seed 2020 (synthetic weights)",
                   "parallelism": "single GPU" if not part else "row-partition x%d" % world,
                   **({} if not part else {"allgather": ag_info, "first_layer_exchange": dict(delta_info, used="deltas of the updated rows" if delta["on"] else "all-gather")})},
        "roofline": {"bound": "hbm", "kernel": "spmm_chunk_kernel (the step's three layer launches: <1> running-sum form for layer 1, <0> plain form for layers 2-3)" if not part else "spmm_chunk_kernel<1>", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "cache_algorithmic_frac": achieved / HBM_PEAK_GBS,
                     "avg_launch_us": spmm_ms * 1e3, "launches_timed": n_timed,
                     "algorithmic_bytes_per_launch": bytes_launch,
                     "compulsory_bytes_per_launch": compulsory_bytes(local_nnz, local_rows),
                     "regime": "cache-resident (4 MB table in L2 / Infinity Cache): `achieved` (algorithmic bytes / time) is a "
                               "cache-bandwidth figure here, NOT an HBM utilisation — see roofline_hbm for that"},
        "extra": dict(aux, bpr_triples_per_s_in_step=T_TRIPLES * a.steps / dt, **({"strong_scaling_hbm_graph": strong} if strong else {})),
    }
    out["extra"]["timed_region"] = {
        "wall_ms": dt * 1e3, "stream_events_ms": region_event_ms, "fixed_cost_us": (dt * 1e3 - region_event_ms) * 1e3,
        "ms_per_step_by_stream_events": region_event_ms / a.steps,
        "note": "`value` / `ms_per_step` use the wall clock between the barriers (the contract); the event pair brackets the same K steps "
                "on their stream: the difference is the region's start / stop cost, independent of K"}
    traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    stored = {}
    if os.path.exists(traffic_file):
        try:
            stored = json.load(open(traffic_file))
        except Exception:
            stored = {}
    # In the cache-resident regime the algorithmic-bytes rate can exceed the HBM peak (every gathered row is an L2 hit), so it is not a
    # roofline FRACTION: it goes to `cache_algorithmic_frac`, and `frac` is what actually crossed the L2 -> fabric boundary per launch
    # (rocprofv3 counters, stored profile) / time / peak — or, without a stored profile, the compulsory bytes / time / peak.  Both <= 1.
    comp = compulsory_bytes(local_nnz, local_rows)
    out["roofline"]["frac"] = comp / (spmm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out["roofline"]["frac_basis"] = "compulsory bytes / time / peak (cache-resident table: the algorithmic rate is in cache_algorithmic_frac)"
    if not part and stored.get("epinion2_spmm_bytes_per_launch"):
        tb = stored["epinion2_spmm_bytes_per_launch"]
        out["roofline"]["traffic"] = tb
        out["roofline"]["traffic_is_stored_profile"] = True
        out["roofline"]["traffic_source"] = stored.get("epinion2_source") or stored.get("source")
        out["roofline"]["l2_miss_traffic_over_compulsory"] = tb / comp
        out["roofline"]["l2_miss_traffic_GBs"] = tb / (spmm_ms * 1e-3) / 1e9
        out["roofline"]["l2_hit_rate_profiled"] = stored.get("epinion2_l2_hit_rate")
        out["roofline"]["frac"] = out["roofline"]["l2_miss_traffic_GBs"] / HBM_PEAK_GBS
        out["roofline"]["frac_basis"] = ("counter traffic (L2-miss bytes per launch, stored rocprofv3 profile) / time / peak — cache-resident "
                                         "table: the algorithmic rate is in cache_algorithmic_frac; the HBM-roofline claim is roofline_hbm")

    if not part:
        # ---- HBM-resident graph: where the roofline fraction is meaningful
        if not a.no_hbm_roofline:
            try:
                from spex_amd.datasets import scaled_graph
                del stepper
                torch.cuda.empty_cache()
                rp, cc, vv, _ = scaled_graph(a.hbm_log2_nodes, device=dev)
                n2, nnz2 = len(rp) - 1, len(cc)
                t_build = time.perf_counter()
                g2 = SpexGraph(rp, cc, vv, device=dev)
                out["extra"]["graph_create_s_hbm_graph"] = time.perf_counter() - t_build   # host packing (threaded) + upload
                del rp, cc, vv
                X = torch.rand(n2, D, device=dev) - 0.5
                Y = torch.empty_like(X)
                A2 = torch.zeros_like(X)
                for _ in range(3):
                    g2.spmm(X, Y=Y, acc_in=A2, acc_out=A2)       # the forward-layer form, as in the propagation
                g2.attach_timer(10)
                for _ in range(10):
                    g2.spmm(X, Y=Y, acc_in=A2, acc_out=A2)
                ms_all = np.asarray(g2.read_timer(), np.float64)
                ms = float(ms_all.mean())
                g2.detach_timer()
                b2 = algorithmic_bytes(nnz2, n2)
                ach = b2 / (ms * 1e-3) / 1e9
                # what a plain device copy of the same 4.3 GB table reaches on this box (the fraction of MEASURED copy
                # bandwidth beside the fraction of the 8 TB/s peak): read + write of X, HIP events on torch's stream
                # (the better of the runtime's device-to-device copy and an elementwise kernel over the same buffers)
                copy_gbs = 0.0
                for fn in (lambda: Y.copy_(X), lambda: torch.mul(X, 1.0, out=Y)):
                    for _ in range(2):
                        fn()
                    copy_gbs = max(copy_gbs, 2.0 * X.numel() * 4 / (time_events(fn, 5) * 1e-3) / 1e9)
                k = round((1 << a.hbm_log2_nodes) / 15593)
                out["roofline_hbm"] = {"bound": "hbm", "kernel": "spmm_chunk_kernel<1>", "achieved": ach,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                       "avg_launch_us": ms * 1e3, "launch_us_min_median_max": [float(ms_all.min() * 1e3),
                                                                                                 float(np.median(ms_all) * 1e3),
                                                                                                 float(ms_all.max() * 1e3)],
                                       "algorithmic_bytes_per_launch": b2,
                                       # the launch timed here is the forward-layer form with the layer mean fused (acc_in read, acc_out written:
                                       # SURVEY 8d "adds N*4d read + write per layer if acc is kept in HBM") — `frac` keeps the base formula
                                       "algorithmic_bytes_incl_fused_mean_streams": b2 + 2 * n2 * D * 4,
                                       "frac_incl_fused_mean_streams": (b2 + 2 * n2 * D * 4) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "copy_bandwidth_GBs": copy_gbs, "frac_of_copy_bandwidth": ach / copy_gbs,
                                       "edges_per_s": nnz2 / (ms * 1e-3),
                                       "workload": "Epinion2 x %d replicas (same degree law, cross-linked), N=%d nodes "
                                                   "~2^%d, nnz=%d, d=64 (X = %.2f GB >> 256 MB Infinity Cache), "
                                                   "long rows=%d" % (k, n2, a.hbm_log2_nodes, nnz2, n2 * D * 4 / 1e9,
                                                                     g2.n_long_rows)}
                prof = (stored.get("hbm_graphs") or {}).get(str(k))
                if prof:        # rocprofv3 --pmc figure of this very graph, stored (profiles/hbm_traffic.json): not measured in this run
                    out["roofline_hbm"]["traffic"] = prof["spmm_bytes_per_launch"]
                    out["roofline_hbm"]["traffic_is_stored_profile"] = True
                    out["roofline_hbm"]["traffic_over_algorithmic"] = prof["spmm_bytes_per_launch"] / b2
                    out["roofline_hbm"]["traffic_GBs"] = prof["spmm_bytes_per_launch"] / (ms * 1e-3) / 1e9
                    out["roofline_hbm"]["l2_hit_rate_profiled"] = prof.get("l2_hit_rate")
                    out["roofline_hbm"]["traffic_source"] = prof.get("source")
                # the exact training step (3 SpMM fwd, scoring, 3 SpMM bwd, Adam over the whole table) on the same graph
                del Y, A2
                st2 = LightGCNStepper(g2, X.mul_(0.1), n2 - (n2 // 15593) * 12407, n_layers=L, lr=lr)
                ub, ib = tu[:256] % 1000, tp[:256] % 1000
                yb2 = (torch.rand(256, device=dev) < 1 / 6).float()
                for _ in range(2):
                    st2.step_bce(ub, ib, yb2)
                ms_big = time_events(lambda: st2.step_bce(ub, ib, yb2), 3)
                out["extra"]["exact_train_step_ms_hbm_graph"] = ms_big
                out["extra"]["exact_train_step_edges_per_s_hbm_graph"] = 2 * L * nnz2 / (ms_big * 1e-3)
                del g2, X, st2
            except Exception as e:
                out["roofline_hbm"] = {"error": repr(e)}

        # ---- CPU baseline: the C restatement on the host cores, same step, bounded samples
        if not a.no_cpu_baseline:
            try:
                from oracle import oracle as O
                cores = min(len(os.sched_getaffinity(0)), 16)   # a 1-GPU box's CPU share is 16 cores
                tuh, tph, tnh = tu.cpu().numpy(), tp.cpu().numpy(), tn.cpu().numpy()

                def cpu_steps(n_threads, budget):
                    O.propagate_mean(rowptr, col, val, E0_host, L, n_threads=n_threads)
                    ts = []
                    t_all = time.perf_counter()
                    while time.perf_counter() - t_all < budget:
                        t1 = time.perf_counter()
                        lo_h = O.propagate_mean(rowptr, col, val, E0_host, L, n_threads=n_threads)
                        O.bpr_sgd(lo_h[:n_u], lo_h[n_u:], E0_host[:n_u], E0_host[n_u:], tuh, tph, tnh, lr, 0.0)
                        ts.append(time.perf_counter() - t1)
                    return np.asarray(ts)
                ts = cpu_steps(cores, 8.0)
                out["cpu_baseline"] = {"value": L * nnz * len(ts) / ts.sum(), "unit": "edges/s", "cores": cores, "kind": "port",
                                       "sample": "%d steps (3-layer propagation + BPR step over %d triples) of the same "
                                                 "Epinion2 workload in %.1f s, C restatement with OpenMP" % (len(ts), T_TRIPLES, ts.sum()),
                                       "ms_per_step_median_p10_p90": [float(np.percentile(ts, q) * 1e3) for q in (50, 10, 90)]}
                ts1 = cpu_steps(1, 5.0)
                out["cpu_baseline"]["one_thread_edges_per_s"] = L * nnz * len(ts1) / ts1.sum()
                out["cpu_baseline"]["one_thread_ms_per_step_median_p10_p90"] = [float(np.percentile(ts1, q) * 1e3) for q in (50, 10, 90)]
                # the exact training step (forward + backward through the propagation + Adam), B = 256
                ubh, ibh = tuh[:256], tph[:256]
                ybh = (np.random.default_rng(5).random(256) < 1 / 6).astype(np.float32)
                W, m_, v_ = E0_host.copy(), np.zeros_like(E0_host), np.zeros_like(E0_host)
                t_csr = O.csr_transpose(rowptr, col, val, n_nodes)
                tt, t_all, k_ = [], time.perf_counter(), 0
                while time.perf_counter() - t_all < 5.0:
                    t1 = time.perf_counter()
                    _, _, gr = O.lightgcn_loss_and_grad(rowptr, col, val, W, n_u, L, ubh, ibh, ybh, n_threads=cores, t_csr=t_csr)
                    k_ += 1
                    O.adam_step(W, gr, m_, v_, k_)
                    tt.append(time.perf_counter() - t1)
                tt = np.asarray(tt)
                out["cpu_baseline"]["train_step_edges_per_s"] = 2 * L * nnz * len(tt) / tt.sum()
                out["cpu_baseline"]["train_step_ms_median_p10_p90"] = [float(np.percentile(tt, q) * 1e3) for q in (50, 10, 90)]
                # what the reference literally executes: torch.sparse.mm on the CPU (stock PyTorch), 3 layers + mean
                A = graph.to_torch_sparse()
                E = torch.from_numpy(E0_host)
                torch.set_num_threads(cores)

                def ref_computer():
                    embs, cur = [E], E
                    for _ in range(L):
                        cur = torch.sparse.mm(A, cur)
                        embs.append(cur)
                    return torch.mean(torch.stack(embs, dim=1), dim=1)
                ref_computer()
                tr_, t_all = [], time.perf_counter()
                while time.perf_counter() - t_all < 5.0:
                    t1 = time.perf_counter()
                    ref_computer()
                    tr_.append(time.perf_counter() - t1)
                tr_ = np.asarray(tr_)
                out["cpu_baseline"]["torch_sparse_mm_propagate_edges_per_s"] = L * nnz * len(tr_) / tr_.sum()
                out["cpu_baseline"]["torch_sparse_mm_ms_per_propagate"] = tr_.sum() / len(tr_) * 1e3
                out["cpu_baseline"]["torch_sparse_mm_ms_median_p10_p90"] = [float(np.percentile(tr_, q) * 1e3) for q in (50, 10, 90)]
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}

    print(json.dumps(out), file=real_stdout, flush=True)
    if part:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
