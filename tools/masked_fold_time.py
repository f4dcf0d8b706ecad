"""Edge dropout (`--dropout 1 --keepprob 0.3`, reference README.md:119-123) with hub rows: the masked SpMM launch with the in-launch
hub fold (spmm_chunk_kernel<.., MASKED, .., FOLD>) against the fix-up form (SPEX_HUB_FOLD=0), per product and per exact training
step, on the Weibo-shaped graph — and the Epinion2 masked launch (no hub: must not move).  us by HIP events."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd.datasets import epinion2_tables, load_epinion2, synthetic_interactions, xavier_uniform_np
from spex_amd.graph import SpexGraph, csr_transpose, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper, edge_dropout_mask
dev = torch.device("cuda:0")


def timed(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(name, csr, n_u, n_i):
    n = n_u + 1 + n_i
    g = SpexGraph(*csr, device=dev)
    t_rp, t_c, t_v, t_eid = csr_transpose(*csr, n)
    gt = SpexGraph(t_rp, t_c, t_v, n_cols=n, edge_id=t_eid, device=dev)
    rng = np.random.default_rng(13)
    E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(n_u + 1, 64, rng), xavier_uniform_np(n_i, 64, rng)])).to(dev)
    ub = torch.from_numpy(rng.integers(0, n_u, 256)).to(dev); ib = torch.from_numpy(rng.integers(0, n_i, 256)).to(dev)
    yb = torch.from_numpy((rng.random(256) < 1 / 6).astype(np.float32)).to(dev)
    acc = torch.zeros(1, device=dev)
    X, Y, A = E0.clone(), torch.empty_like(E0), torch.empty_like(E0)
    for fold in ("1", "0"):
        os.environ["SPEX_HUB_FOLD"] = fold
        mask = edge_dropout_mask(g, 0.3, "philox", 7, 1)
        g.set_edge_mask(*mask); gt.set_edge_mask(*mask)
        t0 = timed(lambda: g.spmm(X, Y=Y))
        t1 = timed(lambda: g.spmm(X, Y=Y, acc_in=X, acc_out=A, acc_div=1.0))
        t2 = timed(lambda: gt.spmm(X, Y=Y, add_in=X, add_div=1.0))
        st = LightGCNStepper(g, E0.clone(), n_u + 1, n_layers=3, lr=1e-3, graph_t=gt)
        k = {"k": 0}

        def step():
            k["k"] += 1
            m = edge_dropout_mask(g, 0.3, "philox", 7, k["k"])
            g.set_edge_mask(*m); gt.set_edge_mask(*m)
            st.step_bce(ub, ib, yb, loss_acc=acc, batch_rows_only=True)
        ts = timed(step, 256)
        g.set_edge_mask(0); gt.set_edge_mask(0)
        tp = timed(lambda: g.spmm(X, Y=Y))
        if fold == "1":
            print("   unmasked <1> %.1f  <2, A^T> %.1f us" % (timed(lambda: g.spmm(X, Y=Y, acc_in=X, acc_out=A, acc_div=1.0)),
                                                           timed(lambda: gt.spmm(X, Y=Y, add_in=X, add_div=1.0))), flush=True)
            keep = (torch.rand(g.nnz, device=dev) < 0.3).to(torch.uint8)
            g.set_edge_mask(1, keep, 0.3, 0)
            print("   injected mask: masked <0> %.1f <1> %.1f us" % (timed(lambda: g.spmm(X, Y=Y)), timed(lambda: g.spmm(X, Y=Y, acc_in=X, acc_out=A, acc_div=1.0))), flush=True)
            g.set_edge_mask(0)
        print("%s hub fold %s: masked <0> %.1f  <1> %.1f  <2, A^T> %.1f us; exact step under dropout %.1f us; unmasked <0> %.1f us"
              % (name, "in launch" if fold == "1" else "fix-up   ", t0, t1, t2, ts, tp), flush=True)
    del os.environ["SPEX_HUB_FOLD"]


u, i = synthetic_interactions(6812, 20000, 400000, seed=7, sigma=1.4)
run("weibo-shape", lightgcn_norm_adj(u.numpy(), i.numpy(), 6812, 20000), 6812, 20000)
tr = load_epinion2()["train"]
run("epinion2   ", lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407), 3185, 12407)
