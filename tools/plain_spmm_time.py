"""Plain-form SpMM launch (spmm_chunk_kernel<0>) and the headline step (3-layer propagation + fused BPR over 2 048 triples) on
Epinion2 and on the Weibo-shaped graph — us by HIP events.  SPEX_LIB=<path> times another build of the library (A/B)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SPEX_LIB"):
    from spex_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["SPEX_LIB"])
from spex_amd.datasets import load_epinion2, synthetic_interactions, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
dev = torch.device("cuda:0")


def timed(fn, n=400, reps=3):
    for _ in range(30): fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def run(name, csr, n_u, n_i):
    g = SpexGraph(*csr, device=dev)
    rng = np.random.default_rng(13)
    E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(n_u + 1, 64, rng), xavier_uniform_np(n_i, 64, rng)])).to(dev)
    X, Y, A = E0.clone(), torch.empty_like(E0), torch.empty_like(E0)
    t0 = timed(lambda: g.spmm(X, Y=Y))
    t1 = timed(lambda: g.spmm(X, Y=Y, acc_in=X, acc_out=A, acc_div=1.0))
    t2 = timed(lambda: g.spmm(X, Y=Y, add_in=X, add_div=1.0))
    st = LightGCNStepper(g, E0.clone(), n_u + 1, n_layers=3, lr=0.05)
    T = 2048
    tu = torch.from_numpy(rng.integers(0, n_u, T)).to(dev); tp = torch.from_numpy(rng.integers(0, n_i, T)).to(dev)
    tn = torch.from_numpy(rng.integers(0, n_i, T)).to(dev)
    ts = timed(lambda: st.step_bpr_sgd(tu, tp, tn), 300)
    ub, ib = tu[:256], tp[:256]
    yb = (torch.rand(256, device=dev) < 1 / 6).float()
    acc = torch.zeros(1, device=dev)
    te = timed(lambda: st.step_bce(ub, ib, yb, loss_acc=acc, batch_rows_only=True), 300)
    print("%s: spmm <0> %.2f  <1> %.2f  <2> %.2f us;  propagate + BPR step %.2f us;  exact BCE step %.2f us" % (name, t0, t1, t2, ts, te), flush=True)


tr = load_epinion2()["train"]
run("epinion2   ", lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407), 3185, 12407)
u, i = synthetic_interactions(6812, 20000, 400000, seed=7, sigma=1.4)
run("weibo-shape", lightgcn_norm_adj(u.numpy(), i.numpy(), 6812, 20000), 6812, 20000)
