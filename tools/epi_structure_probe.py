"""Which structural feature of the Weibo-shaped graph makes the epilogue forms (<1>, <2>) slower than the plain form (<0>)?  The same
interactions with (a) nothing changed, (b) every empty row given one entry, (c) rows capped at 1 024 entries (no hub), (d) both,
(e) rows capped at 64 (no multi-segment row at all).  us per launch by HIP events."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SPEX_LIB"):
    from spex_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["SPEX_LIB"])
from spex_amd.datasets import synthetic_interactions
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
dev = torch.device("cuda:0")


def timed(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


n_u, n_i = 6812, 20000
u, i = synthetic_interactions(n_u, n_i, 400000, seed=7, sigma=1.4)
u, i = u.numpy(), i.numpy()
rng = np.random.default_rng(5)


def variant(fill_empty, cap):
    uu, ii = u.copy(), i.copy()
    if cap:
        keep = np.ones(len(uu), bool)
        for arr in (uu, ii):
            order = np.argsort(arr, kind="stable")
            srt = arr[order]
            start = np.searchsorted(srt, srt, side="left")
            rank = np.arange(len(srt)) - start
            keep[order[rank >= cap]] = False
        uu, ii = uu[keep], ii[keep]
    if fill_empty:
        miss_i = np.setdiff1d(np.arange(n_i), ii)
        miss_u = np.setdiff1d(np.arange(n_u), uu)
        uu = np.concatenate([uu, rng.integers(0, n_u, len(miss_i)), miss_u])
        ii = np.concatenate([ii, miss_i, rng.integers(0, n_i, len(miss_u))])
    return lightgcn_norm_adj(uu, ii, n_u, n_i)


for name, fe, cap in (("as is", False, 0), ("no empty rows", True, 0), ("rows <= 1024", False, 1024), ("both", True, 1024), ("rows <= 64", True, 64)):
    csr = variant(fe, cap)
    deg = np.diff(csr[0])
    g = SpexGraph(*csr, device=dev)
    X = torch.rand(len(deg), 64, device=dev) - 0.5
    Y, A = torch.empty_like(X), torch.empty_like(X)
    print("%-14s nnz %7d empty %5d >64 %5d >1024 %3d | <0> %.1f  <1> %.1f  <2> %.1f us" % (
        name, len(csr[1]), (deg == 0).sum() - 1, (deg > 64).sum(), (deg > 1024).sum(), timed(lambda: g.spmm(X, Y=Y)),
        timed(lambda: g.spmm(X, Y=Y, acc_in=X, acc_out=A, acc_div=1.0)), timed(lambda: g.spmm(X, Y=Y, add_in=X, add_div=1.0))), flush=True)
