#!/usr/bin/env python3
"""Small driver for profiling / A-B timing of the SpMM alone: N launches of the forward-layer form on Epinion2 and on
the HBM-resident Epinion2 x K graph.  Usage: prof_spmm.py [epinion2|hbm] [launches] [log2_nodes]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("SPEX_LIB"):
    from spex_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["SPEX_LIB"])          # A/B against another build of the library
from spex_amd.datasets import load_epinion2, scaled_graph  # noqa: E402
from spex_amd.graph import SpexGraph, lightgcn_norm_adj  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "epinion2"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
if which == "epinion2":
    tr = load_epinion2()["train"]
    rowptr, col, val = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
else:
    rowptr, col, val, _ = scaled_graph(int(sys.argv[3]) if len(sys.argv) > 3 else 23, device=dev)
g = SpexGraph(rowptr, col, val, device=dev)
n, nnz = len(rowptr) - 1, len(col)
X = torch.rand(n, 64, device=dev) - 0.5
Y = torch.empty_like(X)
acc = torch.zeros_like(X)
for _ in range(3):
    g.spmm(X, Y=Y, acc_in=acc, acc_out=acc)
g.attach_timer(launches)
for _ in range(launches):
    g.spmm(X, Y=Y, acc_in=acc, acc_out=acc)      # the forward-layer form (EPI = 1)
ms = g.read_timer()
torch.cuda.synchronize()
alg = nnz * 264 + n * 260
g.attach_timer(launches)
for _ in range(launches):
    g.spmm(X, Y=Y)                                # plain form (EPI = 0)
ms0 = g.read_timer()
print("   plain Y = A X form: %.1f us (min %.1f)" % (ms0.mean() * 1e3, ms0.min() * 1e3))
print("%s: N=%d nnz=%d long_rows=%d segments=%d | main kernel %.1f us (min %.1f) -> %.0f GB/s algorithmic, %.2f G edges/s"
      % (which, n, nnz, g.n_long_rows, g.n_segments, ms.mean() * 1e3, ms.min() * 1e3, alg / ms.mean() / 1e6, nnz / ms.mean() / 1e6))
