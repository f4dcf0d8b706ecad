#!/usr/bin/env python3
"""Small driver for counter passes (rocprofv3 --pmc): N launches of the SpMM on Epinion2 and on the HBM-resident
synthetic graph, nothing else in the process worth profiling.  Usage: prof_spmm.py [epinion2|hbm] [launches] [log2]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
n = len(rowptr) - 1
X = torch.rand(n, 64, device=dev) - 0.5
Y = torch.empty_like(X)
acc = torch.zeros_like(X)
for _ in range(launches):
    g.spmm(X, Y=Y, acc_in=acc, acc_out=acc)      # the forward-layer form (EPI = 1)
torch.cuda.synchronize()
print("done", which, n, len(col), "tasks/long:", g.n_segments, g.n_long_rows)
