#!/usr/bin/env python3
"""How much of the Epinion2 layer time is irregularity?  Same entry count on rows of exactly `deg` entries."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.graph import SpexGraph
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n_cols = 15593
for deg in (64, 32, 16, 128, 27):
    n_rows = 418608 // deg
    col = np.stack([np.sort(rng.choice(n_cols, deg, replace=False)) for _ in range(n_rows)]).reshape(-1).astype(np.int32)
    rowptr = (np.arange(n_rows + 1) * deg).astype(np.int32)
    val = rng.random(len(col)).astype(np.float32)
    g = SpexGraph(rowptr, col, val, n_cols=n_cols, device=dev)
    X = torch.rand(n_cols, 64, device=dev)
    Y = torch.empty(n_rows, 64, device=dev)
    for _ in range(5): g.spmm(X, Y=Y)
    g.attach_timer(50)
    for _ in range(50): g.spmm(X, Y=Y)
    ms = g.read_timer()
    print("rows of %3d entries x %6d rows: %.1f us (min %.1f)" % (deg, n_rows, ms.mean() * 1e3, ms.min() * 1e3))
