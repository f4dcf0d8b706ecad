"""Counter probe: the plain form (<0>) against the backward epilogue form (<2>) of the Epinion2 SpMM launch, 40 launches each —
run under `rocprofv3 --kernel-trace --pmc <counters>`; summarise with tools/pmc_by_kernel.py."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
g = SpexGraph(*lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407), device=dev)
X = torch.rand(15593, 64, device=dev) - 0.5
E = torch.rand(15593, 64, device=dev) - 0.5
Y, A = torch.empty_like(X), torch.empty_like(X)
for _ in range(40):
    g.spmm(X, Y=Y)
for _ in range(40):
    g.spmm(X, Y=Y, add_in=E, add_div=1.0)
for _ in range(40):
    g.spmm(X, Y=Y, acc_in=E, acc_out=A, acc_div=1.0)
torch.cuda.synchronize()
