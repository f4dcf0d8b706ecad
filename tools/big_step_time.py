#!/usr/bin/env python3
"""The exact training step (propagate, BCE, backward, Adam) and the BPR step on an HBM-resident graph (Epinion2 x K)."""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spex_amd.datasets import scaled_graph
from spex_amd.graph import SpexGraph
from spex_amd.trainer import LightGCNStepper

ap = argparse.ArgumentParser()
ap.add_argument("--log2", type=int, default=22)
a = ap.parse_args()
dev = torch.device("cuda:0")
rp, cc, vv, n_u = scaled_graph(a.log2, device=dev)
n, nnz = len(rp) - 1, len(cc)
g = SpexGraph(rp, cc, vv, device=dev)
del rp, cc, vv
E0 = (torch.rand(n, 64, device=dev) - 0.5) * 0.1
st = LightGCNStepper(g, E0, n_u, n_layers=3, lr=1e-3)
rng = np.random.default_rng(0)
out = {"log2_nodes": a.log2, "n": n, "nnz": nnz}
for B in (256, 1 << 16):
    u = torch.from_numpy(rng.integers(0, n_u - 1, B)).to(dev); i = torch.from_numpy(rng.integers(0, n - n_u, B)).to(dev)
    y = (torch.rand(B, device=dev) < 1 / 6).float()
    for _ in range(2): st.step_bce(u, i, y)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): st.step_bce(u, i, y)
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / 5
    out[f"exact_step_ms_B{B}"] = ms
    out[f"exact_step_edges_per_s_B{B}"] = 6 * nnz / (ms * 1e-3)
T = 1 << 20
u = torch.from_numpy(rng.integers(0, n_u - 1, T)).to(dev); p = torch.from_numpy(rng.integers(0, n - n_u, T)).to(dev)
q = torch.from_numpy(rng.integers(0, n - n_u, T)).to(dev)
for _ in range(2): st.step_bpr_sgd(u, p, q)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): st.step_bpr_sgd(u, p, q)
e.record(); e.synchronize()
ms = s.elapsed_time(e) / 5
out["bpr_step_ms_T2e20"] = ms
out["bpr_step_edges_per_s"] = 3 * nnz / (ms * 1e-3)
print(json.dumps(out))
