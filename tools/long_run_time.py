#!/usr/bin/env python3
"""Sustained step rate over long runs: eager launches, eager with host-side flow control (at most ~2 x `window` steps
queued), and HIP-graph replay — for K = 500 ... 8000 steps of the bench step (propagate + fused BPR)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
dev = torch.device("cuda:0")
g = SpexGraph(*csr, device=dev)
E0 = torch.from_numpy(xavier_uniform_np(15593, 64, np.random.default_rng(0))).to(dev)
st = LightGCNStepper(g, E0, 3186)
tu = torch.randint(0, 3185, (2048,), device=dev); tp = torch.randint(0, 12407, (2048,), device=dev); tn = torch.randint(0, 12407, (2048,), device=dev)
step = lambda: st.step_bpr_sgd(tu, tp, tn)

def run(K, mode, window=64):
    for _ in range(50): step()
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    if mode == "graph":
        for _ in range(K): graph.replay()
    else:
        for k in range(K):
            step()
            if mode == "flow" and (k + 1) % window == 0:
                e = torch.cuda.Event(); e.record(); evs.append(e)
                if len(evs) > 2:
                    evs.pop(0).synchronize()          # never more than ~2 windows of launches outstanding
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6

s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
for K in (500, 2000, 8000):
    print("K=%5d  eager %.1f us/step | flow-controlled(64) %.1f | flow(16) %.1f | graph replay %.1f" % (
        K, run(K, "eager"), run(K, "flow", 64), run(K, "flow", 16), run(K, "graph")))
