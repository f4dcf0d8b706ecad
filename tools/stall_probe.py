#!/usr/bin/env python3
"""When does the one-time launch stall happen?  Wall time of consecutive 100-step chunks of the bench step (no syncs)."""
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
import gc
if len(sys.argv) > 1 and sys.argv[1] == 'freeze':
    gc.collect(); gc.freeze()
if len(sys.argv) > 1 and sys.argv[1] == 'disable':
    gc.disable()
torch.cuda.synchronize()
marks = []
t0 = time.perf_counter()
for k in range(6000):
    st.step_bpr_sgd(tu, tp, tn)
    if (k + 1) % 100 == 0:
        marks.append(time.perf_counter())
torch.cuda.synchronize()
prev = t0
out = []
for i, m in enumerate(marks):
    out.append("%d:%.1f" % ((i + 1) * 100, (m - prev) * 1e3)); prev = m
print("ms per 100-step chunk (host time):", " ".join(out))
print("total %.1f us/step" % ((time.perf_counter() - t0) / 6000 * 1e6))
