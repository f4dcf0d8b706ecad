import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spex_amd.datasets import load_epinion2, xavier_uniform_np
from spex_amd.graph import SpexGraph, lightgcn_norm_adj
from spex_amd.trainer import LightGCNStepper
dev = torch.device("cuda:0")
tr = load_epinion2()["train"]
csr = lightgcn_norm_adj(tr[:, 0], tr[:, 1], 3185, 12407)
g = SpexGraph(*csr, device=dev)
rng = np.random.default_rng(0)
E0 = torch.from_numpy(np.concatenate([xavier_uniform_np(3186, 64, rng), xavier_uniform_np(12407, 64, rng)])).to(dev)
st = LightGCNStepper(g, E0, 3186, n_layers=3, lr=1e-3)
T = 2048
tu = torch.from_numpy(rng.integers(0, 3185, T)).to(dev); tp = torch.from_numpy(rng.integers(0, 12407, T)).to(dev); tn = torch.from_numpy(rng.integers(0, 12407, T)).to(dev)
for _ in range(300): st.step_bpr_sgd(tu, tp, tn)
for trial in range(3):
    torch.cuda.synchronize()
    time.sleep(0.002 * trial)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    for e in evs: e.record()              # (torch creates the hipEvent at the FIRST record: keep that out of the timed loop)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    hs = []
    for k in range(20):
        st.step_bpr_sgd(tu, tp, tn)
        evs[k + 1].record()
        hs.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print("trial", trial, "wall %.1f us; per step by events:" % (wall * 1e6), " ".join("%.0f" % (evs[k].elapsed_time(evs[k + 1]) * 1e3) for k in range(20)))
    print("   host time at the end of each step's enqueue (us):", " ".join("%.0f" % (h * 1e6) for h in hs))
